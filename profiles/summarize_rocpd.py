#!/usr/bin/env python3
"""Turns rocprofv3's rocpd SQLite output (kernel trace and/or PMC passes) into the small
text summaries committed under profiles/.   usage: summarize_rocpd.py <results.db> [...]"""
import sqlite3
import sys


def main():
    for path in sys.argv[1:]:
        db = sqlite3.connect(path)
        cur = db.cursor()
        print("## %s" % path)
        try:
            rows = list(cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
            if rows:
                print("kernel,calls,total_us,avg_us,pct")
                for r in rows:
                    print("%s,%d,%.3f,%.3f,%.2f" % (r[0], r[1], r[2], r[3], r[4]))
        except sqlite3.Error:
            pass
        try:
            q = ("select kernel_name,counter_name,count(*),avg(value),sum(value) from counters_collection "
                 "group by kernel_name,counter_name")
            rows = list(cur.execute(q))
            if rows:
                print("kernel,counter,dispatches,avg_per_dispatch,sum")
                for r in rows:
                    print("%s,%s,%d,%.3f,%.3f" % r)
        except sqlite3.Error:
            pass
        print()


if __name__ == "__main__":
    main()
