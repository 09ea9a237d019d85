"""What a 20-step block of bench.py looks like on the device: reads the queue / start / end table that `tools/prof_round5.sh trace`
makes from a rocprofv3 --kernel-trace CSV (one row per k_recon_lf launch), splits it into blocks at idle gaps, and prints per block
the span, every lane's chain (launches, gaps between them, first start, last end) and the launch durations in order.
  python tools/block_timeline.py profiles/r05_block_timeline_dense.csv"""
import csv
import sys

import numpy as np


def main(path, per_block=40):
    rows = list(csv.DictReader(open(path)))
    ev = sorted((int(r["start_ns"]), int(r["end_ns"]), r["queue"]) for r in rows)
    blocks, cur, last_end = [], [], None
    for s, e, q in ev:
        if last_end is not None and s - last_end > 60000:
            blocks.append(cur)
            cur = []
        cur.append((s, e, q))
        last_end = max(last_end or 0, e)
    blocks.append(cur)
    full = [b for b in blocks if len(b) == per_block]
    print("%s: %d launches, %d idle-separated groups, %d of them whole blocks of %d launches" % (path, len(ev), len(blocks), len(full), per_block))
    if not full:
        return
    spans = np.array([max(e for _, e, _ in b) - b[0][0] for b in full]) / 1e3
    print("block span on the device (first start -> last end), us: median %.1f  min %.1f  max %.1f  (= %.2f us a step)"
          % (np.median(spans), spans.min(), spans.max(), np.median(spans) / (per_block / 2)))
    durs = np.array([[e - s for s, e, _ in b] for b in full]) / 1e3
    print("launch duration, us: first %.1f  median %.1f  last %.1f" % (np.median(durs[:, 0]), np.median(durs), np.median(durs[:, -1])))
    off, gaps, chain = [], [], []
    for b in full:
        qs = sorted(set(q for _, _, q in b))
        lanes = [[(s - b[0][0], e - b[0][0]) for s, e, qq in b if qq == q] for q in qs]
        off.append(max(l[0][0] for l in lanes) / 1e3)
        for l in lanes:
            gaps += [(l[i + 1][0] - l[i][1]) / 1e3 for i in range(len(l) - 1)]
            chain.append((l[-1][1] - l[0][0]) / 1e3)
    print("second lane's first launch starts %.1f us after the first lane's (median; %.1f .. %.1f)" % (np.median(off), min(off), max(off)))
    print("gap between consecutive launches of a lane, us: median %.2f  max %.2f" % (np.median(gaps), max(gaps)))
    print("a lane's chain of %d launches, us: median %.1f" % (per_block // 2, np.median(chain)))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
