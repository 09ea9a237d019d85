"""Copies what tools/prof_round4.sh left under gpurun_out/r04p into profiles/ (r04_*) and computes
profiles/r04_pmc_traffic.json from the single-counter PMC passes (FETCH_SIZE doubled per MI355X_MICROARCH.md)."""
import csv, glob, json, os, shutil, sys

SRC = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r04p"
DST = "profiles"
for name in ("bench_default.json", "bench_steps20.json", "bench_twopass.json", "bench_1080p_single.json", "bench_1080p_single_gop16.json",
             "bench_1080p_4streams.json", "bench_720p_single.json", "bench_720p_single_tilekernel.json", "lf_trace_dense.txt",
             "lf_trace_smooth.txt", "lf_trace_720p_single_sb.txt", "pmc_counters_dense_lanes2.txt", "pmc_counters_smooth_lanes2.txt",
             "e2e_sizes.jsonl", "e2e_single_stream_one_piece.jsonl"):
    p = os.path.join(SRC, name)
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, os.path.join(DST, "r04_" + name))
for d in ("stats_lanes1", "stats_default"):
    f = glob.glob(os.path.join(SRC, d, "**", "*kernel_stats.csv"), recursive=True)
    if f:
        shutil.copy(max(f, key=os.path.getmtime), os.path.join(DST, "r04_4k_dense_%s_kernel_stats.csv" % d.split("_")[1]))


def base(name):
    n = name.split("(")[0].strip()
    return (n[5:] if n.startswith("void ") else n).split("<")[0]


def per_kernel(dirname, counter):
    f = glob.glob(os.path.join(SRC, dirname, "**", "*counter_collection.csv"), recursive=True)
    acc = {}
    for row in csv.DictReader(open(max(f, key=os.path.getmtime))):
        if row["Counter_Name"] == counter:
            acc.setdefault(base(row["Kernel_Name"]), []).append(float(row["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


out = {"workload": {"size": "4k", "streams": 4},
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/prof_round4.sh); lanes1: one launch of four "
                 "streams per step, lanes2 (the timed shape): two launches of two streams per step; averages over all launches; FETCH_SIZE "
                 "doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of wide coalesced reads); WRITE_SIZE as reported; units KiB "
                 "in the raw fields"}
for content in ("dense", "smooth"):
    for lanes in (1, 2):
        label = "%s_lanes%d" % (content, lanes)
        try:
            fetch = per_kernel("pmc_FETCH_SIZE_%s_lanes%d" % (content, lanes), "FETCH_SIZE")
            write = per_kernel("pmc_WRITE_SIZE_%s_lanes%d" % (content, lanes), "WRITE_SIZE")
            out[label] = {k: {"fetch_kib_raw": fetch[k][0], "write_kib": write.get(k, (0.0, 0))[0], "launches": fetch[k][1],
                              "hbm_bytes_per_launch": int(round((2 * fetch[k][0] + write.get(k, (0.0, 0))[0]) * 1024))}
                          for k in fetch if k.startswith("k_")}
            out[label]["step_total_bytes"] = sum(v["hbm_bytes_per_launch"] for v in out[label].values()) * lanes
        except Exception as e:
            out[label] = {"error": str(e)}
json.dump(out, open(os.path.join(DST, "r04_pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
