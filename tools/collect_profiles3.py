"""Copies what tools/prof_round3.sh left under gpurun_out/r03 into profiles/ (r03_*) and computes
profiles/r03_pmc_traffic.json from the single-counter PMC passes (FETCH_SIZE doubled per MI355X_MICROARCH.md)."""
import csv, glob, json, os, shutil, sys

SRC = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r03"
DST = "profiles"
for name in ("bench_default.json", "bench_twopass.json", "bench_1080p_single.json", "bench_1080p_single_gop16.json",
             "bench_1080p_4streams.json", "lf_trace_dense.txt", "lf_trace_smooth.txt", "hbm_ceiling.txt", "fe_stages_720p_dense.txt",
             "fe_stages_720p_typical.txt"):
    p = os.path.join(SRC, name)
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, os.path.join(DST, "r03_" + name))
LABEL = {3: "fused", 0: "twopass"}
for fuse in (3, 0):
    for d in ("stats_lanes1", "stats_default"):
        f = glob.glob(os.path.join(SRC, "%s_fuse%d" % (d, fuse), "**", "*kernel_stats.csv"), recursive=True)
        if f:   # (gpurun_out/ keeps the files of earlier runs: the newest one)
            shutil.copy(max(f, key=os.path.getmtime), os.path.join(DST, "r03_4k_dense_%s_%s_kernel_stats.csv" % (d.split("_")[1], LABEL[fuse])))


def per_kernel(dirname, counter):
    f = glob.glob(os.path.join(SRC, dirname, "**", "*counter_collection.csv"), recursive=True)
    acc = {}
    for row in csv.DictReader(open(max(f, key=os.path.getmtime))):
        if row["Counter_Name"] != counter:
            continue
        acc.setdefault(row["Kernel_Name"].split("(")[0], []).append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


out = {"workload": {"size": "4k", "streams_per_launch": 4},
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/prof_round3.sh, THIP_LANES=1 so every "
                 "launch has the 4-stream shape); averages over all launches; FETCH_SIZE doubled per MI355X_MICROARCH.md "
                 "(gfx950 reports half the bytes of wide coalesced reads); WRITE_SIZE as reported; units KiB in the raw fields"}
for content in ("dense", "smooth"):
    for fuse in (3, 0):
        label = "%s_%s" % (content, LABEL[fuse])
        try:
            fetch = per_kernel("pmc_FETCH_SIZE_%s_fuse%d" % (content, fuse), "FETCH_SIZE")
            write = per_kernel("pmc_WRITE_SIZE_%s_fuse%d" % (content, fuse), "WRITE_SIZE")
            out[label] = {k: {"fetch_kib_raw": fetch[k], "write_kib": write.get(k, 0.0),
                              "hbm_bytes_per_launch": int(round((2 * fetch[k] + write.get(k, 0.0)) * 1024))}
                          for k in fetch if k.startswith("k_")}
            out[label]["step_total_bytes"] = sum(v["hbm_bytes_per_launch"] for v in out[label].values())
        except Exception as e:
            out[label] = {"error": str(e)}
    try:
        w = per_kernel("pmc_WRITE_SIZE_%s_fuse0_alwaysstore" % content, "WRITE_SIZE")
        out["%s_twopass_k_loopfilter_write_kib_when_every_row_is_stored" % content] = w.get("k_loopfilter")
    except Exception as e:
        pass
json.dump(out, open(os.path.join(DST, "r03_pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:4000])
