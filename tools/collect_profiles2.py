"""Copies what tools/prof_round2.sh left under gpurun_out/r02 into profiles/ (r02_*) and computes
profiles/r02_pmc_traffic.json from the single-counter PMC passes (FETCH_SIZE doubled per MI355X_MICROARCH.md)."""
import csv, glob, json, os, shutil, sys

SRC = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r02"
DST = "profiles"
for name in ("bench_default.json", "bench_fused.json", "bench_fused_st.json", "bench_1080p_single.json", "bench_1080p_single_gop16.json",
             "bench_1080p_4streams.json", "bench_enc.jsonl", "walk_trace_dense.txt", "walk_trace_smooth.txt",
             "dc_wavefront_time.txt", "valu_rate.txt", "e2e_dense.jsonl", "e2e_dense_device_tokens.jsonl",
             "e2e_dense_device_dc.jsonl"):
    p = os.path.join(SRC, name)
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, os.path.join(DST, "r02_" + name))
for fuse in (0, 1, 2):
    for d in ("stats_lanes1", "stats_default"):
        f = glob.glob(os.path.join(SRC, "%s_fuse%d" % (d, fuse), "**", "*kernel_stats.csv"), recursive=True)
        if f:
            shutil.copy(f[0], os.path.join(DST, "r02_4k_dense_%s_%s_kernel_stats.csv" % (d.split("_")[1], ("twopass", "fused", "fused_st")[fuse])))


def per_kernel(dirname, counter):
    f = glob.glob(os.path.join(SRC, dirname, "**", "*counter_collection.csv"), recursive=True)
    acc = {}
    for row in csv.DictReader(open(f[0])):
        if row["Counter_Name"] != counter:
            continue
        acc.setdefault(row["Kernel_Name"].split("(")[0], []).append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


out = {"workload": {"size": "4k", "content": "dense", "streams_per_launch": 4},
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/prof_round2.sh, THIP_LANES=1 so every "
                 "launch has the 4-stream shape); averages over all launches; FETCH_SIZE doubled per MI355X_MICROARCH.md "
                 "(gfx950 reports half the bytes of wide coalesced reads); WRITE_SIZE as reported; units KiB in the raw fields"}
for fuse, label in ((0, "twopass"), (1, "fused"), (2, "fused_st")):
    try:
        fetch = per_kernel("pmc_FETCH_SIZE_fuse%d" % fuse, "FETCH_SIZE")
        write = per_kernel("pmc_WRITE_SIZE_fuse%d" % fuse, "WRITE_SIZE")
        out[label] = {k: {"fetch_kib_raw": fetch[k], "write_kib": write.get(k, 0.0),
                          "hbm_bytes_per_launch": int(round((2 * fetch[k] + write.get(k, 0.0)) * 1024))}
                      for k in fetch if k.startswith("k_")}
    except Exception as e:
        out[label] = {"error": str(e)}
try:
    hit, miss = per_kernel("pmc_tcc_fuse0", "TCC_HIT_sum"), per_kernel("pmc_tcc_fuse0", "TCC_MISS_sum")
    out["l2_hit_rate_twopass"] = {k: round(hit[k] / (hit[k] + miss[k]), 4) for k in hit if k.startswith("k_")}
except Exception as e:
    out["l2_hit_rate_twopass"] = {"error": str(e)}
json.dump(out, open(os.path.join(DST, "r02_pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
