"""Copies what tools/prof_round.sh left under gpurun_out/r01b into profiles/ (the files profiles/README.md lists)
and recomputes profiles/r01_pmc_traffic.json from the two PMC passes."""
import csv, glob, json, os, shutil, sys

SRC = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r01b"
DST = "profiles"
for a, b in [("bench_default.json", "r01_bench_default.json"), ("bench_smooth.json", "r01_bench_smooth.json"),
             ("bench_1080p.json", "r01_bench_1080p.json"), ("bench_enc.jsonl", "r01_bench_enc.jsonl"),
             ("bench_e2e.jsonl", "r01_bench_e2e.jsonl"), ("bench_e2e_typical.jsonl", "r01_bench_e2e_typical.jsonl")]:
    p = os.path.join(SRC, a)
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, os.path.join(DST, b))
for d, b in [("stats_lanes1", "r01_4k_dense_lanes1_kernel_stats.csv"), ("stats_default", "r01_4k_dense_default_kernel_stats.csv")]:
    f = glob.glob(os.path.join(SRC, d, "**", "*kernel_stats.csv"), recursive=True)
    if f:
        shutil.copy(f[0], os.path.join(DST, b))


def per_kernel(dirname, counter):
    f = glob.glob(os.path.join(SRC, dirname, "**", "*counter_collection.csv"), recursive=True)
    acc = {}
    rows = []
    for row in csv.DictReader(open(f[0])):
        if row["Counter_Name"] != counter:
            continue
        k = row["Kernel_Name"].split("(")[0]
        acc.setdefault(k, []).append(float(row["Counter_Value"]))
        rows.append(row)
    return {k: sum(v) / len(v) for k, v in acc.items()}, rows


try:
    fetch, frows = per_kernel("pmc_fetch", "FETCH_SIZE")
    write, wrows = per_kernel("pmc_write", "WRITE_SIZE")
    out = {"workload": {"size": "4k", "content": "dense", "streams_per_launch": 4}}
    for k in ("k_recon", "k_loopfilter"):
        out[k] = {"fetch_kib_raw": fetch[k], "write_kib": write[k],
                  "hbm_bytes_per_launch": int(round((2 * fetch[k] + write[k]) * 1024))}
    out["method"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/prof_round.sh); averages over all "
                     "launches of the run; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of wide "
                     "coalesced reads); WRITE_SIZE as reported")
    json.dump(out, open(os.path.join(DST, "r01_pmc_traffic.json"), "w"), indent=1)
    with open(os.path.join(DST, "r01_4k_dense_lanes1_pmc.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(frows[0].keys()))
        w.writeheader()
        for r in frows + wrows:
            w.writerow(r)
    print(json.dumps({k: out[k] for k in ("k_recon", "k_loopfilter")}))
except Exception as e:   # PMC passes missing: keep the old files
    print("pmc not refreshed:", e)
