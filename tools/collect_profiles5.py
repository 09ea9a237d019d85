"""Copies what tools/prof_round5.sh left under gpurun_out/r05_* into profiles/ (r05_*): the `final` section's bench lines,
rocprofv3 kernel stats, counter passes and wave timelines; the experiment sections' tables as they are."""
import glob, os, shutil, sys

DST = "profiles"
copied = []


def cp(src, name):
    if os.path.exists(src) and os.path.getsize(src):
        shutil.copy(src, os.path.join(DST, "r05_" + name))
        copied.append("r05_" + name)


F = "gpurun_out/r05_final"
for name in ("bench_default.json", "bench_steps20.json", "bench_twopass.json", "bench_enc.jsonl", "pytest_gpu.txt", "pmc_counters_dense_lanes2.txt",
             "pmc_counters_smooth_lanes2.txt", "lf_trace_dense.txt", "lf_trace_smooth.txt", "torchrun_world1.log"):
    cp(os.path.join(F, name), name)
for d in ("stats_lanes1", "stats_default"):
    f = glob.glob(os.path.join(F, d, "**", "*kernel_stats.csv"), recursive=True)
    if f:
        cp(max(f, key=os.path.getmtime), "4k_dense_%s_kernel_stats.csv" % d.split("_")[1])
f = glob.glob(os.path.join(F, "stats_enc", "**", "*kernel_stats.csv"), recursive=True)
if f:
    cp(max(f, key=os.path.getmtime), "enc_kernel_stats.csv")
for sec, names in (("ab", ("ab256.txt", "ab20.txt")), ("ends", ("block_ends.txt",)), ("trace", ("block_timeline_dense.csv", "block_timeline_smooth.csv")),
                   ("e2e", ("native_1stream.jsonl", "native_4streams.jsonl", "stage_tables.txt"))):
    for n in names:
        cp(os.path.join("gpurun_out/r05_" + sec, n), ("ab_" if sec == "ab" and not n.startswith("ab") else "") + n)
for extra in sys.argv[1:]:      # older session directories of the round: "dir/file=name"
    src, name = extra.split("=")
    cp(src, name)
print("\n".join(copied))
