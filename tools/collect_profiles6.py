"""Copies what tools/prof_round6.sh left under gpurun_out/r06_* into profiles/ (r06_*): the `final` section's bench lines and detail
files, rocprofv3 kernel stats, counter passes; the experiment sections' tables as they are."""
import glob, os, shutil, sys

DST = "profiles"
copied = []


def cp(src, name):
    if os.path.exists(src) and os.path.getsize(src):
        shutil.copy(src, os.path.join(DST, "r06_" + name))
        copied.append("r06_" + name)


F = "gpurun_out/r06_final"
for name in ("bench_default.json", "bench_detail.json", "bench_steps20.json", "bench_detail_steps20.json", "bench_twopass.json", "bench_enc.jsonl",
             "pytest_gpu.txt", "pmc_counters_dense_lanes2.txt", "pmc_counters_smooth_lanes2.txt", "torchrun_world1.log", "native_1stream.jsonl",
             "native_4streams.jsonl"):
    cp(os.path.join(F, name), name)
for d, name in (("stats_lanes1", "4k_dense_lanes1_kernel_stats.csv"), ("stats_default", "4k_dense_default_kernel_stats.csv"),
                ("stats_int16_lanes1", "4k_dense_int16_lanes1_kernel_stats.csv"), ("stats_1080p_single", "1080p_single_kernel_stats.csv"),
                ("stats_enc", "enc_kernel_stats.csv")):
    f = glob.glob(os.path.join(F, d, "**", "*kernel_stats.csv"), recursive=True)
    if f:
        cp(max(f, key=os.path.getmtime), name)
for sec, names in (("ab1", (("ab_spec_int16.txt", "ab_spec_coeffs_and_int16_form.txt"), ("bench_default.json", "bench_first_run_of_the_new_line.json"))),
                   ("ab2", (("ab_pitch.txt", "ab_image_pitch_152.txt"), ("ab_pitch20.txt", "ab_image_pitch_152_steps20.txt"),
                            ("native_1stream.jsonl", "native_1stream_pipeline_default_on.jsonl"), ("pytest_frontend.txt", "pytest_gpu_frontend_take_back.txt"))),
                   ("ab3", (("plain_loop_groups.txt", "plain_loop_groups.txt"),))):
    for src, name in names:
        cp(os.path.join("gpurun_out/r06_" + sec, src), name)
for extra in sys.argv[1:]:      # other session directories of the round: "dir/file=name"
    src, name = extra.split("=")
    cp(src, name)
print("\n".join(copied))
