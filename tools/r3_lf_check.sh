#!/bin/bash
# Round-3 iteration script for the single-wave fused path (THIP_FUSE=3) and the conditional stores of k_loopfilter.
# usage (on the GPU box, from the repo root): bash tools/r3_lf_check.sh [quick|full]    FUSES="0 3" CONTENTS="dense smooth"
mode=${1:-quick}
out=gpurun_out/r3lf
mkdir -p $out
export TMPDIR=/tmp
sel='sequence_small or lane_shared or enqueue or batched'
[ "$mode" = full ] && sel='(sequence or enqueue or batched or grey or dup or lane_shared or static_background or four_concurrent or dc_unprediction or beyond_4k or frame_calls) and not elision and not fused'
THIP_FUSE=3 timeout 1200 python -m pytest tests/test_gpu_frames.py -m gpu -x -q -k "$sel" > $out/pytest_fuse3.log 2>&1
echo "pytest fuse=3 rc=$?" | tee $out/summary.txt
tail -5 $out/pytest_fuse3.log | tee -a $out/summary.txt
for content in ${CONTENTS:-dense smooth}; do
  for fuse in ${FUSES:-0 3}; do
    THIP_FUSE=$fuse timeout 600 python bench.py --steps 256 --warmup 16 --content $content --second-content '' --no-cpu-baseline > $out/bench_${content}_fuse$fuse.json 2> $out/bench_${content}_fuse$fuse.err
    echo "bench $content fuse=$fuse rc=$?" | tee -a $out/summary.txt
    python - <<PY | tee -a $out/summary.txt
import json
try:
    d=json.loads(open("$out/bench_${content}_fuse$fuse.json").read().strip().splitlines()[-1])
    r=d.get("roofline",{})
    print("  fps", d["value"], "ms/step", d["ms_per_step"], "k1_us", r.get("avg_launch_us"), "k2_us", r.get("second_kernel_avg_launch_us"), "pipe", d["pipeline"]["read_roofline_frac"])
except Exception as e:
    print("  (no bench line)", e)
PY
  done
done
if [ -n "$ALT_LIB" ]; then
  for content in ${CONTENTS:-dense smooth}; do
    THIP_LIB=$ALT_LIB THIP_FUSE=0 timeout 600 python bench.py --steps 256 --warmup 16 --content $content --second-content '' --no-cpu-baseline > $out/bench_${content}_alt.json 2> $out/bench_${content}_alt.err
    python - <<PY | tee -a $out/summary.txt
import json
try:
    d=json.loads(open("$out/bench_${content}_alt.json").read().strip().splitlines()[-1])
    r=d.get("roofline",{})
    print("alt lib $content: fps", d["value"], "ms/step", d["ms_per_step"], "k1_us", r.get("avg_launch_us"), "k2_us", r.get("second_kernel_avg_launch_us"), "pipe", d["pipeline"]["read_roofline_frac"])
except Exception as e:
    print("  (no bench line)", e)
PY
  done
fi
