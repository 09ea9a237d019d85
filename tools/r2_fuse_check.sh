#!/bin/bash
# Round-2 iteration script: parity of the row-walking fused path on a subset, then bench lines of both paths.
# usage (on the GPU box, from the repo root): bash tools/r2_row_check.sh [quick|full]
mode=${1:-quick}
out=gpurun_out/r2row
mkdir -p $out
export TMPDIR=/tmp
sel='sequence_small or lane_shared or enqueue or batched'
[ "$mode" = full ] && sel='(sequence or enqueue or batched or grey or dup or lane_shared or static_background) and not elision and not fused and not fused_walk'
THIP_FUSE=1 timeout 900 python -m pytest tests/test_gpu_frames.py -m gpu -x -q -k "$sel" > $out/pytest_fuse2.log 2>&1
echo "pytest fuse2 rc=$?" | tee $out/summary.txt
tail -5 $out/pytest_fuse2.log | tee -a $out/summary.txt
for content in dense smooth; do
  for fuse in ${FUSES:-0 1}; do
    THIP_FUSE=$fuse timeout 600 python bench.py --steps 256 --warmup 16 --content $content --no-cpu-baseline > $out/bench_${content}_fuse$fuse.json 2> $out/bench_${content}_fuse$fuse.err
    echo "bench $content fuse=$fuse rc=$?" | tee -a $out/summary.txt
    python - <<PY | tee -a $out/summary.txt
import json
try:
    d=json.loads(open("$out/bench_${content}_fuse$fuse.json").read().strip().splitlines()[-1])
    r=d.get("roofline",{})
    print("  fps", d["value"], "ms/step", d["ms_per_step"], "recon_us", r.get("avg_launch_us"), "lf_us", r.get("loopfilter_avg_launch_us"), "pipe", d["pipeline"]["read_roofline_frac"])
except Exception as e:
    print("  (no bench line)", e)
PY
  done
done
