#!/bin/bash
# Refreshes the round-2 numbers kept under profiles/: bench lines, rocprofv3 kernel stats, FETCH / WRITE PMC passes
# (one counter per pass), for the default two-pass path and for the fused paths (THIP_FUSE=1: walk, THIP_FUSE=2: super tiles).
# usage (GPU box, repo root): bash tools/prof_round2.sh ; then here: python tools/collect_profiles2.py
export TMPDIR=/tmp
o=gpurun_out/r02
mkdir -p $o
python bench.py > $o/bench_default.json 2> $o/bench_default.err
THIP_FUSE=1 python bench.py --no-cpu-baseline > $o/bench_fused.json 2>/dev/null
THIP_FUSE=2 python bench.py --no-cpu-baseline > $o/bench_fused_st.json 2>/dev/null
python bench.py --size 1080p --streams-per-gpu 1 --no-cpu-baseline --second-content "" > $o/bench_1080p_single.json 2>/dev/null
python bench.py --size 1080p --streams-per-gpu 1 --gop-parallel 16 --no-cpu-baseline --parity-frames 70 > $o/bench_1080p_single_gop16.json 2>/dev/null
python bench.py --size 1080p --no-cpu-baseline --second-content "" > $o/bench_1080p_4streams.json 2>/dev/null
python bench.py --mode enc 2>/dev/null | grep '^{' > $o/bench_enc.jsonl
for fuse in ${FUSES:-0 1 2}; do
  THIP_FUSE=$fuse THIP_LANES=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_lanes1_fuse$fuse -- python bench.py --steps 64 --repeats 2 --min-time 0 --no-cpu-baseline --no-parity --no-profile --second-content "" > $o/stats_lanes1_fuse$fuse.log 2>&1
  THIP_FUSE=$fuse timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_default_fuse$fuse -- python bench.py --steps 64 --repeats 2 --min-time 0 --no-cpu-baseline --no-parity --no-profile --second-content "" > $o/stats_default_fuse$fuse.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    THIP_FUSE=$fuse THIP_LANES=1 timeout 300 rocprofv3 --pmc $c --output-format csv -d $o/pmc_${c}_fuse$fuse -- python bench.py --steps 24 --warmup 4 --repeats 1 --min-time 0 --no-cpu-baseline --no-parity --no-profile --second-content "" > $o/pmc_${c}_fuse$fuse.log 2>&1
    echo "pmc $c fuse=$fuse rc=$?"
  done
done
THIP_LANES=1 timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $o/pmc_tcc_fuse0 -- python bench.py --steps 24 --warmup 4 --repeats 1 --min-time 0 --no-cpu-baseline --no-parity --no-profile --second-content "" > $o/pmc_tcc_fuse0.log 2>&1
echo "pmc tcc rc=$?"
python tools/walk_trace.py --content dense 2>&1 | grep -v amdgpu.ids > $o/walk_trace_dense.txt
python tools/walk_trace.py --content smooth 2>&1 | grep -v amdgpu.ids > $o/walk_trace_smooth.txt
python tools/dc_wavefront_time.py 2>&1 | grep -v amdgpu.ids > $o/dc_wavefront_time.txt
tools/_build/valu_rate > $o/valu_rate.txt 2>&1
python bench.py --mode e2e --threads 1 --loops 10 2>/dev/null | grep '^{' | head -1 > $o/e2e_dense.jsonl
THIP_FE_DEVICE_TOKENS=1 python bench.py --mode e2e --threads 1 --loops 10 2>/dev/null | grep '^{' | head -1 > $o/e2e_dense_device_tokens.jsonl
THIP_FE_DEVICE_DC=1 python bench.py --mode e2e --threads 1 --loops 4 2>/dev/null | grep '^{' | head -1 > $o/e2e_dense_device_dc.jsonl
tail -c 600 $o/bench_default.json
