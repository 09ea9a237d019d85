#!/bin/bash
# Refreshes the round-4 numbers kept under profiles/: bench lines, rocprofv3 kernel stats (default two-lane shape and one lane),
# FETCH / WRITE PMC passes (one counter per pass) for dense and smooth, VALU counters, wave timelines.
#   usage (GPU box, repo root): bash tools/prof_round4.sh ; then here: python tools/collect_profiles4.py
export TMPDIR=/tmp
o=gpurun_out/r04p
mkdir -p $o
python bench.py > $o/bench_default.json 2> $o/bench_default.err
python bench.py --steps 20 --no-cpu-baseline > $o/bench_steps20.json 2>/dev/null
THIP_FUSE=0 python bench.py --no-cpu-baseline --no-1080p > $o/bench_twopass.json 2>/dev/null
python bench.py --size 1080p --streams-per-gpu 1 --no-cpu-baseline --no-1080p --second-content "" > $o/bench_1080p_single.json 2>/dev/null
python bench.py --size 1080p --streams-per-gpu 1 --gop-parallel 16 --no-cpu-baseline --no-1080p --parity-frames 70 > $o/bench_1080p_single_gop16.json 2>/dev/null
python bench.py --size 1080p --no-cpu-baseline --no-1080p --second-content "" > $o/bench_1080p_4streams.json 2>/dev/null
python bench.py --size 720p --streams-per-gpu 1 --no-cpu-baseline --no-1080p --second-content "" > $o/bench_720p_single.json 2>/dev/null
THIP_SB_TILES=0 python bench.py --size 720p --streams-per-gpu 1 --no-cpu-baseline --no-1080p --second-content "" > $o/bench_720p_single_tilekernel.json 2>/dev/null
THIP_LANES=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_lanes1 -- python bench.py --steps 64 --repeats 2 --min-time 0 --no-cpu-baseline --no-parity --no-profile --no-pmc --no-1080p --second-content "" > $o/stats_lanes1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_default -- python bench.py --steps 64 --repeats 2 --min-time 0 --no-cpu-baseline --no-parity --no-profile --no-pmc --no-1080p --second-content "" > $o/stats_default.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  for content in dense smooth; do
    for lanes in 1 2; do
      THIP_LANES=$lanes timeout 300 rocprofv3 --pmc $c --output-format csv -d $o/pmc_${c}_${content}_lanes$lanes -- python bench.py --content $content --steps 24 --warmup 4 --repeats 1 --min-time 0 --no-cpu-baseline --no-parity --no-profile --no-pmc --no-1080p --second-content "" > $o/pmc_${c}_${content}_lanes$lanes.log 2>&1
      echo "pmc $c $content lanes=$lanes rc=$?"
    done
  done
done
LANES=2 bash tools/pmc_r4.sh dense > /dev/null 2>&1; cp gpurun_out/r04/pmc_dense.txt $o/pmc_counters_dense_lanes2.txt
LANES=2 bash tools/pmc_r4.sh smooth > /dev/null 2>&1; cp gpurun_out/r04/pmc_smooth.txt $o/pmc_counters_smooth_lanes2.txt
python tools/lf_trace.py --content dense 2>&1 | grep -v amdgpu.ids > $o/lf_trace_dense.txt
python tools/lf_trace.py --content smooth 2>&1 | grep -v amdgpu.ids > $o/lf_trace_smooth.txt
python tools/lf_trace.py --content dense --size 720p --streams 1 2>&1 | grep -v amdgpu.ids > $o/lf_trace_720p_single_sb.txt
: > $o/e2e_sizes.jsonl
for k in dense typical; do
  for sz in 720p 1080p 4k; do
    for t in 1 4 16; do
      timeout 600 python bench.py --mode e2e --e2e-size $sz --packets $k --threads $t --loops 4 --no-native 2>/dev/null | tail -1 >> $o/e2e_sizes.jsonl
    done
  done
done
# the same three single-stream cells with the round-3 shape of the token-list path (one piece after the packet, one thread)
: > $o/e2e_single_stream_one_piece.jsonl
for sz in 720p 1080p 4k; do
  THIP_FE_GROUPS=1 THIP_FE_WORKER=0 timeout 600 python bench.py --mode e2e --e2e-size $sz --packets dense --threads 1 --loops 4 --no-native 2>/dev/null | tail -1 >> $o/e2e_single_stream_one_piece.jsonl
done
tail -c 600 $o/bench_default.json
