#!/bin/bash
# Refreshes the round-3 numbers kept under profiles/: bench lines, rocprofv3 kernel stats, FETCH / WRITE PMC passes
# (one counter per pass), for the default path (k_recon_lf, THIP_FUSE=3) and the two passes (THIP_FUSE=0), the wave
# timeline of k_recon_lf.   usage (GPU box, repo root): bash tools/prof_round3.sh ; then here: python tools/collect_profiles3.py
export TMPDIR=/tmp
o=gpurun_out/r03
mkdir -p $o
python bench.py > $o/bench_default.json 2> $o/bench_default.err
THIP_FUSE=0 python bench.py --no-cpu-baseline > $o/bench_twopass.json 2>/dev/null
python bench.py --size 1080p --streams-per-gpu 1 --no-cpu-baseline --second-content "" > $o/bench_1080p_single.json 2>/dev/null
python bench.py --size 1080p --streams-per-gpu 1 --gop-parallel 16 --no-cpu-baseline --parity-frames 70 > $o/bench_1080p_single_gop16.json 2>/dev/null
python bench.py --size 1080p --no-cpu-baseline --second-content "" > $o/bench_1080p_4streams.json 2>/dev/null
for fuse in ${FUSES:-3 0}; do
  THIP_FUSE=$fuse THIP_LANES=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_lanes1_fuse$fuse -- python bench.py --steps 64 --repeats 2 --min-time 0 --no-cpu-baseline --no-parity --no-profile --second-content "" > $o/stats_lanes1_fuse$fuse.log 2>&1
  THIP_FUSE=$fuse timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_default_fuse$fuse -- python bench.py --steps 64 --repeats 2 --min-time 0 --no-cpu-baseline --no-parity --no-profile --second-content "" > $o/stats_default_fuse$fuse.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    for content in dense smooth; do
      THIP_FUSE=$fuse THIP_LANES=1 timeout 300 rocprofv3 --pmc $c --output-format csv -d $o/pmc_${c}_${content}_fuse$fuse -- python bench.py --content $content --steps 24 --warmup 4 --repeats 1 --min-time 0 --no-cpu-baseline --no-parity --no-profile --second-content "" > $o/pmc_${c}_${content}_fuse$fuse.log 2>&1
      echo "pmc $c $content fuse=$fuse rc=$?"
    done
  done
done
# k_loopfilter with and without the stores of unchanged rows (tools/_build/libtheora_hip_alwaysstore.so: -DTHIP_LF_ALWAYS_STORE)
if [ -f tools/_build/libtheora_hip_alwaysstore.so ]; then
  for content in dense smooth; do
    THIP_LIB=tools/_build/libtheora_hip_alwaysstore.so THIP_FUSE=0 THIP_LANES=1 timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $o/pmc_WRITE_SIZE_${content}_fuse0_alwaysstore -- python bench.py --content $content --steps 24 --warmup 4 --repeats 1 --min-time 0 --no-cpu-baseline --no-parity --no-profile --second-content "" > $o/pmc_WRITE_SIZE_${content}_fuse0_alwaysstore.log 2>&1
  done
fi
python tools/lf_trace.py --content dense 2>&1 | grep -v amdgpu.ids > $o/lf_trace_dense.txt
python tools/lf_trace.py --content smooth 2>&1 | grep -v amdgpu.ids > $o/lf_trace_smooth.txt
# what the memory system gives: linear streams at several read:write mixes, and k_recon's access pattern (row-major and tiled frames)
{ for m in 60 200 400; do timeout 120 tools/_build/hbm_ceiling $m; done; timeout 120 tools/_build/tile_pattern; } > $o/hbm_ceiling.txt 2>&1
# the host side of th_decode_packetin by stage (option fe_prof)
for k in dense typical; do
  THIP_FE_PROF=1 timeout 300 python bench.py --mode e2e --e2e-size 720p --packets $k --no-native --loops 8 2>&1 | grep -v "^{\|amdgpu.ids" > $o/fe_stages_720p_$k.txt
done
tail -c 700 $o/bench_default.json
