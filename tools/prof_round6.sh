#!/bin/bash
# Round 6 on the GPU box, one script with sections (run from the repository root, e.g. through gpurun):
#   bash tools/prof_round6.sh ab1      parity of the kernel changes, the new bench line, A/B of spec_coeffs by shape, the int16 form
#   bash tools/prof_round6.sh final    the numbers kept under profiles/r06_* (then here: python tools/collect_profiles6.py)
export TMPDIR=/tmp
what=${1:-final}
o=gpurun_out/r06_$what; mkdir -p $o
Q="--second-content '' --no-cpu-baseline --parity-frames 4 --no-1080p --no-e2e --no-pmc --no-wide --no-enc --no-form16"
show() { python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s' % '$1', d['value'], d['ms_per_step'], d['pipeline']['read_roofline_frac'], 'lone_us', (d.get('roofline') or {}).get('avg_launch_us'))
except Exception as e:
    print('%-34s' % '$1', 'FAILED', e)"; }
case $what in
ab1)
  timeout 1500 python -m pytest tests/test_gpu_slots.py tests/test_gpu_costmaps.py -m gpu -q > $o/pytest_kernels.txt 2>&1; tail -3 $o/pytest_kernels.txt
  python bench.py --detail $o/bench_detail.json > $o/bench_default.json 2> $o/bench_default.err; tail -c 600 $o/bench_default.err; wc -c $o/bench_default.json
  for round in 1 2; do
    for spec in 0 1; do
      THIP_SPEC_COEFFS=$spec bash -c "python bench.py --size 1080p --streams-per-gpu 1 --steps 256 $Q" 2>/dev/null | show "1080p_1stream spec=$spec"
      THIP_SPEC_COEFFS=$spec bash -c "python bench.py --size 1080p --streams-per-gpu 4 --steps 256 $Q" 2>/dev/null | show "1080p_4streams spec=$spec"
      THIP_SPEC_COEFFS=$spec bash -c "python bench.py --size 720p --streams-per-gpu 1 --steps 256 $Q" 2>/dev/null | show "720p_1stream spec=$spec"
      THIP_SPEC_COEFFS=$spec THIP_SB_TILES=0 bash -c "python bench.py --size 720p --streams-per-gpu 1 --steps 256 $Q" 2>/dev/null | show "720p_1stream tilekernel spec=$spec"
      THIP_SPEC_COEFFS=$spec bash -c "python bench.py --steps 256 $Q" 2>/dev/null | show "4k_dense spec=$spec"
      THIP_SPEC_COEFFS=$spec bash -c "python bench.py --steps 20 $Q" 2>/dev/null | show "4k_dense steps20 spec=$spec"
      THIP_SPEC_COEFFS=$spec bash -c "python bench.py --steps 256 --content smooth $Q" 2>/dev/null | show "4k_smooth spec=$spec"
    done
    bash -c "python bench.py --form dequant16 --steps 256 $Q" 2>/dev/null | show "4k_dense int16 (no scratch)"
    THIP_LIB=tools/_build/ab/int16fused.so bash -c "python bench.py --form dequant16 --steps 256 $Q" 2>/dev/null | show "4k_dense int16 fused cols (scratch)"
    bash -c "python bench.py --form dequant16 --steps 256 --content smooth $Q" 2>/dev/null | show "4k_smooth int16 (no scratch)"
    THIP_LIB=tools/_build/ab/int16fused.so bash -c "python bench.py --form dequant16 --steps 256 --content smooth $Q" 2>/dev/null | show "4k_smooth int16 fused cols (scratch)"
  done 2>&1 | tee $o/ab_spec_int16.txt ;;
ab2)
  timeout 1500 python -m pytest tests/test_gpu_frontend.py -m gpu -q -x > $o/pytest_frontend.txt 2>&1; tail -3 $o/pytest_frontend.txt
  python bench.py --mode enc > $o/bench_enc.jsonl 2> $o/bench_enc.err; tail -5 $o/bench_enc.err; cut -c1-200 $o/bench_enc.jsonl
  python bench.py --detail $o/bench_detail.json > $o/bench_default.json 2> $o/bench_default.err; tail -c 1500 $o/bench_default.err; wc -c $o/bench_default.json
  python bench.py --steps 20 --detail $o/bench_detail20.json > $o/bench_steps20.json 2> /dev/null; wc -c $o/bench_steps20.json
  AB_ROUNDS=2 bash tools/ab.sh base=tools/_build/ab/base.so pitch152=tools/_build/ab/pitch152.so 2>&1 | tee $o/ab_pitch.txt
  AB_ROUNDS=2 AB_STEPS=20 bash tools/ab.sh base=tools/_build/ab/base.so pitch152=tools/_build/ab/pitch152.so 2>&1 | tee $o/ab_pitch20.txt
  E2E_LOOPS=6 python tools/native_lookahead.py 720p,1080p dense 1 0,8 0,1 > $o/native_1stream.jsonl 2>$o/native.err; cut -c1-250 $o/native_1stream.jsonl ;;
ab3)
  timeout 1500 python -m pytest tests/test_gpu_frames.py tests/test_gpu_levels.py -m gpu -q -x > $o/pytest_kernels.txt 2>&1; tail -3 $o/pytest_kernels.txt
  python bench.py --mode enc > $o/bench_enc.jsonl 2> $o/bench_enc.err; tail -5 $o/bench_enc.err
  python - $o/bench_enc.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    print("%-22s %9.1f %-22s %7.2f us  hbm %.3f" % (d["key"], d["value"], d["unit"], 1e3 * d["ms_per_call"], d["roofline"]["frac"]))
PY
  for round in 1 2; do for g in 4 6 7; do
    echo "== fe_groups $g (round $round)"
    THIP_FE_GROUPS=$g E2E_LOOPS=6 python tools/native_lookahead.py 720p,1080p dense 1 0 0 2>>$o/native.err | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   ', d.get('size_name'), d.get('frames_per_s'))"
  done; done 2>&1 | tee $o/plain_loop_groups.txt ;;
ab4)
  timeout 1800 python -m pytest tests/test_gpu_frontend.py -m gpu -q -x > $o/pytest_frontend.txt 2>&1; tail -3 $o/pytest_frontend.txt
  for round in 1 2; do for t in 0 1; do
    echo "== fe_pair_tail $t (round $round)"
    THIP_FE_PAIR_TAIL=$t E2E_LOOPS=6 python tools/native_lookahead.py 720p,1080p,4k dense,typical 1 0 0 2>>$o/native.err | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   ', d.get('size_name'), d.get('packets'), d.get('frames_per_s'))"
  done; done 2>&1 | tee $o/plain_loop_pair_tail.txt
  python tools/make_clip720.py > /dev/null 2>&1
  for t in 0 1; do echo "== stage table 720p plain loop, fe_pair_tail $t" >> $o/stage_tables.txt; THIP_FE_PAIR_TAIL=$t THIP_FE_PROF=1 examples/decode_bench gpurun_out/clip720.ogv 1 3 >> $o/stage_tables.txt 2>&1; done
  tail -60 $o/stage_tables.txt ;;
ab5)
  timeout 1500 python -m pytest tests/test_gpu_slots.py tests/test_gpu_frames.py -m gpu -q > $o/pytest_slots_frames.txt 2>&1; tail -5 $o/pytest_slots_frames.txt
  for round in 1 2; do for l in 1 0; do echo "== enc_sites_lds $l (round $round)"; THIP_ENC_SITES_LDS=$l python bench.py --mode enc 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    if 'satd_search' in d['key'] or d['key'] in ('fdct', 'cost_maps'): print('   %-18s %9.1f M/s %7.2f us' % (d['key'], d['value'], 1e3 * d['ms_per_call']))"
  done; done 2>&1 | tee $o/enc_sites_lds.txt ;;
ab6)
  timeout 600 python -m pytest tests/test_gpu_slots.py -m gpu -q -k "fdct" 2>&1 | tail -2
  for round in 1 2 3; do for lib in theora_amd/libtheora_hip.so tools/_build/ab/fdct4_lds_zz.so; do echo "== $lib (round $round)"; THIP_LIB=$lib python bench.py --mode enc 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    if d['key'] in ('fdct',): print('   %-18s %9.1f M/s %7.2f us' % (d['key'], d['value'], 1e3 * d['ms_per_call']))"
  done; done 2>&1 | tee $o/fdct4_barrier.txt ;;
half)
  timeout 2400 python -m pytest tests/test_gpu_zz_launch_variants.py -m gpu -q -x -k half > $o/pytest_half.txt 2>&1; tail -25 $o/pytest_half.txt
  for round in 1 2; do
    for ht in 0 1600; do
      THIP_HALF_TILES=$ht bash -c "python bench.py --size 1080p --streams-per-gpu 1 --steps 256 $Q" 2>/dev/null | show "1080p_1stream half_tiles=$ht"
      THIP_HALF_TILES=$ht bash -c "python bench.py --size 720p --streams-per-gpu 2 --steps 256 $Q" 2>/dev/null | show "720p_2streams half_tiles=$ht"
    done
    THIP_HALF_TILES=4000 bash -c "python bench.py --size 1080p --streams-per-gpu 2 --steps 256 $Q" 2>/dev/null | show "1080p_2streams half_tiles=4000"
    THIP_HALF_TILES=0 bash -c "python bench.py --size 1080p --streams-per-gpu 2 --steps 256 $Q" 2>/dev/null | show "1080p_2streams half_tiles=0"
    THIP_HALF_TILES=4000 bash -c "python bench.py --size 1080p --streams-per-gpu 4 --steps 256 $Q" 2>/dev/null | show "1080p_4streams half_tiles=4000"
    THIP_HALF_TILES=0 bash -c "python bench.py --size 1080p --streams-per-gpu 4 --steps 256 $Q" 2>/dev/null | show "1080p_4streams half_tiles=0"
    THIP_SB_TILES=0 THIP_HALF_TILES=600 bash -c "python bench.py --size 720p --streams-per-gpu 1 --steps 256 $Q" 2>/dev/null | show "720p_1stream half instead of sb"
    bash -c "python bench.py --size 720p --streams-per-gpu 1 --steps 256 $Q" 2>/dev/null | show "720p_1stream sb"
  done 2>&1 | tee $o/ab_half.txt ;;
spec2)
  timeout 1500 python -m pytest tests/test_gpu_frames.py tests/test_gpu_levels.py -m gpu -q -x > $o/pytest_kernels.txt 2>&1; tail -3 $o/pytest_kernels.txt
  for round in 1 2 3; do
    for spec in 0 1; do
      THIP_SPEC_COEFFS=$spec bash -c "python bench.py --size 1080p --streams-per-gpu 1 --steps 256 $Q" 2>/dev/null | show "1080p_1stream spec=$spec"
      THIP_SPEC_COEFFS=$spec bash -c "python bench.py --size 1080p --streams-per-gpu 4 --steps 256 $Q" 2>/dev/null | show "1080p_4streams spec=$spec"
      THIP_SPEC_COEFFS=$spec bash -c "python bench.py --steps 256 $Q" 2>/dev/null | show "4k_dense spec=$spec"
      THIP_SPEC_COEFFS=$spec bash -c "python bench.py --steps 20 $Q" 2>/dev/null | show "4k_dense steps20 spec=$spec"
    done
  done 2>&1 | tee $o/ab_spec2.txt
  python tools/lf_trace.py --size 1080p --streams 1 2>&1 | grep -v amdgpu.ids > $o/lf_trace_1080p_single.txt; head -13 $o/lf_trace_1080p_single.txt ;;
soak)
  timeout 400 python tests/soak_take_back.py 7 150 2>&1 | tail -3 | tee $o/soak_take_back.txt
  timeout 400 python tests/soak_frontend.py 11 120 2>&1 | tail -2 | tee $o/soak_frontend.txt
  python bench.py --detail $o/bench_detail.json > $o/bench_default.json 2> $o/bench_default.err; tail -c 300 $o/bench_default.json ;;
final)
  Q2="--no-cpu-baseline --no-parity --no-profile --no-pmc --no-1080p --no-e2e --no-wide --no-enc --no-form16 --second-content ''"
  timeout 2400 python -m pytest tests -m gpu -q > $o/pytest_gpu.txt 2>&1; tail -3 $o/pytest_gpu.txt
  python bench.py --detail $o/bench_detail.json > $o/bench_default.json 2> $o/bench_default.err; wc -c $o/bench_default.json
  python bench.py --steps 20 --detail $o/bench_detail_steps20.json > $o/bench_steps20.json 2>/dev/null; wc -c $o/bench_steps20.json
  THIP_FUSE=0 python bench.py --no-cpu-baseline --no-1080p --no-e2e --no-wide --no-enc --no-form16 --detail $o/bench_detail_twopass.json > $o/bench_twopass.json 2>/dev/null
  python bench.py --mode enc > $o/bench_enc.jsonl 2>$o/bench_enc.err
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_enc -- python bench.py --mode enc > $o/bench_enc_under_rocprof.jsonl 2>$o/stats_enc.log
  THIP_LANES=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_lanes1 -- bash -c "python bench.py --steps 64 --repeats 2 --min-time 0 $Q2" > $o/stats_lanes1.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_default -- bash -c "python bench.py --steps 64 --repeats 2 --min-time 0 $Q2" > $o/stats_default.log 2>&1
  THIP_LANES=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_int16_lanes1 -- bash -c "python bench.py --form dequant16 --steps 64 --repeats 2 --min-time 0 $Q2" > $o/stats_int16_lanes1.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_1080p_single -- bash -c "python bench.py --size 1080p --streams-per-gpu 1 --steps 128 --repeats 2 --min-time 0 $Q2" > $o/stats_1080p_single.log 2>&1
  LANES=2 bash tools/pmc_r4.sh dense > /dev/null 2>&1; cp gpurun_out/r04/pmc_dense.txt $o/pmc_counters_dense_lanes2.txt 2>/dev/null
  LANES=2 bash tools/pmc_r4.sh smooth > /dev/null 2>&1; cp gpurun_out/r04/pmc_smooth.txt $o/pmc_counters_smooth_lanes2.txt 2>/dev/null
  MASTER_ADDR=127.0.0.1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --no-1080p --no-e2e --no-wide --no-enc --no-form16 --no-pmc --cpu-frames 32 --detail $o/bench_detail_torchrun.json > $o/torchrun_world1.log 2>&1
  E2E_LOOPS=6 python tools/native_lookahead.py 720p,1080p,4k dense 1 0,8,16 0,1 > $o/native_1stream.jsonl 2>$o/native.err
  E2E_LOOPS=6 python tools/native_lookahead.py 720p,4k dense 4 0,8 0,1 > $o/native_4streams.jsonl 2>>$o/native.err
  tail -c 600 $o/bench_default.json ;;
esac
