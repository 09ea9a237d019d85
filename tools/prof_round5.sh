#!/bin/bash
# Round 5 on the GPU box, one script with sections (run from the repository root, e.g. through gpurun):
#   bash tools/prof_round5.sh final     the numbers kept under profiles/r05_* (then here: python tools/collect_profiles5.py)
#   bash tools/prof_round5.sh ab        A/B of differently compiled copies of the library under tools/_build/ab/ (tools/ab.sh)
#   bash tools/prof_round5.sh ends      what a 20-step block loses at its ends: runtime wait modes, lanes, streams per launch
#   bash tools/prof_round5.sh trace     rocprofv3 kernel timeline of 20-step blocks (start / end of every k_recon_lf launch)
#   bash tools/prof_round5.sh e2e       th_decode_* end to end from C (examples/decode_bench): look-ahead 0 / 8 / 16, fe_pipeline
export TMPDIR=/tmp
what=${1:-final}
o=gpurun_out/r05_$what; mkdir -p $o
B="python bench.py --content dense --second-content '' --no-cpu-baseline --parity-frames 4 --no-1080p --no-e2e --no-pmc --no-wide --no-enc"
show() { python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['timing']; print('%-26s' % '$1', d['value'], d['ms_per_step'], d['pipeline']['read_roofline_frac'], 'min', t['ms_per_step_min'], 'max', t['ms_per_step_max'], 'submit_us', t['host_submit_us_per_block'], 'idle_sync_us', t['idle_sync_us'])"; }
case $what in
ab)
  L=tools/_build/ab
  args=""; for f in $L/*.so; do args="$args $(basename $f .so)=$f"; done
  bash tools/ab.sh $args > $o/ab256.txt 2>&1
  AB_STEPS=20 bash tools/ab.sh base=$L/base.so cand1=$L/cand1.so > $o/ab20.txt 2>&1
  cat $o/ab256.txt $o/ab20.txt ;;
ends)
  for round in 1 2; do
    bash -c "$B --steps 20" 2>/dev/null | show steps20_default
    ROC_ACTIVE_WAIT_TIMEOUT=100000 bash -c "$B --steps 20" 2>/dev/null | show steps20_active_wait
    THIP_LANES=3 bash -c "$B --steps 20" 2>/dev/null | show steps20_lanes3
    HIP_FORCE_DEV_KERNARG=1 bash -c "$B --steps 20" 2>/dev/null | show steps20_dev_kernarg
    THIP_CHUNK=1 bash -c "$B --steps 20" 2>/dev/null | show steps20_one_stream_per_launch
    bash -c "$B --steps 256" 2>/dev/null | show steps256_default
  done 2>&1 | tee $o/block_ends.txt ;;
trace)
  cd /tmp
  for c in dense smooth; do
    timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$o/trace_$c -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --repeats 40 --min-time 0 --content $c --second-content '' --no-cpu-baseline --no-parity --no-profile --no-1080p --no-e2e --no-pmc --no-wide --no-enc > $GRAFT_REPO_ROOT/$o/trace_$c.log 2>&1
  done
  cd $GRAFT_REPO_ROOT
  for c in dense smooth; do
    f=$(find $o/trace_$c -name "*kernel_trace.csv" | head -1)
    python - "$f" $o/block_timeline_$c.csv <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_recon_lf" in r.get("Kernel_Name", "")]
with open(sys.argv[2], "w") as f:
    f.write("queue,start_ns,end_ns\n")
    for r in rows:
        f.write("%s,%s,%s\n" % (r.get("Queue_Id", ""), r["Start_Timestamp"], r["End_Timestamp"]))
PY
    rm -rf $o/trace_$c
  done ;;
e2e)
  E2E_LOOPS=6 python tools/native_lookahead.py 720p,1080p,4k dense 1 0,8,16 0,1 > $o/native_1stream.jsonl 2>$o/native.err
  E2E_LOOPS=6 python tools/native_lookahead.py 720p,4k dense 4 0,8 0,1 > $o/native_4streams.jsonl 2>>$o/native.err
  python tools/make_clip720.py > /dev/null 2>&1
  for la in 8 16; do for pipe in "" "--pipeline"; do
    echo "== 720p lookahead $la $pipe" >> $o/stage_tables.txt
    THIP_FE_PROF=1 examples/decode_bench gpurun_out/clip720.ogv 1 3 --lookahead $la $pipe >> $o/stage_tables.txt 2>&1
  done; done
  cat $o/native_1stream.jsonl $o/native_4streams.jsonl | cut -c1-250 ;;
final)
  timeout 2400 python -m pytest tests -m gpu -q > $o/pytest_gpu.txt 2>&1; tail -3 $o/pytest_gpu.txt
  python bench.py > $o/bench_default.json 2> $o/bench_default.err
  python bench.py --steps 20 > $o/bench_steps20.json 2>/dev/null
  THIP_FUSE=0 python bench.py --no-cpu-baseline --no-1080p --no-e2e --no-wide --no-enc > $o/bench_twopass.json 2>/dev/null
  python bench.py --mode enc > $o/bench_enc.jsonl 2>/dev/null
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_enc -- python bench.py --mode enc > $o/bench_enc_under_rocprof.jsonl 2>$o/stats_enc.log
  THIP_LANES=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_lanes1 -- python bench.py --steps 64 --repeats 2 --min-time 0 --no-cpu-baseline --no-parity --no-profile --no-pmc --no-1080p --no-e2e --no-wide --no-enc --second-content "" > $o/stats_lanes1.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_default -- python bench.py --steps 64 --repeats 2 --min-time 0 --no-cpu-baseline --no-parity --no-profile --no-pmc --no-1080p --no-e2e --no-wide --no-enc --second-content "" > $o/stats_default.log 2>&1
  LANES=2 bash tools/pmc_r4.sh dense > /dev/null 2>&1; cp gpurun_out/r04/pmc_dense.txt $o/pmc_counters_dense_lanes2.txt 2>/dev/null
  LANES=2 bash tools/pmc_r4.sh smooth > /dev/null 2>&1; cp gpurun_out/r04/pmc_smooth.txt $o/pmc_counters_smooth_lanes2.txt 2>/dev/null
  python tools/lf_trace.py --content dense 2>&1 | grep -v amdgpu.ids > $o/lf_trace_dense.txt
  python tools/lf_trace.py --content smooth 2>&1 | grep -v amdgpu.ids > $o/lf_trace_smooth.txt
  MASTER_ADDR=127.0.0.1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --no-1080p --no-e2e --no-wide --no-enc --no-pmc --cpu-frames 32 > $o/torchrun_world1.log 2>&1
  tail -c 400 $o/bench_default.json ;;
esac
