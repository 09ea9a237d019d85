#!/bin/bash
# One GPU session of round 5 (run through gpurun from the repository root): what a 20-step block loses at its ends.
export TMPDIR=/tmp
o=gpurun_out/r05b; mkdir -p $o
timeout 600 python -m pytest tests/test_gpu_frames.py -m gpu -x -q -k "failed or on_top" > $o/pytest_fault.txt 2>&1; tail -5 $o/pytest_fault.txt
B="python bench.py --steps 20 --content dense --second-content '' --no-cpu-baseline --parity-frames 4 --no-1080p --no-e2e --no-pmc"
show() { python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-28s' % '$1', d['value'], d['ms_per_step'], d['pipeline']['read_roofline_frac'], d['timing'])"; }
for round in 1 2; do
  eval "$B" 2>/dev/null | show default
  ROC_ACTIVE_WAIT_TIMEOUT=100000 bash -c "$B" 2>/dev/null | show active_wait
  THIP_LANES=3 bash -c "$B" 2>/dev/null | show lanes3
  ROC_ACTIVE_WAIT_TIMEOUT=100000 THIP_LANES=3 bash -c "$B" 2>/dev/null | show active_wait_lanes3
  HIP_FORCE_DEV_KERNARG=1 bash -c "$B" 2>/dev/null | show dev_kernarg
  ROC_ACTIVE_WAIT_TIMEOUT=100000 HIP_FORCE_DEV_KERNARG=1 bash -c "$B" 2>/dev/null | show active_wait_dev_kernarg
done 2>&1 | tee $o/block_ends.txt
