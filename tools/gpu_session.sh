#!/bin/bash
# One GPU session of round 5 (run through gpurun from the repository root): streams per launch; the whole GPU suite on the new build
export TMPDIR=/tmp
o=gpurun_out/r05e; mkdir -p $o
B="python bench.py --content dense --second-content '' --no-cpu-baseline --parity-frames 4 --no-1080p --no-e2e --no-pmc --no-wide --no-enc"
show() { python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['timing']; print('%-22s' % '$1', d['value'], d['ms_per_step'], d['pipeline']['read_roofline_frac'], 'min', t['ms_per_step_min'], 'max', t['ms_per_step_max'], 'submit_us', t['host_submit_us_per_block'])"; }
for round in 1 2; do
  bash -c "$B --steps 20" 2>/dev/null | show steps20_chunk8
  THIP_CHUNK=1 bash -c "$B --steps 20" 2>/dev/null | show steps20_chunk1
  THIP_STAGGER=16 bash -c "$B --steps 20" 2>/dev/null | show steps20_stagger16
  THIP_CHUNK=1 bash -c "$B --steps 256" 2>/dev/null | show steps256_chunk1
  bash -c "$B --steps 256" 2>/dev/null | show steps256_chunk8
done 2>&1 | tee $o/chunk.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $o/pytest_gpu.txt 2>&1; tail -5 $o/pytest_gpu.txt
