#!/bin/bash
# A/B of differently compiled copies of the library in ONE session on one box (run-to-run spread between boxes is +-3 %):
#   bash tools/ab.sh name1=path1.so name2=path2.so ...     (THIP_LIB selects the copy; two interleaved rounds, dense and smooth;
#   AB_STEPS=20 for the driver's block length, AB_ROUNDS, AB_CONTENTS to change the defaults)
export TMPDIR=/tmp
for round in $(seq 1 ${AB_ROUNDS:-2}); do
  for kv in "$@"; do
    name=${kv%%=*}; lib=${kv#*=}
    for c in ${AB_CONTENTS:-dense smooth}; do
      THIP_LIB=$lib timeout 300 python bench.py --steps ${AB_STEPS:-256} --content $c --second-content '' --no-cpu-baseline --parity-frames 4 --no-1080p --no-e2e --no-pmc --no-form16 --no-wide --no-enc ${AB_ARGS} 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round $round %-14s %-6s' % ('$name','$c'), d['value'], d['ms_per_step'], d['pipeline']['read_roofline_frac'], (d.get('roofline') or {}).get('avg_launch_us'))"
    done
  done
done
