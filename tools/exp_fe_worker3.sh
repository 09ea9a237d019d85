#!/bin/bash
# stage tables of the token-list path at 1080p and 4K, in one piece and in four groups.   usage (GPU box): bash tools/exp_fe_worker3.sh
export TMPDIR=/tmp
for sz in 1080p 4k; do
 for cfg in "THIP_FE_GROUPS=1" "THIP_FE_WORKER=0 THIP_FE_GROUPS=4" "THIP_FE_WORKER=1 THIP_FE_GROUPS=4"; do
  echo "== $sz dense threads 1 [$cfg]"
  env $cfg THIP_FE_PROF=1 timeout 900 python bench.py --mode e2e --e2e-size $sz --packets dense --threads 1 --loops 3 --no-native 2>&1 | grep -E '^\{|thip front end|ms/frame' | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): print('   fps', json.loads(l)['value'])
    elif ' 0.000 ms' not in l: print(l.rstrip())"
 done
done
