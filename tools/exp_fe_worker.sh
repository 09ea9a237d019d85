#!/bin/bash
# th_decode_* end to end with the token-list path's second thread on (default) and off.   usage (GPU box): bash tools/exp_fe_worker.sh [out]
export TMPDIR=/tmp
out=${1:-gpurun_out/fe_worker.txt}
: > $out
for sz in 720p 1080p 4k; do for pk in dense typical; do for th in 1 4; do for wk in 1 0; do
  THIP_FE_WORKER=$wk timeout 900 python bench.py --mode e2e --e2e-size $sz --packets $pk --threads $th --loops 4 --no-native 2>/dev/null | grep '^{' | head -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('e2e $sz $pk threads $th fe_worker $wk:', d['value'], 'fps')" | tee -a $out
done; done; done; done
