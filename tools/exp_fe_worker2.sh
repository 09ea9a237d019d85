#!/bin/bash
# th_decode_* end to end, token-list path: in one piece after the packet (fe_groups 1) or in groups of indices as they are decoded,
# the DC chain behind the tokens (fe_worker 0) or beside them on the context's second thread (fe_worker 1; fe_worker_pin 0: the thread
# left to the scheduler).   usage (GPU box): bash tools/exp_fe_worker2.sh
export TMPDIR=/tmp
run() { sz=$1; pk=$2; th=$3; shift 3
  echo -n "e2e $sz $pk threads $th [$*]: "
  env "$@" timeout 900 python bench.py --mode e2e --e2e-size $sz --packets $pk --threads $th --loops 4 --no-native 2>/dev/null | grep '^{' | head -1 | python -c "
import sys,json; print(json.loads(sys.stdin.read())['value'], 'fps')"; }
for cfg in "THIP_FE_GROUPS=1 THIP_FE_WORKER=0" "THIP_FE_GROUPS=4 THIP_FE_WORKER=0" "THIP_FE_GROUPS=1 THIP_FE_WORKER=1" "THIP_FE_GROUPS=4 THIP_FE_WORKER=1" "THIP_FE_GROUPS=1 THIP_FE_WORKER=1 THIP_FE_WORKER_PIN=0"; do
  for sz in 720p 1080p 4k; do for th in 1 4; do run $sz dense $th $cfg; done; done
  run 720p typical 1 $cfg; run 4k typical 1 $cfg
done
