#!/bin/bash
# A/B on ONE box (boxes differ by 10 % and more in host speed): options of the token-list path, two interleaved rounds.
#   usage (GPU box): [TH="1 4"] bash tools/exp_fe_ab.sh "THIP_TL_LEVELS=0" "THIP_TL_LEVELS=1"
export TMPDIR=/tmp
run() { sz=$1; pk=$2; th=$3; shift 3
  echo -n "e2e $sz $pk threads $th [$*]: "
  env "$@" timeout 900 python bench.py --mode e2e --e2e-size $sz --packets $pk --threads $th --loops 4 --no-native 2>/dev/null | grep '^{' | head -1 | python -c "
import sys,json; print(json.loads(sys.stdin.read())['value'], 'fps')"; }
for round in 1 2; do
 for sz in 720p 1080p 4k; do for th in ${TH:-1}; do
  for cfg in "$@"; do
   run $sz dense $th $cfg
  done
 done; done
done
