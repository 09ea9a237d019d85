#!/bin/bash
export TMPDIR=/tmp
o=gpurun_out/r04la4
mkdir -p $o
timeout 300 python -m pytest tests/test_gpu_frontend.py -x -q -k "look_ahead or lookahead or walk_left or dump_video" 2>&1 | tail -3 > $o/pytest_lookahead.txt; cat $o/pytest_lookahead.txt
for asg in 1 0; do
  THIP_FE_ASSIGN=$asg THIP_FE_PROF=1 timeout 150 python tools/e2e_lookahead.py 720p,1080p,4k dense 1 0,4,8 > $o/e2e_assign$asg.jsonl 2> $o/stages_assign$asg.txt
done
timeout 100 python tools/e2e_lookahead.py 720p,1080p dense,typical 4 0,4,8 > $o/e2e_4streams.jsonl 2>> $o/err.txt
timeout 100 python tools/e2e_lookahead.py 720p,1080p typical 1 0,4,8 > $o/e2e_typical.jsonl 2>> $o/err.txt
cat $o/e2e_assign1.jsonl $o/e2e_assign0.jsonl $o/e2e_4streams.jsonl $o/e2e_typical.jsonl | cut -c1-140
grep "parser's\|ms/frame," $o/stages_assign1.txt
