#!/bin/bash
# Parser stages on the GPU box's host: 4 and 8 packets ahead, the walk on the parser threads (fe_assign 1) or on the device (0).
export TMPDIR=/tmp
o=gpurun_out/r04la3
mkdir -p $o
for asg in 1 0; do
  THIP_FE_LOOKAHEAD=8 THIP_FE_ASSIGN=$asg THIP_FE_PROF=1 timeout 150 python tools/e2e_lookahead.py 720p,1080p,4k dense 1 4,8 > $o/e2e_assign$asg.jsonl 2> $o/stages_assign$asg.txt
done
THIP_FE_LOOKAHEAD=8 timeout 100 python tools/e2e_lookahead.py 720p,1080p dense,typical 4 0,4,8 > $o/e2e_4streams.jsonl 2>> $o/err.txt
cat $o/e2e_assign1.jsonl $o/e2e_assign0.jsonl $o/e2e_4streams.jsonl | cut -c1-140
grep "parser's\|DCT tokens\|ycbcr_out\|ms/frame," $o/stages_assign1.txt $o/stages_assign0.txt
