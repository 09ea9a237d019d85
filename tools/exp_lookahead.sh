#!/bin/bash
# The look-ahead on the GPU box: the front end's GPU tests, then end to end with 0 / 2 / 4 packets announced ahead.
export TMPDIR=/tmp
o=gpurun_out/r04la
mkdir -p $o
timeout 400 python -m pytest tests/test_gpu_frontend.py -x -q 2>&1 | tail -4 > $o/pytest_gpu_frontend.txt; cat $o/pytest_gpu_frontend.txt
timeout 150 python tools/e2e_lookahead.py 720p,1080p dense,typical 1,4 0,2,4 > $o/e2e_lookahead.jsonl 2> $o/e2e_err.txt
timeout 120 python tools/e2e_lookahead.py 4k dense 1,4 0,2,4 >> $o/e2e_lookahead.jsonl 2>> $o/e2e_err.txt
THIP_FE_PROF=1 timeout 100 python tools/e2e_lookahead.py 720p,1080p,4k dense 1 0,4 2> $o/stage_tables.txt > /dev/null
cat $o/e2e_lookahead.jsonl; tail -5 $o/e2e_err.txt
