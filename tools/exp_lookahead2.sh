#!/bin/bash
# The look-ahead with the walk on the parser threads (k_tok_scatter): its GPU tests, then end to end, A/B against the device's walk.
export TMPDIR=/tmp
o=gpurun_out/r04la2
mkdir -p $o
timeout 300 python -m pytest tests/test_gpu_frontend.py -x -q -k "look_ahead or lookahead or dump_video" 2>&1 | tail -4 > $o/pytest_lookahead.txt; cat $o/pytest_lookahead.txt
timeout 150 python tools/e2e_lookahead.py 720p,1080p dense,typical 1,4 0,4 > $o/e2e_lookahead.jsonl 2> $o/e2e_err.txt
timeout 120 python tools/e2e_lookahead.py 4k dense 1,4 0,4 >> $o/e2e_lookahead.jsonl 2>> $o/e2e_err.txt
THIP_FE_ASSIGN=0 timeout 120 python tools/e2e_lookahead.py 720p,1080p,4k dense 1 4 > $o/e2e_lookahead_device_walk.jsonl 2>> $o/e2e_err.txt
THIP_FE_PROF=1 timeout 100 python tools/e2e_lookahead.py 720p,1080p,4k dense 1 4 2> $o/stage_tables.txt > /dev/null
timeout 200 python tools/native_lookahead.py 720p,1080p,4k dense 1,4 0,4 > $o/native_decode_bench.jsonl 2>> $o/e2e_err.txt
cat $o/e2e_lookahead.jsonl $o/e2e_lookahead_device_walk.jsonl; cut -c1-150 $o/native_decode_bench.jsonl; tail -3 $o/e2e_err.txt
