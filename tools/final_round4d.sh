#!/bin/bash
# Last check of the round's last build: the look-ahead's GPU tests (dump_video_hip with its default of eight packets ahead among them).
export TMPDIR=/tmp
o=gpurun_out/r04fd
mkdir -p $o
timeout 170 python -m pytest tests/test_gpu_frontend.py -x -q -k "look_ahead or lookahead or walk_left or dump_video" 2>&1 | tail -3 > $o/pytest_lookahead.txt; cat $o/pytest_lookahead.txt
