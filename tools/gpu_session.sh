#!/bin/bash
# One GPU session of round 5 (run through gpurun from the repository root): the pipelined look-ahead, the measured lists rule, the new encoder kernels
export TMPDIR=/tmp
o=gpurun_out/r05f; mkdir -p $o
timeout 1200 python -m pytest tests/test_gpu_frontend.py -m gpu -x -q -k "next_frame or promise or measured_per_context or rule_changes" > $o/pytest_frontend_new.txt 2>&1; tail -5 $o/pytest_frontend_new.txt
timeout 600 python -m pytest tests/test_gpu_slots.py -m gpu -x -q -k "fdct or halfpel" > $o/pytest_slots_new.txt 2>&1; tail -3 $o/pytest_slots_new.txt
python tools/native_lookahead.py 720p,1080p,4k dense 1 0,8 0,1 > $o/native_pipeline_1stream.jsonl 2>$o/native.err; cat $o/native_pipeline_1stream.jsonl | cut -c1-260
python tools/native_lookahead.py 720p,4k dense 4 0,8 0,1 > $o/native_pipeline_4streams.jsonl 2>>$o/native.err; cat $o/native_pipeline_4streams.jsonl | cut -c1-260
python bench.py --mode enc > $o/bench_enc.jsonl 2>/dev/null; python - <<'PY'
import json
for l in open("gpurun_out/r05f/bench_enc.jsonl"):
    d=json.loads(l); print(d["metric"][:100], d["value"], d["unit"], d["ms_per_call"], d["roofline"]["frac"])
PY
