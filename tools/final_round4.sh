#!/bin/bash
# End-of-round check on one box: the whole GPU suite, smoke(), the driver's bench line, the end-to-end table.
export TMPDIR=/tmp
o=gpurun_out/r04f
mkdir -p $o
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $o/pytest_gpu.txt; cat $o/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $o/bench_steps20.json; cut -c1-160 $o/bench_steps20.json
: > $o/e2e_sizes.jsonl
for k in dense typical; do for sz in 720p 1080p 4k; do for t in 1 4 16; do
  timeout 600 python bench.py --mode e2e --e2e-size $sz --packets $k --threads $t --loops 4 --no-native 2>/dev/null | tail -1 >> $o/e2e_sizes.jsonl
done; done; done
: > $o/e2e_single_stream_one_piece.jsonl
for sz in 720p 1080p 4k; do
  THIP_FE_GROUPS=1 THIP_FE_WORKER=0 THIP_TL_LEVELS=0 THIP_TL_ALGO=1 timeout 600 python bench.py --mode e2e --e2e-size $sz --packets dense --threads 1 --loops 4 --no-native 2>/dev/null | tail -1 >> $o/e2e_single_stream_one_piece.jsonl
done
python - <<'PY'
import json
for f in ("e2e_sizes", "e2e_single_stream_one_piece"):
    for l in open("gpurun_out/r04f/%s.jsonl" % f):
        d = json.loads(l)
        print(f, d["metric"].split("(")[1].split(",")[0], d["data"].split(",")[1].strip(), "threads", d["host_threads"], d["value"])
PY
