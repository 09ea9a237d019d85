#!/bin/bash
# End-of-round check: the whole GPU suite, smoke(), the rule of fe_assign = 2 on longer runs, the driver's bench line.
export TMPDIR=/tmp
o=gpurun_out/r04fc
mkdir -p $o
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $o/pytest_gpu.txt; cat $o/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
E2E_LOOPS=6 THIP_FE_PROF=1 timeout 150 python tools/e2e_lookahead.py 720p,1080p,4k dense 1 4,8 > $o/e2e_single_auto.jsonl 2> $o/stages_auto.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $o/bench_steps20.json
python - <<'PY'
import json
o = "gpurun_out/r04fc/"
for l in open(o + "e2e_single_auto.jsonl"):
    d = json.loads(l)
    print(d["size"], d["packets"], "streams", d["streams"], "la", d["lookahead"], d["frames_per_s"])
d = json.loads(open(o + "bench_steps20.json").read())
print("bench", d["value"], d["pipeline"], d.get("e2e_720p"))
PY
grep "look-ahead: [0-9]" $o/stages_auto.txt
