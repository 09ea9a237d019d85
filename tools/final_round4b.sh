#!/bin/bash
# End of round 4, second half: the look-ahead end to end (longer runs: the measured rule has settled), the driver's bench line.
export TMPDIR=/tmp
o=gpurun_out/r04fb
mkdir -p $o
for asg in 2 1 0; do
  E2E_LOOPS=5 THIP_FE_ASSIGN=$asg THIP_FE_PROF=1 timeout 200 python tools/e2e_lookahead.py 720p,1080p,4k dense 1 4,8 > $o/e2e_single_assign$asg.jsonl 2> $o/stages_assign$asg.txt
done
E2E_LOOPS=3 timeout 120 python tools/e2e_lookahead.py 720p,1080p,4k dense 1,4 0 > $o/e2e_plain.jsonl 2>> $o/err.txt
E2E_LOOPS=3 timeout 150 python tools/e2e_lookahead.py 720p,1080p dense,typical 4 4,8 > $o/e2e_4streams.jsonl 2>> $o/err.txt
E2E_LOOPS=3 timeout 100 python tools/e2e_lookahead.py 720p,1080p typical 1 0,4,8 > $o/e2e_typical.jsonl 2>> $o/err.txt
timeout 200 python tools/native_lookahead.py 720p,1080p,4k dense 1,4 0,4,8 > $o/native_decode_bench.jsonl 2>> $o/err.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $o/bench_steps20.json
python - <<'PY'
import json
o = "gpurun_out/r04fb/"
for f in ("e2e_single_assign2", "e2e_single_assign1", "e2e_single_assign0", "e2e_plain", "e2e_4streams", "e2e_typical"):
    for l in open(o + f + ".jsonl"):
        d = json.loads(l)
        print(f, d["size"], d["packets"], "streams", d["streams"], "la", d["lookahead"], d["frames_per_s"])
for l in open(o + "native_decode_bench.jsonl"):
    d = json.loads(l)
    print("native", d.get("size_name"), "streams", d.get("streams"), "la", d.get("lookahead"), d.get("frames_per_s"), d.get("rc"))
d = json.loads(open(o + "bench_steps20.json").read())
print("bench", d["value"], d["pipeline"], d.get("e2e_720p"))
PY
grep "look-ahead: [0-9]" $o/stages_assign2.txt
