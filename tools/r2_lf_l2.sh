#!/bin/bash
# (FETCH_SIZE alone per pass: it takes 3 of the 4 TCC counter slots; with TCC_HIT/MISS beside it rocprofv3 never finishes.)
# Does k_loopfilter find k_recon's pixels in the XCD's L2?  Kernel times and FETCH_SIZE / hit rate per launch shape.
export TMPDIR=/tmp
out=gpurun_out/lf_l2
mkdir -p $out
for cfg in "2 8" "2 1" "1 8" "1 1" "4 1"; do
  set -- $cfg
  tag=lanes$1_chunk$2
  THIP_LANES=$1 THIP_CHUNK=$2 python bench.py --steps 256 --warmup 16 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$tag', 'fps', d['value'], 'ms/step', d['ms_per_step'], 'recon_us', r['avg_launch_us'], 'lf_us', r['loopfilter_avg_launch_us'])"
  THIP_LANES=$1 THIP_CHUNK=$2 timeout 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/$tag -- python bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-profile > $out/$tag.log 2>&1
  python - $out/$tag <<'PY'
import csv, glob, sys, collections
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][:24]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        if k.startswith("k_"):
            print("   ", k, {c: round(sum(v) / len(v), 1) for c, v in cs.items()}, "n=%d" % len(next(iter(cs.values()))))
PY
done
