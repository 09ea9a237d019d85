#!/bin/bash
# SQ counter passes over bench.py for whichever path THIP_FUSE selects (one lane so every launch is the 4-stream shape).
# usage: bash tools/r3_pmc.sh <outdir> [bench args]
export TMPDIR=/tmp
out=${1:-gpurun_out/pmc}
shift
mkdir -p $out
i=0
while read -r set; do
  i=$((i+1))
  THIP_LANES=1 timeout 300 rocprofv3 --pmc $set --output-format csv -d $out/pass$i -- python bench.py --steps 24 --warmup 4 --repeats 1 --min-time 0 --no-cpu-baseline --no-parity --no-profile --second-content "" "$@" > $out/pass$i.log 2>&1
  echo "pass $i ($set): rc=$?"
done <<'SETS'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE
SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT
SETS
python - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for d in sorted(glob.glob(out + "/pass*/")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0][:24]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            if k.startswith("k_"):
                print(d.split("/")[-2], k, {c: round(sum(v) / len(v), 1) for c, v in cs.items()}, "n=%d" % len(next(iter(cs.values()))))
PY
