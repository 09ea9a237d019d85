#!/bin/bash
# Round-4 counter passes over k_recon_lf on the bench workload (one group of counters per pass, --pmc alone, as the guide prescribes).
#   usage (GPU box, repo root): bash tools/pmc_r4.sh [dense|smooth] ; results: gpurun_out/r04/pmc_<content>.txt
export TMPDIR=/tmp
content=${1:-dense}
o=gpurun_out/r04
mkdir -p $o
: > $o/pmc_$content.txt
for grp in "VALUBusy" "SQ_INSTS_VALU SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "MemUnitStalled" "SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE"; do
  d=$o/pmc_${content}_$(echo $grp | tr ' ' '_')
  THIP_LANES=${LANES:-1} timeout 300 rocprofv3 --pmc $grp --output-format csv -d $d -- python bench.py --content $content --steps 24 --warmup 4 --repeats 1 --min-time 0 --no-cpu-baseline --no-parity --no-profile --no-pmc --no-1080p --no-e2e --no-wide --no-enc --no-form16 --second-content "" > $d.log 2>&1
  python - "$d" >> $o/pmc_$content.txt <<'PY'
import csv, glob, sys, os
files = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)
acc = {}
for f in files:
    for row in csv.DictReader(open(f)):
        k = (row["Kernel_Name"].split("(")[0].replace("void ", ""), row["Counter_Name"])
        acc.setdefault(k, []).append(float(row["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    if "k_recon" in k or "k_loop" in k:
        print("%-22s %-28s %16.1f  (%d dispatches)" % (k, c, sum(v) / len(v), len(v)))
PY
done
cat $o/pmc_$content.txt
