#!/bin/bash
# PMC passes over bench.py (dense 4K, one lane so every launch is the 4-stream shape).
# usage: tools/pmc_recon.sh <outdir> ; results: <outdir>/passN/*counter_collection.csv
export TMPDIR=/tmp
out=${1:-gpurun_out/pmc}
mkdir -p $out
rocprofv3 -L > $out/counters.txt 2>&1
i=0
while read -r set; do
  i=$((i+1))
  THIP_LANES=1 timeout 300 rocprofv3 --pmc $set --output-format csv -d $out/pass$i -- python bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-profile > $out/pass$i.log 2>&1
  echo "pass $i ($set): rc=$?"
done <<'SETS'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE
TA_BUSY_avr TA_BUSY_max TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum TCC_EA_WRREQ_sum
TCC_TAG_STALL_sum TCC_BUSY_avr TCC_EA_RDREQ_DRAM_sum TCC_EA_WRREQ_64B_sum TCC_EA_WRREQ_STALL_sum TCC_EA_RD_UNCACHED_32B_sum
MemUnitBusy MemUnitStalled WriteUnitStalled VALUBusy L2CacheHit
TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
SETS
python - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for d in sorted(glob.glob(out + "/pass*/")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:24]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            if "k_recon" in k or "k_loopfilter" in k:
                print(d.split("/")[-2], k, {c: round(sum(v) / len(v), 1) for c, v in cs.items()}, "n=%d" % len(next(iter(cs.values()))))
PY
