export TMPDIR=/tmp
o=gpurun_out/r3i; mkdir -p $o
for c in smooth dense; do for dbg in 0 256; do
THIP_DEBUG=$dbg THIP_LANES=1 timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $o/pmc_${c}_$dbg -- python bench.py --content $c --steps 24 --warmup 4 --repeats 1 --min-time 0 --no-cpu-baseline --no-parity --no-profile --no-pmc --second-content "" > $o/pmc_${c}_$dbg.log 2>&1
python - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("$o/pmc_${c}_$dbg/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("k_recon_lf"): acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("$c debug=$dbg", {k: round(sum(v)/len(v)) for k,v in acc.items()})
PY
THIP_DEBUG=$dbg python bench.py --content $c --steps 256 --no-cpu-baseline --no-parity --no-pmc --second-content "" 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   bench $c debug=$dbg', d['ms_per_step'], d['roofline']['avg_launch_us'])"
done; done
