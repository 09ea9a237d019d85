#!/bin/bash
# Refreshes the numbers kept under profiles/: default/secondary bench lines, rocprofv3 kernel stats, FETCH/WRITE PMC passes.
export TMPDIR=/tmp
mkdir -p gpurun_out/r01b
python bench.py > gpurun_out/r01b/bench_default.json 2> gpurun_out/r01b/bench_default.err
python bench.py --content smooth --no-cpu-baseline > gpurun_out/r01b/bench_smooth.json 2>/dev/null
python bench.py --size 1080p --no-cpu-baseline > gpurun_out/r01b/bench_1080p.json 2>/dev/null
python bench.py --mode enc 2>/dev/null | grep '^{' > gpurun_out/r01b/bench_enc.jsonl
python bench.py --mode e2e --threads 1 --loops 20 2>/dev/null | grep "^{" > gpurun_out/r01b/bench_e2e.jsonl
python bench.py --mode e2e --threads 1 --loops 20 --packets typical 2>/dev/null | grep "^{" > gpurun_out/r01b/bench_e2e_typical.jsonl
THIP_LANES=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01b/stats_lanes1 -- python bench.py --steps 128 --no-cpu-baseline > gpurun_out/r01b/stats_lanes1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01b/stats_default -- python bench.py --steps 128 --no-cpu-baseline > gpurun_out/r01b/stats_default.log 2>&1
THIP_LANES=1 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r01b/pmc_fetch -- python bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-profile > gpurun_out/r01b/pmc_fetch.log 2>&1
THIP_LANES=1 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r01b/pmc_write -- python bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-profile > gpurun_out/r01b/pmc_write.log 2>&1
tail -1 gpurun_out/r01b/bench_default.json
find gpurun_out/r01b -name "*kernel_stats.csv" | head; find gpurun_out/r01b -name "*counter_collection.csv" | head
