# final validation of the round: GPU suite, soak, smoke, default bench (the driver's command), torchrun world 1
export TMPDIR=/tmp
o=gpurun_out/r3final; mkdir -p $o
timeout 2700 python -m pytest tests -m gpu -x -q > $o/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $o/pytest_gpu.log
timeout 700 python tests/soak_parity.py 13 500 > $o/soak.log 2>&1; echo "soak rc=$?"; tail -1 $o/soak.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
( time python bench.py ) > $o/bench_default.json 2> $o/bench.err; tail -3 $o/bench.err
python -c "
import json; d=json.loads(open('$o/bench_default.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['pipeline'], d['second_content']['value'], d['second_content']['pipeline_read_roofline_frac'], d['roofline']['traffic'], d['roofline']['frac'], d['parity'].get('matches_committed_crc32'), d['cpu_baseline']['value'])"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 2>&1 | tail -1 | cut -c1-300
