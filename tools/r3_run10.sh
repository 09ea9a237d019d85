export TMPDIR=/tmp
o=gpurun_out/r3h; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_frames.py tests/test_gpu_slots.py -m gpu -x -q -k "loop_filter or sequence_small or lane_shared or full_size or batched or four_concurrent" > $o/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $o/pytest.log
( time python bench.py > $o/bench_default.json 2> $o/bench_default.err ) 2>&1 | grep real
python -c "
import json; d=json.loads(open('$o/bench_default.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['pipeline'], d['second_content']['value'], d['second_content']['pipeline_read_roofline_frac'], d['roofline']['traffic'])"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 > $o/torchrun_world1.log 2>&1; echo "torchrun rc=$?"; tail -c 600 $o/torchrun_world1.log
timeout 600 python tests/soak_parity.py 240 > $o/soak.log 2>&1; echo "soak rc=$?"; tail -3 $o/soak.log
