export TMPDIR=/tmp
o=gpurun_out/r3e; mkdir -p $o
timeout 2400 python -m pytest tests -m gpu -x -q > $o/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $o/pytest_gpu.log
python bench.py > $o/bench_default.json 2> $o/bench_default.err; echo "bench rc=$?"; tail -c 1500 $o/bench_default.json
for sz in 720p 1080p 4k; do for pk in dense typical; do for th in 1 16; do
  python bench.py --mode e2e --e2e-size $sz --packets $pk --threads $th --loops 4 --no-native 2>/dev/null | grep '^{' | head -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('e2e $sz $pk threads $th:', d['value'], 'fps, packet', d['avg_packet_bytes'])" | tee -a $o/e2e_sizes.txt
done; done; done
