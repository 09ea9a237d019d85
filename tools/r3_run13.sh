export TMPDIR=/tmp
o=gpurun_out/r3k; mkdir -p $o
timeout 2400 python -m pytest tests -m gpu -x -q > $o/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $o/pytest_gpu.log
timeout 700 python tests/soak_parity.py 7 500 > $o/soak.log 2>&1; echo "soak rc=$?"; tail -2 $o/soak.log
python bench.py > $o/bench_default.json 2>$o/bench.err; python -c "
import json; d=json.loads(open('$o/bench_default.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['pipeline'], d['second_content']['value'], d['second_content']['pipeline_read_roofline_frac'], d['roofline']['traffic'], d['roofline']['frac'], d['parity'].get('matches_committed_crc32'))"
