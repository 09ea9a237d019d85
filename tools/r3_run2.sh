export TMPDIR=/tmp
out=gpurun_out/r3lf; mkdir -p $out
CONTENTS="" FUSES="" bash tools/r3_lf_check.sh full
python tools/lf_trace.py --content dense > $out/trace_dense.txt 2>&1
python tools/lf_trace.py --content smooth > $out/trace_smooth.txt 2>&1
for lanes in 1 2 3 4; do for c in dense smooth; do
THIP_LANES=$lanes THIP_FUSE=3 python bench.py --steps 256 --content $c --second-content '' --no-cpu-baseline --no-profile --parity-frames 3 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lanes $lanes $c', d['value'], d['ms_per_step'], d['pipeline']['read_roofline_frac'])" >> $out/lanes.txt
done; done
THIP_LANES=4 THIP_FUSE=3 python bench.py --steps 256 --streams-per-gpu 8 --content dense --second-content '' --no-cpu-baseline --no-profile --parity-frames 3 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lanes 4 streams 8 dense', d['value'], d['ms_per_step'], d['pipeline']['read_roofline_frac'])" >> $out/lanes.txt
cat $out/lanes.txt
