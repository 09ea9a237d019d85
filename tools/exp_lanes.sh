export TMPDIR=/tmp
for lanes in 1 2 3 4; do
 for st in 256 20; do
  THIP_LANES=$lanes python bench.py --steps $st --no-cpu-baseline --no-parity --no-pmc --no-profile --no-1080p --no-e2e --second-content "" 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('lanes $lanes steps $st', d['value'], d['ms_per_step'], d['pipeline']['read_roofline_frac'])"
 done
done
THIP_LANES=2 python bench.py --steps 20 --streams-per-gpu 8 --no-cpu-baseline --no-parity --no-pmc --no-profile --no-1080p --no-e2e --second-content "" 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('8 streams', d['value'], d['ms_per_step'], d['pipeline']['read_roofline_frac'])"
