export TMPDIR=/tmp
for pt in 0 2048 4096; do
 for cfg in "1080p 1" "720p 1" "720p 4" "1080p 4" "cif 1"; do
  set -- $cfg
  THIP_PIPE_TILES=$pt python bench.py --size $1 --streams-per-gpu $2 --steps 128 --no-cpu-baseline --parity-frames 40 --no-pmc --no-1080p --second-content "" 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('pipe_tiles $pt', '$1', 'streams $2', d['value'], d['ms_per_step'], d['pipeline']['read_roofline_frac'], d['parity']['bit_exact'])"
 done
done
