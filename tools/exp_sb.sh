export TMPDIR=/tmp
for sbt in 0 1024 8192; do
 for cfg in "1080p 1" "1080p 4" "720p 1" "720p 4"; do
  set -- $cfg
  THIP_SB_TILES=$sbt python bench.py --size $1 --streams-per-gpu $2 --steps 128 --no-cpu-baseline --parity-frames 4 --no-pmc --no-1080p --no-e2e --second-content "" 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('sb_tiles $sbt', '$1', 'streams $2', d['value'], d['ms_per_step'], d['pipeline']['read_roofline_frac'], d['roofline']['avg_launch_us'])"
 done
done
