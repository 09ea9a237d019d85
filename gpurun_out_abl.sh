for dbg in 0 32 64; do
  THIP_LANES=1 THIP_SEG_TILES=1 THIP_DEBUG=$dbg python bench.py --steps 96 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('dbg=$dbg', 'frame_us', r['avg_launch_us'], 'seam_us', r['seam_avg_launch_us'], 'step_ms', d['ms_per_step'])"
done
