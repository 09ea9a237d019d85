mkdir -p gpurun_out
for dbg in 0 1 2 4 8 3 7 15 11; do
  THIP_DEBUG=$dbg python bench.py --steps 96 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('dbg=$dbg', 'recon_us', r['avg_launch_us'], 'lf_us', r['loopfilter_avg_launch_us'], 'step_ms', d['ms_per_step'])"
done
