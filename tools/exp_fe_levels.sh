export TMPDIR=/tmp
for lv in 0 1 0 1; do
 for cfg in "720p 1" "720p 16" "1080p 16"; do
  set -- $cfg
  THIP_FE_LEVELS=$lv THIP_FE_DEVICE_LISTS=0 python bench.py --mode e2e --e2e-size $1 --threads $2 --no-native 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('fe_levels $lv host path', '$1', 'threads $2', d['value'])"
 done
done
