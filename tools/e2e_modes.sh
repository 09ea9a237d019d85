#!/bin/bash
# End to end through th_decode_* with the front end's stages on the host or on the device, 720p, 1 / 16 / 32 host threads:
#   host path (THIP_FE_DEVICE_LISTS=0) | token lists on the device, DC chain on the host (=1) | the default (lists while at most four
#   contexts are alive) | lists and DC on the device (+ THIP_FE_DEVICE_DC=1)
# usage (GPU box, repo root): bash tools/e2e_modes.sh [outfile]
export TMPDIR=/tmp
out=${1:-gpurun_out/e2e_modes.txt}
: > $out
for t in 1 4 8 16 32; do
  for env in "THIP_FE_DEVICE_LISTS=0" "THIP_FE_DEVICE_LISTS=1" "" "THIP_FE_DEVICE_LISTS=1 GPU_MAX_HW_QUEUES=8" "THIP_FE_DEVICE_LISTS=1 THIP_FE_DEVICE_DC=1"; do
    [ "$t" != 16 ] && [ "$env" = "THIP_FE_DEVICE_LISTS=1 GPU_MAX_HW_QUEUES=8" ] && continue
    for k in dense typical; do
      v=$(env $env timeout 300 python bench.py --mode e2e --e2e-size 720p --packets $k --threads $t --no-native --loops 6 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.readline())['value'])")
      echo "e2e 720p $k threads $t [${env:-default: the library chooses}]: $v fps" | tee -a $out
    done
  done
done
THIP_FE_DEVICE_LISTS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/e2e_modes_st -- python bench.py --mode e2e --e2e-size 720p --packets dense --no-native --loops 3 > /dev/null 2>&1
f=$(find gpurun_out/e2e_modes_st -name "*kernel_stats.csv" | head -1)
{ echo "# kernels of the token-list path, 720p dense packets, one thread (rocprofv3 --kernel-trace --stats):"; head -9 $f | cut -d, -f1-4,6,7; } >> $out
