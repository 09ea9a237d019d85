#!/bin/bash
# th_decode_* end to end (packets in host memory -> YUV in host memory) at 720p / 1080p / 4K, dense and typical packets, 1 / 4 / 16
# host threads, the library's defaults.   usage (GPU box, repo root): bash tools/e2e_sizes.sh [outfile]
export TMPDIR=/tmp
out=${1:-gpurun_out/e2e_sizes.txt}
: > $out
for sz in 720p 1080p 4k; do for pk in dense typical; do for th in 1 4 16; do
  timeout 900 python bench.py --mode e2e --e2e-size $sz --packets $pk --threads $th --loops 4 --no-native 2>/dev/null | grep '^{' | head -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('e2e $sz $pk threads $th:', d['value'], 'fps, packet', d['avg_packet_bytes'])" | tee -a $out
done; done; done
