#!/bin/bash
# HIP API time per call of the th_decode_* path (one 720p stream, native thread): where the host
# time of thip_frame_flush / thip_state_ycbcr_out goes.  Output: gpurun_out/api_trace/
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
python tools/make_clip720.py
rm -rf gpurun_out/api_trace; mkdir -p gpurun_out/api_trace
rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d gpurun_out/api_trace -o t -- examples/decode_bench gpurun_out/clip720.ogv 1 8 > gpurun_out/api_trace/run.log 2>&1
find gpurun_out/api_trace -name "*hip_api_stats.csv" | head -1 | xargs -r head -20
find gpurun_out/api_trace -name "*kernel_stats.csv" | head -1 | xargs -r head -8
tail -2 gpurun_out/api_trace/run.log
