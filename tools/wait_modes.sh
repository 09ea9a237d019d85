#!/bin/bash
# How a context should wait for its frame when there are more decoder threads than cores
# (THIP_WAIT_SPIN=1: hipEventSynchronize; default: a few polls, then 20 us sleeps between polls).
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
python tools/make_clip720.py
for m in 1 0; do
  for nt in 1 16 32 64; do
    echo -n "spin $m threads $nt: "
    THIP_WAIT_SPIN=$m examples/decode_bench gpurun_out/clip720.ogv $nt 60 2>/dev/null | tail -1 | cut -c1-110
  done
done
