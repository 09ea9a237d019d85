#!/usr/bin/env python3
"""Phase timeline of k_recon_walk on the bench workload (4K, 4 streams in one launch).

Builds a private copy of the library with -DTHIP_TRACE (lane 0 of every wave stamps s_memrealtime at six
points of every tile it walks), decodes a few frames with THIP_FUSE=1, traces one launch and prints where
a tile's time goes.  Diagnostic only.
  python tools/walk_trace.py [--content dense]
"""
import argparse
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("THIP_LANES", "1")
os.environ["THIP_FUSE"] = "1"
from tools.wave_trace import build_trace_lib   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--content", default="dense")
    ap.add_argument("--streams", type=int, default=4)
    args = ap.parse_args()
    so = build_trace_lib([])
    import torch
    from theora_amd import _lib
    _lib.SO_PATH = so
    import theora_amd
    from theora_amd import synth
    w, h = 3840, 2160
    geom = synth.Geometry(w, h)
    S = args.streams
    keep, descs = [], []
    for gid in range(S):
        rng = np.random.default_rng(1000 + gid)
        frames = [synth.gen_frame(geom, rng, theora_amd.INTRA_FRAME, args.content, flimit=2)]
        frames += [synth.gen_frame(geom, rng, theora_amd.INTER_FRAME, args.content, flimit=2) for _ in range(3)]
        row = []
        for f in frames:
            d, ka = synth.upload_frame(synth.pack_frame(geom, f))
            keep.append(ka)
            row.append(d)
        descs.append(row)
    states = [theora_amd.State(w, h) for _ in range(S)]
    plans = [theora_amd.BatchPlan(states, [descs[s][j] for s in range(S)]) for j in range(4)]
    for i in range(12):
        plans[0 if i == 0 else 1 + i % 3].submit(None)
    theora_amd.synchronize()
    L = _lib.load()
    ntiles = 4080
    buf = torch.zeros((S, ntiles, 8), dtype=torch.int64, device="cuda")
    L.thip_debug_trace_buffer.argtypes = [ctypes.c_void_p]
    L.thip_debug_trace_buffer(ctypes.c_void_p(buf.data_ptr()))
    plans[2].submit(None)
    theora_amd.synchronize()
    L.thip_debug_trace_buffer(ctypes.c_void_p(0))
    t = buf.cpu().numpy().reshape(-1, 8)
    t = t[t[:, 0] != 0]
    t0 = t[:, 0].min()
    us = lambda x: (x.astype(np.float64) - t0) / 100.0          # s_memrealtime: 100 MHz
    P = [us(t[:, i]) for i in range(6)]
    it = (t[:, 7] >> 32) & 0xFF
    print("tiles traced %d; kernel span %.1f us" % (len(t), P[5].max()))
    names = ["begin -> residual done (coefficient round trip + transform)", "residual -> pixels (predictor wait + recon)",
             "pixels -> image + edge published", "published -> left edge consumed (wait for neighbour)",
             "consumed -> cells filtered, stores issued", "whole tile"]
    for rnd in sorted(set(it.tolist())):
        m = it == rnd
        print(" round %d (%d tiles): begins at %.1f .. %.1f us (mean %.1f)" % (rnd, m.sum(), P[0][m].min(), P[0][m].max(), P[0][m].mean()))
        for n, (a, b) in zip(names, [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (0, 5)]):
            d = (P[b] - P[a])[m]
            print("   %-62s mean %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f" % (n, d.mean(), *np.percentile(d, [10, 50, 90]), d.max()))
    grid = np.arange(0, P[5].max(), 1.0)
    busy = [int(np.sum((P[0] <= g) & (P[5] > g))) for g in grid]
    wait = [int(np.sum((P[0] <= g) & (P[1] > g))) for g in grid]
    print(" tiles in progress every 1 us:", " ".join(map(str, busy)))
    print(" of which before 'residual done':", " ".join(map(str, wait)))


if __name__ == "__main__":
    main()
