#!/usr/bin/env python3
"""Wave timeline of k_recon_lf (THIP_FUSE=3) on the bench workload (4K, 4 streams in one launch).

Builds a private copy of the library with -DTHIP_TRACE (lane 0 of every wave stamps s_memrealtime at the nine points
marked in thip_fused.h plus HW_ID / XCC_ID), decodes a few frames, traces one launch and prints where a wave's life goes.
Diagnostic only.   python tools/lf_trace.py [--content dense|smooth] [--streams 4] [-D MACRO=VALUE ...]
"""
import argparse
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("THIP_LANES", "1")
os.environ["THIP_FUSE"] = "3"
from wave_trace import build_trace_lib   # noqa: E402

PHASES = ["start->command words", "->pixels (coefficients, windows, transform)", "->image in LDS, edge units stored",
          "->(nothing: the units need no acknowledgement)", "->neighbours' units there", "->margins filled", "->cells, stores issued",
          "->extra cells / end"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--content", default="dense")
    ap.add_argument("--size", default="4k")
    ap.add_argument("--streams", type=int, default=4)
    ap.add_argument("-D", action="append", default=[])
    args = ap.parse_args()
    so = build_trace_lib(["-D" + d for d in args.D])
    import torch
    from theora_amd import _lib
    _lib.SO_PATH = so
    import theora_amd
    from theora_amd import synth
    w, h = {"4k": (3840, 2160), "1080p": (1920, 1088), "720p": (1280, 720)}[args.size]
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
    nrec = 4 * S * 8 * (geom.ntiles // 8 + 64)      # (k_recon_lf_sb: four waves per work group)
    buf = torch.zeros((nrec, 12), dtype=torch.int64, device="cuda")
    L.thip_debug_trace_buffer.argtypes = [ctypes.c_void_p]
    L.thip_debug_trace_buffer(ctypes.c_void_p(buf.data_ptr()))
    plans[2].submit(None)
    theora_amd.synchronize()
    L.thip_debug_trace_buffer(ctypes.c_void_p(0))
    t = buf.cpu().numpy()
    t = t[t[:, 0] != 0]
    t0 = t[:, 0].min()
    T = (t[:, :9].astype(np.float64) - t0) / 100.0      # s_memrealtime: 100 MHz
    print("waves traced %d; kernel span %.1f us" % (len(t), T[:, 8].max()))
    for i, name in enumerate(PHASES):
        d = T[:, i + 1] - T[:, i]
        print("  %-46s mean %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f us" % (name, d.mean(), *np.percentile(d, [10, 50, 90]), d.max()))
    tres = (t[:, 11].astype(np.float64) - t0) / 100.0
    d = tres - T[:, 1]
    print("    of the pixels: command words -> residual (coefficient round trip, transform)  mean %6.2f  p50 %6.2f" % (d.mean(), np.percentile(d, 50)))
    d = T[:, 2] - tres
    print("                   residual -> pixels (predictor windows awaited, prediction)      mean %6.2f  p50 %6.2f" % (d.mean(), np.percentile(d, 50)))
    life = T[:, 8] - T[:, 0]
    print("  %-46s mean %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f us" % ("whole life", life.mean(), *np.percentile(life, [10, 50, 90]), life.max()))
    grid = np.arange(0, T[:, 8].max(), 1.0)
    print("  resident waves every 1 us:", " ".join("%d" % np.sum((T[:, 0] <= g) & (T[:, 8] > g)) for g in grid))
    xcc = (t[:, 10] & 0xF).astype(int)
    for x in range(8):
        m = xcc == x
        if m.any():
            print("   xcc %d: %5d waves, first start %5.1f last end %5.1f, mean life %.2f" % (x, m.sum(), T[m, 0].min(), T[m, 8].max(), life[m].mean()))
    # life by quarter of the launch (when the wave started)
    q = np.minimum((T[:, 0] / (T[:, 0].max() + 1e-9) * 4).astype(int), 3)
    for i in range(4):
        m = q == i
        print("   waves started in quarter %d of the dispatch: %5d, mean life %.2f, wait for neighbours %.2f" % (i, m.sum(), life[m].mean(), (T[m, 5] - T[m, 4]).mean()))


if __name__ == "__main__":
    main()
