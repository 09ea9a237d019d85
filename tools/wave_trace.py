#!/usr/bin/env python3
"""Wave timeline of k_recon on the bench workload (4K, 4 streams in one launch).

Builds a private copy of the library with -DTHIP_TRACE (lane 0 of every wave records
s_memrealtime at five points plus HW_ID / XCC_ID), decodes a few frames, traces one launch and
prints where a wave's life goes and how full the SIMDs are.  Diagnostic only.
  THIP_LANES=1 python tools/wave_trace.py [--content dense] [--out gpurun_out/trace.npz]
"""
import argparse
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("THIP_LANES", "1")


def build_trace_lib(extra):
    out = os.path.join(ROOT, "tools", "_build", "libtheora_hip_trace.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    csrc = os.path.join(ROOT, "theora_amd", "csrc")
    if not os.path.exists(out) or any(os.path.getmtime(os.path.join(csrc, f)) > os.path.getmtime(out) for f in os.listdir(csrc)):
        from theora_amd import build as thip_build
        thip_build.compile_library(out, ["-DTHIP_TRACE", "-I" + os.path.join(ROOT, "include")] + extra)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--content", default="dense")
    ap.add_argument("--size", default="4k")
    ap.add_argument("--streams", type=int, default=4)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "trace.npz"))
    ap.add_argument("--build-only", action="store_true")
    ap.add_argument("-D", action="append", default=[])
    args = ap.parse_args()
    so = build_trace_lib(["-D" + d for d in args.D])
    if args.build_only:
        return
    import torch
    from theora_amd import _lib
    _lib.SO_PATH = so
    import theora_amd
    from theora_amd import synth
    sizes = {"4k": (3840, 2160), "1080p": (1920, 1088)}
    w, h = sizes[args.size]
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
    ntiles = states[0].get_tiles()[-1] if hasattr(states[0], "get_tiles") else None
    L = _lib.load()
    maxu = 4 * ((3100 + 3) // 4 + 64)
    buf = torch.zeros((S, maxu * 4, 8), dtype=torch.int64, device="cuda")   # generous: gridDim.x*4 <= maxu*4
    L.thip_debug_trace_buffer.argtypes = [ctypes.c_void_p]
    L.thip_debug_trace_buffer(ctypes.c_void_p(buf.data_ptr()))
    plans[2].submit(None)
    theora_amd.synchronize()
    L.thip_debug_trace_buffer(ctypes.c_void_p(0))
    t = buf.cpu().numpy().reshape(-1, 8)
    t = t[t[:, 0] != 0]
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    np.savez_compressed(args.out, t=t)
    analyse(t)


def analyse(t):
    t0 = t[:, 0].min()
    us = lambda x: (x - t0) / 100.0          # s_memrealtime: 100 MHz
    st, h1, iss, dat, end = (us(t[:, i].astype(np.float64)) for i in range(5))
    hw, xcc = t[:, 5], t[:, 6] & 0xF
    cu = (hw >> 8) & 0xF
    sh = (hw >> 12) & 0x1
    se = (hw >> 13) & 0x7
    simd = (hw >> 4) & 0x3
    print("waves traced %d; kernel span (first start -> last end) %.1f us" % (len(t), end.max()))
    ok = dat > 0
    for name, a, b in (("start->cmd words", st, h1), ("cmd words->loads issued", h1, iss), ("issued->data", iss, dat),
                       ("data->stores issued", dat, end), ("whole life", st, end)):
        d = (b - a)[ok]
        print("  %-26s mean %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f us" % (name, d.mean(), *np.percentile(d, [10, 50, 90]), d.max()))
    # occupancy over time
    grid = np.arange(0, end.max(), 0.5)
    occ = [(np.sum((st <= g) & (end > g))) for g in grid]
    print("  resident waves every 0.5us (of %d slots at 4/SIMD):" % (256 * 16))
    print("   ", " ".join("%d" % o for o in occ))
    key = xcc * 1000 + se * 100 + sh * 50 + cu
    nk = len(np.unique(key))
    print("  distinct (xcc,se,sh,cu): %d; waves per XCC: %s" % (nk, np.bincount(xcc.astype(int), minlength=8)))
    # start-time histogram: how the dispatcher feeds waves
    hist, _ = np.histogram(st, bins=np.arange(0, end.max() + 2, 2.0))
    print("  wave starts per 2us:", " ".join(map(str, hist)))
    hist, _ = np.histogram(end, bins=np.arange(0, end.max() + 2, 2.0))
    print("  wave ends   per 2us:", " ".join(map(str, hist)))
    # last XCC to finish
    for x in range(8):
        m = xcc == x
        if m.any():
            print("   xcc %d: first start %.1f last end %.1f, mean life %.2f" % (x, st[m].min(), end[m].max(), (end - st)[m].mean()))


if __name__ == "__main__":
    main()
