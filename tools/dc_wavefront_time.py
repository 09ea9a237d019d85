#!/usr/bin/env python3
"""How long k_dc_unpredict (DC un-prediction as an anti-diagonal wavefront, one work group per plane) takes on
the luma plane of a 720p / 1080p / 4K frame, by content: one reference frame everywhere (a key frame: the pure
wavefront, nh + 2*nv steps), runs of references (typical inter frames), random references (most fragments fall
back to pred_last, whose dependency serialises the rows).  Host time of the same work: ~10 ns per fragment.
  python tools/dc_wavefront_time.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from theora_amd import _lib  # noqa: E402

L = _lib.load()
rng = np.random.default_rng(1)
for name, nh, nv in (("720p", 160, 90), ("1080p", 240, 136), ("4K", 480, 270)):
    n = nh * nv
    for density, refmix in ((1.0, "one"), (0.7, "one"), (0.7, "runs"), (0.7, "random"), (0.2, "random")):
        coded = (rng.random(n) < density).astype(np.uint8)
        if refmix == "one":
            refi = np.full(n, 2, np.uint8)
        elif refmix == "runs":
            refi = np.repeat(rng.integers(0, 3, n // 7 + 1), 7)[:n].astype(np.uint8)
        else:
            refi = rng.integers(0, 3, n).astype(np.uint8)
        flags = torch.from_numpy((coded | (refi << 1)) * coded).cuda()
        tok = torch.from_numpy(rng.integers(-200, 201, n).astype(np.int16)).cuda()
        best = 1e9
        for rep in range(3):
            d = tok.clone()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            assert L.thip_dc_unpredict_plane(d.data_ptr(), flags.data_ptr(), nh, nv) == 0   # synchronous
            best = min(best, time.perf_counter() - t0)
        print("%-6s %3dx%-3d coded %.0f%% refs %-6s : %8.3f ms   (%d fragments: host ~%.2f ms at 10 ns each; pure wavefront = %d steps)"
              % (name, nh, nv, 100 * density, refmix, 1e3 * best, n, n * 1e-5, nh + 2 * nv))
