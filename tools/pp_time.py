#!/usr/bin/env python3
"""Cost of the out-of-loop post-processing on the device (thip_state_postprocess) by level, 720p and 4K, on a textured
picture (de-ringing in all strengths) -- kernel time on the state's stream, launches included.
  python tools/pp_time.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import theora_amd  # noqa: E402
from theora_amd import _lib  # noqa: E402

L = _lib.load()
rng = np.random.default_rng(3)
for name, w, h in (("720p", 1280, 720), ("1080p", 1920, 1088), ("4K", 3840, 2160)):
    st = theora_amd.State(w, h)
    for pli in range(3):
        g = st.planes[pli]
        base = rng.integers(40, 220, (g["nvfrags"], g["nhfrags"]))
        img = np.kron(base, np.ones((8, 8), np.int64)) + rng.integers(-40, 41, (g["height"], g["width"])) * (rng.random((g["height"], g["width"])) < 0.5)
        st.write_plane(0, pli, np.clip(img, 0, 255).astype(np.uint8))
    st.set_ref_idx(0, 0, 0)
    n = st.nfrags
    dc_qis = rng.integers(0, 64, n).astype(np.uint8)
    frag_qi = rng.integers(0, 64, n).astype(np.uint8)
    dcs = np.sort(rng.integers(1, 90, 64))[::-1].astype(np.int32).copy()
    shm = (-rng.integers(0, 6, 64)).astype(np.int32)
    for level in (2, 3, 4, 5, 7):
        best = 1e9
        for rep in range(3):
            theora_amd.synchronize()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            assert L.thip_state_postprocess(st.handle, level, dc_qis.ctypes.data, frag_qi.ctypes.data, dcs.ctypes.data, shm.ctypes.data) == 0
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        print("%-5s level %d: %7.3f ms per frame" % (name, level, 1e3 * best))
    st.close()
