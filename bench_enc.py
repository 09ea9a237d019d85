#!/usr/bin/env python3
"""bench_enc.py -- BASELINE.json config 5: the encoder block kernels on one MI355X.

1920x1088 4:4:4 (97 920 blocks per frame): oc_enc_fdct8x8 on residual blocks, oc_enc_frag_sad
/ satd / satd2 of frame f against frame f-1 at the nine sites of the square search pattern
(lib/mcenc.c:50-53), through the batched C ABI.  Prints one JSON line per kernel with the
GPU rate, the fraction of the HBM roofline (algorithmic bytes of SURVEY.md section 8d:
fDCT 256 B/block, SAD 132 B, SATD 136 B per candidate) and the CPU oracle's rate on a
bounded sample.  Secondary to bench.py (which carries the headline decode metric).
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0


def main():
    import torch
    import theora_amd
    import oracle

    torch.cuda.set_device(0)
    W, H, planes = 1920, 1088, 3
    rng = np.random.default_rng(7)
    # frame f-1 and frame f (f = f-1 shifted by (3,1) + noise), three planes stacked vertically
    prev = rng.integers(0, 256, (H * planes + 16, W + 16)).astype(np.uint8)
    cur = np.roll(prev, (1, 3), (0, 1))
    cur = np.clip(cur.astype(np.int32) + rng.integers(-6, 7, cur.shape), 0, 255).astype(np.uint8)
    stride = prev.shape[1]
    by, bx = np.mgrid[0:H * planes // 8, 0:W // 8]
    base = ((by * 8 + 8) * stride + bx * 8 + 8).reshape(-1).astype(np.int32)      # 97 920 blocks, 8-px margin
    sites = [(0, 0), (-1, -1), (0, -1), (1, -1), (-1, 0), (1, 0), (-1, 1), (0, 1), (1, 1)]
    src_offs = np.tile(base, len(sites))
    ref_offs = np.concatenate([base + dy * stride + dx for dx, dy in sites]).astype(np.int32)
    ref2_offs = (ref_offs + 1).astype(np.int32)
    nblk = base.size
    d_prev, d_cur = torch.from_numpy(prev).cuda(), torch.from_numpy(cur).cuda()
    d_so, d_ro, d_r2 = (torch.from_numpy(a).cuda() for a in (src_offs, ref_offs, ref2_offs))
    resid = rng.integers(-255, 256, (nblk, 64)).astype(np.int16)
    d_res = torch.from_numpy(resid).cuda()

    def timed(fn, reps=20):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / reps

    results = []
    # --- fDCT ---------------------------------------------------------------------------------
    t = timed(lambda: theora_amd.fdct8x8_batch(d_res))
    got = theora_amd.fdct8x8_batch(d_res).cpu().numpy().reshape(-1, 64)
    ncpu = 20000
    t0 = time.perf_counter()
    want = oracle.fdct8x8_batch(resid[:ncpu])
    tc = time.perf_counter() - t0
    assert np.array_equal(got[:ncpu], want)
    results.append(dict(kernel="oc_enc_fdct8x8", units=nblk, unit="blocks", seconds=t, bytes_per_unit=256, cpu_rate=ncpu / tc))
    # --- SAD / SATD / SATD2 ---------------------------------------------------------------------
    for op, bpu in (("sad", 132), ("satd", 136), ("satd2", 136 + 64), ("intra_satd", 72)):
        call = lambda: theora_amd.enc_metric_batch(op, d_cur, d_prev, stride, d_so, d_ro, d_r2, 0)   # noqa: E731
        t = timed(call)
        v, dc = call()
        ncpu = 30000
        t0 = time.perf_counter()
        wv, wdc = oracle.enc_metric_batch(op, cur, prev, stride, src_offs[:ncpu], ref_offs[:ncpu], ref2_offs[:ncpu], 0)
        tc = time.perf_counter() - t0
        assert np.array_equal(v.cpu().numpy()[:ncpu].view(np.uint32), wv)
        results.append(dict(kernel="oc_enc_frag_" + op, units=src_offs.size, unit="(block,candidate)", seconds=t,
                            bytes_per_unit=bpu, cpu_rate=ncpu / tc))
    for r in results:
        gbs = r["units"] * r["bytes_per_unit"] / r["seconds"] / 1e9
        print(json.dumps({
            "metric": r["kernel"] + " throughput", "value": round(r["units"] / r["seconds"] / 1e6, 1), "unit": "M%s/s" % r["unit"],
            "config": {"workload": "1920x1088 4:4:4, %d %s per call, 9-site square pattern" % (r["units"], r["unit"])},
            "ms_per_call": round(1e3 * r["seconds"], 4), "dtype": "u8/i16", "data": "synthetic", "bit_exact_vs_oracle": True,
            "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(gbs / HBM_PEAK_GBS, 4), "alg_bytes_per_unit": r["bytes_per_unit"]},
            "cpu_baseline": {"value": round(r["cpu_rate"] / 1e6, 3), "unit": "M%s/s" % r["unit"], "cores": 1, "kind": "port"}}))


if __name__ == "__main__":
    main()
