"""thip_enc_mb_cost_maps: oc_mb_intra_satd, oc_mb_activity and oc_mb_activity_fast for every macro
block of a frame in one launch, against the oracle's restatement of analyze.c:1152-1251, 1360-1403 -- all three pixel formats,
sizes whose last super block is ragged, BASELINE.json's config-5 size, pictures with flat regions, texture and hard edges (the
three branches of the activity measure), bit-exact."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu

PF_420, PF_422, PF_444 = 0, 2, 3


def _picture(rng, w, h, fmt):
    """luma: flat patches, noise, straight edges in four directions, text-like dots; chroma: gradients + noise"""
    yy, xx = np.mgrid[0:h, 0:w]
    y = np.full((h, w), 90, np.int32)
    kind = rng.integers(0, 6, ((h + 15) // 16, (w + 15) // 16))
    k = np.kron(kind, np.ones((16, 16), np.int64))[:h, :w]
    noise = rng.integers(-40, 41, (h, w))
    y = np.where(k == 1, 128 + noise, y)
    y = np.where(k == 2, np.where((xx // 3) % 2 == 0, 30, 220), y)              # vertical stripes
    y = np.where(k == 3, np.where((yy // 5) % 2 == 0, 40, 200), y)              # horizontal
    y = np.where(k == 4, np.where((xx + yy) % 11 < 5, 20, 240), y)              # diagonal
    y = np.where(k == 5, np.where((xx - yy) % 13 < 6, 60, 180) + rng.integers(-3, 4, (h, w)), y)
    y = np.clip(y + rng.integers(-2, 3, (h, w)), 0, 255).astype(np.uint8)
    cw, ch = (w >> 1 if not (fmt & 1) else w), (h >> 1 if not (fmt & 2) else h)
    cb = np.clip((np.mgrid[0:ch, 0:cw][1] * 255 // max(cw - 1, 1)) + rng.integers(-20, 21, (ch, cw)), 0, 255).astype(np.uint8)
    cr = rng.integers(0, 256, (ch, cw)).astype(np.uint8)
    return [y, cb, cr]


@pytest.mark.parametrize("w,h,fmt", [(64, 48, PF_420), (16, 16, PF_420), (80, 112, PF_422), (176, 144, PF_444), (1280, 720, PF_420),
                                     (1920, 1088, PF_444), (336, 16, PF_420), (16, 272, PF_444)])
def test_cost_maps_match_the_oracle(hip, w, h, fmt):
    import torch
    rng = np.random.default_rng(w * 3 + h + fmt)
    planes = _picture(rng, w, h, fmt)
    want = oracle.mb_cost_maps(planes, w, h, fmt)
    # device planes with a pitch that is not the width, and not a multiple of 4 either
    dev = []
    for p in planes:
        t = torch.zeros((p.shape[0], p.shape[1] + 13), dtype=torch.uint8, device="cuda")
        t[:, :p.shape[1]] = torch.from_numpy(p).cuda()
        dev.append(t)
    got = hip.enc_mb_cost_maps(dev, w, h, fmt)
    names = ("intra_satd", "luma", "activity", "activity_fast")
    for nm, g, wnt in zip(names, got, want):
        g = g.cpu().numpy().view(np.uint32)
        assert np.array_equal(g, wnt), (nm, int((g != wnt).sum()), np.argwhere(g != wnt)[:4].tolist())
    # the picture exercises all three branches of oc_mb_activity
    act = want[2]
    if w * h >= 64 * 48:
        assert (act == 5 << 12).any() or (act < 5 << 12).any()
        assert (act >= 8 << 12).any()


def test_cost_map_argument_checks(hip):
    import ctypes as C
    import torch
    L = hip._lib.load()
    assert L.thip_enc_mb_count(176, 144) == 4 * 6 * 5
    assert L.thip_enc_mb_count(100, 144) == hip._lib.EINVAL
    t = torch.zeros((16, 16), dtype=torch.uint8, device="cuda")
    ptrs = (C.c_void_p * 3)(t.data_ptr(), t.data_ptr(), t.data_ptr())
    strides = (C.c_int32 * 3)(16, 16, 16)
    assert L.thip_enc_mb_cost_maps(ptrs, strides, 16, 16, 1, None, None, None, None) == hip._lib.EINVAL   # reserved format
    assert L.thip_enc_mb_cost_maps(ptrs, strides, 24, 16, 0, None, None, None, None) == hip._lib.EINVAL
    strides = (C.c_int32 * 3)(8, 16, 16)
    assert L.thip_enc_mb_cost_maps(ptrs, strides, 16, 16, 0, None, None, None, None) == hip._lib.EINVAL   # pitch below the width
    assert L.thip_enc_mb_cost_maps(None, strides, 16, 16, 0, None, None, None, None) == hip._lib.EFAULT
