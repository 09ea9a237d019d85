"""Each accel-vtable slot on its own (batched C-ABI form) against the oracle, bit-exact."""
import ctypes as C

import numpy as np
import pytest

import oracle
from tests import util

pytestmark = pytest.mark.gpu


_KEEP = []


def dev(a):
    """Host array -> device tensor; the tensor is kept alive until the test module ends so
    raw data_ptr() values handed to the C ABI stay valid."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    _KEEP.append(t)
    if len(_KEEP) > 4096:
        del _KEEP[:2048]
    return t


def random_blocks(rng, n):
    """Coefficient blocks over the whole int16 range plus realistic sparse ones."""
    x = np.zeros((n, 64), np.int16)
    k = n // 3
    x[:k] = rng.integers(-32768, 32768, (k, 64))
    x[k:2 * k] = rng.integers(-600, 600, (k, 64)) * (rng.random((k, 64)) < 0.3)
    x[2 * k:] = rng.integers(-2000, 2000, (n - 2 * k, 64)) * (rng.random((n - 2 * k, 64)) < 0.08)
    x[-1] = 32767
    x[-2] = -32768
    return x


def test_idct8x8_all_last_zzi(hip):
    rng = np.random.default_rng(0)
    n = 65 * 300 + 17
    x = random_blocks(rng, n)
    lz = (np.arange(n) % 65).astype(np.int32)
    want = oracle.idct8x8_batch(x, lz)
    got = hip.idct8x8_batch(dev(x), dev(lz)).cpu().numpy().reshape(-1, 64)
    assert np.array_equal(want, got)
    # no last_zzi array == the full transform
    want = oracle.idct8x8_batch(x, None)
    got = hip.idct8x8_batch(dev(x), None).cpu().numpy().reshape(-1, 64)
    assert np.array_equal(want, got)


def test_idct_empty_batch(hip):
    from theora_amd import _lib
    assert _lib.load().thip_idct8x8_batch(1, 1, None, 0) == 0
    assert _lib.load().thip_idct8x8_batch(None, None, None, 4) == _lib.EFAULT
    assert _lib.load().thip_idct8x8_batch(1, 1, None, -1) == _lib.EINVAL


def _recon_case(hip, nsrc, seed):
    from theora_amd import _lib
    rng = np.random.default_rng(seed)
    stride, H = 400, 320
    n = 1500
    src = rng.integers(0, 256, (H, stride)).astype(np.uint8)
    dst0 = rng.integers(0, 256, (H, stride)).astype(np.uint8)
    res = rng.integers(-400, 400, (n, 64)).astype(np.int16)
    res[:50] = rng.integers(-32768, 32768, (50, 64))
    # non-overlapping destination blocks on an 8x8 grid; sources anywhere (unaligned)
    cells = rng.permutation((H // 8) * (stride // 8))[:n]
    dst_offs = ((cells // (stride // 8)) * 8 * stride + (cells % (stride // 8)) * 8).astype(np.int32)
    s1 = (rng.integers(0, H - 9, n) * stride + rng.integers(0, stride - 9, n)).astype(np.int32)
    dxy = rng.integers(0, 2, (n, 2))
    s2 = (s1 + dxy[:, 0] * stride + dxy[:, 1]).astype(np.int32)
    want = dst0.copy()
    L = oracle.lib()
    for i in range(n):
        d = want.ctypes.data + int(dst_offs[i])
        r = res[i].ctypes.data
        if nsrc == 0:
            L.orc_frag_recon_intra(d, stride, r)
        elif nsrc == 1:
            L.orc_frag_recon_inter(d, src.ctypes.data + int(s1[i]), stride, r)
        else:
            L.orc_frag_recon_inter2(d, src.ctypes.data + int(s1[i]), src.ctypes.data + int(s2[i]), stride, r)
    d_dst, d_src = dev(dst0), dev(src)
    rc = _lib.load().thip_frag_recon_batch(d_dst.data_ptr(), d_src.data_ptr(), stride, nsrc,
                                           dev(dst_offs).data_ptr(), dev(s1).data_ptr(), dev(s2).data_ptr(),
                                           dev(res).data_ptr(), n)
    assert rc == 0
    assert np.array_equal(want, d_dst.cpu().numpy())


@pytest.mark.parametrize("nsrc", [0, 1, 2])
def test_frag_recon_slots(hip, nsrc):
    _recon_case(hip, nsrc, 20 + nsrc)


def test_frag_copy_list(hip):
    from theora_amd import _lib
    rng = np.random.default_rng(3)
    st = oracle.State(176, 144)
    stride, H = 176, 144
    src = rng.integers(0, 256, (H, stride)).astype(np.uint8)
    dst0 = rng.integers(0, 256, (H, stride)).astype(np.uint8)
    nh, nv = 22, 18
    offs = (np.arange(nv)[:, None] * 8 * stride + np.arange(nh)[None, :] * 8).reshape(-1).astype(np.int32)
    fragis = rng.permutation(nh * nv)[:200].astype(np.int32)
    want = dst0.copy()
    for f in fragis:
        y, x = divmod(int(offs[f]), stride)
        want[y:y + 8, x:x + 8] = src[y:y + 8, x:x + 8]
    d_dst = dev(dst0)
    rc = _lib.load().thip_frag_copy_list_batch(d_dst.data_ptr(), dev(src).data_ptr(), stride,
                                               dev(fragis).data_ptr(), fragis.size, dev(offs).data_ptr())
    assert rc == 0
    assert np.array_equal(want, d_dst.cpu().numpy())


@pytest.mark.parametrize("flimit", [1, 2, 4, 15, 63, 127])
@pytest.mark.parametrize("density", [0.0, 0.1, 0.5, 0.9, 1.0])
def test_loop_filter_plane(hip, flimit, density):
    """The cell decomposition against the reference's sequential raster order
    (state.c:1055-1105) on random pixels and random coded masks, whole plane and row ranges."""
    from theora_amd import _lib
    rng = np.random.default_rng(flimit * 10 + int(density * 10))
    w, h = 176, 144
    st = oracle.State(w, h)
    st.set_ref_idx(0, 0, 0)
    nh, nv = 22, 18
    for (y0, y1) in [(0, nv), (0, 7), (7, nv), (3, 4)]:
        pix = rng.integers(0, 256, (h, w)).astype(np.uint8)
        coded = (rng.random(nh * nv) < density).astype(np.uint8)
        st.coded[:] = 0
        st.coded[:nh * nv] = coded
        st.set_plane(oracle.FRAME_SELF, 0, pix)
        st.loop_filter_rows(flimit, oracle.FRAME_SELF, 0, y0, y1)
        want = st.get_plane(oracle.FRAME_SELF, 0)
        d = dev(pix)
        rc = _lib.load().thip_loop_filter_plane(d.data_ptr(), w, nh, nv, dev(coded).data_ptr(), flimit, y0, y1)
        assert rc == 0
        got = d.cpu().numpy()
        assert np.array_equal(want, got), (flimit, density, y0, y1, int((want != got).sum()))


def test_loop_filter_init_table(hip):
    from theora_amd import _lib
    for fl in list(range(0, 20)) + [31, 63, 64, 65, 100, 126, 127]:
        bv = np.zeros(256, np.int8)
        _lib.load().thip_loop_filter_init(bv.ctypes.data, fl)
        assert np.array_equal(bv, oracle.loop_filter_bv(fl)), fl


def _enc_planes(rng):
    stride, H = 256, 128
    src = rng.integers(0, 256, (H, stride)).astype(np.uint8)
    ref = np.clip(src.astype(np.int32) + rng.integers(-20, 21, (H, stride)), 0, 255).astype(np.uint8)
    ref[:32] = rng.integers(0, 256, (32, stride))
    n = 3000
    so = (rng.integers(0, H - 8, n) * stride + rng.integers(0, stride - 8, n)).astype(np.int32)
    ro = (rng.integers(0, H - 9, n) * stride + rng.integers(0, stride - 9, n)).astype(np.int32)
    dxy = rng.integers(0, 2, (n, 2))
    r2 = (ro + dxy[:, 0] * stride + dxy[:, 1]).astype(np.int32)
    return stride, src, ref, so, ro, r2


@pytest.mark.parametrize("op", ["sad", "sad_thresh", "sad2_thresh", "intra_sad", "satd", "satd2", "intra_satd",
                                "ssd"])
def test_enc_metrics(hip, op):
    rng = np.random.default_rng(sum(map(ord, op)))
    stride, src, ref, so, ro, r2 = _enc_planes(rng)
    for thresh in ([0, 300, 2000, 1 << 30] if "thresh" in op else [0]):
        want, want_dc = oracle.enc_metric_batch(op, src, ref, stride, so, ro, r2, thresh)
        got, got_dc = hip.enc_metric_batch(op, dev(src), dev(ref), stride, dev(so), dev(ro), dev(r2), thresh)
        assert np.array_equal(want, got.cpu().numpy().view(np.uint32)), (op, thresh)
        if "satd" in op:
            assert np.array_equal(want_dc, got_dc.cpu().numpy()), op


@pytest.mark.parametrize("op", ["sad", "satd", "satd_every_lane_its_own_source_block"])
def test_enc_metric_sites(hip, op):
    """The motion-search form (thip_enc_frag_metric_sites_batch): every block against candidate positions in {-1,0,1}^2 around
    one reference position, results candidate-major -- against the oracle called once per (block, candidate) as the
    reference calls oc_enc_frag_sad / oc_enc_frag_satd from mcenc.c:267-330.  The full square pattern in the reference's
    order, a subset in another order, a single candidate; reference positions of every byte alignment; flat, saturated and
    random pictures (the largest coefficients a block can have)."""
    from theora_amd import _lib
    # (SATD: k_enc_sites_satd, the source block shared by a block's three lanes through LDS, is the default since round 6; option
    #  enc_sites_lds = 0 is k_enc_sites<SATD>.  Block counts that end inside a wave's 21 blocks and inside a work group's 84.)
    lds = 0 if op.startswith("satd_") else 1
    op = op.split("_")[0]
    _lib.load().thip_set_option(b"enc_sites_lds", lds)
    rng = np.random.default_rng(11 + len(op))
    stride, H = 272, 136
    src = rng.integers(0, 256, (H, stride)).astype(np.uint8)
    ref = np.clip(src.astype(np.int32) + rng.integers(-20, 21, (H, stride)), 0, 255).astype(np.uint8)
    ref[:40] = rng.integers(0, 256, (40, stride))
    src[40:56] = 255
    ref[40:56] = 0                                   # every difference +255: the largest DC
    src[56:72] = (np.indices((16, stride)).sum(0) & 1) * 255
    ref[56:72] = 255 - src[56:72]                    # checkerboard of +-255: the largest AC coefficient
    n = 4000
    so = (rng.integers(0, H - 8, n) * stride + rng.integers(0, stride - 8, n)).astype(np.int32)
    ro = (rng.integers(1, H - 9, n) * stride + rng.integers(1, stride - 9, n)).astype(np.int32)
    full = [(0, 0), (-1, -1), (0, -1), (1, -1), (-1, 0), (1, 0), (-1, 1), (0, 1), (1, 1)]   # mcenc.c:50-53
    for sites in (full, [(1, 1), (-1, 0), (0, -1), (0, 1)], [(-1, 1)], [(1, 0)]):
        got, got_dc = hip.enc_metric_sites_batch(op, dev(src), dev(ref), stride, dev(so), dev(ro), sites)
        for c, (dx, dy) in enumerate(sites):
            want, want_dc = oracle.enc_metric_batch(op, src, ref, stride, so, (ro + dy * stride + dx).astype(np.int32), ro, 0)
            assert np.array_equal(want, got[c].cpu().numpy().view(np.uint32)), (op, sites, c)
            if op == "satd":
                assert np.array_equal(want_dc, got_dc[c].cpu().numpy()), (op, sites, c)
    for m in (1, 20, 22, 83, 85):
        got, got_dc = hip.enc_metric_sites_batch(op, dev(src), dev(ref), stride, dev(so[:m]), dev(ro[:m]), full)
        for c, (dx, dy) in enumerate(full):
            want, want_dc = oracle.enc_metric_batch(op, src, ref, stride, so[:m], (ro[:m] + dy * stride + dx).astype(np.int32), ro[:m], 0)
            assert np.array_equal(want, got[c].cpu().numpy().view(np.uint32)), (op, m, c)
    _lib.load().thip_set_option(b"enc_sites_lds", 1)
    # argument errors: a position outside the pattern, a position twice, too many candidates, an operation without this form
    L = _lib.load()
    o = dev(np.zeros(9 * n, np.int32))

    def call(opc, dx, dy):
        dx, dy = np.array(dx, np.int8), np.array(dy, np.int8)
        return L.thip_enc_frag_metric_sites_batch(opc, o.data_ptr(), None, dev(src).data_ptr(), dev(ref).data_ptr(), stride, dev(so).data_ptr(),
                                                  dev(ro).data_ptr(), dx.ctypes.data, dy.ctypes.data, len(dx), n)
    assert call(_lib.ENC_OPS["sad"], [2], [0]) == _lib.EINVAL
    assert call(_lib.ENC_OPS["sad"], [0, 0], [1, 1]) == _lib.EINVAL
    assert call(_lib.ENC_OPS["sad"], [0] * 10, [0] * 10) == _lib.EINVAL
    assert call(_lib.ENC_OPS["ssd"], [0], [0]) == _lib.EINVAL


@pytest.mark.parametrize("lanes", [2, 3])
@pytest.mark.parametrize("op", ["satd2", "sad2_thresh"])
def test_enc_metric_halfpel_sites(hip, op, lanes):
    """The half-pel refinement's form (thip_enc_frag_metric_halfpel_batch): every block against the half-pel vectors around its
    whole-pel vector -- against the oracle's restatement of the reference's call pattern (oc_mcenc_ysatd_halfpel_mbrefine,
    mcenc.c:606-657: mvoffset0 / mvoffset1 from the signs of the vector, then oc_enc_frag_satd2 / oc_enc_frag_sad2_thresh on the
    two blocks).  Whole-pel vectors of both signs and zero on both axes (which of the two blocks gets the step, and whether a
    diagonal site pairs {(0,0),(dx,dy)} or {(dx,0),(0,dy)}, follows from them); all eight sites in the reference's order, subsets
    in other orders, single sites; reference positions of every byte alignment; saturated and checkerboard pictures."""
    rng = np.random.default_rng(23 + len(op))
    from theora_amd import _lib
    with util.options(_lib.load(), enc_halfpel_lanes=lanes):     # 2: a lane per side, four sites each (default); 3: a lane per dx
        _halfpel_sites_body(hip, op, rng)


def _halfpel_sites_body(hip, op, rng):
    from theora_amd import _lib
    stride, H = 272, 136
    src = rng.integers(0, 256, (H, stride)).astype(np.uint8)
    ref = np.clip(src.astype(np.int32) + rng.integers(-20, 21, (H, stride)), 0, 255).astype(np.uint8)
    ref[:40] = rng.integers(0, 256, (40, stride))
    src[40:56] = 255
    ref[40:56] = 0
    src[56:72] = (np.indices((16, stride)).sum(0) & 1) * 255
    ref[56:72] = 255 - src[56:72]
    ref[72:88] = (np.indices((16, stride))[1] & 1) * 255          # columns alternate: every horizontal average is 127
    n = 6000
    so = (rng.integers(0, H - 8, n) * stride + rng.integers(0, stride - 8, n)).astype(np.int32)
    ro = (rng.integers(1, H - 9, n) * stride + rng.integers(1, stride - 9, n)).astype(np.int32)
    vx = rng.integers(-3, 4, n)
    vy = rng.integers(-3, 4, n)
    vx[:500] = 0
    vy[250:750] = 0
    vecs = ((vx & 0xFF) | (vy << 8)).astype(np.int16)
    full = [(-1, -1), (0, -1), (1, -1), (-1, 0), (1, 0), (-1, 1), (0, 1), (1, 1)]     # OC_SQUARE_SITES[0], mcenc.c:50-56
    for sites in (full, [(1, 1), (-1, 0), (0, -1), (0, 1)], [(-1, 1)], [(0, 1)], [(1, 0), (1, -1)]):
        got, got_dc = hip.enc_metric_halfpel_batch(op, dev(src), dev(ref), stride, dev(so), dev(ro), dev(vecs), sites)
        want, want_dc = oracle.enc_halfpel_sites(op, src, ref, stride, so, ro, vecs, sites)
        assert np.array_equal(want, got.cpu().numpy().view(np.uint32)), (op, sites, np.argwhere(want != got.cpu().numpy().view(np.uint32))[:4])
        if op == "satd2":
            assert np.array_equal(want_dc, got_dc.cpu().numpy()), (op, sites)
    # the full-size call of BASELINE.json config 5: every block of a 1920x1088 4:4:4 frame, all eight sites
    W5, H5 = 1920 + 16, 3 * 1088 + 16
    prev = rng.integers(0, 256, (H5, W5)).astype(np.uint8)
    cur = np.clip(np.roll(prev, (1, 3), (0, 1)).astype(np.int32) + rng.integers(-6, 7, prev.shape), 0, 255).astype(np.uint8)
    by, bx = np.mgrid[0:3 * 1088 // 8, 0:1920 // 8]
    base = ((by * 8 + 8) * W5 + bx * 8 + 8).reshape(-1).astype(np.int32)
    v5 = ((rng.integers(-2, 3, base.size) & 0xFF) | (rng.integers(-2, 3, base.size) << 8)).astype(np.int16)
    got, got_dc = hip.enc_metric_halfpel_batch(op, dev(cur), dev(prev), W5, dev(base), dev(base), dev(v5), full)
    sel = rng.integers(0, base.size, 30000)
    want, want_dc = oracle.enc_halfpel_sites(op, cur, prev, W5, base[sel], base[sel], v5[sel], full)
    assert np.array_equal(want, got.cpu().numpy().view(np.uint32)[:, sel]), op
    if op == "satd2":
        assert np.array_equal(want_dc, got_dc.cpu().numpy()[:, sel]), op
    # argument errors: the centre is not a half-pel site, a site twice, nine sites, an operation without this form
    L = _lib.load()
    o = dev(np.zeros(8 * n, np.int32))

    def call(opc, dx, dy):
        dx, dy = np.array(dx, np.int8), np.array(dy, np.int8)
        return L.thip_enc_frag_metric_halfpel_batch(opc, o.data_ptr(), None, dev(src).data_ptr(), dev(ref).data_ptr(), stride, dev(so).data_ptr(),
                                                    dev(ro).data_ptr(), dev(vecs).data_ptr(), dx.ctypes.data, dy.ctypes.data, len(dx), n)
    assert call(_lib.ENC_OPS[op], [0], [0]) == _lib.EINVAL
    assert call(_lib.ENC_OPS[op], [1, 1], [1, 1]) == _lib.EINVAL
    assert call(_lib.ENC_OPS[op], [1] * 9, [0] * 9) == _lib.EINVAL
    assert call(_lib.ENC_OPS["satd"], [1], [0]) == _lib.EINVAL
    assert call(_lib.ENC_OPS[op], [1], [0]) == 0


@pytest.mark.parametrize("lanes", [4, 1])
def test_enc_fdct(hip, lanes):
    """oc_enc_fdct8x8 (fdct.c:128-150) in both layouts: four lanes per block (k_enc_fdct4, the default since round 6) and one block per
    lane; block counts that end inside a wave's sixteen blocks and inside a work group."""
    from theora_amd import _lib
    rng = np.random.default_rng(8)
    n = 20000 + 7
    x = rng.integers(-255, 256, (n, 64)).astype(np.int16)
    x[:100] = 0
    x[100:200] = 255
    x[200:300] = -255
    x[300:1000] = rng.integers(-8160, 8161, (700, 64))    # beyond the residual range, still defined
    want = oracle.fdct8x8_batch(x)
    _lib.load().thip_set_option(b"enc_fdct_lanes", lanes)
    try:
        got = hip.fdct8x8_batch(dev(x)).cpu().numpy().reshape(-1, 64)
        assert np.array_equal(want, got)
        for m in (1, 15, 17, 63, 65):
            got = hip.fdct8x8_batch(dev(x[:m])).cpu().numpy().reshape(-1, 64)
            assert np.array_equal(want[:m], got)
    finally:
        _lib.load().thip_set_option(b"enc_fdct_lanes", 4)


def test_enc_sub_copy2_border_ssd(hip):
    from theora_amd import _lib
    import torch
    rng = np.random.default_rng(9)
    stride, src, ref, so, ro, r2 = _enc_planes(rng)
    n = so.size
    L, O = _lib.load(), oracle.lib()
    d_src, d_ref = dev(src), dev(ref)
    # sub / sub_128
    for use_ref in (True, False):
        want = np.empty((n, 64), np.int16)
        for i in range(n):
            if use_ref:
                O.orc_enc_frag_sub(want[i].ctypes.data, src.ctypes.data + int(so[i]), ref.ctypes.data + int(ro[i]), stride)
            else:
                O.orc_enc_frag_sub_128(want[i].ctypes.data, src.ctypes.data + int(so[i]), stride)
        got = torch.empty((n, 64), dtype=torch.int16, device="cuda")
        rc = L.thip_enc_frag_sub_batch(got.data_ptr(), d_src.data_ptr(), d_ref.data_ptr(), stride,
                                       dev(so).data_ptr(), dev(ro).data_ptr() if use_ref else None, n)
        assert rc == 0 and np.array_equal(want, got.cpu().numpy())
    # border ssd
    masks = rng.integers(-2 ** 63, 2 ** 63, n, dtype=np.int64)
    masks[:4] = [0, -1, 1, 1 << 62]
    want = np.array([O.orc_enc_frag_border_ssd(src.ctypes.data + int(so[i]), ref.ctypes.data + int(ro[i]), stride,
                                               int(masks[i])) for i in range(n)], np.uint32)
    got = torch.empty(n, dtype=torch.int32, device="cuda")
    rc = L.thip_enc_frag_border_ssd_batch(got.data_ptr(), d_src.data_ptr(), d_ref.data_ptr(), stride,
                                          dev(so).data_ptr(), dev(ro).data_ptr(), dev(masks).data_ptr(), n)
    assert rc == 0 and np.array_equal(want, got.cpu().numpy().view(np.uint32))
    # copy2 into disjoint 8x8 cells
    H = src.shape[0]
    cells = rng.permutation((H // 8) * (stride // 8))[:400]
    do = ((cells // (stride // 8)) * 8 * stride + (cells % (stride // 8)) * 8).astype(np.int32)
    want = np.zeros_like(src)
    for i in range(400):
        O.orc_enc_frag_copy2(want.ctypes.data + int(do[i]), ref.ctypes.data + int(ro[i]), ref.ctypes.data + int(r2[i]), stride)
    d_dst = dev(np.zeros_like(src))
    rc = L.thip_enc_frag_copy2_batch(d_dst.data_ptr(), d_ref.data_ptr(), stride, dev(do).data_ptr(),
                                     dev(ro[:400]).data_ptr(), dev(r2[:400]).data_ptr(), 400)
    assert rc == 0 and np.array_equal(want, d_dst.cpu().numpy())


def test_enc_quantize(hip):
    """oc_enc_quantize_c + oc_iquant_init (enquant.c:183-248) for every legal step size."""
    from theora_amd import _lib
    import torch
    rng = np.random.default_rng(10)
    L = _lib.load()
    n = 5000
    for trial in range(6):
        dq = rng.integers(8, 4097, 64).astype(np.uint16) if trial else np.full(64, 8, np.uint16)
        if trial == 1:
            dq[:] = 4096
        dct = rng.integers(-32768 if trial > 3 else -9000, 32768 if trial > 3 else 9000, (n, 64)).astype(np.int16)
        want_q, want_nz = oracle.quantize_batch(dct, dq)
        q = torch.empty((n, 64), dtype=torch.int16, device="cuda")
        nz = torch.empty(n, dtype=torch.int32, device="cuda")
        assert L.thip_enc_quantize_batch(q.data_ptr(), nz.data_ptr(), dev(dct).data_ptr(), dev(dq).data_ptr(), n) == 0
        assert np.array_equal(want_q, q.cpu().numpy()) and np.array_equal(want_nz, nz.cpu().numpy()), trial


def test_enc_chain_on_a_caller_stream(hip):
    """thip_set_batch_stream(stream, 0): residual -> forward DCT -> quantiser enqueued back to back
    on a caller-owned stream, no host wait in between, one synchronisation at the end -- the result
    is what the three C slots give one after the other (encfrag.c:21, fdct.c:128, enquant.c:219)."""
    from theora_amd import _lib
    import torch
    rng = np.random.default_rng(12)
    stride, src, ref, so, ro, _ = _enc_planes(rng)
    n = so.size
    L, O = _lib.load(), oracle.lib()
    dq = rng.integers(8, 600, 64).astype(np.uint16)
    want_res = np.empty((n, 64), np.int16)
    for i in range(n):
        O.orc_enc_frag_sub(want_res[i].ctypes.data, src.ctypes.data + int(so[i]), ref.ctypes.data + int(ro[i]), stride)
    want_dct = oracle.fdct8x8_batch(want_res)
    want_q, want_nz = oracle.quantize_batch(want_dct, dq)
    s = torch.cuda.Stream()
    d_src, d_ref, d_so, d_ro, d_dq = dev(src), dev(ref), dev(so), dev(ro), dev(dq)
    res = torch.empty((n, 64), dtype=torch.int16, device="cuda")
    dct = torch.empty_like(res)
    q = torch.empty_like(res)
    nz = torch.empty(n, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    assert L.thip_set_batch_stream(s.cuda_stream, 0) == 0
    try:
        assert L.thip_enc_frag_sub_batch(res.data_ptr(), d_src.data_ptr(), d_ref.data_ptr(), stride, d_so.data_ptr(),
                                         d_ro.data_ptr(), n) == 0
        assert L.thip_enc_fdct8x8_batch(dct.data_ptr(), res.data_ptr(), n) == 0
        assert L.thip_enc_quantize_batch(q.data_ptr(), nz.data_ptr(), dct.data_ptr(), d_dq.data_ptr(), n) == 0
        s.synchronize()
    finally:
        L.thip_set_batch_stream(None, 1)
    assert np.array_equal(q.cpu().numpy(), want_q) and np.array_equal(nz.cpu().numpy(), want_nz)
    assert np.array_equal(dct.cpu().numpy(), want_dct)


def test_enc_enquant_table_slots_and_table_fed_quantiser(hip):
    """oc_enc_enquant_table_init / _fixup (enquant.c:194-217) as host slots, and oc_enc_quantize fed with the
    table they build (encint.h:316-320): equal to the oracle and to the batch that derives the table itself."""
    import ctypes as C
    import torch
    from theora_amd import _lib
    L, O = _lib.load(), oracle.lib()
    rng = np.random.default_rng(21)
    size, align = C.c_size_t(), C.c_int()
    L.thip_enc_opt_data(C.byref(size), C.byref(align))
    assert size.value == 256 and align.value == 16
    for trial in range(5):
        dq = (np.full(64, [8, 4096][trial], np.uint16) if trial < 2 else rng.integers(8, 4097, 64).astype(np.uint16))
        tab = np.zeros(128, np.int16)
        want_tab = np.zeros(128, np.int16)
        L.thip_enc_enquant_table_init(tab.ctypes.data, dq.ctypes.data)
        O.orc_enc_enquant_table_init(want_tab.ctypes.data, dq.ctypes.data)
        assert np.array_equal(tab, want_tab), trial
        n = 4000
        dct = rng.integers(-32768, 32768, (n, 64)).astype(np.int16)
        want_q, want_nz = oracle.quantize_batch(dct, dq)
        q = torch.empty((n, 64), dtype=torch.int16, device="cuda")
        nz = torch.empty(n, dtype=torch.int32, device="cuda")
        assert L.thip_enc_quantize_tab_batch(q.data_ptr(), nz.data_ptr(), dev(dct).data_ptr(), dev(dq).data_ptr(),
                                             dev(tab).data_ptr(), n) == 0
        assert np.array_equal(want_q, q.cpu().numpy()) and np.array_equal(want_nz, nz.cpu().numpy()), trial
    # fixup: the tables of qii > 0 become copies of qii = 0's first entry pair (enquant.c:210-217 copies one oc_iquant)
    tabs = [[[np.full(128, 100 * p + 10 * q + t, np.int16) for t in range(2)] for q in range(3)] for p in range(3)]
    arr = (C.c_void_p * 18)(*[tabs[p][q][t].ctypes.data for p in range(3) for q in range(3) for t in range(2)])
    L.thip_enc_enquant_table_fixup(arr, 3)
    for p in range(3):
        for q in range(3):
            for t in range(2):
                want = np.full(128, 100 * p + 10 * q + t, np.int16)
                if q > 0:
                    want[:2] = 100 * p + t
                assert np.array_equal(tabs[p][q][t], want)


def test_enc_single_block_slots(hip):
    """The oc_enc_opt_vtable slots with the reference's own signatures (host pointers, one block per call:
    thip_enc1_*, what oc_enc_accel_init_hip binds) against their C originals as restated by the oracle."""
    import ctypes as C
    from theora_amd import _lib
    L, O = _lib.load(), oracle.lib()
    rng = np.random.default_rng(22)
    stride = 40
    for trial in range(12):
        src = rng.integers(0, 256, (16, stride)).astype(np.uint8)
        ref = np.clip(src.astype(np.int32) + rng.integers(-30, 31, src.shape), 0, 255).astype(np.uint8)
        ref2 = rng.integers(0, 256, (16, stride)).astype(np.uint8)
        if trial == 0:
            src[:] = 255
            ref[:] = 0
        ps, pr, pr2 = src.ctypes.data + 3 * stride + 5, ref.ctypes.data + 2 * stride + 7, ref2.ctypes.data + 4 * stride + 1
        a, b = np.zeros(64, np.int16), np.zeros(64, np.int16)
        L.thip_enc1_frag_sub(a.ctypes.data, ps, pr, stride)
        O.orc_enc_frag_sub(b.ctypes.data, ps, pr, stride)
        assert np.array_equal(a, b)
        L.thip_enc1_frag_sub_128(a.ctypes.data, ps, stride)
        O.orc_enc_frag_sub_128(b.ctypes.data, ps, stride)
        assert np.array_equal(a, b)
        assert L.thip_enc1_frag_sad(ps, pr, stride) == O.orc_enc_frag_sad(ps, pr, stride)
        for th in (0, 500, 1 << 30):
            assert L.thip_enc1_frag_sad_thresh(ps, pr, stride, th) == O.orc_enc_frag_sad_thresh(ps, pr, stride, th)
            assert L.thip_enc1_frag_sad2_thresh(ps, pr, pr2, stride, th) == O.orc_enc_frag_sad2_thresh(ps, pr, pr2, stride, th)
        assert L.thip_enc1_frag_intra_sad(ps, stride) == O.orc_enc_frag_intra_sad(ps, stride)
        d1, d2 = C.c_int(), C.c_int()
        assert L.thip_enc1_frag_satd(C.byref(d1), ps, pr, stride) == O.orc_enc_frag_satd(C.byref(d2), ps, pr, stride)
        assert d1.value == d2.value
        assert L.thip_enc1_frag_satd2(C.byref(d1), ps, pr, pr2, stride) == O.orc_enc_frag_satd2(C.byref(d2), ps, pr, pr2, stride)
        assert d1.value == d2.value
        assert L.thip_enc1_frag_intra_satd(C.byref(d1), ps, stride) == O.orc_enc_frag_intra_satd(C.byref(d2), ps, stride)
        assert d1.value == d2.value
        assert L.thip_enc1_frag_ssd(ps, pr, stride) == O.orc_enc_frag_ssd(ps, pr, stride)
        mask = int(rng.integers(-2 ** 63, 2 ** 63, dtype=np.int64))
        assert L.thip_enc1_frag_border_ssd(ps, pr, stride, mask) == O.orc_enc_frag_border_ssd(ps, pr, stride, mask)
        o1, o2 = np.zeros((8, stride), np.uint8), np.zeros((8, stride), np.uint8)
        L.thip_enc1_frag_copy2(o1.ctypes.data + 2, pr, pr2, stride)
        O.orc_enc_frag_copy2(o2.ctypes.data + 2, pr, pr2, stride)
        assert np.array_equal(o1, o2)
        x = rng.integers(-255, 256, 64).astype(np.int16)
        L.thip_enc1_fdct8x8(a.ctypes.data, x.ctypes.data)
        O.orc_enc_fdct8x8(b.ctypes.data, x.ctypes.data)
        assert np.array_equal(a, b)
        dq = rng.integers(8, 2000, 64).astype(np.uint16)
        tab = np.zeros(128, np.int16)
        L.thip_enc_enquant_table_init(tab.ctypes.data, dq.ctypes.data)
        qa, qb = np.zeros(64, np.int16), np.zeros(64, np.int16)
        assert L.thip_enc1_quantize(qa.ctypes.data, a.ctypes.data, dq.ctypes.data, tab.ctypes.data) == \
            O.orc_enc_quantize(qb.ctypes.data, b.ctypes.data, dq.ctypes.data, tab.ctypes.data)
        assert np.array_equal(qa, qb)
        res = rng.integers(-300, 301, 64).astype(np.int16)
        L.thip_enc1_frag_recon_intra(o1.ctypes.data + 1, stride, res.ctypes.data)
        O.orc_frag_recon_intra(o2.ctypes.data + 1, stride, res.ctypes.data)
        assert np.array_equal(o1, o2)
        L.thip_enc1_frag_recon_inter(o1.ctypes.data + 9, pr, stride, res.ctypes.data)
        O.orc_frag_recon_inter(o2.ctypes.data + 9, pr, stride, res.ctypes.data)
        assert np.array_equal(o1, o2)


def test_enc_kernels_at_config5_size(hip):
    """BASELINE.json config 5 at its stated size: 1920x1088 4:4:4 (97 920 blocks per frame), the forward DCT
    and the quantiser over every block, SAD / SATD / SATD2 / intra-SATD over the 9-site square pattern of
    mcenc.c:50-53 (881 280 (block, candidate) pairs), all on the GPU; the oracle checks a 30 000-element
    prefix AND a strided sample across the whole batch of each."""
    import torch
    W, H, planes = 1920, 1088, 3
    rng = np.random.default_rng(7)
    prev = rng.integers(0, 256, (H * planes + 16, W + 16)).astype(np.uint8)
    cur = np.roll(prev, (1, 3), (0, 1))
    cur = np.clip(cur.astype(np.int32) + rng.integers(-6, 7, cur.shape), 0, 255).astype(np.uint8)
    stride = prev.shape[1]
    by, bx = np.mgrid[0:H * planes // 8, 0:W // 8]
    base = ((by * 8 + 8) * stride + bx * 8 + 8).reshape(-1).astype(np.int32)
    sites = [(0, 0), (-1, -1), (0, -1), (1, -1), (-1, 0), (1, 0), (-1, 1), (0, 1), (1, 1)]
    src_offs = np.tile(base, len(sites))
    ref_offs = np.concatenate([base + dy * stride + dx for dx, dy in sites]).astype(np.int32)
    ref2_offs = (ref_offs + 1).astype(np.int32)
    nblk = base.size
    assert nblk == 3 * 32640 and src_offs.size == 9 * nblk
    resid = rng.integers(-255, 256, (nblk, 64)).astype(np.int16)
    sample = np.unique(np.concatenate([np.arange(30000), np.arange(0, nblk, 37), [nblk - 1]]))
    got = hip.fdct8x8_batch(dev(resid)).cpu().numpy().reshape(-1, 64)
    assert np.array_equal(got[sample], oracle.fdct8x8_batch(resid[sample]))
    dq = np.clip(np.arange(64) * 3 + 16, 8, 4096).astype(np.uint16)
    gq, gnz = hip.enc_quantize_batch(dev(got), dev(dq))
    wq, wnz = oracle.quantize_batch(got[sample], dq)
    assert np.array_equal(gq.cpu().numpy().reshape(-1, 64)[sample], wq) and np.array_equal(gnz.cpu().numpy()[sample], wnz)
    d_prev, d_cur = dev(prev), dev(cur)
    d_so, d_ro, d_r2 = dev(src_offs), dev(ref_offs), dev(ref2_offs)
    msample = np.unique(np.concatenate([np.arange(30000), np.arange(0, src_offs.size, 97), [src_offs.size - 1]]))
    for op in ("sad", "satd", "satd2", "intra_satd"):
        v, dc = hip.enc_metric_batch(op, d_cur, d_prev, stride, d_so, d_ro, d_r2, 0)
        wv, wdc = oracle.enc_metric_batch(op, cur, prev, stride, src_offs[msample], ref_offs[msample], ref2_offs[msample], 0)
        assert np.array_equal(v.cpu().numpy().view(np.uint32)[msample], wv), op
        if "satd" in op:
            assert np.array_equal(dc.cpu().numpy()[msample], wdc), op
        if op in ("sad", "satd"):   # the same 881 280 pairs through the motion-search form: one launch, candidate-major = the order above
            d_base = dev(base)
            sv, sdc = hip.enc_metric_sites_batch(op, d_cur, d_prev, stride, d_base, d_base, sites)
            assert torch.equal(sv.reshape(-1), v), op
            if op == "satd":
                assert torch.equal(sdc.reshape(-1), dc), op


def _dc_case(rng, w, h, fmt, density, refmix, big):
    """A frame's worth of coded flags, reference indices and token DC values in an oracle state."""
    ost = oracle.State(w, h, fmt)
    n = ost.nfrags
    ost.coded[:] = rng.random(n) < density
    if refmix == "one":
        ost.refi[:] = 2
    elif refmix == "runs":      # long horizontal runs of one reference: the predictor's common cases
        ost.refi[:] = np.repeat(rng.integers(0, 3, n // 7 + 1), 7)[:n]
    else:                       # every neighbour combination, many fragments with no usable neighbour
        ost.refi[:] = rng.integers(0, 3, n)
    lim = 32767 if big else 200
    ost.dc[:] = rng.integers(-lim, lim + 1, n)
    return ost


@pytest.mark.parametrize("w,h,fmt", [(16, 16, 0), (16, 272, 3), (336, 16, 0), (176, 144, 0), (80, 112, 2), (1280, 720, 0),
                                     (3840, 2160, 0)])
def test_dc_unpredict_plane_slot(hip, w, h, fmt):
    """oc_dec_dc_unpredict_mcu_plane (the oc_dec_opt_vtable slot, decode.c:1392-1500) as a device wavefront:
    every plane of random frames -- sparse and dense coded masks, one / runs of / random reference frames (the
    last makes most fragments fall back to pred_last, the dependency that breaks the wavefront), small values
    and full-range ones (16-bit wrap, the 3-neighbour outlier clamp) -- against the oracle, in place."""
    import torch
    from theora_amd import _lib
    L = _lib.load()
    rng = np.random.default_rng(w * 3 + h + fmt)
    cases = [(0.95, "one", False), (0.6, "runs", False), (0.5, "random", True), (0.08, "random", False), (1.0, "random", True)]
    if w >= 1280:
        cases = cases[1:3]
    for density, refmix, big in cases:
        ost = _dc_case(rng, w, h, fmt, density, refmix, big)
        tokens = ost.dc.copy()
        flags = (ost.coded.astype(np.uint8) | (ost.refi.astype(np.uint8) << 1)) * ost.coded.astype(np.uint8)
        ost.dc_unpredict()
        for pli in range(3):
            g = ost.planes[pli]
            lo, hi = g["froffset"], g["froffset"] + g["nfrags"]
            d = torch.from_numpy(tokens[lo:hi].copy()).cuda()
            f = torch.from_numpy(flags[lo:hi].copy()).cuda()
            assert L.thip_dc_unpredict_plane(d.data_ptr(), f.data_ptr(), g["nhfrags"], g["nvfrags"]) == 0
            got = d.cpu().numpy()
            c = ost.coded[lo:hi].astype(bool)
            assert np.array_equal(got[c], ost.dc[lo:hi][c]), (density, refmix, big, pli, int((got[c] != ost.dc[lo:hi][c]).sum()))
        ost.close()
    assert L.thip_dc_unpredict_plane(None, None, 4, 4) == _lib.EFAULT
    d = torch.zeros(8, dtype=torch.int16, device="cuda")
    f = torch.zeros(8, dtype=torch.uint8, device="cuda")
    assert L.thip_dc_unpredict_plane(d.data_ptr(), f.data_ptr(), 1, 1025) == _lib.EIMPL


@pytest.mark.parametrize("lanes", [4, 1])
def test_enc_fdct_quantize_in_one_pass(hip, lanes):
    """(Both kernels: four lanes per block, the default, and one block per lane -- option enc_fq_lanes.)
    thip_enc_fdct_quantize_batch == oc_enc_fdct8x8 followed by oc_enc_quantize (fdct.c:128, enquant.c:219) on the oracle, for
    the residual range, beyond it, the extreme step sizes, with and without the coefficients handed back, with the reciprocals
    derived on the device and with a table of thip_enc_enquant_table_init."""
    from theora_amd import _lib
    import torch
    L = _lib.load()
    rng = np.random.default_rng(21)
    n = 6001          # (not a multiple of a wave)
    with util.options(L, enc_fq_lanes=lanes):
        _fdct_quantize_trials(hip, L, rng, n)
    assert L.thip_enc_fdct_quantize_batch(None, None, None, None, None, None, 0) == 0
    assert L.thip_enc_fdct_quantize_batch(None, None, None, None, None, None, 5) == _lib.EFAULT


def _fdct_quantize_trials(hip, L, rng, n):
    import torch
    for trial in range(4):
        x = rng.integers(-255, 256, (n, 64)).astype(np.int16)
        if trial == 2:
            x[:500] = rng.integers(-8160, 8161, (500, 64))
        dq = rng.integers(8, 4097, 64).astype(np.uint16) if trial else np.full(64, 8, np.uint16)
        if trial == 3:
            dq[:] = 4096
        want_dct = oracle.fdct8x8_batch(x)
        want_q, want_nz = oracle.quantize_batch(want_dct, dq)
        q, nz, dct = hip.enc_fdct_quantize_batch(dev(x), dev(dq), want_dct=True)
        assert np.array_equal(dct.cpu().numpy().reshape(-1, 64), want_dct), trial
        assert np.array_equal(q.cpu().numpy().reshape(-1, 64), want_q) and np.array_equal(nz.cpu().numpy(), want_nz), trial
        q2, nz2 = hip.enc_fdct_quantize_batch(dev(x), dev(dq))
        assert torch.equal(q2, q) and torch.equal(nz2, nz)
        enq = np.zeros(128, np.int16)
        L.thip_enc_enquant_table_init(enq.ctypes.data, dq.ctypes.data)
        q3 = torch.empty((n, 64), dtype=torch.int16, device="cuda")
        nz3 = torch.empty(n, dtype=torch.int32, device="cuda")
        assert L.thip_enc_fdct_quantize_batch(q3.data_ptr(), nz3.data_ptr(), None, dev(x).data_ptr(), dev(dq).data_ptr(), dev(enq).data_ptr(), n) == 0
        assert torch.equal(q3.reshape(-1), q.reshape(-1)) and torch.equal(nz3, nz)
