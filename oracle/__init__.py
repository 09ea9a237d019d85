"""ctypes binding of the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product (theora_amd/) never does.  PARITY UNPINNED: see theora_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libtheora_oracle.so")

FRAME_GOLD, FRAME_PREV, FRAME_SELF, FRAME_NONE = 0, 1, 2, 3
INTRA_FRAME, INTER_FRAME = 0, 1
PF_420, PF_422, PF_444 = 0, 2, 3


def build(force=False):
    """Compile the C restatement with gcc (seconds)."""
    src = os.path.join(_HERE, "theora_oracle.c")
    hdr = os.path.join(_HERE, "theora_oracle.h")
    newest = max(os.path.getmtime(src), os.path.getmtime(hdr), os.path.getmtime(os.path.join(_HERE, "Makefile")))
    if (not force and os.path.exists(_SO) and os.path.getmtime(_SO) >= newest
            and os.path.exists(_SO_SIMD) and os.path.getmtime(_SO_SIMD) >= newest):
        return _SO
    subprocess.check_call(["make", "-s", "-B", "-C", _HERE])
    return _SO


class _PlaneGeom(C.Structure):
    _fields_ = [("nhfrags", C.c_int), ("nvfrags", C.c_int), ("froffset", C.c_ssize_t),
                ("nfrags", C.c_ssize_t), ("width", C.c_int), ("height", C.c_int),
                ("stride", C.c_int)]


class _State(C.Structure):
    _fields_ = [("frame_width", C.c_int), ("frame_height", C.c_int), ("pixel_fmt", C.c_int),
                ("hdec", C.c_int), ("vdec", C.c_int), ("fplanes", _PlaneGeom * 3),
                ("nfrags", C.c_ssize_t),
                ("coded", C.POINTER(C.c_uint8)), ("refi", C.POINTER(C.c_uint8)),
                ("mvs", C.POINTER(C.c_int16)), ("dc", C.POINTER(C.c_int16)),
                ("frag_buf_offs", C.POINTER(C.c_ssize_t)),
                ("ref_slab", C.POINTER(C.c_uint8)), ("ref_frame_sz", C.c_size_t),
                ("ref_plane_data", (C.POINTER(C.c_uint8) * 3) * 3),
                ("ref_frame_idx", C.c_int * 3),
                ("ref_frame_data", C.POINTER(C.c_uint8) * 3),
                ("plane_off", C.c_ssize_t * 3)]


_SO_SIMD = os.path.join(_HERE, "_build", "libtheora_oracle_simd.so")
_lib = None
_lib_simd = None


def lib(simd=False):
    """The oracle library; simd=True: the build whose inverse transform, reconstruction loops and loop-filter edges are SSE2
    intrinsics (oracle/Makefile, -DORC_SIMD) -- bench.py's vectorised CPU baseline, equal to the scalar one value for value
    (tests/test_oracle.py)."""
    global _lib, _lib_simd
    if simd:
        if _lib_simd is None:
            build()
            if not os.path.exists(_SO_SIMD):
                subprocess.check_call(["make", "-s", "-B", "-C", _HERE])
            _lib_simd = _bind(C.CDLL(_SO_SIMD))
        return _lib_simd
    if _lib is not None:
        return _lib
    build()
    _lib = _bind(C.CDLL(_SO))
    return _lib


def _bind(L):
    P = C.c_void_p
    L.orc_state_new.restype = C.POINTER(_State)
    L.orc_state_new.argtypes = [C.c_int] * 3
    L.orc_state_free.argtypes = [C.POINTER(_State)]
    L.orc_state_get_plane.argtypes = [C.POINTER(_State), C.c_int, C.c_int, P]
    L.orc_state_set_plane.argtypes = [C.POINTER(_State), C.c_int, C.c_int, P]
    L.orc_state_set_ref_idx.argtypes = [C.POINTER(_State)] + [C.c_int] * 3
    L.orc_decode_frame.restype = C.c_int
    L.orc_decode_frame.argtypes = [C.POINTER(_State), C.c_int, P, P, P, P, P, P, C.c_ssize_t, C.c_int]
    L.orc_sb_order.restype = C.c_ssize_t
    L.orc_sb_order.argtypes = [C.POINTER(_State), C.c_int, P]
    L.orc_dc_unpredict_rows.restype = C.c_ssize_t
    L.orc_dc_unpredict_rows.argtypes = [C.POINTER(_State), C.c_int, C.c_int, C.c_int, P]
    L.orc_state_loop_filter_frag_rows.argtypes = [C.POINTER(_State), P, C.c_int, C.c_int, C.c_int, C.c_int]
    L.orc_loop_filter_init.argtypes = [P, C.c_int]
    L.orc_idct8x8_batch.argtypes = [P, P, P, C.c_ssize_t]
    L.orc_idct8x8.argtypes = [P, P, C.c_int]
    L.orc_idct8x8_full.argtypes = [P, P]
    L.orc_enc_fdct8x8_batch.argtypes = [P, P, C.c_ssize_t]
    L.orc_enc_quantize_batch.argtypes = [P, P, P, P, C.c_ssize_t]
    L.orc_enc_metric_batch.argtypes = [C.c_int, P, P, P, P, C.c_int, P, P, P, C.c_uint, C.c_ssize_t]
    L.orc_mv_offsets.restype = C.c_int
    L.orc_mv_offsets.argtypes = [P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    L.orc_mb_cost_maps.restype = C.c_int
    L.orc_mb_cost_maps.argtypes = [P, P, C.c_int, C.c_int, C.c_int, P, P, P, P]
    L.orc_frag_recon_intra.argtypes = [P, C.c_int, P]
    L.orc_frag_recon_inter.argtypes = [P, P, C.c_int, P]
    L.orc_frag_recon_inter2.argtypes = [P, P, P, C.c_int, P]
    L.orc_enc_frag_sub.argtypes = [P, P, P, C.c_int]
    L.orc_enc_frag_sub_128.argtypes = [P, P, C.c_int]
    L.orc_enc_frag_copy2.argtypes = [P, P, P, C.c_int]
    L.orc_enc_frag_border_ssd.restype = C.c_uint
    L.orc_enc_frag_border_ssd.argtypes = [P, P, C.c_int, C.c_int64]
    L.orc_postprocess_frame.argtypes = [C.POINTER(_State), C.c_int, C.c_int, C.c_int, P, P, P, P, P, P]
    L.orc_pp_deblock_frag_rows.argtypes = [P, C.c_int, P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, P, P, P, C.c_int, C.c_int]
    L.orc_pp_dering_frag_rows.argtypes = [P, C.c_int, C.c_int, C.c_int, C.c_int, P, P, P, P, C.c_int, C.c_int, C.c_int, C.c_int]
    # the single-block encoder slots (tests/test_gpu_slots.py compares thip_enc1_* with them one call at a time)
    U = C.c_uint
    for name, res, args in (("orc_enc_frag_sad", U, [P, P, C.c_int]), ("orc_enc_frag_sad_thresh", U, [P, P, C.c_int, U]),
                            ("orc_enc_frag_sad2_thresh", U, [P, P, P, C.c_int, U]), ("orc_enc_frag_intra_sad", U, [P, C.c_int]),
                            ("orc_enc_frag_satd", U, [P, P, P, C.c_int]), ("orc_enc_frag_satd2", U, [P, P, P, P, C.c_int]),
                            ("orc_enc_frag_intra_satd", U, [P, P, C.c_int]), ("orc_enc_frag_ssd", U, [P, P, C.c_int]),
                            ("orc_enc_fdct8x8", None, [P, P]), ("orc_enc_enquant_table_init", None, [P, P]),
                            ("orc_enc_quantize", C.c_int, [P, P, P, P])):
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    return L


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


class State:
    """The oracle's restatement of the path-relevant part of oc_theora_state."""

    def __init__(self, frame_width, frame_height, pixel_fmt=PF_420, simd=False):
        self._L = lib(simd)
        self._st = self._L.orc_state_new(frame_width, frame_height, pixel_fmt)
        if not self._st:
            raise ValueError("invalid frame geometry")
        s = self._st.contents
        self.frame_width, self.frame_height, self.pixel_fmt = frame_width, frame_height, pixel_fmt
        self.hdec, self.vdec = s.hdec, s.vdec
        self.nfrags = s.nfrags
        self.planes = [dict(nhfrags=g.nhfrags, nvfrags=g.nvfrags, froffset=g.froffset,
                            nfrags=g.nfrags, width=g.width, height=g.height, stride=g.stride)
                       for g in s.fplanes]
        n = self.nfrags
        self.coded = np.ctypeslib.as_array(s.coded, (n,))
        self.refi = np.ctypeslib.as_array(s.refi, (n,))
        self.mvs = np.ctypeslib.as_array(s.mvs, (n,))
        self.dc = np.ctypeslib.as_array(s.dc, (n,))

    def close(self):
        if self._st:
            self._L.orc_state_free(self._st)
            self._st = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def ref_frame_idx(self):
        return list(self._st.contents.ref_frame_idx)

    def set_ref_idx(self, gold, prev, self_):
        self._L.orc_state_set_ref_idx(self._st, gold, prev, self_)

    def get_plane(self, slot, pli):
        g = self.planes[pli]
        out = np.empty((g["height"], g["width"]), np.uint8)
        self._L.orc_state_get_plane(self._st, slot, pli, _p(out))
        return out

    def set_plane(self, slot, pli, arr):
        g = self.planes[pli]
        a = _c(arr, np.uint8)
        assert a.shape == (g["height"], g["width"])
        self._L.orc_state_set_plane(self._st, slot, pli, _p(a))

    def sb_order(self, pli):
        out = np.empty(self.planes[pli]["nfrags"], np.int64)
        n = self._L.orc_sb_order(self._st, pli, _p(out))
        assert n == out.size
        return out

    def dc_unpredict(self):
        """Whole-frame DC un-prediction (decode.c:1392-1500) on self.dc in place."""
        for pli in range(3):
            pred_last = np.zeros(3, np.int32)
            self._L.orc_dc_unpredict_rows(self._st, pli, 0, self.planes[pli]["nvfrags"], _p(pred_last))

    def postprocess(self, slot, level, loop_filter, dc_qis, frag_qi, pp_dc_scale, pp_sharp_mod):
        """Out-of-loop post-processing (decode.c:1608-1957) of the frame in `slot`, driven MCU by MCU with the
        reference's row delays (decode.c:2858-2945).  Returns ([Y, Cb, Cr] in bitstream row order, variances)."""
        sizes = [g["width"] * g["height"] for g in self.planes]
        out = np.zeros(sum(sizes), np.uint8)
        var = np.zeros(self.nfrags, np.int32)
        self._L.orc_postprocess_frame(self._st, slot, level, int(bool(loop_filter)), _p(_c(dc_qis, np.uint8)),
                                      _p(_c(frag_qi, np.uint8)), _p(_c(pp_dc_scale, np.int32)), _p(_c(pp_sharp_mod, np.int32)),
                                      _p(out), _p(var))
        planes, off = [], 0
        for g, n in zip(self.planes, sizes):
            planes.append(out[off:off + n].reshape(g["height"], g["width"]).copy())
            off += n
        return planes, var

    def loop_filter_rows(self, flimit, slot, pli, fragy0, fragy_end):
        bv = np.zeros(256, np.int8)
        self._L.orc_loop_filter_init(_p(bv), flimit)
        self._L.orc_state_loop_filter_frag_rows(self._st, _p(bv), slot, pli, fragy0, fragy_end)

    def decode_frame(self, frame_type, coded_fragis, ncoded, coeffs, last_zzi, dc_quant,
                     uncoded_fragis, flimit):
        """coeffs: int16 [n,64] natural order, raw DC in [:,0] (dequantised AC)."""
        cf = _c(coded_fragis, np.int64)
        nc = _c(ncoded, np.int64)
        co = _c(coeffs, np.int16).reshape(-1, 64) if len(cf) else np.zeros((0, 64), np.int16)
        lz = _c(last_zzi, np.uint8)
        dq = _c(dc_quant, np.uint16)
        uf = _c(uncoded_fragis, np.int64)
        assert co.shape[0] == cf.size == lz.size == dq.size
        return self._L.orc_decode_frame(self._st, frame_type, _p(cf), _p(nc), _p(co), _p(lz), _p(dq),
                                        _p(uf), uf.size, flimit)


def idct8x8_batch(x, last_zzi=None):
    x = _c(x, np.int16).reshape(-1, 64)
    y = np.empty_like(x)
    lz = None if last_zzi is None else _c(last_zzi, np.int32)
    lib().orc_idct8x8_batch(_p(y), _p(x), _p(lz), x.shape[0])
    return y


def idct8x8_full_batch(x):
    x = _c(x, np.int16).reshape(-1, 64).copy()
    y = np.empty_like(x)
    L = lib()
    for i in range(x.shape[0]):
        L.orc_idct8x8_full(_p(y[i]), _p(x[i]))
    return y


def fdct8x8_batch(x):
    x = _c(x, np.int16).reshape(-1, 64)
    y = np.empty_like(x)
    lib().orc_enc_fdct8x8_batch(_p(y), _p(x), x.shape[0])
    return y


def quantize_batch(dct, dequant):
    """oc_enc_quantize_c over [n,64] zig-zag-ordered coefficients with one 64-entry table."""
    d = _c(dct, np.int16).reshape(-1, 64)
    dq = _c(dequant, np.uint16)
    q = np.empty_like(d)
    nz = np.empty(d.shape[0], np.int32)
    lib().orc_enc_quantize_batch(_p(q), _p(nz), _p(d), _p(dq), d.shape[0])
    return q, nz


METRIC_OPS = dict(sad=0, sad_thresh=1, sad2_thresh=2, intra_sad=3, satd=4, satd2=5, intra_satd=6, ssd=7)


def enc_metric_batch(op, src_plane, ref_plane, ystride, src_offs, ref_offs=None, ref2_offs=None, thresh=0):
    so = _c(src_offs, np.int32)
    ro = None if ref_offs is None else _c(ref_offs, np.int32)
    r2 = None if ref2_offs is None else _c(ref2_offs, np.int32)
    out = np.empty(so.size, np.uint32)
    dc = np.zeros(so.size, np.int32)
    sp = _c(src_plane, np.uint8)
    rp = sp if ref_plane is None else _c(ref_plane, np.uint8)
    lib().orc_enc_metric_batch(METRIC_OPS[op], _p(out), _p(dc), _p(sp), _p(rp), ystride, _p(so), _p(ro),
                               _p(r2), thresh, so.size)
    return out, dc


def halfpel_mvoffsets(vx, vy, dx, dy, ystride):
    """The two block offsets the half-pel refinement hands to oc_enc_frag_satd2 / oc_enc_frag_sad2_thresh for the half-pel vector
    2 * (vx, vy) + (dx, dy), relative to the block at the whole-pel vector (the reference's mvoffset_base):
    oc_mcenc_ysatd_halfpel_mbrefine, mcenc.c:606-657 (offset_y[site] = dy * ystride, :624-626; xmask / ymask = OC_SIGNMASK(((vec <<
    1) + d) ^ d), :644-645; mvoffset0 = (dx & xmask) + (offset_y & ymask), mvoffset1 = (dx & ~xmask) + (offset_y & ~ymask), :646-647).
    vx, vy: integer arrays; returns (mvoffset0, mvoffset1) as int64 arrays."""
    vx, vy = np.asarray(vx, np.int64), np.asarray(vy, np.int64)
    xmask = np.where((((vx << 1) + dx) ^ dx) < 0, -1, 0).astype(np.int64)
    ymask = np.where((((vy << 1) + dy) ^ dy) < 0, -1, 0).astype(np.int64)
    oy = dy * ystride
    return (dx & xmask) + (oy & ymask), (dx & ~xmask) + (oy & ~ymask)


def enc_halfpel_sites(op, src_plane, ref_plane, ystride, src_offs, ref_offs, vecs, sites):
    """What the refinement computes for every block and every half-pel site of `sites` = [(dx, dy), ...] (mcenc.c:551-657): the
    reference's own slot (orc_enc_metric_batch: oc_enc_frag_satd2_c, encfrag.c:323-328, or oc_enc_frag_sad2_thresh_c with a
    threshold nothing reaches, encfrag.c:62-86) on the two blocks of halfpel_mvoffsets.  vecs: int16, x & 0xFF | y << 8
    (state.h:232-240).  Returns (values, dc) as [len(sites), nblocks] arrays."""
    v = np.asarray(vecs).astype(np.int16)
    vx = (v & 0xFF).astype(np.int8).astype(np.int64)
    vy = (v >> 8).astype(np.int8).astype(np.int64)
    ro = np.asarray(ref_offs, np.int64)
    vals, dcs = [], []
    for dx, dy in sites:
        o0, o1 = halfpel_mvoffsets(vx, vy, dx, dy, ystride)
        a, d = enc_metric_batch(op, src_plane, ref_plane, ystride, src_offs, (ro + o0).astype(np.int32), (ro + o1).astype(np.int32),
                                0xFFFFFFFF if op == "sad2_thresh" else 0)
        vals.append(a)
        dcs.append(d)
    return np.stack(vals), np.stack(dcs)


def mb_cost_maps(planes, frame_width, frame_height, pixel_fmt):
    """oc_mb_intra_satd / oc_mb_activity / oc_mb_activity_fast over a whole frame (three unpadded planes, bitstream row order).
    Returns (intra_satd [nmbs,12], luma [nmbs], activity [nmbs,4], activity_fast [nmbs,4]) in the reference's macro-block order."""
    pl = [_c(p, np.uint8) for p in planes]
    nsbw, nsbh = ((frame_width >> 3) + 3) >> 2, ((frame_height >> 3) + 3) >> 2
    nmbs = 4 * nsbw * nsbh
    ptrs = (C.c_void_p * 3)(*[p.ctypes.data for p in pl])
    strides = (C.c_int * 3)(*[p.shape[1] for p in pl])
    satd = np.zeros((nmbs, 12), np.uint32)
    luma = np.zeros(nmbs, np.uint32)
    act = np.zeros((nmbs, 4), np.uint32)
    fast = np.zeros((nmbs, 4), np.uint32)
    rc = lib().orc_mb_cost_maps(ptrs, strides, frame_width, frame_height, pixel_fmt, _p(satd), _p(luma), _p(act), _p(fast))
    assert rc == 0
    return satd, luma, act, fast


def mv_offsets(ystride, qpx, qpy, dx, dy):
    offs = np.zeros(2, np.int32)
    n = lib().orc_mv_offsets(_p(offs), ystride, qpx, qpy, dx, dy)
    return n, int(offs[0]), int(offs[1])


def loop_filter_bv(flimit):
    bv = np.zeros(256, np.int8)
    lib().orc_loop_filter_init(_p(bv), flimit)
    return bv
