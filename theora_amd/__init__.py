"""theora_amd -- MI355X (gfx950) backend for libtheora's per-fragment reconstruction path.

The product is the C-ABI library ``libtheora_hip.so`` (include/theora_hip.h, sources in
theora_amd/csrc/).  This package is the thin Python harness above it used by tests and
bench.py: ctypes bindings, host-side packing of fragment command streams into the
backend's HBM layout, and device-buffer plumbing through torch.  It never falls back to a
CPU implementation.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import (DUPFRAME, FRAME_GOLD, FRAME_PREV, FRAME_SELF, INTER_FRAME, INTRA_FRAME,  # noqa: F401
                   COEFFS_DEQUANT16, COEFFS_LEVELS, MAX_BATCH, SLOT_GROUP, SLOT_GROUP_BYTES, SLOT_WIDE, TILE_FRAGS, FrameDesc,
                   PlaneGeom, TheoraHipError, TileGeom)

TILE_BLOCKS = SLOT_GROUP   # slots per coefficient group

PF_420, PF_422, PF_444 = 0, 2, 3


def version():
    return _lib.load().thip_version_string().decode()


def _ptr(t):
    """Device (torch tensor) or host (numpy array) address, or NULL."""
    if t is None:
        return None
    if isinstance(t, np.ndarray):
        return t.ctypes.data
    return t.data_ptr()


# ---------------------------------------------------------------------------------------
# host-side packing of a frame's fragment command stream (layout of include/theora_hip.h)
# ---------------------------------------------------------------------------------------
def info_words(npos, pos, refi, last_zzi, mvx, mvy, dc, dc_quant, qii=None):
    """[npos,2] uint32 frag_info array (include/theora_hip.h): coded fragments at tile
    positions `pos`; every other position stays 0 (uncoded / outside the plane).  dc is the raw
    DC coefficient (used for the DC-only fragments, whose value has no coefficient slot),
    dc_quant the per-fragment DC quantiser: the dequantisation itself (state.c:967-979) is the
    kernel's.  qii given: the LEVELS form -- word 0 carries it, word 1 the raw DC of every block."""
    out = np.zeros((npos, 2), np.uint32)
    lz = np.asarray(last_zzi, np.uint32)
    w0 = np.uint32(_lib.INFO_CODED) | ((np.asarray(refi, np.uint32) & 3) << 1) | (lz << 8)
    w0 |= np.where(lz < 2, np.uint32(_lib.INFO_DC_ONLY), np.uint32(0))
    w0 |= (np.asarray(mvx, np.int32).astype(np.uint32) & 0xFF) << 16
    w0 |= (np.asarray(mvy, np.int32).astype(np.uint32) & 0xFF) << 24
    if qii is not None:
        w0 |= (np.asarray(qii, np.uint32) & 3) << _lib.INFO_QII_SHIFT
    out[pos, 0] = w0
    w1 = (np.asarray(dc_quant, np.int64).astype(np.uint32) & 0xFFFF) << 16
    w1 |= np.where((lz < 2) | (qii is not None), np.asarray(dc, np.int32).astype(np.uint32) & 0xFFFF, 0).astype(np.uint32)
    out[pos, 1] = w1
    return out


def pack_tiles(coeffs):
    """[n,64] natural-order int16 blocks -> the backend's tile layout (int16, flat); see
    include/theora_hip.h: group q=2j+h of slot i at group(i//64)*4096 + q*512 + (i%64)*8
    int16s, holding {x[2j][c], x[2j+1][c]} for c=4h..4h+3."""
    co = np.asarray(coeffs, np.int16).reshape(-1, 8, 8)
    n = co.shape[0]
    ntiles = (n + TILE_BLOCKS - 1) // TILE_BLOCKS
    pad = np.zeros((ntiles * TILE_BLOCKS, 8, 8), np.int16)
    pad[:n] = co
    # [tile, lane, j, half, h, cc] -> [tile, j, h, lane, cc, half]
    t = pad.reshape(ntiles, TILE_BLOCKS, 4, 2, 2, 4).transpose(0, 2, 4, 1, 5, 3)
    return np.ascontiguousarray(t).reshape(-1)


def pack_units(levels, wide, first_unit, nunits):
    """The LEVELS form's slot array (include/theora_hip.h): levels [n,64] natural-order int16, wide [n] bool (the block's
    tile is wide), first_unit [n] the block's first 64-byte unit.  Returns the array as uint8, whole groups of 64 units."""
    lv = np.asarray(levels, np.int16).reshape(-1, 8, 8)
    wide = np.asarray(wide, bool)
    first_unit = np.asarray(first_unit, np.int64)
    ngroups = (int(nunits) + 63) // 64
    out = np.zeros(ngroups * 4096, np.uint8)

    def addr(unit, piece):     # byte address of a unit's piece
        return (unit >> 6) * 4096 + piece * 1024 + (unit & 63) * 16
    nar = np.nonzero(~wide)[0]
    if nar.size:
        # piece j, dword d: bytes x[2j][2d], x[2j][2d+1], x[2j+1][2d], x[2j+1][2d+1]
        b = lv[nar].astype(np.int8).reshape(-1, 4, 2, 4, 2).transpose(0, 1, 3, 2, 4)      # [n, j, d, parity, e]
        b = np.ascontiguousarray(b).reshape(-1, 4, 16).view(np.uint8)
        for j in range(4):
            a = addr(first_unit[nar], j)
            out[(a[:, None] + np.arange(16)[None, :]).reshape(-1)] = b[:, j].reshape(-1)
    wd = np.nonzero(wide)[0]
    if wd.size:
        # piece q = 2j+h: the int16 pairs {x[2j][c], x[2j+1][c]}, c = 4h..4h+3; pieces 0-3 in the first unit, 4-7 in the second
        t = lv[wd].reshape(-1, 4, 2, 2, 4).transpose(0, 1, 3, 4, 2)                        # [n, j, h, cc, parity]
        t = np.ascontiguousarray(t).reshape(-1, 8, 8).view(np.uint8).reshape(-1, 8, 16)
        for q in range(8):
            a = addr(first_unit[wd] + (q >> 2), q & 3)
            out[(a[:, None] + np.arange(16)[None, :]).reshape(-1)] = t[:, q].reshape(-1)
    return out


def pack_dequant_tables(dequant):
    """[3][3][2][64] zig-zag-ordered uint16 tables -> the 18 x 64 slot-order array the kernels multiply with
    (what thip_pack_dequant_table does for one table): out[(j*8 + c)*2 + p] = table[zig-zag index of (2j+p, c)]."""
    from .synth import FZIG_ZAG
    dq = np.ascontiguousarray(dequant, np.uint16).reshape(18, 64)
    nat = np.zeros((18, 64), np.uint16)
    nat[:, FZIG_ZAG] = dq
    return np.ascontiguousarray(nat.reshape(18, 4, 2, 8).transpose(0, 1, 3, 2)).reshape(18, 64)


def unpack_tiles(tiles, n):
    t = np.asarray(tiles, np.int16).reshape(-1, 4, 2, TILE_BLOCKS, 4, 2).transpose(0, 3, 1, 5, 2, 4)
    return np.ascontiguousarray(t).reshape(-1, 64)[:n]


class State:
    """Device side of one stream (thip_state): three resident frames + the reference ring."""

    def __init__(self, frame_width, frame_height, pixel_fmt=PF_420, device=None):
        self._L = _lib.load()
        h = C.c_void_p()
        if device is None:      # the calling thread's current HIP device
            _lib.check(self._L.thip_state_create(C.byref(h), frame_width, frame_height, pixel_fmt),
                       "thip_state_create")
        else:
            _lib.check(self._L.thip_state_create_on(C.byref(h), int(device), frame_width, frame_height, pixel_fmt),
                       "thip_state_create_on")
        self.device = self._L.thip_state_device(h)
        self._h = h
        geom = (PlaneGeom * 3)()
        nfrags, fbytes = C.c_int64(), C.c_int64()
        _lib.check(self._L.thip_state_get_geom(h, geom, C.byref(nfrags), C.byref(fbytes)), "get_geom")
        self.planes = [dict((f, getattr(g, f)) for f, _ in PlaneGeom._fields_) for g in geom]
        self.nfrags = nfrags.value
        self.frame_bytes = fbytes.value
        self.frame_width, self.frame_height, self.pixel_fmt = frame_width, frame_height, pixel_fmt
        tg = TileGeom()
        _lib.check(self._L.thip_state_get_tiles(h, C.byref(tg)), "get_tiles")
        self.tiles_x, self.tiles_y, self.tile_off = list(tg.tiles_x), list(tg.tiles_y), list(tg.tile_off)
        self.ntiles = tg.ntiles

    def frag_pos(self, fragi):
        return self._L.thip_state_frag_pos(self._h, int(fragi))

    @property
    def handle(self):
        return self._h

    def close(self):
        if getattr(self, "_h", None):
            self._L.thip_state_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def ref_idx(self, which):
        return self._L.thip_state_ref_idx(self._h, which)

    def check_fault(self):
        """thip_state_check_fault: waits for the state's stream; 0 nothing wrong, 1 the newest frame was decoded again (a tile
        hand-over had failed), raises on THIP_EFAULT.  For callers that only bracket their work with synchronize()."""
        return _lib.check(self._L.thip_state_check_fault(self._h), "thip_state_check_fault")

    def set_ref_idx(self, gold, prev, self_):
        _lib.check(self._L.thip_state_set_ref_idx(self._h, gold, prev, self_), "set_ref_idx")

    def read_plane(self, bufi, pli):
        g = self.planes[pli]
        out = np.empty((g["height"], g["width"]), np.uint8)
        _lib.check(self._L.thip_state_read_plane(self._h, bufi, pli, out.ctypes.data), "read_plane")
        return out

    def write_plane(self, bufi, pli, arr):
        g = self.planes[pli]
        a = np.ascontiguousarray(arr, np.uint8)
        assert a.shape == (g["height"], g["width"])
        _lib.check(self._L.thip_state_write_plane(self._h, bufi, pli, a.ctypes.data), "write_plane")

    def ycbcr_out(self):
        """th_decode_ycbcr_out: the last decoded frame, display (top-down) row order."""
        outs = [np.empty((g["height"], g["width"]), np.uint8) for g in self.planes]
        ptrs = (C.c_void_p * 3)(*[o.ctypes.data for o in outs])
        strides = (C.c_int32 * 3)(*[g["width"] for g in self.planes])
        _lib.check(self._L.thip_state_ycbcr_out(self._h, ptrs, strides), "ycbcr_out")
        return outs

    def ycbcr_map(self):
        """The same frame without the copy: numpy views of the library's pinned image (valid
        until the second following frame has been decoded; do not write)."""
        ptrs = (C.c_void_p * 3)()
        strides = (C.c_int32 * 3)()
        _lib.check(self._L.thip_state_ycbcr_map(self._h, ptrs, strides), "ycbcr_map")
        outs = []
        for pli, g in enumerate(self.planes):
            buf = (C.c_ubyte * (strides[pli] * g["height"])).from_address(ptrs[pli])
            outs.append(np.frombuffer(buf, np.uint8).reshape(g["height"], strides[pli])[:, :g["width"]])
        return outs

    def set_eager_output(self, on):
        _lib.check(self._L.thip_state_set_eager_output(self._h, int(bool(on))), "set_eager_output")

    # ---- host-enqueue form: the vtable slots, one fragment at a time --------------------
    def frame_begin(self, frame_type):
        _lib.check(self._L.thip_frame_begin(self._h, frame_type), "frame_begin")

    def frag_recon(self, fragi, pli, dct_coeffs, last_zzi, dc_quant, refi, mv):
        """oc_state_frag_recon slot; dct_coeffs is an int16[128] numpy array (zeroed on return)."""
        assert dct_coeffs.dtype == np.int16 and dct_coeffs.size >= 128
        _lib.check(self._L.thip_state_frag_recon(self._h, fragi, pli, dct_coeffs.ctypes.data, last_zzi,
                                                 dc_quant, refi, mv), "state_frag_recon")

    def frag_recon_levels(self, fragi, pli, levels, last_zzi, dc_quant, qii, refi, mv):
        """thip_state_frag_recon_levels: `levels` is an int16[128] numpy array holding the quantised levels (zeroed on return)."""
        assert levels.dtype == np.int16 and levels.size >= 128
        _lib.check(self._L.thip_state_frag_recon_levels(self._h, fragi, pli, levels.ctypes.data, last_zzi, dc_quant, qii, refi, mv),
                   "state_frag_recon_levels")

    def frame_dequant_table(self, sel, table_zz):
        t = np.ascontiguousarray(table_zz, np.uint16)
        assert t.size == 64
        _lib.check(self._L.thip_frame_dequant_table(self._h, sel, t.ctypes.data), "frame_dequant_table")

    def frag_copy_list(self, fragis):
        a = np.ascontiguousarray(fragis, np.int64)
        _lib.check(self._L.thip_frag_copy_list(self._h, a.ctypes.data, a.size), "frag_copy_list")

    def loop_filter_frag_rows(self, flimit, refi, pli, fragy0, fragy_end):
        _lib.check(self._L.thip_state_loop_filter_frag_rows(self._h, flimit, refi, pli, fragy0, fragy_end),
                   "loop_filter_frag_rows")

    def frame_flush(self):
        return _lib.check(self._L.thip_frame_flush(self._h), "frame_flush")


def decode_frames(states, descs, stream=None):
    """thip_decode_frames over parallel lists of State and FrameDesc; returns per-stream results."""
    L = _lib.load()
    n = len(states)
    assert n == len(descs)
    hs = (C.c_void_p * n)(*[s.handle for s in states])
    ds = (FrameDesc * n)(*descs)
    res = (C.c_int32 * n)()
    _lib.check(L.thip_decode_frames(hs, ds, n, stream, res), "thip_decode_frames")
    return list(res)


class BatchPlan:
    """A pre-marshalled thip_decode_frames call (bench.py's timed loop reuses these so the
    Python/ctypes overhead per step is one foreign call)."""

    def __init__(self, states, descs):
        self._L = _lib.load()
        self.n = len(states)
        self.hs = (C.c_void_p * self.n)(*[s.handle for s in states])
        self.ds = (FrameDesc * self.n)(*descs)
        self.res = (C.c_int32 * self.n)()

    def submit(self, stream=None):
        rc = self._L.thip_decode_frames(self.hs, self.ds, self.n, stream, self.res)
        if rc < 0:
            raise TheoraHipError("thip_decode_frames failed: %d" % rc)


def synchronize():
    """thip_synchronize: waits for the library's streams; raises while some state's fault word is set (it only reports: the state's
    own State.check_fault() / ycbcr_out / read_plane repair or clear it)."""
    _lib.check(_lib.load().thip_synchronize(), "thip_synchronize")


def make_desc(info_dev, coeffs_dev, slot0_dev, nslots, ncoded, frame_type, flimit, dc_tokens_dev=None, dequant_dev=None):
    """dequant_dev given: the LEVELS form (coeffs_dev = units, slot0 = first units | SLOT_WIDE, nslots = units)."""
    return FrameDesc(_ptr(info_dev), _ptr(coeffs_dev), _ptr(slot0_dev), nslots, ncoded, frame_type, flimit,
                     _ptr(dc_tokens_dev), COEFFS_LEVELS if dequant_dev is not None else COEFFS_DEQUANT16, _ptr(dequant_dev))


def profile_enable(on):
    _lib.check(_lib.load().thip_profile_enable(int(on)), "profile_enable")


def profile_reset():
    _lib.check(_lib.load().thip_profile_reset(), "profile_reset")


def profile_read():
    n = (C.c_int64 * _lib.NKERNELS)()
    ms = (C.c_double * _lib.NKERNELS)()
    _lib.check(_lib.load().thip_profile_read(n, ms), "profile_read")
    return list(n), list(ms)


# ---------------------------------------------------------------------------------------
# batched single slots (device tensors in, device tensors out)
# ---------------------------------------------------------------------------------------
def idct8x8_batch(x_dev, last_zzi_dev=None):
    import torch
    y = torch.empty_like(x_dev)
    n = x_dev.numel() // 64
    _lib.check(_lib.load().thip_idct8x8_batch(_ptr(y), _ptr(x_dev), _ptr(last_zzi_dev), n), "idct8x8_batch")
    return y


def fdct8x8_batch(x_dev):
    import torch
    y = torch.empty_like(x_dev)
    _lib.check(_lib.load().thip_enc_fdct8x8_batch(_ptr(y), _ptr(x_dev), x_dev.numel() // 64), "fdct8x8_batch")
    return y


def enc_quantize_batch(dct_dev, dequant_dev):
    """oc_enc_quantize over [n,64] zig-zag-ordered int16 blocks; returns (qdct, nonzero)."""
    import torch
    n = dct_dev.numel() // 64
    q = torch.empty_like(dct_dev)
    nz = torch.empty(n, dtype=torch.int32, device=dct_dev.device)
    _lib.check(_lib.load().thip_enc_quantize_batch(_ptr(q), _ptr(nz), _ptr(dct_dev), _ptr(dequant_dev), n),
               "enc_quantize_batch")
    return q, nz


def enc_fdct_quantize_batch(x_dev, dequant_dev, want_dct=False):
    """thip_enc_fdct_quantize_batch over [n,64] natural-order int16 residual blocks; returns (qdct, nonzero[, dct])."""
    import torch
    n = x_dev.numel() // 64
    q = torch.empty_like(x_dev)
    nz = torch.empty(n, dtype=torch.int32, device=x_dev.device)
    dct = torch.empty_like(x_dev) if want_dct else None
    _lib.check(_lib.load().thip_enc_fdct_quantize_batch(_ptr(q), _ptr(nz), _ptr(dct), _ptr(x_dev), _ptr(dequant_dev), None, n),
               "enc_fdct_quantize_batch")
    return (q, nz, dct) if want_dct else (q, nz)


def enc_metric_batch(op, src_plane, ref_plane, ystride, src_offs, ref_offs=None, ref2_offs=None, thresh=0):
    import torch
    n = src_offs.numel()
    out = torch.empty(n, dtype=torch.int32, device=src_offs.device)
    dc = torch.empty(n, dtype=torch.int32, device=src_offs.device)   # (the kernel writes every element: 0 for the SAD family)
    _lib.check(_lib.load().thip_enc_frag_metric_batch(
        _lib.ENC_OPS[op], _ptr(out), _ptr(dc), _ptr(src_plane), _ptr(ref_plane), ystride, _ptr(src_offs),
        _ptr(ref_offs), _ptr(ref2_offs), thresh, n), "enc_frag_metric_batch")
    return out, dc


def enc_mb_cost_maps(planes, frame_width, frame_height, pixel_fmt):
    """thip_enc_mb_cost_maps over three device planes ([H, stride] uint8 tensors, bitstream row order).  Returns
    (intra_satd [nmbs,12], luma [nmbs], activity [nmbs,4], activity_fast [nmbs,4]) as int32 device tensors."""
    import torch
    L = _lib.load()
    nmbs = L.thip_enc_mb_count(frame_width, frame_height)
    dev = planes[0].device
    satd = torch.empty((nmbs, 12), dtype=torch.int32, device=dev)
    luma = torch.empty(nmbs, dtype=torch.int32, device=dev)
    act = torch.empty((nmbs, 4), dtype=torch.int32, device=dev)
    fast = torch.empty((nmbs, 4), dtype=torch.int32, device=dev)
    ptrs = (C.c_void_p * 3)(*[_ptr(p) for p in planes])
    strides = (C.c_int32 * 3)(*[int(p.stride(0)) for p in planes])
    _lib.check(L.thip_enc_mb_cost_maps(ptrs, strides, frame_width, frame_height, pixel_fmt, _ptr(satd), _ptr(luma), _ptr(act),
                                       _ptr(fast)), "enc_mb_cost_maps")
    return satd, luma, act, fast


def enc_metric_halfpel_batch(op, src_plane, ref_plane, ystride, src_offs, ref_offs, vecs, sites):
    """thip_enc_frag_metric_halfpel_batch: every block against the half-pel vectors 2 * vec + (dx, dy), `sites` = [(dx, dy), ...]
    without (0, 0), around its whole-pel vector (vecs: int16 tensor, x & 0xFF | y << 8; ref_offs: the block at that vector).
    op "satd2" or "sad2_thresh".  Returns (values, dc), both [len(sites), nblocks]; dc is None for "sad2_thresh"."""
    import numpy as np
    import torch
    n = src_offs.numel()
    ns = len(sites)
    out = torch.empty((ns, n), dtype=torch.int32, device=src_offs.device)
    dc = torch.empty((ns, n), dtype=torch.int32, device=src_offs.device) if op == "satd2" else None
    dx = np.array([s[0] for s in sites], np.int8)
    dy = np.array([s[1] for s in sites], np.int8)
    _lib.check(_lib.load().thip_enc_frag_metric_halfpel_batch(
        _lib.ENC_OPS[op], _ptr(out), _ptr(dc), _ptr(src_plane), _ptr(ref_plane), ystride, _ptr(src_offs), _ptr(ref_offs), _ptr(vecs),
        dx.ctypes.data, dy.ctypes.data, ns, n), "enc_frag_metric_halfpel_batch")
    return out, dc


def enc_metric_sites_batch(op, src_plane, ref_plane, ystride, src_offs, ref_offs, sites):
    """thip_enc_frag_metric_sites_batch: every block against the candidate positions `sites` = [(dx, dy), ...] around its
    reference position.  Returns (values, dc), both [len(sites), nblocks] (candidate-major); dc is None for "sad"."""
    import numpy as np
    import torch
    n = src_offs.numel()
    ns = len(sites)
    out = torch.empty((ns, n), dtype=torch.int32, device=src_offs.device)
    dc = torch.empty((ns, n), dtype=torch.int32, device=src_offs.device) if op == "satd" else None
    dx = np.array([s[0] for s in sites], np.int8)
    dy = np.array([s[1] for s in sites], np.int8)
    _lib.check(_lib.load().thip_enc_frag_metric_sites_batch(
        _lib.ENC_OPS[op], _ptr(out), _ptr(dc), _ptr(src_plane), _ptr(ref_plane), ystride, _ptr(src_offs), _ptr(ref_offs),
        dx.ctypes.data, dy.ctypes.data, ns, n), "enc_frag_metric_sites_batch")
    return out, dc
