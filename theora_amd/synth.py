"""Seeded synthetic fragment command streams.

Generates, for a given coded frame size, what libtheora's front end hands to the
reconstruction path for one frame: coded / uncoded fragment lists in coded order, the
per-fragment reference index and motion vector, the quantised LEVELS of every coded block with the
frame's dequantisation tables and each block's qii -- and their product, the dequantised coefficients
with the raw DC that oc_state_frag_recon receives (decode.c:1573) --, last_zzi, dc_quant and the
loop-filter limit.  Used by tests (against the oracle) and by
bench.py.  Pure numpy; no oracle and no device code in here.
"""
import numpy as np

from . import FRAME_GOLD, FRAME_PREV, FRAME_SELF, INTER_FRAME, INTRA_FRAME, PF_420

# zig-zag index -> natural position (lib/internal.c:27)
FZIG_ZAG = np.array([
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,
    7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31,
    39, 46, 53, 60, 61, 54, 47, 55, 62, 63], np.int64)

# (row, col) of the k-th block along the 4x4 Hilbert curve of a super block (state.c:134-139)
_HILBERT_RC = np.array([(0, 0), (0, 1), (1, 1), (1, 0), (2, 0), (3, 0), (3, 1), (2, 1),
                        (2, 2), (3, 2), (3, 3), (2, 3), (1, 3), (1, 2), (0, 2), (0, 3)], np.int64)


class Geometry:
    """Fragment-plane geometry of a stream (state.c:424-475)."""

    def __init__(self, frame_width, frame_height, pixel_fmt=PF_420):
        assert frame_width % 16 == 0 and frame_height % 16 == 0
        self.frame_width, self.frame_height, self.pixel_fmt = frame_width, frame_height, pixel_fmt
        self.hdec = int(not (pixel_fmt & 1))
        self.vdec = int(not (pixel_fmt & 2))
        yh, yv = frame_width >> 3, frame_height >> 3
        self.nh = [yh, (yh + self.hdec) >> self.hdec, (yh + self.hdec) >> self.hdec]
        self.nv = [yv, (yv + self.vdec) >> self.vdec, (yv + self.vdec) >> self.vdec]
        self.pl_nfrags = [self.nh[p] * self.nv[p] for p in range(3)]
        self.froffset = [0, self.pl_nfrags[0], self.pl_nfrags[0] + self.pl_nfrags[1]]
        self.nfrags = sum(self.pl_nfrags)
        self.nhmb, self.nvmb = yh // 2, yv // 2
        self._sb = [self._sb_order(p) for p in range(3)]
        self.coded_order = np.concatenate(self._sb)          # all fragments, coded order
        self.plane_of = np.concatenate([np.full(self.pl_nfrags[p], p, np.int8) for p in range(3)])
        # macro block of every fragment (raster fragment index -> mb index)
        mb = []
        for p in range(3):
            fy, fx = np.divmod(np.arange(self.pl_nfrags[p]), self.nh[p])
            sy = 1 if (p == 0 or not self.vdec) else 0
            sx = 1 if (p == 0 or not self.hdec) else 0
            mb.append((fy >> sy) * self.nhmb + (fx >> sx))
        self.mb_of = np.concatenate(mb)
        # work tiles of the device (include/theora_hip.h): 4 super blocks of one SB row
        self.tiles_x = [(self.nh[p] + 15) // 16 for p in range(3)]
        self.tiles_y = [(self.nv[p] + 3) // 4 for p in range(3)]
        self.tile_off = [0, self.tiles_x[0] * self.tiles_y[0],
                         self.tiles_x[0] * self.tiles_y[0] + self.tiles_x[1] * self.tiles_y[1]]
        self.ntiles = self.tile_off[2] + self.tiles_x[2] * self.tiles_y[2]
        hinv = np.zeros((4, 4), np.int64)
        for k, (r, c) in enumerate(_HILBERT_RC):
            hinv[r, c] = k
        pos = []
        for p in range(3):
            by, bx = np.divmod(np.arange(self.pl_nfrags[p]), self.nh[p])
            pos.append((self.tile_off[p] + (by >> 2) * self.tiles_x[p] + (bx >> 4)) * 64
                       + ((bx >> 2) & 3) * 16 + hinv[by & 3, bx & 3])
        self.frag_pos = np.concatenate(pos)       # raster fragment index -> tile*64+lane

    def _sb_order(self, pli):
        nh, nv = self.nh[pli], self.nv[pli]
        sby, sbx = np.meshgrid(np.arange(0, nv, 4), np.arange(0, nh, 4), indexing="ij")
        fy = sby.reshape(-1, 1) + _HILBERT_RC[:, 0]
        fx = sbx.reshape(-1, 1) + _HILBERT_RC[:, 1]
        ok = (fy < nv) & (fx < nh)
        return (self.froffset[pli] + fy * nh + fx)[ok]

    def sb_order(self, pli):
        return self._sb[pli]


CLASSES = {
    # p_coded: P-frame coded fraction; mode mix of coded MBs (intra, golden, zero-mv, whole-pel, half-pel);
    # p_dc_only / p_zz10: share of DC-only and <=10-coefficient blocks; amp: coefficient amplitude
    "dense": dict(p_coded=1.0, intra=0.04, golden=0.01, zeromv=0.10, halfpel=0.25, p_dc_only=0.0,
                  p_zz10=0.0, amp=40, edge_mv=0.0, extreme=0.0),
    "smooth": dict(p_coded=0.66, intra=0.039, golden=0.003, zeromv=0.086, halfpel=0.025, p_dc_only=0.80,
                   p_zz10=0.15, amp=24, edge_mv=0.0, extreme=0.0),
    "mixed": dict(p_coded=0.6, intra=0.15, golden=0.15, zeromv=0.15, halfpel=0.4, p_dc_only=0.3,
                  p_zz10=0.3, amp=300, edge_mv=0.3, extreme=0.02, big_levels=0.002),
    # a quarter of the picture changes (smooth statistics inside), the rest is a static background
    "static_bg": dict(p_coded=0.9, intra=0.039, golden=0.003, zeromv=0.086, halfpel=0.025, p_dc_only=0.80,
                      p_zz10=0.15, amp=24, edge_mv=0.0, extreme=0.0, window=0.25),
    # diagnostic classes: each isolates one access pattern of k_recon (bench.py --content)
    "static_1pct": dict(p_coded=0.9, intra=0.039, golden=0.003, zeromv=0.086, halfpel=0.025, p_dc_only=0.80,
                        p_zz10=0.15, amp=24, edge_mv=0.0, extreme=0.0, window=0.01),   # almost nothing changes
    "skip": dict(p_coded=0.0, intra=0.0, golden=0.0, zeromv=1.0, halfpel=0.0, p_dc_only=1.0,
                 p_zz10=0.0, amp=24, edge_mv=0.0, extreme=0.0),            # copy prev -> self only
    "zeromv_dc": dict(p_coded=1.0, intra=0.0, golden=0.0, zeromv=1.0, halfpel=0.0, p_dc_only=1.0,
                      p_zz10=0.0, amp=24, edge_mv=0.0, extreme=0.0),       # aligned predictor + DC, no slots
    "intra_dense": dict(p_coded=1.0, intra=1.0, golden=0.0, zeromv=1.0, halfpel=0.0, p_dc_only=0.0,
                        p_zz10=0.0, amp=40, edge_mv=0.0, extreme=0.0),     # coefficients + iDCT, no predictor
}


def gen_frame(geom, rng, frame_type, content="mixed", flimit=None, global_mv=None):
    """One frame's command stream as plain numpy arrays (see module docstring)."""
    P = CLASSES[content] if isinstance(content, str) else content
    N = geom.nfrags
    nmb = geom.nhmb * geom.nvmb
    intra_frame = frame_type == INTRA_FRAME
    # --- macro-block level decisions -----------------------------------------------------
    if intra_frame:
        mb_coded_p = np.ones(nmb)
        mb_refi = np.full(nmb, FRAME_SELF, np.uint8)
        mb_mvx = np.zeros(nmb, np.int32)
        mb_mvy = np.zeros(nmb, np.int32)
    else:
        u = rng.random(nmb)
        mb_refi = np.full(nmb, FRAME_PREV, np.uint8)
        mb_refi[u < P["intra"]] = FRAME_SELF
        mb_refi[(u >= P["intra"]) & (u < P["intra"] + P["golden"])] = FRAME_GOLD
        gx, gy = global_mv if global_mv is not None else (int(rng.integers(-12, 13)), int(rng.integers(-8, 9)))
        # whole-pel vectors are even in half-pel units
        mb_mvx = 2 * (gx // 2 + rng.integers(-2, 3, nmb))
        mb_mvy = 2 * (gy // 2 + rng.integers(-2, 3, nmb))
        half = rng.random(nmb) < P["halfpel"]
        mb_mvx = mb_mvx + np.where(half, rng.integers(-1, 2, nmb), 0)
        mb_mvy = mb_mvy + np.where(half, rng.integers(-1, 2, nmb), 0)
        zero = rng.random(nmb) < P["zeromv"]
        mb_mvx[zero] = 0
        mb_mvy[zero] = 0
        if P["edge_mv"]:
            big = rng.random(nmb) < P["edge_mv"]
            mb_mvx = np.where(big, rng.integers(-31, 32, nmb), mb_mvx)
            mb_mvy = np.where(big, rng.integers(-31, 32, nmb), mb_mvy)
        mb_mvx = np.clip(mb_mvx, -31, 31)
        mb_mvy = np.clip(mb_mvy, -31, 31)
        mb_mvx[mb_refi == FRAME_SELF] = 0
        mb_mvy[mb_refi == FRAME_SELF] = 0
        # spatially coherent coded probability
        mb_coded_p = np.clip(P["p_coded"] + 0.5 * (rng.random(nmb) - 0.5) * (P["p_coded"] < 1.0), 0.0, 1.0)
        if P.get("window"):
            # only a centred window of the picture (this fraction of its area) changes: static background
            f = float(P["window"]) ** 0.5
            my, mx = np.divmod(np.arange(nmb), geom.nhmb)
            inside = (np.abs(mx - geom.nhmb / 2) <= f * geom.nhmb / 2) & (np.abs(my - geom.nvmb / 2) <= f * geom.nvmb / 2)
            mb_coded_p = np.where(inside, mb_coded_p, 0.0)
    # --- per fragment -------------------------------------------------------------------
    refi = mb_refi[geom.mb_of]
    mvx = mb_mvx[geom.mb_of].astype(np.int32)
    mvy = mb_mvy[geom.mb_of].astype(np.int32)
    coded = rng.random(N) < mb_coded_p[geom.mb_of]
    if not intra_frame and P["p_coded"] >= 1.0:
        coded[:] = True
    if intra_frame:
        coded[:] = True
    refi = np.where(coded, refi, 3).astype(np.uint8)      # OC_FRAME_NONE for uncoded (decode.c:658)
    order = geom.coded_order
    is_coded = coded[order]
    coded_fragis = order[is_coded]
    uncoded_fragis = order[~is_coded][::-1].copy()        # the reference stores them reversed (state.h:423-426)
    ncoded = [int(coded[geom.froffset[p]:geom.froffset[p] + geom.pl_nfrags[p]].sum()) for p in range(3)]
    n = coded_fragis.size
    # --- coefficients: quantised levels x the frame's dequantisation tables (decode.c:1537-1538, 1573) ----------------
    u = rng.random(n)
    ncoef = np.where(u < P["p_dc_only"], 1,
                     np.where(u < P["p_dc_only"] + P["p_zz10"], rng.integers(2, 11, n), rng.integers(11, 65, n)))
    if content == "dense":
        ncoef[:] = 64
    zz = np.arange(64)
    # tables: dequant[plane][qii][qti][zig-zag index], falling with the index like the amplitudes the classes ask for
    # (amp / (1 + zz/4)); one to three qi per frame (nqis), the finer ones with smaller factors; inter tables a bit flatter
    nqis = int(rng.integers(1, 4))
    amp = np.maximum(P["amp"] / (1.0 + 0.25 * zz), 1.0)
    dequant = np.zeros((3, 3, 2, 64), np.uint16)
    for pl_ in range(3):
        for qi_ in range(3):
            for qt_ in range(2):
                f = (1.0 + 0.15 * pl_) * (1.0 - 0.3 * qi_) * (1.0 - 0.2 * qt_)
                dequant[pl_, qi_, qt_] = np.clip(np.rint(amp * f), 1, 65535)
    qti = (refi[coded_fragis] != FRAME_SELF).astype(np.int64)
    pl = geom.plane_of[coded_fragis].astype(np.int64)
    qii = rng.integers(0, nqis, n).astype(np.int64)
    # levels: small integers (products of two draws, -9..9), now and then a large one (a level beyond eight bits makes its
    # tile a wide one in the levels form)
    Z = rng.integers(-3, 4, (n, 64)).astype(np.int32) * rng.integers(1, 4, (n, 64))
    if P.get("big_levels"):
        big = rng.random((n, 64)) < P["big_levels"]
        Z = np.where(big, Z * rng.integers(8, 64, (n, 64)), Z)
    Z[zz[None, :] >= ncoef[:, None]] = 0
    last_zzi = np.minimum(ncoef, 63).astype(np.uint8)
    dc_only_eob0 = (ncoef == 1) & (rng.random(n) < 0.5)
    last_zzi[dc_only_eob0] = 0                            # an EOB run reaching the block: last_zzi==0
    if P["extreme"]:
        ex = rng.random(n) < P["extreme"]
        Z[ex] = rng.integers(-32768, 32768, (int(ex.sum()), 64))
        # blocks whose last_zzi claims fewer coefficients than are present: the reference's
        # iDCT variants ignore the rest (idct.c:234-277)
        last_zzi[ex] = rng.integers(0, 64, int(ex.sum()))
        if rng.random() < 0.5:
            dequant[1, 0, 1, 1:] = rng.integers(1, 65536, 63)     # the whole 16-bit range of factors
    levels = np.zeros((n, 64), np.int16)
    levels[:, FZIG_ZAG] = Z.astype(np.int16)
    # what the reference's token expansion hands to the slot: (ogg_int16_t)(coeff * ac_quant[zzi]), decode.c:1573
    prod = (Z.astype(np.int64) * dequant[pl, qii, qti].astype(np.int64)).astype(np.int16)
    coeffs = np.zeros((n, 64), np.int16)
    coeffs[:, FZIG_ZAG] = prod
    dq_table = rng.integers(8, 120, (3, 2)).astype(np.uint16)
    if P["extreme"]:
        dq_table[2, 1] = 65535
    dc_quant = dq_table[pl, qti]
    # raw (un-predicted, not yet dequantised) DC; keep dc*dc_quant inside the range real
    # streams use unless the class asks for extremes
    dc_target = rng.integers(-1500, 1500, n)
    coeffs[:, 0] = (dc_target // dc_quant.astype(np.int64)).astype(np.int16)
    if P["extreme"]:
        exd = rng.random(n) < P["extreme"]
        coeffs[exd, 0] = rng.integers(-32768, 32768, int(exd.sum()))
    levels[:, 0] = coeffs[:, 0]
    if flimit is None:
        flimit = int(rng.choice([0, 2, 4, 15, 63]))
    return dict(frame_type=frame_type, coded_fragis=coded_fragis.astype(np.int64), ncoded=ncoded,
                uncoded_fragis=uncoded_fragis.astype(np.int64), refi=refi, mvx=mvx, mvy=mvy,
                coeffs=coeffs, last_zzi=last_zzi, dc_quant=dc_quant.astype(np.uint16), flimit=flimit,
                levels=levels, qii=qii.astype(np.uint8), dequant=dequant)


def widen_tiles(geom, frame, frac, rng):
    """A copy of `frame` in which a share `frac` of the tiles that hold a block with coefficients got ONE level beyond eight bits
    (300 at zig-zag index 1 of one of their blocks): in the levels form such a tile is wide (include/theora_hip.h THIP_SLOT_WIDE).
    Returns (frame, number of tiles widened)."""
    out = dict(frame)
    cf = frame["coded_fragis"]
    has = np.nonzero(frame["last_zzi"] >= 2)[0]            # coded blocks that carry AC coefficients
    if has.size == 0:
        return out, 0
    tile = geom.frag_pos[cf[has]] // 64
    tiles, first = np.unique(tile, return_index=True)
    pick = rng.random(tiles.size) < frac
    blocks = has[first[pick]]
    levels = np.array(frame["levels"], np.int16, copy=True)
    levels[blocks, FZIG_ZAG[1]] = 300
    out["levels"] = levels
    co = dequantise(geom, out)
    out["coeffs"] = co
    return out, int(pick.sum())


def nothing_coded(geom, frame, frame_type=None):
    """The frame with no coded fragment at all (decode.c:2764-2772: a duplicate of the previous frame)."""
    out = dict(frame)
    out.update(coded_fragis=np.zeros(0, np.int64), ncoded=[0, 0, 0], uncoded_fragis=geom.coded_order[::-1].copy(),
               coeffs=np.zeros((0, 64), np.int16), last_zzi=np.zeros(0, np.uint8), dc_quant=np.zeros(0, np.uint16),
               levels=np.zeros((0, 64), np.int16), qii=np.zeros(0, np.uint8))
    if frame_type is not None:
        out["frame_type"] = frame_type
    return out


def dequantise(geom, frame):
    """The slot's dequantised coefficients from a frame's levels, qii and tables: (ogg_int16_t)(coeff * ac_quant[zzi]) with
    ac_quant = dequant[plane][qii][qti], qti = the block is not intra (decode.c:1537-1538, 1573); the DC stays raw."""
    cf = frame["coded_fragis"]
    qti = (frame["refi"][cf] != FRAME_SELF).astype(np.int64)
    pl = geom.plane_of[cf].astype(np.int64)
    tab = np.asarray(frame["dequant"], np.int64)[pl, np.asarray(frame["qii"], np.int64), qti]      # [n, 64] zig-zag order
    lv = np.asarray(frame["levels"], np.int64).reshape(-1, 64)
    co = np.zeros_like(lv)
    co[:, FZIG_ZAG] = lv[:, FZIG_ZAG] * tab
    co = co.astype(np.int16)
    co[:, 0] = lv[:, 0].astype(np.int16)
    return co


def pack_frame(geom, frame, form=None):
    """numpy command stream -> the device layout of include/theora_hip.h (host arrays).  form: "levels" (the quantised
    levels + the dequantisation tables, the default when the frame carries them) or "dequant16" (the slot's dequantised
    int16 coefficients)."""
    from . import SLOT_WIDE, info_words, pack_dequant_tables, pack_tiles, pack_units
    if form is None:
        form = "levels" if "levels" in frame else "dequant16"
    cf = frame["coded_fragis"]
    pos = geom.frag_pos[cf]
    assert (np.diff(pos) > 0).all(), "coded order must equal tile/lane order"
    lz = frame["last_zzi"]
    co = np.asarray(frame["coeffs"], np.int16).reshape(-1, 64)   # AC dequantised, DC raw (the slot's _dct_coeffs)
    has = lz >= 2
    common = dict(ncoded=int(cf.size), frame_type=frame["frame_type"], flimit=frame["flimit"])
    if form == "dequant16":
        info = info_words(geom.ntiles * 64, pos, frame["refi"][cf], lz, frame["mvx"][cf], frame["mvy"][cf], co[:, 0],
                          frame["dc_quant"])
        # first slot of every tile: number of coefficient-carrying fragments in earlier tiles
        per_tile = np.bincount(pos[has] >> 6, minlength=geom.ntiles)
        slot0 = np.concatenate([[0], np.cumsum(per_tile)[:-1]]).astype(np.uint32)
        return dict(info=info, coeffs=pack_tiles(co[has]), slot0=slot0, nslots=int(has.sum()), **common)
    assert form == "levels"
    lv = np.asarray(frame["levels"], np.int16).reshape(-1, 64).copy()
    lv[:, 0] = 0                                                  # the DC rides in command word 1
    info = info_words(geom.ntiles * 64, pos, frame["refi"][cf], lz, frame["mvx"][cf], frame["mvy"][cf], co[:, 0],
                      frame["dc_quant"], qii=frame["qii"])
    tile = pos[has] >> 6
    big = (np.abs(lv[has].astype(np.int32)) > 127).any(axis=1) | (lv[has] == -128).any(axis=1)   # (-128 fits, but keep the range symmetric)
    wide_tile = np.zeros(geom.ntiles, bool)
    wide_tile[tile[big]] = True
    per_tile = np.bincount(tile, minlength=geom.ntiles) * np.where(wide_tile, 2, 1)          # units
    unit0 = np.concatenate([[0], np.cumsum(per_tile)[:-1]]).astype(np.int64)
    rank = np.arange(tile.size) - np.searchsorted(tile, tile, side="left")                   # rank of a block inside its tile
    wide = wide_tile[tile]
    first_unit = unit0[tile] + rank * np.where(wide, 2, 1)
    nunits = int(per_tile.sum())
    slot0 = (unit0.astype(np.uint32) | np.where(wide_tile, np.uint32(SLOT_WIDE), np.uint32(0))).astype(np.uint32)
    return dict(info=info, coeffs=pack_units(lv[has], wide, first_unit, nunits), slot0=slot0, nslots=nunits,
                dequant=pack_dequant_tables(frame["dequant"]), wide_tiles=int(wide_tile.sum()), **common)


def upload_frame(packed, device="cuda"):
    """Host arrays -> HBM (torch is only the allocator here).  Returns (FrameDesc, keepalive)."""
    import torch

    from . import make_desc

    def dev(a, dt):
        if a.size == 0:
            return None
        return torch.from_numpy(np.ascontiguousarray(a).view(dt)).to(device)
    info = dev(packed["info"].reshape(-1), np.int32)
    coeffs = dev(packed["coeffs"], np.uint8 if "dequant" in packed else np.int16)
    slot0 = dev(packed["slot0"], np.int32)
    dq = dev(packed["dequant"].reshape(-1), np.int16) if "dequant" in packed else None
    desc = make_desc(info, coeffs, slot0, packed["nslots"], packed["ncoded"], packed["frame_type"],
                     packed["flimit"], dequant_dev=dq)
    return desc, (info, coeffs, slot0, dq)


def algorithmic_bytes(geom, frame):
    """B_alg of SURVEY.md section 8(d) for one frame: 128*n_coded + 8*N + 64*(n_coded_inter +
    n_uncoded) + 64*N, and B_read = B_alg - 64*N."""
    n_coded = int(frame["coded_fragis"].size)
    n_unc = int(frame["uncoded_fragis"].size)
    n_inter = int((frame["refi"][frame["coded_fragis"]] != FRAME_SELF).sum())
    N = geom.nfrags
    b_alg = 128 * n_coded + 8 * N + 64 * (n_inter + n_unc) + 64 * N
    return b_alg, b_alg - 64 * N
