"""spec_model.py -- TEST INFRASTRUCTURE ONLY.

An independent restatement of the NORMATIVE SPECIFICATION text of the Theora format
(/root/reference/doc/spec/spec.tex), written from the spec's enumerated procedures and
NOT from libtheora's C.  It exists to cross-check oracle/theora_oracle.c (which follows the
C line by line): two restatements made from two different descriptions agreeing bit for bit
is the strongest pin available while the reference cannot be built here (no libogg).

Covered: the 1D/2D inverse DCT (spec.tex:6255-6593, "The Inverse DCT"), the DC-only rule
and the per-pixel reconstruction (spec.tex:6751-7120, "The Complete Reconstruction
Algorithm"), the intra / whole-pixel / half-pixel predictors (spec.tex:5789-6083) and the
loop filter with its raster ordering (spec.tex:7125-7560).  Plain numpy; slow paths are
pure Python and meant for small cases.
"""
import numpy as np

# Table "16-bit Approximations of Sines and Cosines", spec.tex:6353-6368
C1 = S7 = 64277
C2 = S6 = 60547
C3 = S5 = 54491
C4 = S4 = 46341
C5 = S3 = 36410
C6 = S2 = 25080
C7 = S1 = 12785


def trunc16(v):
    """'Truncate to a 16-bit signed representation by dropping any higher-order bits.'"""
    v = np.asarray(v, np.int64)
    return ((v + 32768) & 0xFFFF) - 32768


def idct_1d(Y):
    """Section 7.9.3.1 'The 1D Inverse DCT', steps 1-56 (spec.tex:6370-6499), on the last
    axis (length 8) of an int64 array."""
    Y = np.asarray(Y, np.int64)
    y = [Y[..., i] for i in range(8)]
    T = [None] * 8
    T[0] = y[0] + y[4]
    T[0] = trunc16(T[0])
    T[0] = (C4 * T[0]) >> 16
    T[1] = y[0] - y[4]
    T[1] = trunc16(T[1])
    T[1] = (C4 * T[1]) >> 16
    T[2] = ((C6 * y[2]) >> 16) - ((S6 * y[6]) >> 16)
    T[3] = ((S6 * y[2]) >> 16) + ((C6 * y[6]) >> 16)
    T[4] = ((C7 * y[1]) >> 16) - ((S7 * y[7]) >> 16)
    T[5] = ((C3 * y[5]) >> 16) - ((S3 * y[3]) >> 16)
    T[6] = ((S3 * y[5]) >> 16) + ((C3 * y[3]) >> 16)
    T[7] = ((S7 * y[1]) >> 16) + ((C7 * y[7]) >> 16)
    R = T[4] + T[5]
    T[5] = T[4] - T[5]
    T[5] = trunc16(T[5])
    T[5] = (C4 * T[5]) >> 16
    T[4] = R
    R = T[7] + T[6]
    T[6] = T[7] - T[6]
    T[6] = trunc16(T[6])
    T[6] = (C4 * T[6]) >> 16
    T[7] = R
    R = T[0] + T[3]
    T[3] = T[0] - T[3]
    T[0] = R
    R = T[1] + T[2]
    T[2] = T[1] - T[2]
    T[1] = R
    R = T[6] + T[5]
    T[5] = T[6] - T[5]
    T[6] = R
    X = [trunc16(T[0] + T[7]), trunc16(T[1] + T[6]), trunc16(T[2] + T[5]), trunc16(T[3] + T[4]),
         trunc16(T[3] - T[4]), trunc16(T[2] - T[5]), trunc16(T[1] - T[6]), trunc16(T[0] - T[7])]
    return np.stack(X, axis=-1)


def idct_2d(DQC):
    """Section 7.9.3.2 'The 2D Inverse DCT' (spec.tex:6501-6593): rows, then columns of the
    result, then (x+8)>>4.  DQC [..., 64] natural order; returns RES [..., 8, 8]."""
    d = np.asarray(DQC, np.int64).reshape(np.shape(DQC)[:-1] + (8, 8))
    res = idct_1d(d)                                # each row ri: Y[ci] = DQC[ri*8+ci]
    res = np.swapaxes(idct_1d(np.swapaxes(res, -1, -2)), -1, -2)   # each column ci: Y[ri] = RES[ri][ci]
    return trunc16((res + 8) >> 4)


def residual(coeffs, ncoeffs, dc_quant):
    """Steps of spec.tex:7036-7068: NCOEFFS<2 -> DC=(COEFFS[0]*QMAT[0]+15)>>5 truncated to
    16 bits, replicated; otherwise dequantise and run the 2D transform.  `coeffs` natural
    order with the AC already dequantised and the raw DC in [...,0] (the form the reference's
    front end hands over, decode.c:1573-1581)."""
    co = np.asarray(coeffs, np.int64).reshape(-1, 64).copy()
    n = np.asarray(ncoeffs).reshape(-1)
    dq = np.asarray(dc_quant, np.int64).reshape(-1)
    out = np.empty((co.shape[0], 8, 8), np.int64)
    dconly = n < 2
    dc = trunc16((co[:, 0] * dq + 15) >> 5)
    out[dconly] = dc[dconly, None, None]
    full = ~dconly
    co[:, 0] = trunc16(co[:, 0] * dq)      # dequantised values are 16-bit quantities (spec.tex:6084-6209)
    out[full] = idct_2d(co[full])
    return out


def split_mv(v, subsampled):
    """MVX/MVX2 of spec.tex:7003-7021: the vector is in half-pel units (quarter-pel on a
    subsampled chroma axis); integer part truncating toward zero, second one away from zero."""
    den = 4 if subsampled else 2
    a = abs(int(v))
    s = (v > 0) - (v < 0)
    return (a // den) * s, -((-a) // den) * s


def predict(refp, bx, by, mvx, mvy, mvx2, mvy2):
    """Whole-pixel (spec.tex:5849-5947) or half-pixel (spec.tex:5949-6083) predictor with the
    spec's coordinate clamping; refp[ry][rx], row 0 at the bottom."""
    rph, rpw = refp.shape
    pred = np.empty((8, 8), np.int64)
    for j in range(8):
        ry1 = min(max(by + mvy + j, 0), rph - 1)
        ry2 = min(max(by + mvy2 + j, 0), rph - 1)
        for i in range(8):
            rx1 = min(max(bx + mvx + i, 0), rpw - 1)
            rx2 = min(max(bx + mvx2 + i, 0), rpw - 1)
            if mvx == mvx2 and mvy == mvy2:
                pred[j, i] = refp[ry1, rx1]
            else:
                pred[j, i] = (int(refp[ry1, rx1]) + int(refp[ry2, rx2])) >> 1
    return pred


def reconstruct_block(pred, res):
    """spec.tex:7097-7118: P = PRED + RES clamped to 0..255."""
    return np.clip(np.asarray(pred, np.int64) + np.asarray(res, np.int64), 0, 255).astype(np.uint8)


def lflim(R, L):
    """The piecewise definition of spec.tex:7140-7148."""
    if R <= -2 * L:
        return 0
    if R <= -L:
        return -R - 2 * L
    if R < L:
        return R
    if R < 2 * L:
        return -R + 2 * L
    return 0


def _clamp(p):
    return 0 if p < 0 else 255 if p > 255 else p


def filter_horizontal(recp, fx, fy, L):
    """Section 7.10.1 'Horizontal Filter' (spec.tex:7163-7240), in place."""
    for by in range(8):
        row = recp[fy + by]
        R = (int(row[fx]) - 3 * int(row[fx + 1]) + 3 * int(row[fx + 2]) - int(row[fx + 3]) + 4) >> 3
        f = lflim(R, L)
        p1 = _clamp(int(row[fx + 1]) + f)
        p2 = _clamp(int(row[fx + 2]) - f)
        row[fx + 1] = p1
        row[fx + 2] = p2


def filter_vertical(recp, fx, fy, L):
    """Section 7.10.2 'Vertical Filter' (spec.tex:7242-7338), in place."""
    for bx in range(8):
        x = fx + bx
        R = (int(recp[fy][x]) - 3 * int(recp[fy + 1][x]) + 3 * int(recp[fy + 2][x]) - int(recp[fy + 3][x]) + 4) >> 3
        f = lflim(R, L)
        p1 = _clamp(int(recp[fy + 1][x]) + f)
        p2 = _clamp(int(recp[fy + 2][x]) - f)
        recp[fy + 1][x] = p1
        recp[fy + 2][x] = p2


def loop_filter_plane(recp, bcoded, L):
    """Section 7.10.3 'Complete Loop Filter' (spec.tex:7340-7530) for one plane: blocks in
    raster order; left edge, bottom edge, then right / top edges towards uncoded
    neighbours.  recp [RPH][RPW] uint8 (row 0 = bottom), bcoded [nv][nh]."""
    recp = np.array(recp, np.int64)
    rph, rpw = recp.shape
    nv, nh = bcoded.shape
    if L == 0:
        return recp.astype(np.uint8)
    for r in range(nv):
        for c in range(nh):
            if not bcoded[r, c]:
                continue
            bx, by = c * 8, r * 8
            if bx > 0:
                filter_horizontal(recp, bx - 2, by, L)
            if by > 0:
                filter_vertical(recp, bx, by - 2, L)
            if bx + 8 < rpw and not bcoded[r, c + 1]:
                filter_horizontal(recp, bx + 6, by, L)
            if by + 8 < rph and not bcoded[r + 1, c]:
                filter_vertical(recp, bx, by + 6, L)
    return recp.astype(np.uint8)
