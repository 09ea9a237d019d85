"""streamgen.py -- TEST INFRASTRUCTURE: a seeded generator of syntactically valid Theora
packets (three header packets + data packets) together with the ground truth of what
they mean (coded flags, modes, vectors, qi indices, coefficients, last_zzi).

It is NOT an encoder: coefficients are random, not the transform of any picture.  It
exists because the reference encoder cannot be built here and the repository holds no
sample streams; written from the bitstream specification (doc/spec/spec.tex section 6-7)
independently of the C++ front end it tests (theora_amd/csrc/thip_frontend.cpp).  Every
syntax element the decoder knows is exercised: all eight mode schemes, both vector
codings, 4MV with uncoded luma blocks, 1-3 qi values, every DCT token, EOB runs crossing
lists and planes, pure zero runs reaching coefficient 63, custom Huffman trees, all three
pixel formats.
"""
import heapq

import numpy as np

from theora_amd import synth

ZIGZAG = synth.FZIG_ZAG
MB_ORDER = [(0, 0), (1, 0), (1, 1), (0, 1)]
MODE_ALPHABETS = [[3, 4, 2, 0, 1, 5, 6, 7], [3, 4, 0, 2, 1, 5, 6, 7], [3, 2, 4, 0, 1, 5, 6, 7],
                  [3, 2, 0, 4, 1, 5, 6, 7], [0, 3, 4, 2, 1, 5, 6, 7], [0, 5, 3, 4, 2, 1, 6, 7]]
MODE_REFI = [1, 2, 1, 1, 1, 0, 0, 1]          # THIP_FRAME_*: GOLD 0, PREV 1, SELF 2
(INTER_NOMV, INTRA, INTER_MV, INTER_MV_LAST, INTER_MV_LAST2, GOLDEN_NOMV, GOLDEN_MV, INTER_MV_FOUR) = range(8)


class BitWriter:
    def __init__(self):
        self.bits = []

    def write(self, value, nbits):
        for i in range(nbits - 1, -1, -1):
            self.bits.append((value >> i) & 1)

    def code(self, s):
        self.bits.extend(int(c) for c in s)

    def bytes(self):
        b = self.bits + [0] * (-len(self.bits) % 8)
        return bytes(int("".join(map(str, b[i:i + 8])), 2) for i in range(0, len(b), 8))


def ilog(v):
    return int(v).bit_length()


# ---------------------------------------------------------------------------------------
# setup header contents
# ---------------------------------------------------------------------------------------
class Setup:
    def __init__(self, rng):
        self.lflims = [int(v) for v in np.clip(np.round(np.linspace(40, 0, 64) + rng.integers(-2, 3, 64)), 0, 127)]
        self.acscale = [int(v) for v in np.round(np.geomspace(500, 12, 64))]
        self.dcscale = [int(v) for v in np.round(np.geomspace(220, 10, 64))]
        base = np.add.outer(np.arange(8), np.arange(8))
        self.bms = [np.clip(16 + 6 * base + rng.integers(0, 6, (8, 8)), 1, 255).reshape(-1),
                    np.clip(17 + 9 * base + rng.integers(0, 6, (8, 8)), 1, 255).reshape(-1),
                    np.clip(16 + 2 * base + rng.integers(0, 4, (8, 8)), 1, 255).reshape(-1),
                    np.clip(20 + 3 * base + rng.integers(0, 9, (8, 8)), 1, 255).reshape(-1)]
        # quant ranges per (qti, pli): sizes summing to 63 and base-matrix indices; some sets are copies
        self.qr = {}
        self.qr_how = {}
        self.qr[(0, 0)] = ([20, 43], [0, 3, 1])
        self.qr_how[(0, 0)] = "new"
        self.qr[(0, 1)] = ([63], [1, 1])
        self.qr_how[(0, 1)] = "new"
        self.qr[(0, 2)] = self.qr[(0, 1)]
        self.qr_how[(0, 2)] = "prev"            # NEWQR=0, most recent set
        self.qr[(1, 0)] = ([1, 31, 31], [2, 2, 3, 0])
        self.qr_how[(1, 0)] = "new"
        self.qr[(1, 1)] = self.qr[(0, 1)]
        self.qr_how[(1, 1)] = "same_plane"      # NEWQR=0, RPQR=1
        self.qr[(1, 2)] = self.qr[(1, 1)]
        self.qr_how[(1, 2)] = "prev"
        # 80 Huffman trees from random token weights
        self.codes = []
        for _ in range(80):
            w = rng.random(32) ** 3 + 1e-3
            self.codes.append(huffman_codes(w))

    def qmat(self, qti, pli, qi):
        """spec 6.4.3; natural order."""
        sizes, bmis = self.qr[(qti, pli)]
        qri, start = 0, 0
        while qri < len(sizes) - 1 and qi > start + sizes[qri]:
            start += sizes[qri]
            qri += 1
        size = sizes[qri]
        end = start + size
        bmi, bmj = self.bms[bmis[qri]].astype(np.int64), self.bms[bmis[qri + 1]].astype(np.int64)
        bm = (2 * (end - qi) * bmi + 2 * (qi - start) * bmj + size) // (2 * size)
        out = np.empty(64, np.int64)
        for ci in range(64):
            qmin = (16 if qti == 0 else 32) if ci == 0 else (8 if qti == 0 else 16)
            qscale = self.dcscale[qi] if ci == 0 else self.acscale[qi]
            out[ci] = max(qmin, min((qscale * int(bm[ci]) // 100) * 4, 4096))
        return out

    def pack(self):
        bw = BitWriter()
        bw.write(0x82, 8)
        for c in b"theora":
            bw.write(c, 8)
        nb = max(1, max(ilog(v) for v in self.lflims))
        bw.write(nb, 3)
        for v in self.lflims:
            bw.write(v, nb)
        for arr in (self.acscale, self.dcscale):
            nb = max(ilog(v) for v in arr)
            bw.write(nb - 1, 4)
            for v in arr:
                bw.write(v, nb)
        nbms = len(self.bms)
        bw.write(nbms - 1, 9)
        for bm in self.bms:
            for v in bm:
                bw.write(int(v), 8)
        for qti in range(2):
            for pli in range(3):
                how = self.qr_how[(qti, pli)]
                if qti > 0 or pli > 0:
                    bw.write(1 if how == "new" else 0, 1)
                if how != "new":
                    if qti > 0:
                        bw.write(1 if how == "same_plane" else 0, 1)
                    continue
                sizes, bmis = self.qr[(qti, pli)]
                qi = 0
                bw.write(bmis[0], ilog(nbms - 1))
                for k, sz in enumerate(sizes):
                    bw.write(sz - 1, ilog(62 - qi))
                    qi += sz
                    bw.write(bmis[k + 1], ilog(nbms - 1))
                assert qi == 63
        for codes in self.codes:
            write_tree(bw, codes)
        return bw.bytes()


def huffman_codes(weights):
    """dict token -> bit string, a full prefix code over all 32 tokens."""
    heap = [(float(w), i, ("leaf", i)) for i, w in enumerate(weights)]
    heapq.heapify(heap)
    n = len(heap)
    while len(heap) > 1:
        a = heapq.heappop(heap)
        b = heapq.heappop(heap)
        heapq.heappush(heap, (a[0] + b[0], n, ("node", a[2], b[2])))
        n += 1
    codes = {}

    def walk(t, prefix):
        if t[0] == "leaf":
            codes[t[1]] = prefix
        else:
            walk(t[1], prefix + "0")
            walk(t[2], prefix + "1")
    walk(heap[0][2], "")
    return codes


def write_tree(bw, codes):
    """spec 6.4.4: pre-order, 0 = internal node, 1 + 5-bit token = leaf."""
    inv = {c: t for t, c in codes.items()}

    def rec(prefix):
        if prefix in inv:
            bw.write(1, 1)
            bw.write(inv[prefix], 5)
        else:
            bw.write(0, 1)
            rec(prefix + "0")
            rec(prefix + "1")
    rec("")


# ---------------------------------------------------------------------------------------
# primitive codes
# ---------------------------------------------------------------------------------------
def write_long_runs(bw, bits, rng=None):
    """spec 7.2.1 (inverse)."""
    n = len(bits)
    if n == 0:
        return
    i = 0
    bw.write(int(bits[0]), 1)
    cur = int(bits[0])
    while i < n:
        j = i
        while j < n and int(bits[j]) == cur and j - i < 4129:
            j += 1
        run = j - i
        for code, start, nb in (("0", 1, 0), ("10", 2, 1), ("110", 4, 1), ("1110", 6, 2), ("11110", 10, 3),
                                ("111110", 18, 4), ("111111", 34, 12)):
            if run < start + (1 << nb):
                bw.code(code)
                bw.write(run - start, nb)
                break
        i = j
        if i >= n:
            break
        if run == 4129:
            cur = int(bits[i])
            bw.write(cur, 1)
        else:
            cur = 1 - cur
            assert int(bits[i]) == cur


def write_short_runs(bw, bits):
    """spec 7.2.2 (inverse)."""
    n = len(bits)
    if n == 0:
        return
    i = 0
    cur = int(bits[0])
    bw.write(cur, 1)
    while i < n:
        j = i
        while j < n and int(bits[j]) == cur and j - i < 30:
            j += 1
        run = j - i
        for code, start, nb in (("0", 1, 1), ("10", 3, 1), ("110", 5, 1), ("1110", 7, 2), ("11110", 11, 2),
                                ("11111", 15, 4)):
            if run < start + (1 << nb):
                bw.code(code)
                bw.write(run - start, nb)
                break
        i = j
        if i < n and int(bits[i]) == cur:
            # a run longer than 30 cannot be expressed: force a flip in the data instead
            raise ValueError("short run longer than 30")
        cur = 1 - cur


def write_mv(bw, v, mvmode):
    if mvmode:
        bw.write(abs(v), 5)
        bw.write(1 if v < 0 else 0, 1)
        return
    a = abs(v)
    if a == 0:
        bw.code("000")
    elif a == 1:
        bw.code("001" if v > 0 else "010")
    else:
        if a < 4:
            bw.code("011" if a == 2 else "100")
        elif a < 8:
            bw.code("101")
            bw.write(a - 4, 2)
        elif a < 16:
            bw.code("110")
            bw.write(a - 8, 3)
        else:
            bw.code("111")
            bw.write(a - 16, 4)
        bw.write(1 if v < 0 else 0, 1)


def value_token(v):
    """(token, extra value, extra bits) of a lone coefficient value, |v| <= 580."""
    a, s = abs(v), 1 if v < 0 else 0
    if a == 1:
        return (10 if s else 9, 0, 0)
    if a == 2:
        return (12 if s else 11, 0, 0)
    if a <= 6:
        return (10 + a, s, 1)
    for tok, lo, nb in ((17, 7, 1), (18, 9, 2), (19, 13, 3), (20, 21, 4), (21, 37, 5), (22, 69, 9)):
        if a < lo + (1 << nb):
            return (tok, (s << nb) | (a - lo), nb + 1)
    raise ValueError(v)


def eob_token(run):
    if run <= 3:
        return (run - 1, 0, 0)
    if run <= 7:
        return (3, run - 4, 2)
    if run <= 15:
        return (4, run - 8, 3)
    if run <= 31:
        return (5, run - 16, 4)
    return (6, run, 12)


def tokenize_block(rng, nz):
    """nz: list of (zzi, value) with increasing zzi.  Returns [(zzi_read, token, extra, nbits)]
    WITHOUT the final EOB, plus the index at which the block would read its next token
    (64 if it is full).  Randomly picks among the legal ways of writing zero runs."""
    out = []
    cur = 0
    for z, v in nz:
        gap = z - cur
        a = abs(v)
        s = 1 if v < 0 else 0
        combo = None
        if gap > 0 and rng.random() < 0.7:
            if a == 1 and gap <= 5:
                combo = (22 + gap, s, 1)
            elif a == 1 and gap <= 9:
                combo = (28, (s << 2) | (gap - 6), 3)
            elif a == 1 and gap <= 17:
                combo = (29, (s << 3) | (gap - 10), 4)
            elif a in (2, 3) and gap == 1:
                combo = (30, (s << 1) | (a - 2), 2)
            elif a in (2, 3) and gap in (2, 3):
                combo = (31, (s << 2) | ((a - 2) << 1) | (gap - 2), 3)
        if combo:
            out.append((cur, combo[0], combo[1], combo[2]))
        else:
            while gap > 0:      # pure zero runs, possibly split
                r = gap if rng.random() < 0.6 else int(rng.integers(1, gap + 1))
                if r <= 8 and rng.random() < 0.7:
                    out.append((cur, 7, r - 1, 3))
                else:
                    out.append((cur, 8, r - 1, 6))
                cur += r
                gap -= r
            t = value_token(v)
            out.append((cur, t[0], t[1], t[2]))
        cur = z + 1
    return out, cur


# ---------------------------------------------------------------------------------------
# stream
# ---------------------------------------------------------------------------------------
class Stream:
    def __init__(self, width, height, fmt, seed, kfgshift=6, trees="random", probe_kwargs=None):
        """trees="random": the 80 Huffman trees come from random token weights, so that frequent
        tokens get long codes too (what the parity tests want).  trees="matched": the trees of each
        index group are built from the token statistics of probe frames of the same kind of content
        (frame() arguments in probe_kwargs), as an encoder's are -- what a throughput measurement
        wants, since code length is what a Huffman decoder's speed depends on."""
        self.rng = np.random.default_rng(seed)
        self.w, self.h, self.fmt = width, height, fmt
        self.geom = synth.Geometry(width, height, fmt)
        self.setup = Setup(self.rng)
        self.tok_hist = np.zeros((5, 32), np.int64)
        if trees == "matched":
            probe = Stream(width, height, fmt, seed + 7919, kfgshift)
            for ft in (0, 1, 1):
                probe.frame(ft, **(probe_kwargs or {}))
            self.setup.codes = []
            for hg in range(5):
                for _ in range(16):
                    w = (probe.tok_hist[hg] + 0.5) * (1 + 0.2 * self.rng.random(32))
                    self.setup.codes.append(huffman_codes(w))
        elif trees != "random":
            raise ValueError(trees)
        self.kfgshift = kfgshift
        g = self.geom
        # super blocks (all planes) as ranges of coded order, macro blocks in coded order
        self.sb_ranges = []
        k = 0
        for p in range(3):
            for sby in range(0, g.nv[p], 4):
                for sbx in range(0, g.nh[p], 4):
                    n = min(4, g.nv[p] - sby) * min(4, g.nh[p] - sbx)
                    self.sb_ranges.append((k, k + n))
                    k += n
        self.mbs = []
        yh, yv = g.nh[0], g.nv[0]
        for sby in range(0, yv, 4):
            for sbx in range(0, yh, 4):
                for (my, mx) in MB_ORDER:
                    y, x = sby + 2 * my, sbx + 2 * mx
                    if y >= yv or x >= yh:
                        continue
                    luma = [(y + i) * yh + x + j for i in range(2) for j in range(2)]
                    cx, cy = x >> g.hdec, y >> g.vdec
                    ncx, ncy = (1 if g.hdec else 2), (1 if g.vdec else 2)
                    chroma = []
                    for c in (1, 2):
                        slots = [-1] * 4
                        for i in range(ncy):
                            for j in range(ncx):
                                slots[i * 2 + j] = g.froffset[c] + (cy + i) * g.nh[c] + cx + j
                        chroma.append(slots)
                    self.mbs.append((luma, chroma))
        self.qmats = {}

    def qmat_zz(self, qti, pli, qi):
        key = (qti, pli, qi)
        if key not in self.qmats:
            self.qmats[key] = self.setup.qmat(qti, pli, qi)[ZIGZAG]
        return self.qmats[key]

    # ---- headers -------------------------------------------------------------------------
    def header_packets(self):
        bw = BitWriter()
        bw.write(0x80, 8)
        for c in b"theora":
            bw.write(c, 8)
        for v, n in ((3, 8), (2, 8), (1, 8), (self.w >> 4, 16), (self.h >> 4, 16), (self.w, 24), (self.h, 24),
                     (0, 8), (0, 8), (30, 32), (1, 32), (1, 24), (1, 24), (0, 8), (0, 24), (32, 6),
                     (self.kfgshift, 5), (self.fmt, 2), (0, 3)):
            bw.write(v, n)
        info = bw.bytes()
        vendor = b"theora-hip streamgen"
        comment = bytes([0x81]) + b"theora" + len(vendor).to_bytes(4, "little") + vendor + (1).to_bytes(4, "little") \
            + (9).to_bytes(4, "little") + b"TITLE=gen"
        return [info, comment, self.setup.pack()]

    # ---- one data packet ---------------------------------------------------------------------
    def frame(self, frame_type, density=0.6, nqis=None, force_qis=None, p_dc_only=0.3, p_empty=0.15):
        """Returns (packet bytes, truth dict)."""
        rng, g = self.rng, self.geom
        N = g.nfrags
        order = g.coded_order
        bw = BitWriter()
        bw.write(0, 1)
        bw.write(frame_type, 1)
        if force_qis is not None:
            qis = list(force_qis)
        else:
            nq = int(nqis if nqis is not None else rng.integers(1, 4))
            qis = [int(v) for v in rng.choice(64, nq, replace=False)]
        for i, q in enumerate(qis):
            bw.write(q, 6)
            if i < 2:
                bw.write(1 if i + 1 < len(qis) else 0, 1)
        coded = np.zeros(N, bool)
        if frame_type == 0:
            bw.write(0, 3)
            coded[:] = True
        else:
            # 7.3: super blocks uncoded / fully coded / partially coded
            nsb = len(self.sb_ranges)
            state = rng.choice(3, nsb, p=[max(0.0, 1 - density) * 0.7, density * 0.6, 1 - max(0.0, 1 - density) * 0.7 - density * 0.6])
            sbp = (state == 2).astype(np.uint8)
            write_long_runs(bw, sbp)
            sbf = (state[state != 2] == 1).astype(np.uint8)
            write_long_runs(bw, sbf)
            bbits = []
            for s, (a, b) in enumerate(self.sb_ranges):
                if state[s] == 1:
                    coded[order[a:b]] = True
                elif state[s] == 2:
                    bits = rng.random(b - a) < 0.5
                    coded[order[a:b]] = bits
                    bbits.extend(bits.astype(np.uint8))
            # runs longer than 30 cannot be written in a short-run string: break them
            i = 0
            while i < len(bbits):
                j = i
                while j < len(bbits) and bbits[j] == bbits[i]:
                    j += 1
                if j - i > 30:
                    bbits[i + 30] ^= 1
                    j = i + 30
                i = j
            # the flips above must be reflected in `coded`
            k = 0
            for s, (a, b) in enumerate(self.sb_ranges):
                if state[s] == 2:
                    coded[order[a:b]] = np.array(bbits[k:k + b - a], bool)
                    k += b - a
            write_short_runs(bw, bbits)
        ncoded_total = int(coded.sum())
        refi = np.full(N, 3, np.uint8)
        mvx = np.zeros(N, np.int32)
        mvy = np.zeros(N, np.int32)
        frag_mode = np.zeros(N, np.int32)
        if frame_type == 0:
            refi[:] = 2
            frag_mode[:] = INTRA
        elif ncoded_total:
            # 7.4 modes
            scheme = int(rng.integers(0, 8))
            bw.write(scheme, 3)
            if scheme == 0:
                alphabet = [int(v) for v in rng.permutation(8)]
                for mode in range(8):
                    bw.write(alphabet.index(mode), 3)
            elif scheme != 7:
                alphabet = MODE_ALPHABETS[scheme - 1]
            modes = []
            for (luma, chroma) in self.mbs:
                if coded[luma].any():
                    mode = int(rng.choice(self.mode_choices)) if getattr(self, "mode_choices", None) else int(rng.integers(0, 8))
                    if scheme != 7:
                        mi = alphabet.index(mode)
                        bw.code("1" * mi + ("0" if mi < 7 else ""))
                    else:
                        bw.write(mode, 3)
                else:
                    mode = INTER_NOMV
                modes.append(mode)
            # 7.5 motion vectors
            mvmode = int(rng.integers(0, 2))
            bw.write(mvmode, 1)
            last1, last2 = (0, 0), (0, 0)

            def rnd_mv():
                if getattr(self, "mv_choices", None):      # diagnostic knob: draw each component from a list
                    return (int(rng.choice(self.mv_choices)), int(rng.choice(self.mv_choices)))
                return (int(rng.integers(-31, 32)), int(rng.integers(-31, 32))) if rng.random() < 0.5 else \
                    (int(rng.integers(-4, 5)), int(rng.integers(-4, 5)))
            for (luma, chroma), mode in zip(self.mbs, modes):
                mv = (0, 0)
                lmv = [(0, 0)] * 4
                if mode == INTER_MV_FOUR:
                    for k in range(4):
                        if coded[luma[k]]:
                            lmv[k] = rnd_mv()
                            write_mv(bw, lmv[k][0], mvmode)
                            write_mv(bw, lmv[k][1], mvmode)
                            mv = lmv[k]
                    last2, last1 = last1, mv
                elif mode == GOLDEN_MV:
                    mv = rnd_mv()
                    write_mv(bw, mv[0], mvmode)
                    write_mv(bw, mv[1], mvmode)
                elif mode == INTER_MV_LAST2:
                    mv = last2
                    last2, last1 = last1, mv
                elif mode == INTER_MV_LAST:
                    mv = last1
                elif mode == INTER_MV:
                    mv = rnd_mv()
                    write_mv(bw, mv[0], mvmode)
                    write_mv(bw, mv[1], mvmode)
                    last2, last1 = last1, mv
                for k in range(4):
                    f = luma[k]
                    refi[f] = MODE_REFI[mode]
                    frag_mode[f] = mode
                    mvx[f], mvy[f] = lmv[k] if mode == INTER_MV_FOUR else mv
                for c in range(2):
                    for k in range(4):
                        f = chroma[c][k]
                        if f < 0:
                            continue
                        refi[f] = MODE_REFI[mode]
                        frag_mode[f] = mode
                        cmv = mv
                        if mode == INTER_MV_FOUR:
                            def rdiv(v, sh):
                                h = 1 << (sh - 1)
                                return (v + h) >> sh if v >= 0 else -((-v + h) >> sh)
                            if g.hdec and g.vdec:
                                cmv = (rdiv(sum(m[0] for m in lmv), 2), rdiv(sum(m[1] for m in lmv), 2))
                            elif g.hdec:
                                a = 0 if k == 0 else 2
                                cmv = (rdiv(lmv[a][0] + lmv[a + 1][0], 1), rdiv(lmv[a][1] + lmv[a + 1][1], 1))
                            else:
                                cmv = lmv[k]
                        mvx[f], mvy[f] = cmv
            refi[~coded] = 3
        truth = dict(frame_type=frame_type, qis=qis, coded=coded.copy())
        if ncoded_total == 0:
            truth["dup"] = True
            return bw.bytes(), truth
        truth["dup"] = False
        cf = order[coded[order]]
        # 7.6 block-level qi
        qii = np.zeros(N, np.int64)
        if len(qis) > 1:
            qii[cf] = rng.integers(0, len(qis), cf.size)
            for q in range(len(qis) - 1):
                sel = cf[qii[cf] >= q]
                write_long_runs(bw, (qii[sel] > q).astype(np.uint8))
        # 7.7 coefficients and tokens
        n = cf.size
        plane_of = g.plane_of[cf].astype(np.int64)
        blocks = []          # per coded block: token list, next index
        qcoef = np.zeros((n, 64), np.int64)   # quantised, zig-zag order
        last_zzi = np.zeros(n, np.int64)
        for i in range(n):
            u = rng.random()
            if u < p_empty or (getattr(self, "chroma_empty", False) and plane_of[i] > 0):
                nzpos = []           # (chroma_empty: diagnostic knob, grey pictures)
            elif u < p_empty + p_dc_only or (getattr(self, "chroma_dc_only", False) and plane_of[i] > 0):
                nzpos = [0]          # (chroma_dc_only: diagnostic knob, piecewise-constant chroma)
            else:
                cnt = int(rng.integers(1, 12)) if rng.random() < 0.8 else int(rng.integers(12, 65))
                nzpos = sorted(rng.choice(64, cnt, replace=False).tolist())
            nz = []
            for z in nzpos:
                r = rng.random()
                mag = 1 if r < 0.45 else int(rng.integers(2, 7)) if r < 0.8 else int(rng.integers(7, 69)) if r < 0.97 \
                    else int(rng.integers(69, 581))
                if getattr(self, "max_mag", None):
                    mag = min(mag, int(self.max_mag))   # diagnostic knob: keep the transform inside its valid range
                v = mag if rng.random() < 0.5 else -mag
                nz.append((z, v))
                qcoef[i, z] = v
            toks, nxt = tokenize_block(rng, nz)
            if nxt < 64 and rng.random() < 0.06:
                # finish with a pure zero run that reaches coefficient 63 instead of an EOB
                toks.append((nxt, 8, 64 - nxt - 1, 6))
                nxt = 64
            blocks.append((toks, nxt))
        # the index at which each block reads its final token (the reference's last_zzi)
        for i, (toks, nxt) in enumerate(blocks):
            last_zzi[i] = nxt if nxt < 64 else toks[-1][0]
        # global token order: index by index, plane by plane, coded order inside
        by_z = [dict((t[0], t) for t in toks) for toks, _ in blocks]
        per_plane = [np.nonzero(plane_of == p)[0] for p in range(3)]
        events = []          # (z, plane, block, kind, token tuple)
        for z in range(64):
            for p in range(3):
                for i in per_plane[p]:
                    t = by_z[i].get(z)
                    if t is not None:
                        events.append((z, p, int(i), "tok", t))
                    elif blocks[i][1] == z:
                        events.append((z, p, int(i), "eob", None))
        # merge consecutive EOBs into runs (they may cross planes and indices)
        lists = {}
        k = 0
        while k < len(events):
            z, p, i, kind, t = events[k]
            if kind == "tok":
                lists.setdefault((z, p), []).append(t[1:])
                k += 1
                continue
            j = k
            while j < len(events) and events[j][3] == "eob":
                j += 1
            total = j - k
            while total > 0:
                r = int(min(total, rng.integers(1, 40) if rng.random() < 0.8 else rng.integers(1, 4096)))
                tok = eob_token(r) if rng.random() < 0.85 else (6, r, 12)
                zz, pp = events[j - total][0], events[j - total][1]
                lists.setdefault((zz, pp), []).append(tok)
                total -= r
            k = j
        htis = None
        for z in range(64):
            if z < 2:
                htis = (int(rng.integers(0, 16)), int(rng.integers(0, 16)))
                bw.write(htis[0], 4)
                bw.write(htis[1], 4)
            hg = 0 if z == 0 else 1 if z <= 5 else 2 if z <= 14 else 3 if z <= 27 else 4
            for p in range(3):
                codes = self.setup.codes[16 * hg + (htis[0] if p == 0 else htis[1])]
                for (tok, extra, nb) in lists.get((z, p), []):
                    self.tok_hist[hg, tok] += 1
                    bw.code(codes[tok])
                    bw.write(extra, nb)
        truth.update(coded_fragis=cf, qii=qii, refi=refi, mvx=mvx, mvy=mvy, frag_mode=frag_mode, qcoef=qcoef,
                     last_zzi=last_zzi, flimit=self.setup.lflims[qis[0]])
        return bw.bytes(), truth

    def oracle_inputs(self, truth, ost):
        """Ground truth -> the arguments of oracle.State.decode_frame (DC un-prediction done
        by the oracle's restatement of decode.c:1392-1500)."""
        g = self.geom
        cf = truth["coded_fragis"]
        n = cf.size
        qis = truth["qis"]
        plane_of = g.plane_of[cf].astype(np.int64)
        qti = (truth["frag_mode"][cf] != INTRA).astype(np.int64)
        coeffs = np.zeros((n, 64), np.int16)
        dcq = np.zeros(n, np.uint16)
        for i in range(n):
            acq = self.qmat_zz(int(qti[i]), int(plane_of[i]), qis[int(truth["qii"][cf[i]])])
            coeffs[i, ZIGZAG] = (truth["qcoef"][i] * acq).astype(np.int64).astype(np.int16)
            dcq[i] = self.qmat_zz(int(qti[i]), int(plane_of[i]), qis[0])[0]
        # DC un-prediction on the coded residuals
        ost.coded[:] = truth["coded"]
        ost.refi[:] = truth["refi"]
        ost.dc[:] = 0
        ost.dc[cf] = truth["qcoef"][:, 0].astype(np.int16)
        ost.dc_unpredict()
        coeffs[:, 0] = ost.dc[cf]
        ost.mvs[:] = ((truth["mvx"] & 0xFF) | (truth["mvy"] << 8)).astype(np.int16)
        ncoded = [int(truth["coded"][g.froffset[p]:g.froffset[p] + g.pl_nfrags[p]].sum()) for p in range(3)]
        unc = g.coded_order[~truth["coded"][g.coded_order]]
        return dict(frame_type=truth["frame_type"], coded_fragis=cf, ncoded=ncoded, coeffs=coeffs,
                    last_zzi=truth["last_zzi"].astype(np.uint8), dc_quant=dcq, uncoded_fragis=unc,
                    flimit=truth["flimit"])
