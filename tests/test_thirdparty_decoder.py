"""Cross-check against a third-party Theora decoder that happens to be in this image: the
`kaleido` package bundles a headless Chromium 88 whose media stack (FFmpeg's own Theora decoder,
an implementation independent of libtheora and of this repository) plays Ogg/Theora.  Streams
from tests/streamgen.py are wrapped with tests/oggmux.py, played in that browser, read back
from a canvas, and compared with the oracle's pictures converted to RGB.

What this pins, and what it does not.  It is NOT the reference decoder, so the oracle stays
"parity unpinned" in the sense of the build contract.  But it is a decoder neither derived from
this repository's reading of the specification nor sharing code with it, and it agrees with the
oracle (and so with the HIP path, which equals the oracle bit for bit) to within RGB rounding on
every frame of multi-frame sequences that use all eight coding modes, both vector codings, 4MV,
golden-frame prediction, vectors far outside the frame, 1-3 qi values per frame with block-level
qi, custom quantisation matrices and Huffman trees, EOB runs, and the loop filter at any limit.
Two differences exist and are by design of FFmpeg, not errors here: coefficients outside the
range a real encoder produces (FFmpeg's inverse DCT keeps 32-bit intermediates where the
specification truncates to 16 bits) -- the generator is therefore asked for in-range values --
and RGB conversion, which only lets in-gamut pixels be compared (4:4:4 streams, so that no chroma
resampling is involved).

Skips when kaleido / its Chromium cannot run (e.g. a box without the package)."""
import base64
import json
import os

import numpy as np
import pytest

import oracle
from tests import oggmux, streamgen

JS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "thirdparty", "chromium_theora.js")


@pytest.fixture(scope="module")
def browser():
    try:
        from kaleido.scopes.plotly import PlotlyScope
        scope = PlotlyScope(plotlyjs=JS)
        probe = json.loads(scope.transform({"data": [{"ogv": "", "nframes": 0, "fps": 30}], "layout": {}}, format="json"))
    except Exception as e:   # noqa: BLE001 -- any failure to start the browser means "not available here"
        pytest.skip("no usable kaleido/Chromium: %r" % (e,))
    if "frames" not in probe:
        pytest.skip("kaleido/Chromium did not run the stand-in script")
    return scope


def play(scope, ogv, nframes):
    out = json.loads(scope.transform({"data": [{"ogv": base64.b64encode(ogv).decode(), "nframes": nframes, "fps": 30}],
                                      "layout": {}}, format="json"))
    assert "error" not in out, out.get("error")
    return out


def make_clip(w, h, seed, nframes, lflim, max_mag, grey, force_qis=None, fmt=3, chroma_dc_only=False, trees="random"):
    from theora_amd.decoder import Decoder
    st = streamgen.Stream(w, h, fmt, seed=seed, trees=trees)         # 4:4:4 unless told otherwise
    st.chroma_dc_only = chroma_dc_only
    st.setup.lflims = [lflim] * 64
    st.max_mag = max_mag
    st.chroma_empty = grey
    hdr = st.header_packets()
    from theora_amd import _lib
    L = _lib.load()
    old = L.thip_option(b"fe_trace_backend")
    L.thip_set_option(b"fe_trace_backend", 1)
    try:
        dec = Decoder(hdr)                              # slot-trace context: only used for the granule positions
    finally:
        L.thip_set_option(b"fe_trace_backend", old)
    ost = oracle.State(w, h, fmt)
    ls = oggmux.LogicalStream(0x7E0 + seed)
    for k, p in enumerate(hdr):
        ls.add_packet(p, granulepos=0, flush=(k == 0 or k == len(hdr) - 1))
    want, modes = [], set()
    for f in range(nframes):
        pkt, truth = st.frame(0 if f % 5 == 0 else 1, density=0.8, p_empty=0.05, force_qis=force_qis)
        rc, gp = dec.packetin(pkt)
        ls.add_packet(pkt, granulepos=gp, flush=True, eos=(f == nframes - 1))
        if not truth["dup"]:
            assert ost.decode_frame(**st.oracle_inputs(truth, ost)) == 0
            modes |= set(int(m) for m in truth["frag_mode"][truth["coded"]])
        want.append([ost.get_plane(oracle.FRAME_PREV, p)[::-1].astype(np.float64) for p in range(3)])
    dec.close()
    return b"".join(ls.pages), want, modes


def _flat3x3(c):
    p = np.pad(c, 1, mode="edge")
    m = np.ones(c.shape, bool)
    for dy in range(3):
        for dx in range(3):
            m &= p[dy:dy + c.shape[0], dx:dx + c.shape[1]] == c
    return m


def compare(out, want, w, h):
    """Per frame: (mean, worst 8x8-block mean) absolute RGB error over comparable pixels, and their
    share.  Comparable = in gamut and, for subsampled chroma, with a flat 3x3 chroma neighbourhood
    (there the browser's chroma resampling filter, whatever it is, returns the sample itself)."""
    res = []
    for f, planes in enumerate(want):
        rgb = np.frombuffer(base64.b64decode(out["frames"][f]), np.uint8).reshape(h, w, 3).astype(np.float64)
        Y, Cb, Cr = planes
        sub = np.ones(Y.shape, bool)
        if Cb.shape != Y.shape:
            ry, rx = Y.shape[0] // Cb.shape[0], Y.shape[1] // Cb.shape[1]
            sub = np.repeat(np.repeat(_flat3x3(Cb) & _flat3x3(Cr), ry, 0), rx, 1)
            Cb, Cr = np.repeat(np.repeat(Cb, ry, 0), rx, 1), np.repeat(np.repeat(Cr, ry, 0), rx, 1)
        raw = np.stack([1.164383 * (Y - 16) + 1.596027 * (Cr - 128),
                        1.164383 * (Y - 16) - 0.391762 * (Cb - 128) - 0.812968 * (Cr - 128),
                        1.164383 * (Y - 16) + 2.017232 * (Cb - 128)], -1)          # BT.601, limited range
        ok = ((raw > 6) & (raw < 249)).all(-1) & sub
        err = np.abs(np.clip(np.round(raw), 0, 255) - rgb).max(-1)
        err[~ok] = np.nan
        blocks = [np.nanmean(err[y:y + 8, x:x + 8]) for y in range(0, h, 8) for x in range(0, w, 8)
                  if not np.isnan(err[y:y + 8, x:x + 8]).all()]
        res.append((float(np.nanmean(err)), float(max(blocks)), float(ok.mean())))
    return res


@pytest.mark.parametrize("seed,lflim,max_mag,grey,qis", [
    (5, 0, 4, True, [50]),          # no loop filter, small coefficients, grey: everything but the filter
    (7, 6, 5, True, None),          # loop filter, 1-3 random qi per frame with block-level qi (any quantiser: small levels)
    (9, 4, 30, True, [52, 44]),     # larger levels with fine quantisers (products stay in the transform's valid range)
    (11, 30, 12, True, [20]),       # strong loop filter, coarse quantiser
    (13, 3, 8, False, [45, 38]),    # colour
])
def test_ffmpeg_in_chromium_agrees_with_the_oracle(browser, seed, lflim, max_mag, grey, qis):
    w, h, n = 64, 48, 10
    ogv, want, modes = make_clip(w, h, seed, n, lflim, max_mag, grey, qis)
    out = play(browser, ogv, n)
    assert (out["w"], out["h"]) == (w, h) and abs(out["duration"] - n / 30.0) < 1e-3
    assert len(out["frames"]) == n
    assert len(modes) >= 6                                   # the sequence really used the coding modes
    # The browser is asked for frame f by seeking to its mid-time; next to a key frame its Ogg seek
    # sometimes lands one frame late.  So every picture it returns must BE one of the oracle's
    # pictures (to within RGB rounding) at index f or f+-1, and nearly all of them at index f --
    # which, inter frames depending on all their predecessors, covers the whole sequence.
    exact = 0
    seen = {}
    for f in range(n):
        one = {"frames": [out["frames"][f]] * n}
        scores = compare(one, want, w, h)
        g = min(range(n), key=lambda i: scores[i][0])
        mean, worst_block, share = scores[g]
        assert abs(g - f) <= 1, (f, g)
        assert share > 0.15, (f, share)                           # enough in-gamut pixels to mean something
        assert mean < (0.15 if grey else 0.6), (f, g, mean)     # RGB rounding only
        assert worst_block < 1.5, (f, g, worst_block)           # no 8x8 block is off by more than conversion noise
        exact += g == f
        if grey and g == f:
            # grey pictures make the comparison EXACT: R = G = B is a function of Y alone and, its slope
            # being above 1, an injective one.  So over all in-gamut pixels the browser's R must be a
            # single-valued, injective function of the oracle's Y -- any +-1 disagreement in a decoded luma
            # sample would give one Y two different R.  (No formula for the conversion is assumed.)
            rgb = np.frombuffer(base64.b64decode(out["frames"][f]), np.uint8).reshape(h, w, 3)
            Y = want[f][0].astype(np.int64)
            sel = (Y >= 24) & (Y <= 228)
            assert (rgb[..., 0][sel] == rgb[..., 1][sel]).all() and (rgb[..., 1][sel] == rgb[..., 2][sel]).all()
            for y, r in zip(Y[sel].ravel(), rgb[..., 0][sel].ravel()):
                seen.setdefault(int(y), set()).add(int(r))
    assert exact >= n - 2
    if grey:
        assert len(seen) > 100                                        # most luma levels occurred
        assert all(len(v) == 1 for v in seen.values()), {k: v for k, v in seen.items() if len(v) > 1}
        rs = [next(iter(v)) for _, v in sorted(seen.items())]
        assert all(b > a for a, b in zip(rs, rs[1:]))                 # strictly increasing: injective


@pytest.mark.parametrize("fmt", [0, 2])
def test_subsampled_formats(browser, fmt):
    """4:2:0 and 4:2:2: chroma vectors are derived from the luma ones (averaged over the macro block
    for 4MV, quarter-pel units).  With DC-only chroma blocks the chroma planes are piecewise constant,
    so wherever a 3x3 chroma neighbourhood is flat the browser's resampling is exact and the pixel can
    be compared; a wrong chroma vector moves the flat regions."""
    w, h, n = 64, 48, 8
    # (4:2:2 with Huffman trees built from the content's statistics, 4:2:0 with random ones)
    ogv, want, modes = make_clip(w, h, 3, n, 0, 5, False, [50], fmt=fmt, chroma_dc_only=True,
                                 trees="matched" if fmt == 2 else "random")
    out = play(browser, ogv, n)
    assert len(out["frames"]) == n and len(modes) >= 6
    exact = 0
    for f in range(n):
        scores = compare({"frames": [out["frames"][f]] * n}, want, w, h)
        g = min(range(n), key=lambda i: scores[i][0])
        assert abs(g - f) <= 1 and scores[g][2] > 0.25, (f, g, scores[g])
        assert scores[g][0] < 0.5 and scores[g][1] < 1.5, (f, g, scores[g])
        exact += g == f
    assert exact >= n - 2


def test_the_check_has_teeth(browser):
    """The comparison notices a real decoding difference: the same clip compared with pictures of a
    differently seeded clip fails by a wide margin."""
    w, h, n = 64, 48, 4
    ogv, want, _ = make_clip(w, h, 21, n, 0, 6, True, [50])
    _, other, _ = make_clip(w, h, 22, n, 0, 6, True, [50])
    out = play(browser, ogv, n)
    assert max(m for m, _, _ in compare(out, want, w, h)) < 0.15
    assert min(m for m, _, _ in compare(out, other, w, h)) > 5.0
