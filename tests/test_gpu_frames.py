"""Frame-level parity on the GPU: thip_decode_frames (recon + copy + loop filter + ring)
against the oracle's restatement of the reference's MCU loop, bit-exact."""
import numpy as np
import pytest

import oracle
from theora_amd import synth
from tests import util

pytestmark = pytest.mark.gpu

PF_420, PF_422, PF_444 = 0, 2, 3


@pytest.mark.parametrize("w,h,fmt", [(176, 144, PF_420), (64, 48, PF_444), (80, 112, PF_422),
                                     (16, 16, PF_420), (336, 16, PF_420), (16, 272, PF_444)])
@pytest.mark.parametrize("content", ["mixed", "smooth", "dense"])
def test_sequence_small(hip, w, h, fmt, content):
    rep = util.run_sequence(hip, w, h, fmt, nframes=12, content=content, seed=w * 7 + h + fmt, kf_interval=5)
    assert not rep, rep[:3]


@pytest.mark.parametrize("p_dc_only", [0.97, 0.75, 0.5, 0.0])
def test_lane_shared_transform_boundaries(hip, p_dc_only):
    """k_recon picks 4, 2 or 1 lanes per block from the number of coefficient-owning lanes in a tile
    (<=16, <=32, more).  Fully coded frames whose DC-only share puts that number around 2, 16, 32 and
    at 64, with blocks whose last_zzi claims fewer coefficients than are present (the masking each
    path must apply), decode bit-exactly."""
    content = dict(p_coded=1.0, intra=0.2, golden=0.1, zeromv=0.2, halfpel=0.4, p_dc_only=p_dc_only, p_zz10=0.4 * (1 - p_dc_only),
                   amp=200, edge_mv=0.2, extreme=0.1)
    rep = util.run_sequence(hip, 256, 96, PF_420, nframes=8, content=content, seed=int(p_dc_only * 100), kf_interval=4)
    assert not rep, rep[:3]


@pytest.mark.parametrize("w,h,fmt", [(256, 160, PF_420), (176, 144, PF_444), (336, 96, PF_422), (64, 48, PF_420)])
@pytest.mark.parametrize("window", [0.25, 0.6, 0.02])
def test_static_background_blocks_stay_in_place(hip, w, h, fmt, window):
    """Blocks that are uncoded in two consecutive frames and whose edge neighbours were uncoded in the first
    of them are not copied: the ring buffer they would be copied into holds them already (k_recon,
    launch_chunk).  A centred window of the picture changes, the rest is static, over key frames (which
    break the two-buffer rotation), golden references and edge vectors; every plane of every frame is
    compared, with the elision on (default) -- the same sequences with it off are the other tests."""
    content = dict(synth.CLASSES["mixed"], p_coded=0.85, window=window)
    rep = util.run_sequence(hip, w, h, fmt, nframes=14, content=content, seed=int(window * 100) + w, kf_interval=6)
    assert not rep, rep[:3]


def test_static_background_through_the_slots_and_across_dup_frames(hip):
    """The same through the one-fragment-at-a-time slots, and with frames in which nothing is coded in
    between (decode.c:2764-2772: the state does not move, so the next frame still finds the frame before
    the previous one in its buffer)."""
    content = dict(synth.CLASSES["mixed"], p_coded=0.85, window=0.3)
    rep = util.run_sequence(hip, 176, 144, PF_420, nframes=9, content=content, seed=77, kf_interval=5, enqueue=True)
    assert not rep, rep[:3]
    w, h = 256, 96
    geom = synth.Geometry(w, h, PF_420)
    rng = np.random.default_rng(5)
    ost = oracle.State(w, h, PF_420)
    gst = hip.State(w, h, PF_420)
    for f in range(12):
        fr = synth.gen_frame(geom, rng, hip.INTRA_FRAME if f == 0 else hip.INTER_FRAME, content)
        if f in (4, 5, 9):   # nothing coded
            fr = synth.nothing_coded(geom, fr)
        rc_o = util.oracle_apply(ost, fr)
        desc, ka = synth.upload_frame(synth.pack_frame(geom, fr))
        assert hip.decode_frames([gst], [desc])[0] == rc_o
        assert not util.planes_equal(ost, gst), f


def test_static_blocks_are_copied_again_after_the_caller_wrote_a_buffer(hip):
    """thip_state_write_plane into the buffer the next frame will be decoded into: the library must stop
    assuming that buffer still holds the frame before the previous one."""
    w, h = 256, 96
    content = dict(synth.CLASSES["mixed"], p_coded=0.85, window=0.1)
    geom = synth.Geometry(w, h, PF_420)
    rng = np.random.default_rng(8)
    ost = oracle.State(w, h, PF_420)
    gst = hip.State(w, h, PF_420)
    for f in range(8):
        fr = synth.gen_frame(geom, rng, hip.INTRA_FRAME if f == 0 else hip.INTER_FRAME, content)
        util.oracle_apply(ost, fr)
        desc, ka = synth.upload_frame(synth.pack_frame(geom, fr))
        hip.decode_frames([gst], [desc])
        assert not util.planes_equal(ost, gst), f
        if f == 5:   # scribble over the buffer that is neither GOLD nor PREV
            nxt = [b for b in range(3) if b not in (gst.ref_idx(hip.FRAME_GOLD), gst.ref_idx(hip.FRAME_PREV))][0]
            for pli in range(3):
                g = gst.planes[pli]
                gst.write_plane(nxt, pli, rng.integers(0, 256, (g["height"], g["width"])).astype(np.uint8))


def test_sequence_720p(hip):
    """BASELINE.json config 2: one 720p stream, more than a key-frame interval of 64, every plane of
    every frame compared."""
    rep = util.run_sequence(hip, 1280, 720, PF_420, nframes=66, content="mixed", seed=3, kf_interval=64)
    assert not rep, rep[:3]


@pytest.mark.parametrize("w,h,fmt,content,n", [(3840, 2160, PF_420, "smooth", 20), (3840, 2160, PF_420, "mixed", 20),
                                                 (3840, 2160, PF_420, "dense", 10), (1920, 1088, PF_444, "mixed", 20),
                                                 (1920, 1088, PF_422, "smooth", 20), (3840, 2160, PF_420, "static_bg", 20),
                                                 (1920, 1088, PF_420, "static_1pct", 20)])
def test_full_size_sequences(hip, w, h, fmt, content, n):
    """BASELINE.json's sizes against the oracle directly (it decodes a 4K frame in ~40 ms): every
    plane of every frame, key-frame interval 16, all three content classes and pixel formats, and the two
    static-background classes (blocks left in place, whole tiles and filter waves skipped)."""
    rep = util.run_sequence(hip, w, h, fmt, nframes=n, content=content, seed=w + n, kf_interval=16)
    assert not rep, rep[:3]


@pytest.mark.parametrize("w,h,fmt", [(7680, 4320, PF_420), (8192, 16, PF_444), (16, 8192, PF_422)])
def test_beyond_4k_and_extreme_shapes(hip, w, h, fmt):
    """Larger than BASELINE.json's largest size (8K: 64 800 luma tiles, plane offsets beyond 32 MB) and the two
    degenerate shapes -- one tile row 512 tiles wide, 1024 fragment rows one partial tile wide: key frame + two inter
    frames, every plane against the oracle."""
    rep = util.run_sequence(hip, w, h, fmt, nframes=3, content="mixed", seed=w + h, kf_interval=16)
    assert not rep, rep[:3]


def test_sequence_1080p_smooth(hip):
    rep = util.run_sequence(hip, 1920, 1088, PF_420, nframes=4, content="smooth", seed=4, kf_interval=64)
    assert not rep, rep[:3]


def test_config3_as_written(hip):
    """BASELINE.json config 3 literally: 1920x1080 (coded 1920x1088) 4:2:0, key-frame interval 64, 130 frames (two interval
    boundaries), ONE stream on one state, the frames in flight behind each other; every 8th frame and the last are
    compared with the oracle."""
    assert util.run_sequence(hip, 1920, 1088, PF_420, 130, "smooth", seed=1080, kf_interval=64, check_every=8) == []


def test_gop_parallel_through_the_c_abi(hip):
    """A stream's key-frame intervals are independent (a key frame resets both references, decode.c:2947-2955), so a caller
    may decode G of them side by side on G states -- bench.py --gop-parallel, DESIGN.md section 5c.  Here through the C ABI:
    12 intervals of 9 frames, G = 4 states, state g takes intervals g, g + 4, g + 8 (three rounds, i.e. every state crosses two
    interval boundaries), the four states' frames of a step in ONE thip_decode_frames call; every frame of every interval
    against ONE oracle that decodes the stream sequentially."""
    import zlib
    w, h, K, G, rounds = 336, 272, 9, 4, 3
    geom = synth.Geometry(w, h, PF_420)
    rng = np.random.default_rng(4242)
    ost = oracle.State(w, h, PF_420)
    frames, want = [], []
    for i in range(G * rounds * K):
        fr = synth.gen_frame(geom, rng, hip.INTRA_FRAME if i % K == 0 else hip.INTER_FRAME, ["mixed", "smooth", "dense"][(i // K) % 3])
        util.oracle_apply(ost, fr)
        frames.append(fr)
        want.append([ost.get_plane(oracle.FRAME_PREV, p).copy() for p in range(3)])
    states = [hip.State(w, h, PF_420) for _ in range(G)]
    keep = []
    for r in range(rounds):
        for j in range(K):
            idx = [(r * G + g) * K + j for g in range(G)]          # frame j of interval r*G + g for state g
            descs = []
            for i in idx:
                d, ka = synth.upload_frame(synth.pack_frame(geom, frames[i]))
                keep.append(ka)
                descs.append(d)
            assert list(hip.decode_frames(states, descs)) == [0] * G
            for g, i in enumerate(idx):
                for p in range(3):
                    got = states[g].read_plane(states[g].ref_idx(hip.FRAME_PREV), p)
                    assert np.array_equal(got, want[i][p]), (r, j, g, p)


def test_enqueue_path_matches(hip):
    """The one-fragment-at-a-time vtable slots (thip_state_frag_recon, thip_frag_copy_list,
    thip_state_loop_filter_frag_rows) driven in the reference's MCU order."""
    rep = util.run_sequence(hip, 176, 144, PF_420, nframes=7, content="mixed", seed=11, kf_interval=4,
                            enqueue=True)
    assert not rep, rep[:3]
    rep = util.run_sequence(hip, 64, 80, PF_444, nframes=5, content="mixed", seed=12, kf_interval=4,
                            enqueue=True)
    assert not rep, rep[:3]


@pytest.mark.parametrize("w,h,fmt", [(336, 272, PF_420), (176, 144, PF_444), (1280, 720, PF_420)])
def test_partial_loop_filter_row_ranges_through_the_fused_pass(hip, w, h, fmt):
    """thip_state_loop_filter_frag_rows with a fragment-row range that is NOT the whole plane (state.c:1066: only the edges the
    fragments of rows [fragy0, fragy_end) trigger), on frames that take k_recon_lf: the range reaches the cells as lf_y0 / lf_y1,
    including the cells a tile finishes for the tile above (its rows 30, 31) and the row pair a tile's last cell row adds below
    itself (rows 28, 29), and the cells on XCD band boundaries.  Oracle: the frame decoded without the filter, then
    oc_state_loop_filter_frag_rows over the same ranges."""
    geom = synth.Geometry(w, h, fmt)
    rng = np.random.default_rng(w * 7 + h + fmt)
    ost = oracle.State(w, h, fmt)
    gst = hip.State(w, h, fmt)
    buf = np.zeros(128, np.int16)
    for f in range(4):
        ftype = hip.INTRA_FRAME if f == 0 else hip.INTER_FRAME
        fr = synth.gen_frame(geom, rng, ftype, "mixed" if f % 2 else "smooth", flimit=[3, 1, 7, 2][f])
        ranges = []
        for pli in range(3):
            nv = geom.nv[pli]
            a = int(rng.integers(0, nv))
            b = int(rng.integers(a, nv + 1))
            ranges.append([(0, nv), (a, b), (0, b), (a, nv)][f] if nv > 1 else (0, nv))
        # oracle: no filter, then the ranges
        fl = fr["flimit"]
        fr0 = dict(fr)
        fr0["flimit"] = 0
        assert util.oracle_apply(ost, fr0) == 0
        for pli in range(3):
            if ranges[pli][1] > ranges[pli][0]:
                ost.loop_filter_rows(fl, oracle.FRAME_PREV, pli, ranges[pli][0], ranges[pli][1])
            ost.set_plane(oracle.FRAME_PREV, pli, ost.get_plane(oracle.FRAME_PREV, pli))   # (the borders again, now of the filtered picture)
        # device: the enqueue slots, one loop-filter call per plane with the range
        gst.frame_begin(fr["frame_type"])
        coded = np.zeros(geom.nfrags, bool)
        coded[fr["coded_fragis"]] = True
        for slot, fi in enumerate(fr["coded_fragis"]):
            fi = int(fi)
            pli = 0 if fi < geom.froffset[1] else (1 if fi < geom.froffset[2] else 2)
            buf[:64] = fr["coeffs"][slot]
            mv = int((int(fr["mvx"][fi]) & 0xFF) | (int(fr["mvy"][fi]) << 8))
            mv = (mv + 0x8000) % 0x10000 - 0x8000
            gst.frag_recon(fi, pli, buf, int(fr["last_zzi"][slot]), int(fr["dc_quant"][slot]), int(fr["refi"][fi]), mv)
        unc = np.nonzero(~coded)[0]
        if unc.size:
            gst.frag_copy_list(unc)
        for pli in range(3):
            if ranges[pli][1] > ranges[pli][0]:
                gst.loop_filter_frag_rows(fl, hip.FRAME_SELF, pli, ranges[pli][0], ranges[pli][1])
        assert gst.frame_flush() == 0
        bad = util.planes_equal(ost, gst)
        assert not bad, (f, ranges, bad)


def test_frame_calls_and_enqueue_calls_by_turns_on_one_state(hip):
    """A state fed through the enqueue slots runs on a context stream, one decoded with thip_decode_frames on a batch
    lane: frames of ONE state that alternate between the two are ordered behind each other with an event (every
    inter frame reads what the call before it, on the other stream, wrote).  Compared every fourth frame only.  (The
    Python-driven slots are far slower than the GPU, so this checks the path, not the race.)"""
    rep = util.run_sequence(hip, 336, 272, PF_420, nframes=16, content="mixed", seed=21, kf_interval=8, enqueue="alternate",
                            check_every=4)
    assert not rep, rep[:3]


def test_start_on_inter_frame_uses_grey_dummy(hip):
    """decode.c:2757-2762 / :2053-2080: no keyframe yet -> references are 0x80."""
    w, h = 96, 64
    geom = synth.Geometry(w, h, PF_420)
    rng = np.random.default_rng(5)
    ost = oracle.State(w, h, PF_420)
    gst = hip.State(w, h, PF_420)
    for f in range(3):
        fr = synth.gen_frame(geom, rng, hip.INTER_FRAME, "mixed")
        util.oracle_apply(ost, fr)
        desc, ka = synth.upload_frame(synth.pack_frame(geom, fr))
        hip.decode_frames([gst], [desc])
        assert not util.planes_equal(ost, gst)


def test_dup_frame_leaves_state_alone(hip):
    w, h = 64, 64
    geom = synth.Geometry(w, h, PF_420)
    rng = np.random.default_rng(6)
    ost = oracle.State(w, h, PF_420)
    gst = hip.State(w, h, PF_420)
    fr = synth.gen_frame(geom, rng, hip.INTRA_FRAME, "mixed")
    util.oracle_apply(ost, fr)
    desc, ka = synth.upload_frame(synth.pack_frame(geom, fr))
    hip.decode_frames([gst], [desc])
    before = [gst.ref_idx(k) for k in range(3)]
    empty = synth.nothing_coded(geom, fr, hip.INTER_FRAME)
    assert util.oracle_apply(ost, empty) == 1
    desc2, ka2 = synth.upload_frame(synth.pack_frame(geom, empty))
    assert hip.decode_frames([gst], [desc2]) == [hip.DUPFRAME]
    assert before == [gst.ref_idx(k) for k in range(3)]
    assert not util.planes_equal(ost, gst)


def test_batched_streams(hip):
    """More streams than THIP_MAX_BATCH, different sizes, lock-step frames."""
    sizes = [(64, 48), (176, 144), (32, 32), (128, 96), (64, 48), (48, 80), (96, 96), (160, 64), (64, 64),
             (32, 112)]
    geoms = [synth.Geometry(w, h, PF_420) for w, h in sizes]
    rngs = [np.random.default_rng(100 + i) for i in range(len(sizes))]
    osts = [oracle.State(w, h, PF_420) for w, h in sizes]
    gsts = [hip.State(w, h, PF_420) for w, h in sizes]
    for f in range(5):
        descs, keep = [], []
        for i in range(len(sizes)):
            fr = synth.gen_frame(geoms[i], rngs[i], hip.INTRA_FRAME if f == 0 else hip.INTER_FRAME, "mixed")
            util.oracle_apply(osts[i], fr)
            d, ka = synth.upload_frame(synth.pack_frame(geoms[i], fr))
            descs.append(d)
            keep.append(ka)
        hip.decode_frames(gsts, descs)
        for i in range(len(sizes)):
            assert not util.planes_equal(osts[i], gsts[i]), (f, i)


def test_batched_streams_of_mixed_formats_and_sizes(hip):
    """One thip_decode_frames call over streams that share nothing: 1080p 4:2:0 next to a single-tile 16x16 4:4:4, a 4:2:2 picture,
    one tile row (2048x16), one tile column (16x1040), 640x480 -- every stream with its own XCD bands (some of them empty), its own
    loop-filter limit (0 included) and content class, over a key frame and five inter frames."""
    cases = [(1920, 1088, PF_420, "smooth"), (16, 16, PF_444, "dense"), (336, 272, PF_422, "mixed"), (2048, 16, PF_420, "mixed"),
             (16, 1040, PF_444, "smooth"), (640, 480, PF_420, "dense")]
    geoms = [synth.Geometry(w, h, f) for w, h, f, _ in cases]
    rngs = [np.random.default_rng(900 + i) for i in range(len(cases))]
    osts = [oracle.State(w, h, f) for w, h, f, _ in cases]
    gsts = [hip.State(w, h, f) for w, h, f, _ in cases]
    for f in range(6):
        descs, keep = [], []
        for i, c in enumerate(cases):
            fr = synth.gen_frame(geoms[i], rngs[i], hip.INTRA_FRAME if f == 0 else hip.INTER_FRAME, c[3], flimit=[2, 0, 63, 4, 15, 7][(i + f) % 6])
            util.oracle_apply(osts[i], fr)
            d, ka = synth.upload_frame(synth.pack_frame(geoms[i], fr))
            descs.append(d)
            keep.append(ka)
        hip.decode_frames(gsts, descs)
        for i in range(len(cases)):
            assert not util.planes_equal(osts[i], gsts[i]), (f, i)


def test_ycbcr_out_is_top_down(hip):
    w, h = 64, 48
    geom = synth.Geometry(w, h, PF_420)
    rng = np.random.default_rng(9)
    ost = oracle.State(w, h, PF_420)
    gst = hip.State(w, h, PF_420)
    fr = synth.gen_frame(geom, rng, hip.INTRA_FRAME, "mixed")
    util.oracle_apply(ost, fr)
    desc, ka = synth.upload_frame(synth.pack_frame(geom, fr))
    hip.decode_frames([gst], [desc])
    outs = gst.ycbcr_out()
    for pli in range(3):
        assert np.array_equal(outs[pli], ost.get_plane(oracle.FRAME_PREV, pli)[::-1])


@pytest.mark.parametrize("eager", [False, True])
@pytest.mark.parametrize("w,h,fmt", [(64, 48, PF_420), (80, 112, PF_422), (48, 32, PF_444)])
def test_ycbcr_map_pinned_images(hip, w, h, fmt, eager):
    """thip_state_ycbcr_map: the library's pinned image equals the copy of thip_state_ycbcr_out, with
    and without eager output; two images alternate, so the view handed out for frame n is still frame
    n after frame n+1 has been decoded, and a DUP frame leaves the current view alone."""
    geom = synth.Geometry(w, h, fmt)
    rng = np.random.default_rng(w + h)
    ost = oracle.State(w, h, fmt)
    gst = hip.State(w, h, fmt)
    gst.set_eager_output(eager)
    prev_view, prev_want = None, None
    for i in range(5):
        fr = synth.gen_frame(geom, rng, hip.INTRA_FRAME if i == 0 else hip.INTER_FRAME, "mixed")
        util.oracle_apply(ost, fr)
        desc, ka = synth.upload_frame(synth.pack_frame(geom, fr))
        hip.decode_frames([gst], [desc])
        want = [ost.get_plane(oracle.FRAME_PREV, pli)[::-1].copy() for pli in range(3)]
        view = gst.ycbcr_map()
        copy = gst.ycbcr_out()
        for pli in range(3):
            assert np.array_equal(view[pli], want[pli])
            assert np.array_equal(copy[pli], want[pli])
            if prev_view is not None:
                assert np.array_equal(prev_view[pli], prev_want[pli])   # the older image is untouched
        prev_view, prev_want = view, want
    # nothing coded: DUP frame, the picture and its image stay (decode.c:2764-2772)
    empty = synth.nothing_coded(geom, fr, hip.INTER_FRAME)
    desc2, ka2 = synth.upload_frame(synth.pack_frame(geom, empty))
    assert hip.decode_frames([gst], [desc2]) == [hip.DUPFRAME]
    view2 = gst.ycbcr_map()
    for pli in range(3):
        assert np.array_equal(view2[pli], prev_want[pli])


def test_parity_check_has_teeth(hip):
    """Negative control: a frame decoded with a different loop-filter limit on the GPU side
    must be reported as a mismatch by the same comparison the other tests rely on."""
    w, h = 176, 144
    geom = synth.Geometry(w, h, PF_420)
    rng = np.random.default_rng(77)
    ost = oracle.State(w, h, PF_420)
    gst = hip.State(w, h, PF_420)
    fr = synth.gen_frame(geom, rng, hip.INTRA_FRAME, "mixed", flimit=15)
    util.oracle_apply(ost, fr)
    wrong = dict(fr)
    wrong["flimit"] = 14
    desc, ka = synth.upload_frame(synth.pack_frame(geom, wrong))
    hip.decode_frames([gst], [desc])
    assert util.planes_equal(ost, gst), "comparison failed to notice a different filter limit"


def test_static_block_elision_forced_on_every_frame(hip):
    """THIP_SKIP_STATIC=2 lifts the "most of the frame is uncoded" condition, so that every inter frame of
    the sequence tests of this file (scattered uncoded blocks, all content classes, the slots, batches, DUP
    frames, the grey start) takes the path that leaves untouched blocks where they are."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, THIP_SKIP_STATIC="2")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_frames.py", "-m", "gpu", "-x", "-q",
                        "-k", "(sequence_small or static_background or enqueue or batched or grey or dup or lane_shared) "
                              "and not elision and not fused"], cwd=root,
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("content", ["dense", "smooth"])
def test_four_concurrent_4k_streams_one_call(hip, content):
    """BASELINE.json config 4's per-GPU share exactly as bench.py times it: four 4K 4:2:0 streams with
    different content decoded in lock step, ONE thip_decode_frames call per step (the library spreads
    them over its two lanes), nine frames with a key frame inside the run; every plane of every stream
    after every step against the oracle."""
    w, h, S, nframes = 3840, 2160, 4, 9
    geom = synth.Geometry(w, h, PF_420)
    rngs = [np.random.default_rng(4000 + 17 * i) for i in range(S)]
    osts = [oracle.State(w, h, PF_420) for _ in range(S)]
    gsts = [hip.State(w, h, PF_420) for _ in range(S)]
    for f in range(nframes):
        descs, keep = [], []
        for i in range(S):
            ftype = hip.INTRA_FRAME if f in (0, 6) else hip.INTER_FRAME
            fr = synth.gen_frame(geom, rngs[i], ftype, content, flimit=2 if i != 3 else 5)
            assert util.oracle_apply(osts[i], fr) == 0
            d, ka = synth.upload_frame(synth.pack_frame(geom, fr))
            descs.append(d)
            keep.append(ka)
        assert hip.decode_frames(gsts, descs) == [0] * S
        for i in range(S):
            assert not util.planes_equal(osts[i], gsts[i]), (f, i)


def test_state_on_a_chosen_device(hip):
    """thip_state_create_on: device -1 = the current device, an explicit device number is honoured and
    reported by thip_state_device, a device the node does not have is TH_EINVAL; frames decode there."""
    import ctypes as C
    from theora_amd import _lib
    L = _lib.load()
    n = L.thip_device_count()
    assert n >= 1
    h = C.c_void_p()
    assert L.thip_state_create_on(C.byref(h), n, 64, 48, PF_420) == _lib.EINVAL
    assert L.thip_state_create_on(C.byref(h), -1, 64, 48, PF_420) == 0
    assert L.thip_state_device(h) >= 0
    L.thip_state_free(h)
    for dev in sorted({0, n - 1}):
        w, hgt = 176, 144
        geom = synth.Geometry(w, hgt, PF_420)
        rng = np.random.default_rng(31 + dev)
        ost = oracle.State(w, hgt, PF_420)
        gst = hip.State(w, hgt, PF_420, device=dev)
        assert gst.device == dev
        for f in range(4):
            fr = synth.gen_frame(geom, rng, hip.INTRA_FRAME if f == 0 else hip.INTER_FRAME, "mixed")
            util.oracle_apply(ost, fr)
            if f == 3:      # the slots stage in pinned host memory of the state's device
                assert util.enqueue_frame(hip, gst, geom, fr) == 0
            else:
                import torch
                with torch.cuda.device(dev):
                    desc, ka = synth.upload_frame(synth.pack_frame(geom, fr))
                hip.decode_frames([gst], [desc])
            assert not util.planes_equal(ost, gst), (dev, f)


def test_streams_spread_over_two_devices_in_one_call(hip):
    """A batch whose states live on different GPUs of the node in ONE thip_decode_frames call (skipped on a
    single-GPU box): stream i on device i mod 2, every plane against the oracle."""
    import torch
    from theora_amd import _lib
    if _lib.load().thip_device_count() < 2:
        pytest.skip("needs two GPUs")
    w, h, S = 256, 160, 4
    geom = synth.Geometry(w, h, PF_420)
    rngs = [np.random.default_rng(900 + i) for i in range(S)]
    osts = [oracle.State(w, h, PF_420) for _ in range(S)]
    gsts = [hip.State(w, h, PF_420, device=i % 2) for i in range(S)]
    for f in range(6):
        descs, keep = [], []
        for i in range(S):
            fr = synth.gen_frame(geom, rngs[i], hip.INTRA_FRAME if f == 0 else hip.INTER_FRAME, "mixed")
            util.oracle_apply(osts[i], fr)
            with torch.cuda.device(i % 2):
                d, ka = synth.upload_frame(synth.pack_frame(geom, fr))
            descs.append(d)
            keep.append(ka)
        hip.decode_frames(gsts, descs)
        for i in range(S):
            assert not util.planes_equal(osts[i], gsts[i]), (f, i)


def _predicted_tokens(geom, fr, fmt):
    """The DC values a bitstream would carry for this frame: the frame's final DCs pushed back through the
    predictor (an encoder's forward prediction, here by inverting the oracle's un-prediction fragment by
    fragment in raster order)."""
    ost = oracle.State(geom.frame_width, geom.frame_height, fmt)
    n = geom.nfrags
    cf = fr["coded_fragis"]
    coded = np.zeros(n, bool)
    coded[cf] = True
    final = np.zeros(n, np.int16)
    final[cf] = np.asarray(fr["coeffs"], np.int16).reshape(-1, 64)[:, 0]
    ost.coded[:] = coded
    ost.refi[:] = fr["refi"]
    # un-predicting zeros gives minus nothing useful in general (the predictor is not linear), so build the
    # tokens greedily: token = final - pred(final neighbours); the oracle's own pass then reproduces `final`
    tokens = np.zeros(n, np.int16)
    for pli in range(3):
        nh, nv, fro = geom.nh[pli], geom.nv[pli], geom.froffset[pli]
        pl = [0, 0, 0]
        for y in range(nv):
            for x in range(nh):
                i = fro + y * nh + x
                if not coded[i]:
                    continue
                r = int(fr["refi"][i])

                def ok(xx, yy):
                    j = fro + yy * nh + xx
                    return 0 <= xx < nh and yy >= 0 and coded[j] and int(fr["refi"][j]) == r
                m = (1 if ok(x - 1, y) else 0) | (2 if ok(x - 1, y - 1) else 0) | (4 if ok(x, y - 1) else 0) | (8 if ok(x + 1, y - 1) else 0)
                d = lambda xx, yy: int(final[fro + yy * nh + xx])   # noqa: E731
                tr = lambda a, b: int(a / b) if a * b >= 0 or a % b == 0 else -((-a) // b)   # noqa: E731  (C division)
                if m == 0:
                    p = pl[r]
                elif m in (1, 3):
                    p = d(x - 1, y)
                elif m == 2:
                    p = d(x - 1, y - 1)
                elif m in (4, 6, 12):
                    p = d(x, y - 1)
                elif m == 5:
                    p = tr(d(x - 1, y) + d(x, y - 1), 2)
                elif m == 8:
                    p = d(x + 1, y - 1)
                elif m in (9, 11, 13):
                    p = tr(75 * d(x - 1, y) + 53 * d(x + 1, y - 1), 128)
                elif m == 10:
                    p = tr(d(x - 1, y - 1) + d(x + 1, y - 1), 2)
                elif m == 14:
                    p = tr(3 * (d(x - 1, y - 1) + d(x + 1, y - 1)) + 10 * d(x, y - 1), 16)
                else:
                    p0, p1, p2 = d(x - 1, y), d(x - 1, y - 1), d(x, y - 1)
                    p = tr(29 * (p0 + p2) - 26 * p1, 32)
                    if abs(p - p2) > 128:
                        p = p2
                    elif abs(p - p0) > 128:
                        p = p0
                    elif abs(p - p1) > 128:
                        p = p1
                tokens[i] = np.int16((int(final[i]) - p + 32768) % 65536 - 32768)
                pl[r] = int(final[i])
    ost.dc[:] = tokens
    ost.dc_unpredict()
    assert np.array_equal(ost.dc[coded], final[coded]), "token construction does not invert the oracle's un-prediction"
    ost.close()
    return tokens


@pytest.mark.parametrize("enqueue", [False, True])
@pytest.mark.parametrize("w,h,fmt", [(176, 144, PF_420), (80, 112, PF_422), (64, 48, PF_444)])
def test_dc_unprediction_on_the_device(hip, w, h, fmt, enqueue):
    """thip_frame_desc.dc_tokens / thip_state_set_device_dc: the frame's DC values arrive as the bitstream
    carries them (before oc_dec_dc_unpredict_mcu_plane, decode.c:1392-1500), with zeros where the command
    stream would carry the final DC; the launch un-predicts on the device and the pictures equal the
    oracle's, which was given the final values."""
    import torch
    geom = synth.Geometry(w, h, fmt)
    rng = np.random.default_rng(w + h + fmt + int(enqueue))
    ost = oracle.State(w, h, fmt)
    gst = hip.State(w, h, fmt)
    if enqueue:
        from theora_amd import _lib
        assert _lib.load().thip_state_set_device_dc(gst.handle, 1) == 0
    keep = []
    for f in range(6):
        fr = synth.gen_frame(geom, rng, hip.INTRA_FRAME if f % 4 == 0 else hip.INTER_FRAME, "mixed")
        util.oracle_apply(ost, fr)
        tokens = _predicted_tokens(geom, fr, fmt)
        blind = dict(fr)
        co = np.asarray(fr["coeffs"], np.int16).reshape(-1, 64).copy()
        co[:, 0] = tokens[fr["coded_fragis"]] if enqueue else 0     # the slots receive the token DC; the descriptor path nothing
        blind["coeffs"] = co
        if enqueue:
            assert util.enqueue_frame(hip, gst, geom, blind) == 0
        else:
            packed = synth.pack_frame(geom, blind)
            desc, ka = synth.upload_frame(packed)
            dct = torch.from_numpy(tokens).cuda()
            desc.dc_tokens = dct.data_ptr()
            keep.append((ka, dct))
            assert hip.decode_frames([gst], [desc]) == [0]
        assert not util.planes_equal(ost, gst), f


def _pp_case(rng, w, h, fmt, smooth):
    ost = oracle.State(w, h, fmt)
    ost.set_ref_idx(0, 0, 0)
    planes = []
    for pli in range(3):
        g = ost.planes[pli]
        if smooth:   # flat blocks with small steps: the de-blocking conditions fire, the variances stay low
            base = rng.integers(60, 200, (g["nvfrags"], g["nhfrags"]))
            img = np.kron(base, np.ones((8, 8), np.int64)) + rng.integers(-2, 3, (g["height"], g["width"]))
        else:        # texture: high variances, de-ringing in all its strengths
            base = rng.integers(40, 220, (g["nvfrags"], g["nhfrags"]))
            img = np.kron(base, np.ones((8, 8), np.int64)) + rng.integers(-40, 41, (g["height"], g["width"])) * (rng.random((g["height"], g["width"])) < 0.5)
        a = np.clip(img, 0, 255).astype(np.uint8)
        ost.set_plane(oracle.FRAME_PREV, pli, a)
        planes.append(a)
    return ost, planes


@pytest.mark.parametrize("w,h,fmt", [(64, 48, PF_420), (48, 80, PF_444), (80, 64, PF_422), (16, 16, PF_420), (176, 144, PF_420),
                                     (336, 272, PF_420), (1280, 720, PF_420), (3840, 2160, PF_420)])
def test_postprocessing(hip, w, h, fmt):
    """TH_DECCTL_SET_PPLEVEL's filters on the device (thip_state_postprocess: k_pp_hedge, k_pp_vedge, k_pp_dering;
    decode.c:1608-1957) against the oracle driven MCU by MCU as th_decode_packetin drives them: every level, flat
    and textured pictures (all three de-ringing strengths, one and three passes), random per-fragment quantiser
    indices, every plane of the post-processed picture; the decoded frame itself must stay untouched."""
    import ctypes as C
    from theora_amd import _lib
    L = _lib.load()
    rng = np.random.default_rng(w + 3 * h + fmt)
    levels = range(0, 8) if w < 1000 else (7,)
    for smooth in (True, False):
        ost, planes = _pp_case(rng, w, h, fmt, smooth)
        gst = hip.State(w, h, fmt)
        for pli in range(3):
            gst.write_plane(0, pli, planes[pli])
        gst.set_ref_idx(0, 0, 0)
        n = ost.nfrags
        dc_qis = rng.integers(0, 64, n).astype(np.uint8)
        frag_qi = rng.integers(0, 64, n).astype(np.uint8)
        dcs = np.sort(rng.integers(1, 90, 64))[::-1].astype(np.int32).copy()
        shm = (-rng.integers(0, 6, 64)).astype(np.int32)
        for level in levels:
            want, _ = ost.postprocess(oracle.FRAME_PREV, level, 1, dc_qis, frag_qi, dcs, shm)
            assert L.thip_state_postprocess(gst.handle, level, dc_qis.ctypes.data, frag_qi.ctypes.data, dcs.ctypes.data,
                                            shm.ctypes.data) == 0
            if level >= 2:
                for pli in range(3):
                    g = gst.planes[pli]
                    got = np.empty((g["height"], g["width"]), np.uint8)
                    assert L.thip_state_read_pp_plane(gst.handle, pli, got.ctypes.data) == 0
                    assert np.array_equal(got, want[pli]), (smooth, level, pli, int((got != want[pli]).sum()))
            out = gst.ycbcr_out()            # what th_decode_ycbcr_out hands out: the post-processed picture, top row first
            for pli in range(3):
                assert np.array_equal(out[pli], want[pli][::-1]), (smooth, level, pli)
                assert np.array_equal(gst.read_plane(0, pli), planes[pli])   # the reference frame is not touched
        if not smooth and w >= 64:
            want7, var = ost.postprocess(oracle.FRAME_PREV, 7, 1, dc_qis, frag_qi, dcs, shm)
            assert (var > 5 * 384).any() and (var > 384).any()   # the strong and the three-pass branches were taken
        ost.close()


def test_a_failed_hand_over_is_decoded_again(hip, capfd):
    """k_recon_lf hands tile edges between concurrently running work groups and bounds every wait; a wait that runs out sets the
    state's pinned fault word (thip_fused.h).  Option debug = 512 makes tile 1 of every stream tag its units with the wrong serial
    number, so its neighbours give up: a frame that came through the enqueue slots (its command stream lives in the state's own
    staging) is then decoded again with the two passes by the next synchronising call and the picture is still bit-exact; a frame
    whose descriptors are the caller's cannot be, and the call says THIP_EFAULT -- once."""
    L = hip._lib.load()
    import ctypes as C
    w, h = 512, 256
    geom = synth.Geometry(w, h, PF_420)
    rng = np.random.default_rng(321)
    ost = oracle.State(w, h, PF_420)
    gst = hip.State(w, h, PF_420)
    n0 = C.c_int()
    L.thip_get_option(b"faults_recovered", C.byref(n0))
    try:
        for f in range(4):
            fr = synth.gen_frame(geom, rng, hip.INTRA_FRAME if f == 0 else hip.INTER_FRAME, "dense", flimit=3)
            util.oracle_apply(ost, fr)
            L.thip_set_option(b"debug", 512 if f in (1, 2) else 0)
            util.enqueue_frame(hip, gst, geom, fr)
            if f == 2:
                outs = gst.ycbcr_out()     # (this path notices it, too, and sends the right picture)
                for pli in range(3):
                    assert np.array_equal(outs[pli], ost.get_plane(oracle.FRAME_PREV, pli)[::-1]), (f, pli)
            assert not util.planes_equal(ost, gst), f
        n1 = C.c_int()
        L.thip_get_option(b"faults_recovered", C.byref(n1))
        assert n1.value - n0.value == 2
        assert "decoding the frame again" in capfd.readouterr().err
        # the caller's descriptors: no second attempt, THIP_EFAULT once, then the state carries on (from a key frame)
        fr = synth.gen_frame(geom, rng, hip.INTER_FRAME, "dense", flimit=3)
        desc, ka = synth.upload_frame(synth.pack_frame(geom, fr))
        L.thip_set_option(b"debug", 512)
        hip.decode_frames([gst], [desc])
        L.thip_set_option(b"debug", 0)
        with pytest.raises(hip.TheoraHipError):
            gst.read_plane(gst.ref_idx(hip.FRAME_PREV), 0)
        gst.read_plane(gst.ref_idx(hip.FRAME_PREV), 0)            # reported once
        ost2 = oracle.State(w, h, PF_420)
        for f in range(2):
            fr = synth.gen_frame(geom, rng, hip.INTRA_FRAME if f == 0 else hip.INTER_FRAME, "mixed", flimit=3)
            util.oracle_apply(ost2, fr)
            desc, ka = synth.upload_frame(synth.pack_frame(geom, fr))
            hip.decode_frames([gst], [desc])
            assert not util.planes_equal(ost2, gst), f
    finally:
        L.thip_set_option(b"debug", 0)


def test_a_frame_decoded_on_top_of_a_failed_one_is_reported_not_repeated(hip, capfd):
    """The kernels record WHICH launch's wait ran out.  A caller that enqueues frame N (whose hand-over fails) and frame N + 1 before
    any synchronising call has decoded N + 1 against a wrong reference: repeating N + 1 -- the only frame the state can still
    repeat -- would not make it right, so the next synchronising call says THIP_EFAULT (once) instead of counting a recovery, and
    the state is good again from its next key frame."""
    L = hip._lib.load()
    import ctypes as C
    w, h = 512, 256
    geom = synth.Geometry(w, h, PF_420)
    rng = np.random.default_rng(654)
    ost = oracle.State(w, h, PF_420)
    gst = hip.State(w, h, PF_420)
    n0 = C.c_int()
    L.thip_get_option(b"faults_recovered", C.byref(n0))
    try:
        for f in range(3):
            fr = synth.gen_frame(geom, rng, hip.INTRA_FRAME if f == 0 else hip.INTER_FRAME, "dense", flimit=3)
            util.oracle_apply(ost, fr)
            L.thip_set_option(b"debug", 512 if f == 1 else 0)
            util.enqueue_frame(hip, gst, geom, fr)      # (no synchronising call between frame 1 and frame 2)
        L.thip_set_option(b"debug", 0)
        with pytest.raises(hip.TheoraHipError):
            gst.read_plane(gst.ref_idx(hip.FRAME_PREV), 0)
        gst.read_plane(gst.ref_idx(hip.FRAME_PREV), 0)            # reported once
        n1 = C.c_int()
        L.thip_get_option(b"faults_recovered", C.byref(n1))
        assert n1.value == n0.value
        assert "could not be decoded again" in capfd.readouterr().err
        ost2 = oracle.State(w, h, PF_420)
        for f in range(2):
            fr = synth.gen_frame(geom, rng, hip.INTRA_FRAME if f == 0 else hip.INTER_FRAME, "mixed", flimit=3)
            util.oracle_apply(ost2, fr)
            util.enqueue_frame(hip, gst, geom, fr)
            assert not util.planes_equal(ost2, gst), f
    finally:
        L.thip_set_option(b"debug", 0)


def test_a_failed_hand_over_on_the_callers_descriptors(hip, capfd):
    """thip_decode_frames on descriptors that are the caller's (the multi-stream server's and bench.py's path), two states in one
    call.  By default the library cannot repeat such a frame: each state's next synchronising call returns THIP_EFAULT once.  With
    option redo_descs = 1 -- the caller's promise that the buffers behind a descriptor stay as they are until the state's next
    synchronising call -- the frame is decoded again with the two passes and both pictures are the oracle's."""
    L = hip._lib.load()
    import ctypes as C
    w, h = 512, 256
    geom = synth.Geometry(w, h, PF_420)
    rng = np.random.default_rng(987)
    osts = [oracle.State(w, h, PF_420) for _ in range(2)]
    gsts = [hip.State(w, h, PF_420) for _ in range(2)]
    keep = []
    n0 = C.c_int()
    L.thip_get_option(b"faults_recovered", C.byref(n0))
    try:
        L.thip_set_option(b"redo_descs", 1)
        for f in range(3):
            descs = []
            for i in range(2):
                fr = synth.gen_frame(geom, rng, hip.INTRA_FRAME if f == 0 else hip.INTER_FRAME, "dense", flimit=3)
                util.oracle_apply(osts[i], fr)
                desc, ka = synth.upload_frame(synth.pack_frame(geom, fr, form="dequant16" if f & 1 else None))
                keep.append(ka)
                descs.append(desc)
            L.thip_set_option(b"debug", 512 if f >= 1 else 0)
            hip.decode_frames(gsts, descs)
            L.thip_set_option(b"debug", 0)
            for i in range(2):
                assert not util.planes_equal(osts[i], gsts[i]), (f, i)
        n1 = C.c_int()
        L.thip_get_option(b"faults_recovered", C.byref(n1))
        assert n1.value - n0.value == 4
        assert "decoding the frame again" in capfd.readouterr().err
        # without the promise: THIP_EFAULT once per state, pictures wrong until the next key frame
        L.thip_set_option(b"redo_descs", 0)
        descs = []
        for i in range(2):
            fr = synth.gen_frame(geom, rng, hip.INTER_FRAME, "dense", flimit=3)
            desc, ka = synth.upload_frame(synth.pack_frame(geom, fr))
            keep.append(ka)
            descs.append(desc)
        L.thip_set_option(b"debug", 512)
        hip.decode_frames(gsts, descs)
        L.thip_set_option(b"debug", 0)
        assert L.thip_synchronize() == -1   # THIP_EFAULT: reports, repairs nothing
        for i in range(2):
            with pytest.raises(hip.TheoraHipError):
                gsts[i].read_plane(gsts[i].ref_idx(hip.FRAME_PREV), 0)
            gsts[i].read_plane(gsts[i].ref_idx(hip.FRAME_PREV), 0)
        assert L.thip_synchronize() == 0
    finally:
        L.thip_set_option(b"debug", 0)
        L.thip_set_option(b"redo_descs", 0)


@pytest.mark.parametrize("content,skip_static,ahead_is_key", [("mixed", 1, False), ("static_bg", 2, False), ("smooth", 2, True)])
def test_a_frame_decoded_ahead_is_taken_back_at_the_c_abi(hip, content, skip_static, ahead_is_key):
    """thip_state_ring_mark / thip_state_ring_rewind (round 6; what th_decode_packetin does when, with option fe_pipeline, another
    packet comes than the one th_decode_ycbcr_out decoded ahead): a frame -- an inter frame, or a key frame, which also moves the
    golden reference (decode.c:2947-2955) -- decoded after the mark never happened once the ring is put back: the references are
    the marked ones, the marked frame is the picture again, the next frames decode bit-exactly against an oracle that never saw
    the discarded one, and no static-block shortcut is taken on what the discarded frame left in its buffer (skip_static = 2
    applies the shortcut wherever it is allowed)."""
    import ctypes as C
    from theora_amd import _lib
    L = _lib.load()
    w, h = 336, 176
    geom = synth.Geometry(w, h)
    rng = np.random.default_rng(4711)
    ost, gst = oracle.State(w, h), hip.State(w, h)
    keep = []

    def both(fr, oracle_too=True):
        if oracle_too:
            assert util.oracle_apply(ost, fr) == 0
        d, ka = synth.upload_frame(synth.pack_frame(geom, fr))
        keep.append(ka)
        assert hip.decode_frames([gst], [d])[0] == 0
    with util.options(L, skip_static=skip_static):
        both(synth.gen_frame(geom, rng, hip.INTRA_FRAME, content, flimit=3))
        for _ in range(2):
            both(synth.gen_frame(geom, rng, hip.INTER_FRAME, content, flimit=3))
        assert not util.planes_equal(ost, gst)
        ring = [gst.ref_idx(k) for k in range(3)]
        mark = (C.c_int64 * 8)()
        assert L.thip_state_ring_mark(gst.handle, mark) == 0
        assert L.thip_state_ring_rewind(gst.handle, mark) == 0                      # nothing decoded since: nothing happens
        both(synth.gen_frame(geom, rng, hip.INTRA_FRAME if ahead_is_key else hip.INTER_FRAME, content, flimit=3), oracle_too=False)
        assert [gst.ref_idx(k) for k in range(3)] != ring
        assert L.thip_state_ring_rewind(gst.handle, mark) == 0
        assert [gst.ref_idx(k) for k in range(3)] == ring
        assert not util.planes_equal(ost, gst)                                    # the marked frame is the newest again
        got = gst.ycbcr_out()
        for pli in range(3):
            assert np.array_equal(got[pli], ost.get_plane(oracle.FRAME_PREV, pli)[::-1]), pli
        for _ in range(4):
            both(synth.gen_frame(geom, rng, hip.INTER_FRAME, content, flimit=3))
            assert not util.planes_equal(ost, gst)
            assert ost.ref_frame_idx == [gst.ref_idx(k) for k in range(3)]
        bad = (C.c_int64 * 8)()
        assert L.thip_state_ring_rewind(gst.handle, bad) == _lib.EINVAL              # not a mark
        assert L.thip_state_ring_mark(None, mark) == _lib.EFAULT
    gst.close()
    ost.close()


def test_check_fault_repairs_for_a_caller_that_only_synchronises(hip, capfd):
    """thip_state_check_fault: a caller that keeps its frames on the device and brackets its work with thip_synchronize
    only never reaches the calls that repair a failed hand-over -- thip_synchronize REPORTS (THIP_EFAULT while a state's word is
    set) and leaves the state to its owner.  The owner's thip_state_check_fault waits for the state's stream and does the repair: 1
    = the newest frame was decoded again (bit-exact), after which thip_synchronize is quiet; 0 when nothing was wrong."""
    L = hip._lib.load()
    w, h = 512, 256
    geom = synth.Geometry(w, h, PF_420)
    rng = np.random.default_rng(99)
    ost, gst = oracle.State(w, h, PF_420), hip.State(w, h, PF_420)
    try:
        for f in range(3):
            fr = synth.gen_frame(geom, rng, hip.INTRA_FRAME if f == 0 else hip.INTER_FRAME, "dense", flimit=3)
            util.oracle_apply(ost, fr)
            L.thip_set_option(b"debug", 512 if f == 1 else 0)
            util.enqueue_frame(hip, gst, geom, fr)
            if f == 1:
                L.thip_set_option(b"debug", 0)
                with pytest.raises(hip.TheoraHipError):
                    hip.synchronize()                 # reports, repairs nothing
                assert gst.check_fault() == 1         # the owner's call: decoded again with the two passes
                hip.synchronize()
                assert "decoding the frame again" in capfd.readouterr().err
            else:
                assert gst.check_fault() == 0
            assert not util.planes_equal(ost, gst), f
    finally:
        L.thip_set_option(b"debug", 0)
    gst.close()
    ost.close()
