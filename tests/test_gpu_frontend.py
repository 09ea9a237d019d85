"""End to end through the th_decode_* API: Theora packets -> the library's own front end
(bit reader, Huffman, modes, vectors, tokens, DC un-prediction, dequantisation) -> HIP
reconstruction -> th_decode_ycbcr_out, against the oracle fed with the generator's ground
truth.  The packets come from tests/streamgen.py (the reference encoder cannot be built
here and no sample streams exist)."""
import numpy as np
import pytest

import oracle
from tests import streamgen, util

pytestmark = pytest.mark.gpu


def run_stream(hip, w, h, fmt, seed, nframes, kf=5, trees="random", device_dc=False, device_tokens=False, device_lists=False,
               lookahead=0):
    import ctypes as C
    from theora_amd.decoder import Decoder
    st = streamgen.Stream(w, h, fmt, seed, trees=trees)
    dec = Decoder(st.header_packets())
    if device_dc:   # TH_DECCTL_THIP_SET_DEVICE_DC: the DC prediction is undone on the GPU, not in th_decode_packetin
        on = C.c_int(1)
        assert dec._L.th_decode_ctl(dec._dec, 0x7102, C.byref(on), C.sizeof(on)) == 0
    if device_tokens:   # TH_DECCTL_THIP_SET_DEVICE_TOKENS: token expansion + AC dequantisation on the GPU
        on = C.c_int(1)
        assert dec._L.th_decode_ctl(dec._dec, 0x7103, C.byref(on), C.sizeof(on)) == 0
    if device_lists is not None:   # TH_DECCTL_THIP_SET_DEVICE_LISTS: everything behind the entropy decoder on the GPU, or (0) the
        on = C.c_int(int(bool(device_lists)))   # host's own token walk; None leaves the choice to the library (few contexts: the lists)
        assert dec._L.th_decode_ctl(dec._dec, 0x7104, C.byref(on), C.sizeof(on)) == 0
    assert dec.info.frame_width == w and dec.info.frame_height == h and dec.info.pixel_fmt == fmt
    assert dec.comment.vendor == b"theora-hip streamgen"
    ost = oracle.State(w, h, fmt)
    nnew = 0
    ahead = []   # lookahead > 0: the packets are made up front and announced that far ahead (TH_DECCTL_THIP_PREFETCH_PACKET)
    if lookahead:
        ahead = [st.frame(0 if f % kf == 0 else 1, density=[0.9, 0.5, 0.15][f % 3]) for f in range(nframes)]
    announced = taken = 0
    for f in range(nframes):
        ftype = 0 if f % kf == 0 else 1
        if lookahead:
            pkt, truth = ahead[f]
            while announced < nframes and announced < f + lookahead:
                announced = max(announced, f)
                if dec.prefetch(ahead[announced][0]):
                    taken += 1
                elif len(ahead[announced][0]):
                    break
                announced += 1
        else:
            pkt, truth = st.frame(ftype, density=[0.9, 0.5, 0.15][f % 3])
        rc, gp = dec.packetin(pkt)
        if truth["dup"]:
            assert rc == 1, f
        else:
            assert rc == 0, f
            nnew += 1
            args = st.oracle_inputs(truth, ost)
            assert ost.decode_frame(**args) == 0
        got = dec.ycbcr_out()
        for pli in range(3):
            want = ost.get_plane(oracle.FRAME_PREV, pli)[::-1]
            assert np.array_equal(got[pli], want), (f, pli, int((got[pli] != want).sum()))
    dec.close()
    if lookahead:
        assert taken >= nnew - 1, (taken, nnew)   # (the announcements were taken: the test ran what it says)
    return nnew


@pytest.mark.parametrize("lists", [True, False, None])
@pytest.mark.parametrize("w,h,fmt,ahead", [(64, 48, 0, 1), (176, 144, 0, 3), (48, 64, 3, 2), (80, 48, 2, 4), (336, 32, 0, 6),
                                           (1280, 720, 0, 3), (1920, 1088, 0, 2)])
def test_packets_decode_bit_exact_with_a_look_ahead(hip, w, h, fmt, ahead, lists):
    """TH_DECCTL_THIP_PREFETCH_PACKET: the packets announced `ahead` packets before their th_decode_packetin -- entropy decoder,
    DC chain and the packing of the token lists on parser threads, th_decode_packetin adopts the frame and hands it to the device
    (token lists in one piece, or the host's own token walk, or as the library chooses) -- give the oracle's pictures, frame by
    frame, dropped frames and more announcements than slots included."""
    assert run_stream(hip, w, h, fmt, seed=w + 3 * h + fmt, nframes=12 if w < 1000 else 5, device_lists=lists, lookahead=ahead,
                      trees="matched" if w >= 1000 else "random") >= 4


@pytest.mark.parametrize("w,h,fmt", [(64, 48, 0), (176, 144, 0)])
def test_look_ahead_rule_changes_sides_against_the_oracle(hip, w, h, fmt):
    """Option fe_assign = 2 (the default) measures per stream who pairs tokens and fragments -- the announced packets' parsers
    (k_tok_scatter on the device) or the device's walk (k_tok_assign) -- and changes sides while the stream runs: after
    4 * 8 + 8 adopted frames 24 frames are timed each way (the frames announced under the other rule let through first), the
    faster rule is kept for fe_assign_settle frames (shortened here: 1024 by default) and the measurement begins again.  Long
    enough to cross every one of those points, eight packets announced ahead, EVERY frame compared with the oracle: the frames
    around a change of sides were parsed under one rule and handed over under the other's successor.  The counters say that both
    changes happened."""
    import ctypes as C
    L = hip._lib.load()
    to_dev0, to_par0, to_dev1, to_par1 = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    ad0, ad1 = C.c_int(), C.c_int()
    L.thip_get_option(b"fe_assign_to_device", C.byref(to_dev0))
    L.thip_get_option(b"fe_assign_to_parsers", C.byref(to_par0))
    L.thip_get_option(b"fe_lookahead_adopted", C.byref(ad0))
    with util.options(L, fe_assign=2, fe_assign_settle=6, fe_lookahead=8):
        assert run_stream(hip, w, h, fmt, seed=7 * w + h + fmt, nframes=170, kf=17, device_lists=True, lookahead=8, trees="matched") >= 120
    L.thip_get_option(b"fe_assign_to_device", C.byref(to_dev1))
    L.thip_get_option(b"fe_assign_to_parsers", C.byref(to_par1))
    L.thip_get_option(b"fe_lookahead_adopted", C.byref(ad1))
    assert ad1.value - ad0.value >= 120, (ad0.value, ad1.value)     # (dropped frames leave the announcements in place)
    assert to_dev1.value > to_dev0.value and to_par1.value > to_par0.value, (to_dev0.value, to_dev1.value, to_par0.value, to_par1.value)


@pytest.mark.parametrize("lists", [True, False])
@pytest.mark.parametrize("w,h,fmt,ahead,n", [(64, 48, 0, 4, 60), (176, 144, 0, 8, 40), (16, 16, 0, 3, 60), (48, 64, 3, 2, 30), (1280, 720, 0, 4, 8)])
def test_next_frame_on_the_device_before_this_picture_is_waited_for(hip, w, h, fmt, ahead, n, lists):
    """Option fe_pipeline: th_decode_ycbcr_out(N) hands the next announced packet's frame to the device before it waits for picture
    N (thip_state_ycbcr_map_begin / _end: the two pictures live in the state's two host images) and th_decode_packetin(N + 1) finds
    the work done.  Every picture against the oracle -- the one handed out while the next frame is already being decoded included --,
    return codes and dropped frames (zero-byte packets between announced ones: the 16 x 16 stream has them) as ever; the counter
    says that frames did go ahead."""
    import ctypes as C
    L = hip._lib.load()
    c0, c1 = C.c_int(), C.c_int()
    L.thip_get_option(b"fe_pipelined", C.byref(c0))
    with util.options(L, fe_pipeline=1):
        assert run_stream(hip, w, h, fmt, seed=5 * w + h + fmt, nframes=n, kf=7, device_lists=lists, lookahead=ahead,
                          trees="matched" if w >= 1000 else "random") >= n // 3
    L.thip_get_option(b"fe_pipelined", C.byref(c1))
    assert c1.value - c0.value >= n // 3, (c0.value, c1.value)


@pytest.mark.parametrize("w,h,fmt,dup_between", [(64, 48, 0, False), (176, 144, 0, True), (80, 48, 3, False)])
def test_a_frame_decoded_ahead_is_taken_back_when_another_packet_comes(hip, w, h, fmt, dup_between):
    """fe_pipeline (on by default since round 6): th_decode_ycbcr_out hands the announced packet's frame to the device before it
    waits for its own picture -- and an announcement stays a hint (theoradec.h:279-302 promises nothing about what comes next): a
    th_decode_packetin that brings ANOTHER packet takes the frame decoded ahead back (reference ring, counters) and decodes its own,
    bit-exact with an oracle that never saw the announced packet; a zero-byte packet in between changes nothing about that.  Every
    frame against the oracle, granule positions against a decoder that was never told about the skipped packet."""
    import ctypes as C
    import time
    from theora_amd.decoder import Decoder
    L = hip._lib.load()

    def counter(name):
        v = C.c_int()
        assert L.thip_get_option(name, C.byref(v)) == 0
        return v.value
    st = streamgen.Stream(w, h, fmt, seed=77 + w, trees="random")
    hdr = st.header_packets()
    made = [st.frame(0 if f % 6 == 0 else 1, density=0.9, p_empty=0.0, nqis=(2 if f % 3 == 1 else None)) for f in range(14)]
    skipped = (1, 8)                           # announced, decoded ahead, and then NOT handed in
    with util.options(L, fe_pipeline=1):
        dec, ref = Decoder(hdr), Decoder(hdr)  # ref: the same packets without any announcement
        ost = oracle.State(w, h, fmt)
        back0, ahead0 = counter(b"fe_pipeline_taken_back"), counter(b"fe_pipelined")

        def step(i):
            pkt, truth = made[i]
            rc, gp = dec.packetin(pkt)
            rc2, gp2 = ref.packetin(pkt)
            assert (rc, gp) == (rc2, gp2), (i, rc, gp, rc2, gp2)
            if not truth["dup"]:
                assert ost.decode_frame(**st.oracle_inputs(truth, ost)) == 0
            got = dec.ycbcr_out()
            ref.ycbcr_out()
            for pli in range(3):
                assert np.array_equal(got[pli], ost.get_plane(oracle.FRAME_PREV, pli)[::-1]), (i, pli)
        i = 0
        while i < len(made):
            step(i)
            if i + 1 in skipped:
                assert dec.prefetch(made[i + 1][0])
                if i + 2 < len(made):
                    dec.prefetch(made[i + 2][0])     # (a second announcement behind it: dropped with the first)
                time.sleep(0.05)                 # (the parser is done: the next th_decode_ycbcr_out takes the frame ahead)
                got = dec.ycbcr_out()            # the picture of frame i, once more -- and frame i + 1 goes to the device
                for pli in range(3):
                    assert np.array_equal(got[pli], ost.get_plane(oracle.FRAME_PREV, pli)[::-1]), (i, pli)
                if dup_between:                  # a dropped frame (decode.c:2746) between the two: the picture stays, the counters move
                    rc, gp = dec.packetin(b"")
                    rc2, gp2 = ref.packetin(b"")
                    assert (rc, gp) == (rc2, gp2), (i, rc, gp, rc2, gp2)
                i += 2                           # ... and the packet after it comes instead
            else:
                i += 1
        assert counter(b"fe_pipelined") - ahead0 >= len(skipped)
        assert counter(b"fe_pipeline_taken_back") - back0 == len(skipped)
    dec.close()
    ref.close()
    ost.close()


@pytest.mark.parametrize("w,h,fmt", [(64, 48, 0), (176, 144, 2)])
def test_lists_or_host_walk_is_measured_per_context_against_the_oracle(hip, w, h, fmt):
    """The plain th_decode_* loop with the library left to choose (TH_DECCTL_THIP_SET_DEVICE_LISTS not called, option fe_device_lists
    = -1): since round 5 a context MEASURES whether its token lists go to the device or the host walks them -- 16 inter frames
    each way, the faster kept for fe_assign_settle frames (shortened here), and again -- so a stream changes sides while it runs.
    Every frame against the oracle; the counters say that it went both ways."""
    import ctypes as C
    L = hip._lib.load()

    def counter(name):
        v = C.c_int()
        assert L.thip_get_option(name, C.byref(v)) == 0
        return v.value
    before = (counter(b"fe_lists_to_device"), counter(b"fe_lists_to_host"))
    with util.options(L, fe_device_lists=-1, fe_lists_rule=1, fe_assign_settle=5):
        assert run_stream(hip, w, h, fmt, seed=11 * w + h + fmt, nframes=150, kf=30, device_lists=None, trees="matched") >= 110
    after = (counter(b"fe_lists_to_device"), counter(b"fe_lists_to_host"))
    assert after[0] > before[0] and after[1] > before[1], (before, after)


@pytest.mark.parametrize("levels", [1, 0])
@pytest.mark.parametrize("w,h,fmt,ahead", [(176, 144, 0, 3), (48, 64, 3, 2), (336, 32, 0, 4), (1280, 720, 0, 3)])
def test_look_ahead_with_the_walk_left_to_the_device(hip, w, h, fmt, ahead, levels):
    """Option fe_assign = 0: an announced packet's parser packs the lists and leaves the walk (which token belongs to which
    fragment) to the device (thip_state_token_lists_begin: k_tok_assign / k_tok_walk) instead of doing it itself
    (thip_state_token_lists_begin_assigned: k_tok_scatter, the default, what the other look-ahead tests run); and both forms of
    the coefficient slots (tl_levels) behind the host's walk."""
    L = hip._lib.load()
    with util.options(L, fe_assign=0 if levels else 1, tl_levels=levels):
        assert run_stream(hip, w, h, fmt, seed=w + 5 * h + fmt, nframes=10 if w < 1000 else 5, device_lists=True, lookahead=ahead,
                          trees="matched" if w >= 1000 else "random") >= 4


@pytest.mark.parametrize("w,h,fmt", [(64, 48, 0), (176, 144, 0), (48, 64, 3), (80, 48, 2), (16, 16, 0), (336, 32, 0)])
def test_packets_decode_bit_exact(hip, w, h, fmt):
    assert run_stream(hip, w, h, fmt, seed=w + 3 * h + fmt, nframes=9) >= 6


@pytest.mark.parametrize("w,h,fmt", [(64, 48, 0), (176, 144, 0), (48, 64, 3), (80, 48, 2), (16, 16, 0), (1280, 720, 0)])
def test_packets_decode_bit_exact_with_dc_unprediction_on_the_gpu(hip, w, h, fmt):
    """The same streams with spec 7.8 / decode.c:1392-1500 left to the backend (k_dc_unpredict): the host
    front end hands the slots the DC values as the tokens carry them."""
    assert run_stream(hip, w, h, fmt, seed=w + 3 * h + fmt, nframes=9 if w < 1000 else 4, device_dc=True) >= 3


@pytest.mark.parametrize("w,h,fmt", [(64, 48, 0), (176, 144, 0), (48, 64, 3), (80, 48, 2), (336, 32, 0), (1280, 720, 0)])
def test_packets_decode_bit_exact_with_levels_through_the_slot(hip, w, h, fmt):
    """Option fe_levels: the host's own token walk hands thip_state_frag_recon_levels the quantised levels and the frame's
    tables; the multiplication of decode.c:1573 is the reconstruction kernel's (streams with one to three qi per frame,
    custom matrices, intra and inter tables)."""
    L = hip._lib.load()
    L.thip_set_option(b"fe_levels", 1)
    try:
        assert run_stream(hip, w, h, fmt, seed=w + 3 * h + fmt, nframes=9 if w < 1000 else 4, device_lists=False) >= 3
    finally:
        L.thip_set_option(b"fe_levels", 0)


@pytest.mark.parametrize("device_dc", [False, True])
@pytest.mark.parametrize("w,h,fmt", [(64, 48, 0), (176, 144, 0), (48, 64, 3), (80, 48, 2), (16, 16, 0), (1280, 720, 0)])
def test_packets_decode_bit_exact_with_token_expansion_on_the_gpu(hip, w, h, fmt, device_dc):
    """The same streams with decode.c:1540-1581 (tokens -> 64 dequantised coefficients in natural order) left to
    the backend (k_expand_tokens): the host only delimits each fragment's tokens; also together with the DC
    un-prediction on the GPU."""
    assert run_stream(hip, w, h, fmt, seed=w + 3 * h + fmt, nframes=9 if w < 1000 else 4, device_dc=device_dc,
                      device_tokens=True) >= 3


@pytest.mark.parametrize("device_dc", [False, True])
@pytest.mark.parametrize("w,h,fmt", [(64, 48, 0), (176, 144, 0), (48, 64, 3), (80, 48, 2), (16, 16, 0), (336, 32, 0),
                                     (1280, 720, 0), (1920, 1088, 0)])
def test_packets_decode_bit_exact_with_the_token_lists_on_the_gpu(hip, w, h, fmt, device_dc):
    """The same streams with everything behind the entropy decoder left to the backend
    (thip_state_decode_token_lists): the token lists per (plane, zig-zag index) go to the GPU as they are; which
    token belongs to which fragment (EOB runs crossing lists and planes included), expansion, dequantisation,
    command words and coefficient slots are the device's (k_tok_assign, k_tok_slots, k_tok_write); the DC
    prediction undone by the front end (thip_token_lists.dc) or, device_dc, on the GPU too (k_dc_wave)."""
    assert run_stream(hip, w, h, fmt, seed=w + 3 * h + fmt, nframes=9 if w < 1000 else 4, device_lists=True,
                      device_dc=device_dc) >= 3


@pytest.mark.parametrize("groups,worker,levels,algo", [(1, 0, 1, 2), (1, 1, 0, 1), (4, 0, 0, 2), (9, 1, 1, 1), (2, 1, 1, 2), (4, 1, 1, 1), (7, 0, 1, 1)])
@pytest.mark.parametrize("w,h,fmt", [(64, 48, 0), (176, 144, 0), (48, 64, 3), (336, 32, 0), (1280, 720, 0), (1920, 1088, 0)])
def test_packets_decode_bit_exact_with_the_token_lists_in_groups(hip, w, h, fmt, groups, worker, levels, algo):
    """The token-list path's options.  tl_levels: the device writes the coefficient slots as dequantised int16 (0, round 3's
    form) or in the levels form (1, the default: int8 units, tiles with a level beyond eight bits as int16 over two units --
    the generator's streams have both kinds of tile --, k_recon_lf<LEVELS> dequantises).  fe_groups: the lists go to the device in one piece after the packet's last bit
    (1: thip_state_token_lists_begin) or in groups of zig-zag indices while the caller still decodes
    (thip_state_token_lists_staging / _open / _append, k_tok_assign launched once per group, the fragments' positions kept
    between the launches; the default of 6 -- {3, 10, 28, 48, 64} -- is what the other tests run).  fe_worker: the DC prediction undone by the caller
    behind the tokens (0) or by the context's second thread beside them."""
    L = hip._lib.load()
    L.thip_set_option(b"fe_groups", groups)
    L.thip_set_option(b"fe_worker", worker)
    L.thip_set_option(b"tl_levels", levels)
    L.thip_set_option(b"tl_algo", algo)   # 1: k_tok_assign; 2: k_tok_rank + k_tok_walk (the default picks by plane size)
    try:
        assert run_stream(hip, w, h, fmt, seed=w + 3 * h + fmt, nframes=9 if w < 1000 else 4, device_lists=True) >= 3
    finally:
        L.thip_set_option(b"fe_groups", 6)
        L.thip_set_option(b"fe_worker", 2)
        L.thip_set_option(b"tl_levels", 1)
        L.thip_set_option(b"tl_algo", 0)


def test_token_list_groups_come_in_order(hip):
    """thip_state_token_lists_append: the first group starts at index 0, each at the end of the one before, _finish wants
    them all; _abort gives an opened frame up."""
    import ctypes as C

    class TL(C.Structure):
        _fields_ = [("frame_type", C.c_int32), ("flimit", C.c_int32), ("tokens", C.c_void_p), ("ntokens", C.c_int64),
                    ("list_off", C.c_uint32 * 64 * 3), ("list_len", C.c_uint32 * 64 * 3), ("eob_carry", C.c_uint32 * 64 * 3),
                    ("arrivals", C.c_uint32 * 64 * 3), ("coded", C.c_void_p), ("frag_meta", C.c_void_p),
                    ("ncoded", C.c_int32 * 3), ("dequant", C.c_void_p), ("dc_quant", C.c_uint16 * 2 * 3), ("dc", C.c_void_p)]

    L = hip._lib.load()
    st = hip.State(64, 48, 0)
    tl = TL()
    tl.frame_type = 1   # an inter frame with nothing coded
    z = (C.c_uint32 * 64 * 3)()
    EINVAL = hip._lib.EINVAL
    assert L.thip_state_token_lists_abort(st.handle) == EINVAL                       # nothing opened
    assert L.thip_state_token_lists_append(st.handle, 0, 64, None, 0, z, z, z, z) == EINVAL
    assert L.thip_state_token_lists_open(st.handle, C.byref(tl)) == 0
    assert L.thip_state_token_lists_open(st.handle, C.byref(tl)) == EINVAL           # one frame at a time
    assert L.thip_state_token_lists_append(st.handle, 1, 2, None, 0, z, z, z, z) == EINVAL   # starts at 0
    assert L.thip_state_token_lists_append(st.handle, 0, 0, None, 0, z, z, z, z) == EINVAL   # not empty
    assert L.thip_state_token_lists_append(st.handle, 0, 5, None, 0, z, z, z, z) == 0
    assert L.thip_state_token_lists_append(st.handle, 6, 64, None, 0, z, z, z, z) == EINVAL  # a gap
    assert L.thip_state_token_lists_finish(st.handle, None) == EINVAL                # indices 5..63 still to come
    assert L.thip_state_token_lists_append(st.handle, 5, 65, None, 0, z, z, z, z) == EINVAL
    assert L.thip_state_token_lists_append(st.handle, 5, 64, None, 0, z, z, z, z) == 0
    assert L.thip_state_token_lists_finish(st.handle, None) == hip._lib.DUPFRAME
    assert L.thip_state_token_lists_open(st.handle, C.byref(tl)) == 0
    assert L.thip_state_token_lists_abort(st.handle) == 0
    assert L.thip_state_token_lists_open(st.handle, C.byref(tl)) == 0
    assert L.thip_state_token_lists_append(st.handle, 0, 64, None, 0, z, z, z, z) == 0
    assert L.thip_state_token_lists_finish(st.handle, None) == hip._lib.DUPFRAME
    st.close()


@pytest.mark.parametrize("lists", [False, None])
def test_packets_decode_bit_exact_720p(hip, lists):
    """BASELINE.json's 720p size through the whole API (key frame + inter frames of three densities),
    with Huffman trees matched to the content: the short-code fast path of the token loop at scale; on the host path and as the
    library chooses by itself (one context alive: the token lists on the device)."""
    assert run_stream(hip, 1280, 720, 0, seed=720, nframes=4, kf=3, trees="matched", device_lists=lists) >= 3


@pytest.mark.parametrize("mode", ["host", "device_dc", "device_lists", "device_lists_dc", "device_lists_lookahead"])
def test_packets_decode_bit_exact_4k(hip, mode):
    """BASELINE.json's 4K size (3840x2160 4:2:0, 194 400 fragments) through th_decode_*: a key frame and two inter frames with
    matched Huffman trees, by the host front end, with the DC un-prediction on the GPU (k_dc_wave: 480 x 270 luma fragments, the
    64 rows in flight in LDS), and with TH_DECCTL_THIP_SET_DEVICE_LISTS: the token lists themselves on the GPU, the key frame's luma
    plane (129 600 coded fragments) with k_tok_assign's rank -> fragment map in memory instead of LDS."""
    assert run_stream(hip, 3840, 2160, 0, seed=2160, nframes=3, kf=3, trees="matched", device_dc=mode in ("device_dc", "device_lists_dc"),
                      device_lists=mode in ("device_lists", "device_lists_dc", "device_lists_lookahead"),
                      lookahead=2 if mode == "device_lists_lookahead" else 0) == 3


def test_empty_packet_is_dup_frame(hip):
    from theora_amd.decoder import Decoder
    st = streamgen.Stream(64, 48, 0, seed=5)
    dec = Decoder(st.header_packets())
    pkt, truth = st.frame(0)
    assert dec.packetin(pkt)[0] == 0
    before = [p.copy() for p in dec.ycbcr_out()]
    assert dec.packetin(b"")[0] == 1
    after = dec.ycbcr_out()
    assert all(np.array_equal(a, b) for a, b in zip(before, after))


def test_header_errors(hip):
    import ctypes as C
    from theora_amd import _lib
    from theora_amd.decoder import Decoder, _packet
    L = _lib.load()
    st = streamgen.Stream(64, 48, 0, seed=6)
    hp = st.header_packets()
    info, tc, setup = _lib.ThInfo(), _lib.ThComment(), C.c_void_p()
    L.th_info_init(C.byref(info))
    L.th_comment_init(C.byref(tc))
    op, k1 = _packet(hp[1], bos=0)
    assert L.th_decode_headerin(C.byref(info), C.byref(tc), C.byref(setup), C.byref(op)) == -20   # TH_EBADHEADER: comment before info (decinfo.c:226)
    bad = bytearray(hp[0])
    bad[3] ^= 0xFF
    op, k2 = _packet(bytes(bad), bos=1)
    assert L.th_decode_headerin(C.byref(info), C.byref(tc), C.byref(setup), C.byref(op)) == -21   # TH_ENOTFORMAT: not "theora" (decinfo.c:213)
    with pytest.raises(Exception):
        Decoder([hp[0], hp[2]])


def test_corrupted_packet_changes_the_picture(hip):
    """Negative control for the comparison above: flip one bit in the token area of a
    keyframe and the decoded picture must differ from the clean decode."""
    from theora_amd.decoder import Decoder
    st = streamgen.Stream(64, 48, 0, seed=8)
    hp = st.header_packets()
    pkt, truth = st.frame(0)
    a = Decoder(hp)
    a.packetin(pkt)
    clean = a.ycbcr_out()
    bad = bytearray(pkt)
    bad[len(bad) // 2] ^= 0x10
    b = Decoder(hp)
    b.packetin(bytes(bad))
    dirty = b.ycbcr_out()
    assert any(not np.array_equal(x, y) for x, y in zip(clean, dirty))


def test_contexts_on_concurrent_host_threads(hip):
    """Decoder contexts are independent (no mutable globals on the path, SURVEY 8b
    "Threading"): four streams decoded by four host threads at once give the pictures the same
    packets give one after the other."""
    import threading
    from theora_amd.decoder import Decoder
    geoms = [(176, 144, 0), (64, 48, 3), (336, 32, 0), (80, 48, 2)]
    streams = []
    for i, (w, h, fmt) in enumerate(geoms):
        st = streamgen.Stream(w, h, fmt, seed=40 + i)
        hdr = st.header_packets()
        pkts = [st.frame(0 if f % 4 == 0 else 1, density=0.6)[0] for f in range(8)]
        streams.append((hdr, pkts))

    def decode(hdr, pkts, out, reps):
        for _ in range(reps):
            dec = Decoder(hdr)
            frames = []
            for p in pkts:
                dec.packetin(p)
                frames.append([pl.copy() for pl in dec.ycbcr_out()])
            dec.close()
            out.append(frames)

    serial = []
    for hdr, pkts in streams:
        o = []
        decode(hdr, pkts, o, 1)
        serial.append(o[0])
    outs = [[] for _ in streams]
    ths = [threading.Thread(target=decode, args=(hdr, pkts, outs[i], 3)) for i, (hdr, pkts) in enumerate(streams)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for i in range(len(streams)):
        assert len(outs[i]) == 3
        for rep in outs[i]:
            for fa, fb in zip(rep, serial[i]):
                assert all(np.array_equal(a, b) for a, b in zip(fa, fb)), i


@pytest.mark.parametrize("algo,groups", [(0, 4), (2, 4), (2, 1), (1, 9)])
def test_mutated_packets_through_the_device_path(hip, algo, groups):
    """The same kind of damage tests/test_frontend_fuzz.py applies on the host, through the real
    backend: whatever th_decode_packetin accepts must reconstruct without a device fault (vectors
    pointing anywhere are clamped reads; slots and tiles are assigned by the library itself; EOB runs
    of any length, lists longer or shorter than their arrivals -- with either of the device's two walks over
    the token lists, in one piece or in groups) and the context must stay usable -- a clean key frame
    afterwards decodes bit-exactly."""
    from theora_amd.decoder import Decoder
    from theora_amd._lib import TheoraHipError
    L = hip._lib.load()
    L.thip_set_option(b"tl_algo", algo)
    L.thip_set_option(b"fe_groups", groups)
    try:
        _mutated_packets(hip)
    finally:
        L.thip_set_option(b"tl_algo", 0)
        L.thip_set_option(b"fe_groups", 6)


def _mutated_packets(hip):
    from theora_amd.decoder import Decoder
    from theora_amd._lib import TheoraHipError
    rng = np.random.default_rng(7)
    st = streamgen.Stream(176, 144, 0, seed=77)
    dec = Decoder(st.header_packets())
    ost = oracle.State(176, 144, 0)
    clean = [st.frame(0 if f % 3 == 0 else 1, density=0.7) for f in range(6)]
    accepted = 0
    for it in range(60):
        pkt = bytearray(clean[it % 6][0])
        kind = it % 3
        if kind == 0:
            for _ in range(3):
                pkt[int(rng.integers(1, len(pkt)))] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:
            del pkt[int(rng.integers(2, len(pkt))):]
        else:
            a = int(rng.integers(1, len(pkt)))
            pkt[a:a + 40] = bytes(rng.integers(0, 256, min(40, len(pkt) - a), dtype=np.uint8))
        try:
            rc, _ = dec.packetin(bytes(pkt))
            accepted += rc == 0
            dec.ycbcr_out()
        except TheoraHipError:
            pass
    assert accepted > 10
    pkt, truth = st.frame(0, density=0.9)          # a clean key frame resynchronises everything
    assert dec.packetin(pkt)[0] == 0
    args = st.oracle_inputs(truth, ost)
    assert ost.decode_frame(**args) == 0
    got = dec.ycbcr_out()
    for pli in range(3):
        assert np.array_equal(got[pli], ost.get_plane(oracle.FRAME_PREV, pli)[::-1])
    dec.close()


def test_dump_video_example_on_an_ogg_file(hip, tmp_path):
    """examples/dump_video_hip.c (the reference's dump_video, against this library only): an Ogg
    file with a Theora stream multiplexed with another logical stream -> YUV4MPEG2, one FRAME per
    data packet including duplicates, pictures bit-identical to the oracle's."""
    import os
    import subprocess
    from tests import oggmux
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "dump_video_hip"
    cc = subprocess.run(["gcc", "-O1", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "dump_video_hip.c"),
                         "-L" + os.path.join(root, "theora_amd"), "-ltheora_hip",
                         "-Wl,-rpath," + os.path.join(root, "theora_amd"), "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)],
                        capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr[-2000:]
    w, h, fmt = 176, 144, 0
    st = streamgen.Stream(w, h, fmt, seed=123)
    ost = oracle.State(w, h, fmt)
    video = oggmux.LogicalStream(0x5EED, max_segs=40)
    other = oggmux.LogicalStream(0xA0D10, max_segs=9)
    hdr = st.header_packets()
    for k, p in enumerate(hdr):
        video.add_packet(p, granulepos=0, flush=(k == 0 or k == len(hdr) - 1))
    other.add_packet(b"\x01vorbis-not-really" + bytes(30), flush=True)
    want = []
    for f in range(9):
        pkt, truth = st.frame(0 if f % 4 == 0 else 1, density=0.6)
        video.add_packet(pkt, granulepos=f + 1)
        other.add_packet(bytes([f]) * (50 + 37 * f))
        if not truth["dup"]:
            assert ost.decode_frame(**st.oracle_inputs(truth, ost)) == 0
        want.append([ost.get_plane(oracle.FRAME_PREV, pli)[::-1].copy() for pli in range(3)])
    ogv = tmp_path / "clip.ogv"
    ogv.write_bytes(oggmux.interleave(video.finish(), other.finish()))
    out = tmp_path / "clip.y4m"
    r = subprocess.run([str(exe), "-o", str(out), str(ogv)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    data = out.read_bytes()
    head, _, rest = data.partition(b"\n")
    assert head == b"YUV4MPEG2 C420jpeg W176 H144 F30:1 Ip A1:1"        # the form of dump_video.c:507-510
    fsz = 6 + w * h * 3 // 2
    assert len(rest) == fsz * len(want)
    for f, planes in enumerate(want):
        rec = rest[f * fsz:(f + 1) * fsz]
        assert rec[:6] == b"FRAME\n"
        y = np.frombuffer(rec, np.uint8, w * h, 6).reshape(h, w)
        cb = np.frombuffer(rec, np.uint8, w * h // 4, 6 + w * h).reshape(h // 2, w // 2)
        cr = np.frombuffer(rec, np.uint8, w * h // 4, 6 + w * h * 5 // 4).reshape(h // 2, w // 2)
        assert np.array_equal(y, planes[0]) and np.array_equal(cb, planes[1]) and np.array_equal(cr, planes[2]), f
    assert "9 frames" in r.stderr
    # (that was the tool's default: eight packets read ahead and announced to the library; the plain loop writes the same file)
    out0 = tmp_path / "clip0.y4m"
    r = subprocess.run([str(exe), "--lookahead", "0", "-o", str(out0), str(ogv)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and out0.read_bytes() == data


def test_hip_decoder_agrees_with_ffmpeg_in_chromium(hip):
    """The whole product chain -- Ogg demultiplexer, th_decode_* front end, HIP reconstruction,
    th_decode_ycbcr_out -- against FFmpeg's Theora decoder in the Chromium that the kaleido package
    bundles (tests/test_thirdparty_decoder.py explains the method and its limits).  Skips where
    that browser is not available."""
    import base64
    import json
    from tests import test_thirdparty_decoder as T
    from theora_amd.decoder import Decoder, ogg_packets
    try:
        from kaleido.scopes.plotly import PlotlyScope
        scope = PlotlyScope(plotlyjs=T.JS)
        probe = json.loads(scope.transform({"data": [{"ogv": "", "nframes": 0, "fps": 30}], "layout": {}}, format="json"))
        assert "frames" in probe
    except Exception as e:   # noqa: BLE001
        pytest.skip("no usable kaleido/Chromium here: %r" % (e,))
    w, h, n = 64, 48, 10
    ogv, want, modes = T.make_clip(w, h, 31, n, 5, 6, False, [48, 40])
    pkts, stats = ogg_packets(ogv)
    assert stats == (0, 0)
    dec = Decoder([p[1] for p in pkts[:3]])
    mine = []
    for p in pkts[3:]:
        dec.packetin(p[1])
        mine.append([pl.astype(np.float64) for pl in dec.ycbcr_out()])
    dec.close()
    assert len(mine) == n
    for a, b in zip(mine, want):                           # HIP == oracle, bit for bit
        assert all(np.array_equal(x, y) for x, y in zip(a, b))
    out = T.play(scope, ogv, n)
    exact = 0
    for f in range(n):
        scores = T.compare({"frames": [out["frames"][f]] * n}, mine, w, h)
        g = min(range(n), key=lambda i: scores[i][0])
        assert abs(g - f) <= 1 and scores[g][0] < 0.6 and scores[g][1] < 1.5, (f, g, scores[g])
        exact += g == f
    assert exact >= n - 2


def test_stripe_callback(hip):
    """TH_DECCTL_SET_STRIPE_CB (theoradec.h:79-92): called once per decoded frame with the whole
    row range and the buffer th_decode_ycbcr_out would return; not called for duplicate frames; a
    NULL function switches it off."""
    import ctypes as C
    from theora_amd import _lib
    from theora_amd.decoder import Decoder
    st = streamgen.Stream(64, 48, 0, seed=9)
    dec = Decoder(st.header_packets())
    L = _lib.load()
    calls = []
    FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(_lib.ThImgPlane), C.c_int, C.c_int)

    def on_stripe(ctx, buf, y0, y1):
        luma = np.ctypeslib.as_array(buf[0].data, (buf[0].height, buf[0].stride))[:, :buf[0].width].copy()
        calls.append((ctx, y0, y1, luma))
    cb = FN(on_stripe)

    class StripeCb(C.Structure):
        _fields_ = [("ctx", C.c_void_p), ("fn", FN)]
    s = StripeCb(0x1234, cb)
    assert L.th_decode_ctl(dec._dec, 7, C.byref(s), C.sizeof(s)) == 0
    assert L.th_decode_ctl(dec._dec, 7, C.byref(s), 3) == -10          # TH_EINVAL
    pkt, _ = st.frame(0)
    assert dec.packetin(pkt)[0] == 0
    assert len(calls) == 1 and calls[0][:3] == (0x1234, 0, 6)
    assert np.array_equal(calls[0][3], dec.ycbcr_out()[0])
    assert dec.packetin(b"")[0] == 1                                   # duplicate frame: no callback
    assert len(calls) == 1
    off = StripeCb(0, FN())
    assert L.th_decode_ctl(dec._dec, 7, C.byref(off), C.sizeof(off)) == 0
    pkt, _ = st.frame(1, density=0.9)
    dec.packetin(pkt)
    assert len(calls) == 1
    dec.close()


def test_contexts_release_their_device_memory(hip):
    """th_decode_alloc ... th_decode_free in a loop: device memory in use afterwards is what it was
    before (three frames, the coded map, staging and the pinned output image all go back)."""
    import torch
    from theora_amd.decoder import Decoder
    st = streamgen.Stream(176, 144, 0, seed=2)
    hdr = st.header_packets()
    pkts = [st.frame(0)[0], st.frame(1, density=0.8)[0]]

    def cycle():
        dec = Decoder(hdr)
        for p in pkts:
            dec.packetin(p)
            dec.ycbcr_out()
        dec.close()
    for _ in range(3):
        cycle()                       # first uses create streams, load code objects, warm allocator pools
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(60):
        cycle()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 8 << 20, (free0, free1)


def _pp_tables(setup):
    """pp_dc_scale and pp_sharp_mod from the setup header, following quant.c:88 (the inter / Cr pass is the last to
    write it) and decode.c:398-409, independently of the library."""
    from tests.streamgen import ZIGZAG
    dcs, shm = np.zeros(64, np.int32), np.zeros(64, np.int32)
    for qi in range(64):
        sizes, bmis = setup.qr[(1, 2)]
        qri, start = 0, 0
        while qri < len(sizes) - 1 and qi > start + sizes[qri]:
            start += sizes[qri]
            qri += 1
        size, end = sizes[qri], start + sizes[qri]
        base0 = (2 * (end - qi) * int(setup.bms[bmis[qri]][0]) + 2 * (qi - start) * int(setup.bms[bmis[qri + 1]][0]) + size) // (2 * size)
        dcs[qi] = setup.dcscale[qi] * base0 // 160
        qsum = 0
        for qti in range(2):
            for pli in range(3):
                zz = setup.qmat(qti, pli, qi)[ZIGZAG]     # dequant table in zig-zag order
                qsum += int(zz[12] + zz[17] + zz[18] + zz[24]) << (1 if pli == 0 else 0)
        shm[qi] = -(qsum >> 11)
    return dcs, shm


@pytest.mark.parametrize("w,h,fmt,level", [(176, 144, 0, 7), (64, 48, 0, 2), (48, 64, 3, 4), (80, 48, 2, 6), (176, 144, 0, 1),
                                           (336, 32, 0, 7)])
def test_postprocessing_through_th_decode_ctl(hip, w, h, fmt, level):
    """TH_DECCTL_SET_PPLEVEL on real packets: the level is set before the first key frame; th_decode_ycbcr_out must
    hand out the oracle's post-processed picture (oracle.State.postprocess on the oracle's own decode, the DC
    quantiser indices tracked as decode.c:1220-1243 does, frag_qi = qis[qii] with stale qii for uncoded blocks)
    while the references keep decoding bit-exactly underneath (inter frames follow)."""
    import ctypes as C
    from theora_amd.decoder import Decoder
    st = streamgen.Stream(w, h, fmt, seed=w + h + level)
    dec = Decoder(st.header_packets())
    mx = C.c_int(-1)
    assert dec._L.th_decode_ctl(dec._dec, 1, C.byref(mx), C.sizeof(mx)) == 0 and mx.value == 7     # TH_DECCTL_GET_PPLEVEL_MAX
    bad = C.c_int(8)
    assert dec._L.th_decode_ctl(dec._dec, 3, C.byref(bad), C.sizeof(bad)) == -10                   # TH_EINVAL
    lv = C.c_int(level)
    assert dec._L.th_decode_ctl(dec._dec, 3, C.byref(lv), C.sizeof(lv)) == 0                       # TH_DECCTL_SET_PPLEVEL
    dcs, shm = _pp_tables(st.setup)
    ost = oracle.State(w, h, fmt)
    n = ost.nfrags
    dc_qis, qii_persist, qis_persist = None, np.zeros(n, np.int64), [0, 0, 0]
    for f in range(8):
        pkt, truth = st.frame(0 if f % 5 == 0 else 1, density=[0.9, 0.5, 0.15][f % 3])
        rc, _ = dec.packetin(pkt)
        if truth["dup"]:
            assert rc == 1
        else:
            assert rc == 0
            assert ost.decode_frame(**st.oracle_inputs(truth, ost)) == 0
            cf = truth["coded_fragis"]
            for k, q in enumerate(truth["qis"]):
                qis_persist[k] = int(q)
            qii_persist[cf] = truth["qii"][cf]
            if dc_qis is None:
                if truth["frame_type"] == 0:
                    dc_qis = np.full(n, truth["qis"][0], np.uint8)
            else:
                dc_qis[cf] = truth["qis"][0]
            if level >= 2 and dc_qis is not None:
                frag_qi = np.array(qis_persist, np.uint8)[qii_persist]
                want, _ = ost.postprocess(oracle.FRAME_PREV, level, truth["flimit"] != 0, dc_qis, frag_qi, dcs, shm)
            else:
                want = [ost.get_plane(oracle.FRAME_PREV, p) for p in range(3)]
        got = dec.ycbcr_out()
        for pli in range(3):
            assert np.array_equal(got[pli], want[pli][::-1]), (f, pli, int((got[pli] != want[pli][::-1]).sum()))
    dec.close()
