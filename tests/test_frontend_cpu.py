"""The host logic of th_decode_packetin without a GPU: a context in slot-trace mode
(THIP_FE_TRACE_BACKEND=1, include/theoradec_hip.h) parses packets completely -- headers, coded
flags, modes, vectors, block qi, DCT tokens, DC un-prediction, dequantisation -- and records the
accel-vtable slot calls it would make.  They must equal what the packet generator's ground
truth turns into through the oracle's own DC un-prediction, and feeding them to the oracle
must give the oracle's picture."""
import os

import numpy as np
import pytest

import oracle
from tests import streamgen, util


@pytest.fixture()
def trace_env():
    """Contexts allocated inside the test record the slot calls instead of running them (option fe_trace_backend, the
    THIP_FE_TRACE_BACKEND of the environment): no device needed."""
    from theora_amd import _lib
    L = _lib.load()
    old = L.thip_option(b"fe_trace_backend")
    assert L.thip_set_option(b"fe_trace_backend", 1) == 0
    yield
    L.thip_set_option(b"fe_trace_backend", old)


@pytest.mark.parametrize("trees", ["random", "matched"])
@pytest.mark.parametrize("w,h,fmt", [(64, 48, 0), (176, 144, 0), (48, 64, 3), (80, 48, 2), (16, 16, 0)])
def test_slot_calls_match_ground_truth(trace_env, w, h, fmt, trees):
    from theora_amd.decoder import Decoder
    st = streamgen.Stream(w, h, fmt, seed=w * 5 + h + fmt, trees=trees)
    dec = Decoder(st.header_packets())
    ost = oracle.State(w, h, fmt)          # ground truth -> oracle
    ost2 = oracle.State(w, h, fmt)         # recorded slot calls -> oracle
    checked = 0
    for f in range(10):
        pkt, truth = st.frame(0 if f % 4 == 0 else 1, density=[0.9, 0.5, 0.15][f % 3])
        rc, _ = dec.packetin(pkt)
        if truth["dup"]:
            assert rc == 1
            continue
        assert rc == 0
        want = st.oracle_inputs(truth, ost)
        assert ost.decode_frame(**want) == 0
        got = dec.slot_trace()
        cf = want["coded_fragis"]
        assert got["frame_type"] == want["frame_type"] and got["flimit"] == want["flimit"]
        assert np.array_equal(got["fragi"], cf)
        assert np.array_equal(got["coeffs"], want["coeffs"])
        assert np.array_equal(got["last_zzi"], want["last_zzi"])
        assert np.array_equal(got["dc_quant"], want["dc_quant"])
        assert np.array_equal(np.sort(got["uncoded"]), np.sort(want["uncoded_fragis"]))
        assert np.array_equal(got["refi"], truth["refi"][cf])
        mv = ((truth["mvx"] & 0xFF) | (truth["mvy"] << 8)).astype(np.int16)
        inter = truth["refi"][cf] != oracle.FRAME_SELF
        assert np.array_equal(got["mv"][inter], mv[cf][inter])
        # and the recorded calls reproduce the picture
        ost2.refi[:] = oracle.FRAME_NONE
        ost2.refi[got["fragi"]] = got["refi"]
        ost2.mvs[:] = 0
        ost2.mvs[got["fragi"]] = got["mv"]
        ncoded = [int((got["pli"] == p).sum()) for p in range(3)]
        assert ost2.decode_frame(got["frame_type"], got["fragi"], ncoded, got["coeffs"], got["last_zzi"],
                                 got["dc_quant"], got["uncoded"], got["flimit"]) == 0
        for pli in range(3):
            assert np.array_equal(ost2.get_plane(oracle.FRAME_PREV, pli), ost.get_plane(oracle.FRAME_PREV, pli))
        checked += 1
    assert checked >= 6
    dec.close()


def test_trace_request_needs_a_trace_context(trace_env):
    from theora_amd.decoder import Decoder
    st = streamgen.Stream(32, 32, 0, seed=1)
    dec = Decoder(st.header_packets())
    pkt, truth = st.frame(0)
    assert dec.packetin(pkt)[0] == 0
    assert dec.slot_trace()["fragi"].size == int(truth["coded"].sum())
    # a truncated packet must neither crash nor report more blocks than exist
    assert dec.packetin(pkt[: len(pkt) // 3])[0] in (0, 1)
    t = dec.slot_trace()
    assert t["fragi"].size + t["uncoded"].size == truth["coded"].size
    dec.close()


def test_small_codec_h_helpers(trace_env):
    """th_version_*, th_packet_isheader / _iskeyframe, th_granule_time, th_comment_* (codec.h;
    internal.c:189-210, state.c:1259, info.c) -- the bits of the API players call around the decoder."""
    import ctypes as C
    from theora_amd import _lib
    from theora_amd.decoder import Decoder, _packet
    L = _lib.load()
    assert L.th_version_number() == 0x030201 and b"theora" in L.th_version_string()
    st = streamgen.Stream(32, 32, 0, seed=4)
    hdr = st.header_packets()
    dec = Decoder(hdr)
    key, _ = st.frame(0)
    delta, _ = st.frame(1, density=0.9)
    for data, want_hdr, want_key in ((hdr[0], 1, -1), (hdr[2], 1, -1), (key, 0, 1), (delta, 0, 0), (b"", 0, 0)):
        op, keep = _packet(data)
        assert L.th_packet_isheader(C.byref(op)) == want_hdr
        assert L.th_packet_iskeyframe(C.byref(op)) == want_key
    rc, gp0 = dec.packetin(key)
    rc, gp1 = dec.packetin(delta)
    assert L.th_granule_frame(dec._dec, gp0) == 0 and L.th_granule_frame(dec._dec, gp1) == 1
    assert abs(L.th_granule_time(dec._dec, gp1) - 2 / 30.0) < 1e-12 and L.th_granule_time(dec._dec, -1) == -1
    tc = dec.comment
    assert L.th_comment_query_count(C.byref(tc), b"title") == 1                      # case-insensitive tag
    assert C.string_at(L.th_comment_query(C.byref(tc), b"TITLE", 0)) == b"gen"
    assert L.th_comment_query(C.byref(tc), b"TITLE", 1) is None and L.th_comment_query(C.byref(tc), b"ARTIST", 0) is None
    L.th_comment_add_tag(C.byref(tc), b"Artist", b"nobody")
    L.th_comment_add(C.byref(tc), b"ARTIST=somebody")
    assert L.th_comment_query_count(C.byref(tc), b"artist") == 2
    assert C.string_at(L.th_comment_query(C.byref(tc), b"artist", 1)) == b"somebody"
    dec.close()


def test_headerin_return_codes_follow_decinfo(trace_env):
    """th_decode_headerin's codes and the order of its checks (decinfo.c:182-258, codec.h:77-93):
    TH_EBADHEADER = -20, TH_ENOTFORMAT = -21; the codec string is checked before anything that
    depends on the packet type; a data packet is 'not Theora' only while no info header has been seen."""
    import ctypes as C
    from theora_amd import _lib
    from theora_amd.decoder import _packet
    L = _lib.load()
    EBADHEADER, ENOTFORMAT, EFAULT = -20, -21, -1
    st = streamgen.Stream(64, 48, 0, seed=6)
    hp = st.header_packets()
    data, _ = st.frame(0)

    def fresh():
        info, tc, setup = _lib.ThInfo(), _lib.ThComment(), C.c_void_p()
        L.th_info_init(C.byref(info))
        L.th_comment_init(C.byref(tc))
        return info, tc, setup

    def hin(ctx, pkt, bos=0, tc_null=False, setup_null=False):
        info, tc, setup = ctx
        op, keep = _packet(pkt, bos=bos)
        return L.th_decode_headerin(C.byref(info), None if tc_null else C.byref(tc),
                                    None if setup_null else C.byref(setup), C.byref(op))

    ctx = fresh()
    assert hin(ctx, data) == ENOTFORMAT                      # data packet, nothing seen (decinfo.c:196)
    assert hin(ctx, hp[1]) == EBADHEADER                     # comment before info (:226)
    assert hin(ctx, hp[2]) == EBADHEADER                     # setup before info (:238)
    bad = bytearray(hp[0]); bad[3] ^= 0xFF
    assert hin(ctx, bytes(bad), bos=1) == ENOTFORMAT         # not "theora" (:213)
    bad = bytearray(hp[1]); bad[2] ^= 0xFF
    assert hin(ctx, bytes(bad)) == ENOTFORMAT                # the codec string comes before the type's state checks
    assert hin(ctx, hp[0], bos=0) == EBADHEADER              # info header must be b_o_s (:218)
    assert hin(ctx, bytes([0x83]) + b"theora") == EBADHEADER  # unknown header type (:251)
    assert hin(ctx, hp[0], bos=1) == 3
    assert hin(ctx, hp[0], bos=1) == EBADHEADER              # a second info header
    assert hin(ctx, data) == EBADHEADER                      # data packet, info seen, comment missing (:202)
    assert hin(ctx, data, tc_null=True) == EFAULT            # (:199)
    assert hin(ctx, hp[1]) == 2
    assert hin(ctx, data) == EBADHEADER                      # setup missing (:207)
    assert hin(ctx, data, setup_null=True) == EFAULT         # (:205)
    assert hin(ctx, hp[2]) == 1
    assert hin(ctx, data) == 0                               # all three seen: header decode ends
    L.th_setup_free(ctx[2])
    L.th_comment_clear(C.byref(ctx[1]))


def test_dropped_frame_before_any_frame_shows_grey(trace_env):
    """decode.c:2757-2772: a dropped frame with no reference yet initialises the mid-grey dummy frame,
    and that is what th_decode_ycbcr_out shows."""
    from theora_amd.decoder import Decoder
    st = streamgen.Stream(32, 32, 0, seed=3)
    dec = Decoder(st.header_packets())
    assert dec.packetin(b"")[0] == 1
    for pl in dec.ycbcr_out():
        assert pl.size and np.all(pl == 0x80)
    dec.close()


@pytest.mark.parametrize("ahead,assign", [(1, 2), (3, 1), (6, 2), (4, 0)])
@pytest.mark.parametrize("w,h,fmt", [(176, 144, 0), (48, 64, 3), (80, 48, 2)])
def test_announced_packets_give_the_same_slot_calls(trace_env, w, h, fmt, ahead, assign):
    """TH_DECCTL_THIP_PREFETCH_PACKET (include/theoradec_hip.h): packets announced ahead of their th_decode_packetin are parsed
    on threads of their own and adopted; return codes, granule positions and every recorded slot call must equal those of a
    context that was never told anything -- key frames, inter frames, dropped frames, several qi per frame, more announcements
    than slots, an announcement that does not match what comes, and a context freed with announcements outstanding.  With
    option fe_assign at 1 or 2 the parsers also pair tokens and fragments for the device while they decode (decode_token_list<true>);
    in slot-trace mode every adopted frame's pairing is applied token by token, as k_tok_scatter applies it, and compared with the
    coefficients of the host's own fragment-order walk -- a mismatch makes th_decode_packetin fail."""
    from theora_amd import _lib
    from theora_amd.decoder import Decoder
    L = _lib.load()
    with util.options(L, fe_assign=assign):
        st = streamgen.Stream(w, h, fmt, seed=w + 3 * h + fmt, trees="matched")
        hdr = st.header_packets()
        pk = []
        for f in range(18):
            if f in (7, 13):
                pk.append(b"")                      # a dropped frame (decode.c:2746)
            else:
                pk.append(st.frame(0 if f % 6 == 0 else 1, density=[0.9, 0.5, 0.15][f % 3])[0])
        plain, fast = Decoder(hdr), Decoder(hdr)
        nxt, taken = 0, 0
        for i, p in enumerate(pk):
            while nxt < len(pk) and nxt < i + ahead:
                if nxt < i:
                    nxt = i
                q = pk[nxt]
                if nxt == 10 and len(q) > 8:        # announce something else than what will come: dropped, parsed the ordinary way
                    q = bytes(q[:-4]) + b"\x55\xAA\x55\xAA"
                if fast.prefetch(q):
                    taken += 1
                elif len(q):
                    break                           # no slot free (option fe_lookahead: eight)
                nxt += 1
            ra, rb = plain.packetin(p), fast.packetin(p)
            assert ra == rb, (i, ra, rb)
            if ra[0] == 0:
                ta, tb = plain.slot_trace(), fast.slot_trace()
                for k in ta:
                    assert np.array_equal(ta[k], tb[k]), (i, k)
        assert taken >= 10
        fast.prefetch(pk[0])
        fast.prefetch(pk[1])
        fast.close()                                # announcements outstanding: waited for, nothing leaks, nothing hangs
        plain.close()


def test_prefetch_request_arguments_and_refusals(trace_env):
    """TH_DECCTL_THIP_PREFETCH_PACKET: TH_EFAULT without a context or a buffer, TH_EINVAL for a buffer that is not an ogg_packet;
    1 ("not taken", harmless) for an empty packet, with option fe_lookahead at 0, and once the slots are full -- and the
    packets decode the same whatever was refused."""
    import ctypes as C
    from theora_amd import _lib
    from theora_amd.decoder import Decoder, TH_DECCTL_THIP_PREFETCH_PACKET, _packet
    L = _lib.load()
    st = streamgen.Stream(64, 48, 0, seed=3)
    hdr = st.header_packets()
    pk = [st.frame(0 if f == 0 else 1, density=0.6)[0] for f in range(12)]
    dec, ref = Decoder(hdr), Decoder(hdr)
    op, keep = _packet(pk[0])
    assert L.th_decode_ctl(None, TH_DECCTL_THIP_PREFETCH_PACKET, C.byref(op), C.sizeof(op)) == _lib.EFAULT
    assert L.th_decode_ctl(dec._dec, TH_DECCTL_THIP_PREFETCH_PACKET, None, C.sizeof(op)) == _lib.EFAULT
    assert L.th_decode_ctl(dec._dec, TH_DECCTL_THIP_PREFETCH_PACKET, C.byref(op), 4) == _lib.EINVAL
    assert dec.prefetch(b"") is False                       # a dropped frame: nothing to parse
    old = L.thip_option(b"fe_lookahead")
    L.thip_set_option(b"fe_lookahead", 0)
    assert dec.prefetch(pk[0]) is False                     # announcements are not taken
    L.thip_set_option(b"fe_lookahead", 3)
    try:
        taken = [dec.prefetch(p) for p in pk[:5]]
        assert taken == [True, True, True, False, False]    # three slots
        for i, p in enumerate(pk):
            ra, rb = ref.packetin(p), dec.packetin(p)
            assert ra == rb
            ta, tb = ref.slot_trace(), dec.slot_trace()
            for k in ta:
                assert np.array_equal(ta[k], tb[k]), (i, k)
    finally:
        L.thip_set_option(b"fe_lookahead", old)
    dec.close()
    ref.close()


def test_the_measured_rule_changes_sides_in_a_long_stream(trace_env):
    """Option fe_assign = 2 across its switch points (thip_frontend.cpp, fe_pair_rule) on the CPU, slot-trace mode: 170 frames,
    eight announced ahead, the settled phase shortened to six frames so that a second measurement begins inside the stream.  Every
    adopted frame's pairing is checked against the host's own walk (a mismatch fails th_decode_packetin), every slot call equals
    the plain loop's, dropped frames leave the announcements in place, and the counters show both changes of sides."""
    import ctypes as C
    from theora_amd import _lib
    from theora_amd.decoder import Decoder
    L = _lib.load()

    def counter(name):
        v = C.c_int()
        assert L.thip_get_option(name, C.byref(v)) == 0
        return v.value
    before = [counter(n) for n in (b"fe_assign_to_device", b"fe_assign_to_parsers", b"fe_lookahead_adopted")]
    st = streamgen.Stream(64, 48, 0, seed=99, trees="matched")
    hdr = st.header_packets()
    pk = [st.frame(0 if f % 17 == 0 else 1, density=[0.9, 0.5, 0.15][f % 3])[0] for f in range(170)]
    with util.options(L, fe_assign=2, fe_assign_settle=6, fe_lookahead=8):
        plain, fast = Decoder(hdr), Decoder(hdr)
        nxt = 0
        for i, p in enumerate(pk):
            while nxt < len(pk) and nxt < i + 8:
                nxt = max(nxt, i)
                if not fast.prefetch(pk[nxt]) and len(pk[nxt]):
                    break
                nxt += 1
            ra, rb = plain.packetin(p), fast.packetin(p)
            assert ra == rb, (i, ra, rb)
            if ra[0] == 0:
                ta, tb = plain.slot_trace(), fast.slot_trace()
                for k in ta:
                    assert np.array_equal(ta[k], tb[k]), (i, k)
        fast.close()
        plain.close()
    after = [counter(n) for n in (b"fe_assign_to_device", b"fe_assign_to_parsers", b"fe_lookahead_adopted")]
    nonempty = sum(1 for p in pk if len(p))
    assert after[2] - before[2] >= nonempty - 12, (before, after, nonempty)     # (all but the first few and the frames without coded blocks)
    assert after[0] > before[0] and after[1] > before[1], (before, after)


def test_the_plain_clone_of_the_token_loop():
    """decode_token_list has two clones of one body, chosen when the library is loaded: compiled for BMI2 where the CPU has it (every
    box these tests run on), plain otherwise.  THIP_FE_NO_BMI2=1 selects the plain one: the ground-truth test above once more, in a
    process of its own."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, THIP_FE_NO_BMI2="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", os.path.join(root, "tests", "test_frontend_cpu.py"),
                        "-k", "test_slot_calls_match_ground_truth and matched"], capture_output=True, text=True, env=env, cwd=root, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-500:]
    assert "5 passed" in r.stdout, r.stdout[-500:]
