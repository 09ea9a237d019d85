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
from tests import streamgen


@pytest.fixture()
def trace_env():
    old = os.environ.get("THIP_FE_TRACE_BACKEND")
    os.environ["THIP_FE_TRACE_BACKEND"] = "1"
    yield
    if old is None:
        del os.environ["THIP_FE_TRACE_BACKEND"]
    else:
        os.environ["THIP_FE_TRACE_BACKEND"] = old


@pytest.mark.parametrize("w,h,fmt", [(64, 48, 0), (176, 144, 0), (48, 64, 3), (80, 48, 2), (16, 16, 0)])
def test_slot_calls_match_ground_truth(trace_env, w, h, fmt):
    from theora_amd.decoder import Decoder
    st = streamgen.Stream(w, h, fmt, seed=w * 5 + h + fmt)
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
