"""Extra assurance for fe_pipeline's take-back (a script, not collected by pytest): random generator streams through th_decode_* with
packets announced ahead, some of which are then NOT handed in (th_decode_ycbcr_out has decoded them ahead by then: they are taken
back), zero-byte packets in between, second th_decode_ycbcr_out calls -- every picture and every granule position against the oracle
and against a decoder context that was never told about the skipped packets.
  python tests/soak_take_back.py <seed> <seconds>"""
import ctypes as C
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
import theora_amd
from tests import streamgen
from theora_amd.decoder import Decoder

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
limit = float(sys.argv[2]) if len(sys.argv) > 2 else 120
L = theora_amd._lib.load()


def counter(name):
    v = C.c_int()
    assert L.thip_get_option(name, C.byref(v)) == 0
    return v.value


t0, cases, frames = time.time(), 0, 0
back0, ahead0 = counter(b"fe_pipeline_taken_back"), counter(b"fe_pipelined")
L.thip_set_option(b"fe_pipeline", 1)
while time.time() - t0 < limit:
    w = int(rng.choice([32, 64, 176, 336])); h = int(rng.choice([48, 80, 144]))
    fmt = int(rng.choice([0, 2, 3]))
    st = streamgen.Stream(w, h, fmt, seed=int(rng.integers(1 << 30)), trees=str(rng.choice(["random", "matched"])))
    hdr = st.header_packets()
    n = int(rng.integers(8, 24))
    kf = int(rng.integers(3, 7))
    made = [st.frame(0 if f % kf == 0 else 1, density=float(rng.choice([0.9, 0.5, 0.15])), p_empty=0.0,
                     nqis=(int(rng.integers(1, 4)) if rng.random() < 0.5 else None)) for f in range(n)]
    L.thip_set_option(b"fe_assign", int(rng.integers(3)))
    dec, ref, ost = Decoder(hdr), Decoder(hdr), oracle.State(w, h, fmt)

    def check(i):
        got = dec.ycbcr_out()
        for pli in range(3):
            assert np.array_equal(got[pli], ost.get_plane(oracle.FRAME_PREV, pli)[::-1]), (cases, i, pli)
    i = 0
    while i < n:
        pkt, truth = made[i]
        a, b = dec.packetin(pkt), ref.packetin(pkt)
        assert a == b, (cases, i, a, b)
        if not truth["dup"]:
            assert ost.decode_frame(**st.oracle_inputs(truth, ost)) == 0
        check(i)
        ref.ycbcr_out()
        frames += 1
        step = 1
        if i + 2 < n and rng.random() < 0.5:
            # announce what follows (one to three packets), let the parsers finish, and ask for the picture again: the first
            # announced frame goes to the device there
            k = int(rng.integers(1, 4))
            for j in range(i + 1, min(n, i + 1 + k)):
                dec.prefetch(made[j][0])
            time.sleep(0.02)
            check(i)
            if rng.random() < 0.3:
                a, b = dec.packetin(b""), ref.packetin(b"")   # a dropped frame in between
                assert a == b, (cases, i, a, b)
                check(i)
                ref.ycbcr_out()
            if rng.random() < 0.6:
                step = int(rng.integers(2, 4))               # ... and then another packet comes: the frame ahead is taken back
                if made[min(i + step, n - 1)][1]["frame_type"] != 0 and rng.random() < 0.3:
                    step = 1
        i += step
    dec.close(); ref.close(); ost.close()
    cases += 1
L.thip_set_option(b"fe_assign", 2)
print("take-back soak: %d streams, %d frames bit-exact, %d frames decoded ahead, %d of them taken back, %.0f s"
      % (cases, frames, counter(b"fe_pipelined") - ahead0, counter(b"fe_pipeline_taken_back") - back0, time.time() - t0))
