"""Extra assurance for the look-ahead (a script, not collected by pytest; no GPU needed): random generator streams -- sizes, pixel
formats, trees, densities, dropped frames, 1..9 packets announced ahead, the three settings of fe_assign -- through two slot-trace
contexts, one told nothing, one with its packets announced: return codes, granule positions and every recorded slot call must be
equal, and every adopted frame's pairing of tokens and fragments passes the library's own check against the fragment-order walk
(a wrong pairing is TH_EFAULT).   python tests/soak_lookahead_cpu.py <seed> <streams>   (round 4's last build: 180 streams, 2 340 frames)"""
import os
import sys, random, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from theora_amd import _lib
from theora_amd.decoder import Decoder
from tests import streamgen
L = _lib.load()
L.thip_set_option(b"fe_trace_backend", 1)
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
nchk = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    w, h = 16 * rnd.randint(1, 24), 16 * rnd.randint(1, 18)
    fmt = rnd.choice([0, 2, 3])
    trees = rnd.choice(["random", "matched"])
    st = streamgen.Stream(w, h, fmt, seed=rnd.randint(0, 10**6), trees=trees)
    hdr = st.header_packets()
    pk = []
    for f in range(14):
        if rnd.random() < 0.08:
            pk.append(b"")
        else:
            pk.append(st.frame(0 if (f % rnd.randint(2, 7) == 0) else 1, density=rnd.choice([0.95, 0.6, 0.3, 0.05]))[0])
    a, b = Decoder(hdr), Decoder(hdr)
    ahead = rnd.randint(1, 9)
    L.thip_set_option(b"fe_assign", rnd.choice([0, 1, 2]))
    nxt = 0
    for i, p in enumerate(pk):
        while nxt < len(pk) and nxt < i + ahead:
            nxt = max(nxt, i)
            if not b.prefetch(pk[nxt]) and len(pk[nxt]):
                break
            nxt += 1
        ra, rb = a.packetin(p), b.packetin(p)
        assert ra == rb, (it, i, ra, rb)
        if ra[0] == 0:
            ta, tb = a.slot_trace(), b.slot_trace()
            for k in ta:
                assert np.array_equal(ta[k], tb[k]), (it, i, k)
            nchk += 1
    a.close(); b.close()
print("soak ok:", nchk, "frames compared")
