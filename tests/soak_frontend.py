"""Extra front-end assurance (a script, not collected by pytest): random generator streams through th_decode_*
against the oracle fed with the generator's ground truth, for a given time.
  python tests/soak_frontend.py <seed> <seconds>     (end of round 1: 3 196 streams, all bit-exact)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import theora_amd
from tests import test_gpu_frontend as T

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
limit = float(sys.argv[2]) if len(sys.argv) > 2 else 120
t0, cases = time.time(), 0
while time.time() - t0 < limit:
    w = int(rng.choice([16, 32, 64, 176, 336])); h = int(rng.choice([16, 48, 80, 144]))
    fmt = int(rng.choice([0, 2, 3]))
    trees = str(rng.choice(["random", "matched"]))
    seed = int(rng.integers(1 << 30))
    # the path behind the entropy decoder: the host's own walk, the token lists on the device (in one piece or in groups of indices,
    # the DC chain on the caller's thread or on the context's second one), or the library's choice
    lists = [False, True, True, None][int(rng.integers(4))]
    L = theora_amd._lib.load()
    L.thip_set_option(b"fe_groups", int(rng.choice([1, 2, 3, 4, 5, 6, 7, 9])))
    L.thip_set_option(b"fe_worker", int(rng.integers(2)))
    L.thip_set_option(b"tl_levels", int(rng.integers(2)))
    L.thip_set_option(b"tl_algo", int(rng.integers(0, 3)))
    L.thip_set_option(b"tl_walk_threads", int(rng.choice([0, 256, 512, 1024])))
    # ... and the packets announced ahead or not (TH_DECCTL_THIP_PREFETCH_PACKET), their parsers pairing tokens and fragments (1), the
    # device walking the lists (0), or the measured rule (2)
    L.thip_set_option(b"fe_assign", int(rng.integers(3)))
    T.run_stream(theora_amd, w, h, fmt, seed=seed, nframes=int(rng.integers(4, 14)), kf=int(rng.integers(2, 6)), trees=trees,
                 device_lists=lists, lookahead=int(rng.choice([0, 0, 1, 3, 8])))
    cases += 1
print("front-end soak: %d streams bit-exact, %.0f s" % (cases, time.time() - t0))
