"""Extra parity assurance (a script, not collected by pytest; it lives under tests/ because it uses the oracle):
  python tests/soak_parity.py <seed> <seconds>     (end of round 1: 4 643 sequences, 0 mismatches)
 random seeds x content classes x sizes against the oracle, with the
static-block elision in its default mode and forced onto every frame (THIP_SKIP_STATIC=2 in the environment); since round 4 also
the coefficient form (levels, int16, by turns), levels beyond eight bits at random rates (wide tiles), the kernel a launch takes
(THIP_SB_TILES: k_recon_lf_sb below that many tiles) and the levels form of the enqueue slot."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import theora_amd
from theora_amd import synth
from tests import util

bad = 0
t0 = time.time()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
cases = 0
while time.time() - t0 < (float(sys.argv[2]) if len(sys.argv) > 2 else 120):
    w = int(rng.choice([64, 176, 336, 640, 1280, 1920])); h = int(rng.choice([48, 144, 272, 368, 720, 1088]))
    fmt = int(rng.choice([0, 2, 3]))
    cls = str(rng.choice(["mixed", "smooth", "dense", "static_bg", "static_1pct"]))
    content = dict(synth.CLASSES[cls])
    if rng.random() < 0.5:
        content["window"] = float(rng.choice([0.05, 0.3, 0.7]))
        content["p_coded"] = float(rng.choice([0.3, 0.85, 1.0]))
    seed = int(rng.integers(1 << 30))
    content["big_levels"] = float(rng.choice([0.0, 0.0005, 0.01, 0.3]))
    form = [None, "dequant16", "alternate"][int(rng.integers(3))]
    small = w * h <= 336 * 272
    enq = [False, False, False, True, "levels", "levels_alternate", "alternate"][int(rng.integers(7))] if small else False
    theora_amd._lib.load().thip_set_option(b"sb_tiles", int(rng.choice([0, 600, 1 << 30])))
    rep = util.run_sequence(theora_amd, w, h, fmt, nframes=int(rng.integers(6, 16)), content=content, seed=seed,
                            kf_interval=int(rng.integers(2, 9)), enqueue=enq, form=form)
    cases += 1
    if rep:
        bad += 1
        print("MISMATCH", w, h, fmt, cls, content, seed, form, enq, rep[:2])
print("soak: %d cases, %d mismatching, %.0f s, THIP_SKIP_STATIC=%s" % (cases, bad, time.time() - t0, os.environ.get("THIP_SKIP_STATIC", "default")))
sys.exit(1 if bad else 0)
