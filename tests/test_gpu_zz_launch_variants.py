"""The launch shape that is not the default: the two passes k_recon + k_loopfilter for every frame (option fuse = 0; the default
is k_recon_lf, which every other GPU test file exercises and which hands frames with static blocks or without a loop filter to
the two passes).  The sequence tests of test_gpu_frames.py in a child process with THIP_FUSE=0 in its environment, in a file of
its own, collected last."""
import pytest
pytestmark = pytest.mark.gpu


def test_two_pass_variant(hip):
    """THIP_FUSE=0: k_recon + k_loopfilter for every frame -- all formats and sizes from 16x16 to 8K, ragged tiles, slots,
    batches, DUP frames, the grey start, four 4K streams in one call, DC values from the device, the loop-filter row
    ranges of the enqueue slot."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, THIP_FUSE="0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sel = ("(sequence or enqueue or batched or grey or dup or lane_shared or static_background or four_concurrent "
           "or dc_unprediction or beyond_4k or frame_calls) and not elision and not fused")
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_frames.py", "-m", "gpu", "-x", "-q", "-k", sel],
                       cwd=root, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_fuse_option_switches_paths_inside_one_process(hip):
    """thip_set_option("fuse", ...) takes effect with the next call: the same sequence decoded with the option flipped between
    frames (fused, two passes, fused, ...) on ONE state stays bit-exact -- the two paths leave identical frames and
    coded maps behind."""
    import numpy as np
    import oracle
    from theora_amd import _lib, synth
    from tests import util
    L = _lib.load()
    old = L.thip_option(b"fuse")
    try:
        for (w, h, fmt) in [(336, 272, 0), (176, 144, 3)]:
            geom = synth.Geometry(w, h, fmt)
            rng = np.random.default_rng(77)
            ost, gst, keep = oracle.State(w, h, fmt), hip.State(w, h, fmt), []
            for f in range(12):
                fr = synth.gen_frame(geom, rng, hip.INTRA_FRAME if f % 6 == 0 else hip.INTER_FRAME, "mixed", flimit=[2, 7, 30][f % 3])
                util.oracle_apply(ost, fr)
                L.thip_set_option(b"fuse", 3 if f % 2 == 0 else 0)
                desc, ka = synth.upload_frame(synth.pack_frame(geom, fr))
                keep.append(ka)
                hip.decode_frames([gst], [desc])
                assert not util.planes_equal(ost, gst), (w, h, f)
    finally:
        L.thip_set_option(b"fuse", old)


def test_half_tile_variant(hip):
    """k_recon_lf_h (round 6: two super blocks a wave, two lanes a block -- the launches between k_recon_lf_sb's and k_recon_lf's)
    forced for EVERY fused launch (THIP_SB_TILES=0 THIP_HALF_TILES=huge): the sequence tests of test_gpu_frames.py -- all formats
    and sizes from 16x16 to 8K, ragged tiles, plane widths that end inside a half tile, batches, the grey start, four 4K streams in
    one call, the loop filter's row ranges -- the levels-form tests, and the forced failure of a tile hand-over with its recovery."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, THIP_SB_TILES="0", THIP_HALF_TILES="100000000")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sel = ("sequence or enqueue or batched or grey or dup or lane_shared or static_background or four_concurrent "
           "or dc_unprediction or beyond_4k or frame_calls or hand_over or failed or taken_back or config3")
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_frames.py", "tests/test_gpu_levels.py", "-m", "gpu", "-x", "-q", "-k",
                        "(" + sel + " or levels or wide or form) and not one_tile_per_wave"],   # (that test sets sb_tiles itself)
                       cwd=root, env=env, capture_output=True, text=True, timeout=2400)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
