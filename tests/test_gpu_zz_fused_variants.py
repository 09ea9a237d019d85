"""The launch shapes that are not the default: the two passes k_recon + k_loopfilter (THIP_FUSE=0; the default is k_recon_lf, which
every other GPU test file exercises, and which hands frames with static blocks to the two passes) and the earlier fused designs
(THIP_FUSE=1: k_recon_walk + k_lf_seams, THIP_FUSE=2: k_recon_st + k_lf_st_seams): the sequence tests of test_gpu_frames.py in
child processes with the switch set.  In a file of their own,
collected last: these kernels hand data between concurrently running work groups (bounded waits), and a failure here must not
keep the rest of the suite from running under `pytest -x`."""
import numpy as np
import pytest
import oracle
from theora_amd import synth
from tests import util
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("waves,groups", [(8, 0), (16, 64), (1, 8), (3, 40), (5, 2048)])
def test_fused_walk_variant(hip, waves, groups):
    """THIP_FUSE=1 selects k_recon_walk + k_lf_seams: the waves of a work group deal out a range of tiles,
    hand the tile edges to each other through LDS, filter every cell that does not lie on a tile-row
    boundary and write the frame once; the second kernel filters the boundary rows and the cells on the
    cuts between two groups' ranges.  The sequence tests of test_gpu_frames.py in a child process with the switch
    on, for several shapes: the default, a single wave walking alone, few large ranges, more groups
    than tiles (cuts everywhere)."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, THIP_FUSE="1", THIP_WALK_WAVES=str(waves))
    if groups:
        env["THIP_WALK_WGS"] = str(groups)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sel = "(sequence or enqueue or batched or grey or dup or lane_shared or static_background) and not elision and not fused"
    if groups:
        sel = "(sequence_small or sequence_1080p or enqueue or batched or lane_shared) and not elision and not fused"
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_frames.py", "-m", "gpu", "-x", "-q", "-k", sel],
                       cwd=root, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_fused_super_tile_variant(hip):
    """THIP_FUSE=2 selects k_recon_st + k_lf_st_seams: one work group per super tile of 2 x 4 tiles, the filter
    cells inside it closed in LDS, the left edge handed from group to group through L2, the frame written once; the
    second kernel filters the cell rows between two super-tile rows.  The sequence tests of test_gpu_frames.py (all formats
    and sizes, slots, batches, DUP frames, the grey start, four 4K streams in one call, DC values from the
    device) in a child process with the switch on."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, THIP_FUSE="2")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sel = ("(sequence or enqueue or batched or grey or dup or lane_shared or static_background or four_concurrent "
           "or dc_unprediction) and not elision and not fused")
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_frames.py", "-m", "gpu", "-x", "-q", "-k", sel],
                       cwd=root, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_fused_single_wave_tiles_variant(hip):
    """THIP_FUSE=3 selects k_recon_lf (theora_amd/csrc/thip_fused.h): one wave per tile reconstructs it, hands its last block
    column and block row to the tiles on its right and below through per-tile records in L2 (device-scope stores where the
    reader sits on another XCD's band), takes its left / upper neighbours' edges the same way and closes all 64 filter cells
    of its region -- the frame is written once and there is no second kernel.  The sequence tests of test_gpu_frames.py (all
    formats and sizes from 16x16 to 8K, ragged tiles, slots, batches, DUP frames, the grey start, four 4K streams in one
    call, DC values from the device, the loop-filter row ranges of the enqueue slot) in a child process with the switch on."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, THIP_FUSE="3")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sel = ("(sequence or enqueue or batched or grey or dup or lane_shared or static_background or four_concurrent "
           "or dc_unprediction or beyond_4k or frame_calls) and not elision and not fused")
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_frames.py", "-m", "gpu", "-x", "-q", "-k", sel],
                       cwd=root, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_two_pass_variant(hip):
    """THIP_FUSE=0: k_recon + k_loopfilter for every frame (the default uses them only for frames that leave static blocks in
    place or have no loop filter).  The sequence tests of test_gpu_frames.py in a child process."""
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
