"""The LEVELS form of the fragment command stream (include/theora_hip.h: THIP_COEFFS_LEVELS): quantised levels in 64-byte
int8 units (int16 in wide tiles) + the frame's dequantisation tables, `(ogg_int16_t)(coeff*ac_quant[zzi])` of
decode.c:1573-1574 done inside k_recon_lf / k_recon.  Every test compares with the oracle, which receives the products the
reference's token expansion would hand to oc_state_frag_recon."""
import ctypes as C

import numpy as np
import pytest

import oracle
from theora_amd import synth
from tests import util

pytestmark = pytest.mark.gpu

PF_420, PF_422, PF_444 = 0, 2, 3


@pytest.mark.parametrize("content", ["mixed", "smooth", "dense"])
@pytest.mark.parametrize("form", ["dequant16", "alternate"])
@pytest.mark.parametrize("w,h,fmt", [(176, 144, PF_420), (80, 112, PF_422), (336, 16, PF_444)])
def test_both_forms_and_by_turns_on_one_state(hip, w, h, fmt, content, form):
    """The int16 form of round 1-3 still decodes, and a state may receive the two forms by turns (util.run_sequence's default
    is the levels form: every other sequence test of the suite runs on it)."""
    rep = util.run_sequence(hip, w, h, fmt, nframes=10, content=content, seed=w + h + fmt, kf_interval=4, form=form)
    assert not rep, rep[:3]


@pytest.mark.parametrize("fuse", [3, 0])
@pytest.mark.parametrize("p_dc_only", [0.9, 0.6, 0.0])
@pytest.mark.parametrize("big", [0.0, 0.003, 0.3])
def test_narrow_and_wide_tiles_on_every_transform_path(hip, fuse, p_dc_only, big):
    """Tiles with <= 16, <= 32 and more coefficient-owning lanes (four, two, one lane per block) x tiles whose levels fit
    eight bits, some that do not, nearly all that do not (int16 units), with three qi per frame, intra and inter tables, the
    factors' whole 16-bit range (class 'extreme') and products that wrap: one pass (k_recon_lf) and two (k_recon)."""
    content = dict(p_coded=1.0, intra=0.3, golden=0.1, zeromv=0.2, halfpel=0.4, p_dc_only=p_dc_only, p_zz10=0.4 * (1 - p_dc_only),
                   amp=120, edge_mv=0.2, extreme=0.05, big_levels=big)
    hip._lib.load().thip_set_option(b"fuse", fuse)
    try:
        rep = util.run_sequence(hip, 384, 96, PF_420, nframes=8, content=content, seed=int(p_dc_only * 10 + big * 1000), kf_interval=4)
    finally:
        hip._lib.load().thip_set_option(b"fuse", 3)
    assert not rep, rep[:3]


def test_every_level_value_at_every_position(hip):
    """All 256 byte values at all 63 AC positions of a narrow unit, against tables of distinct 16-bit factors (a swapped byte,
    a missed sign extension or a transposed table entry cannot hide), and the int16 range in wide units."""
    w, h = 256, 64
    geom = synth.Geometry(w, h, PF_444)
    rng = np.random.default_rng(3)
    for wide in (False, True):
        ost, gst = oracle.State(w, h, PF_444), hip.State(w, h, PF_444)
        for f in range(3):
            fr = synth.gen_frame(geom, rng, hip.INTRA_FRAME if f == 0 else hip.INTER_FRAME, "dense", flimit=3)
            n = fr["levels"].shape[0]
            lv = np.zeros((n, 64), np.int16)
            if wide:
                lv[:, 1:] = rng.integers(-32768, 32768, (n, 63))
            else:
                k = np.arange(n)[:, None] * 7 + np.arange(63)[None, :] * 29 + f * 111
                lv[:, 1:] = ((k % 256) - 128).astype(np.int16)
                lv[:, 1:][lv[:, 1:] == -128] = 127
            lv[:, 0] = fr["levels"][:, 0]
            fr["levels"] = lv
            fr["dequant"] = rng.permutation(65535)[:3 * 3 * 2 * 64].reshape(3, 3, 2, 64).astype(np.uint16) + 1
            fr["last_zzi"][:] = 63
            fr["coeffs"] = synth.dequantise(geom, fr)
            packed = synth.pack_frame(geom, fr)
            assert (packed["wide_tiles"] > 0) == wide
            util.oracle_apply(ost, fr)
            desc, ka = synth.upload_frame(packed)
            hip.decode_frames([gst], [desc])
            assert not util.planes_equal(ost, gst), (wide, f)


def test_the_two_forms_in_one_call(hip):
    """Streams of one thip_decode_frames call need not share the form (the library cuts its launches where it changes)."""
    sizes = [(176, 144, PF_420), (64, 48, PF_444), (176, 144, PF_420), (80, 112, PF_422), (336, 32, PF_420)]
    geoms = [synth.Geometry(*s) for s in sizes]
    rngs = [np.random.default_rng(40 + i) for i in range(len(sizes))]
    osts = [oracle.State(*s) for s in sizes]
    gsts = [hip.State(*s) for s in sizes]
    keep = []
    for f in range(6):
        descs = []
        for i in range(len(sizes)):
            fr = synth.gen_frame(geoms[i], rngs[i], hip.INTRA_FRAME if f % 4 == 0 else hip.INTER_FRAME, "mixed", flimit=[2, 9][i % 2])
            util.oracle_apply(osts[i], fr)
            d, ka = synth.upload_frame(synth.pack_frame(geoms[i], fr, ("levels", "dequant16")[(i + f) % 2]))
            keep.append(ka)
            descs.append(d)
        hip.decode_frames(gsts, descs)
        for i in range(len(sizes)):
            assert not util.planes_equal(osts[i], gsts[i]), (f, i)


def test_levels_form_argument_checks(hip):
    w, h = 64, 48
    geom = synth.Geometry(w, h)
    rng = np.random.default_rng(9)
    fr = synth.gen_frame(geom, rng, hip.INTRA_FRAME, "mixed")
    gst = hip.State(w, h)
    desc, ka = synth.upload_frame(synth.pack_frame(geom, fr))
    L = hip._lib.load()

    def call(d):
        hs = (C.c_void_p * 1)(gst.handle)
        ds = (hip.FrameDesc * 1)(d)
        return L.thip_decode_frames(hs, ds, 1, None, None)
    bad = hip.FrameDesc.from_buffer_copy(desc)
    bad.dequant = None
    assert call(bad) == hip._lib.EFAULT
    bad = hip.FrameDesc.from_buffer_copy(desc)
    bad.coeff_format = 2
    assert call(bad) == hip._lib.EINVAL
    bad = hip.FrameDesc.from_buffer_copy(desc)
    bad.nslots = 2 * desc.ncoded + 1
    assert call(bad) == hip._lib.EINVAL
    assert gst.ref_idx(hip.FRAME_SELF) == -1          # nothing was decoded
    assert call(desc) == 0


@pytest.mark.parametrize("sb_tiles", [0, 1 << 30])
@pytest.mark.parametrize("form", ["levels", "dequant16"])
@pytest.mark.parametrize("w,h,fmt,content", [(176, 144, PF_420, "mixed"), (80, 112, PF_422, "dense"), (336, 48, PF_444, "mixed"),
                                             (16, 16, PF_420, "mixed"), (1280, 720, PF_420, "smooth"), (208, 272, PF_420, "dense")])
def test_one_tile_per_wave_and_one_super_block_per_wave(hip, sb_tiles, form, w, h, fmt, content):
    """k_recon_lf (a tile of 64 blocks per wave) and k_recon_lf_sb (a super block per wave, four lanes per block: the kernel of
    launches that leave the chip empty, option sb_tiles) on the same sequences: ragged super blocks, every pixel format, both
    coefficient forms, wide tiles, band boundaries in every plane."""
    hip._lib.load().thip_set_option(b"sb_tiles", sb_tiles)
    try:
        rep = util.run_sequence(hip, w, h, fmt, nframes=7, content=content, seed=w + 3 * h + fmt, kf_interval=4, form=form)
    finally:
        hip._lib.load().thip_set_option(b"sb_tiles", 600)
    assert not rep, rep[:3]


@pytest.mark.parametrize("w,h,fmt,content", [(176, 144, PF_420, "mixed"), (80, 112, PF_422, "dense"), (336, 48, PF_444, "mixed"), (256, 96, PF_420, "smooth")])
@pytest.mark.parametrize("mode", ["levels", "levels_alternate"])
def test_levels_through_the_enqueue_slot(hip, w, h, fmt, content, mode):
    """thip_state_frag_recon_levels: the slot fed with quantised levels in the reference's MCU order -- narrow tiles, tiles that turn
    wide after some of their blocks have been packed (class 'mixed': a level beyond eight bits now and then), frames by turns
    with thip_decode_frames on the same state."""
    rep = util.run_sequence(hip, w, h, fmt, nframes=8, content=content, seed=w * 5 + h, kf_interval=4, enqueue=mode)
    assert not rep, rep[:3]


def test_a_tile_turns_wide_with_its_last_block(hip):
    """Every slot-owning block of every tile narrow except the LAST one that arrives: all units of the tile are rewritten."""
    w, h = 256, 64
    geom = synth.Geometry(w, h, PF_420)
    rng = np.random.default_rng(8)
    ost, gst = oracle.State(w, h, PF_420), hip.State(w, h, PF_420)
    for f in range(3):
        fr = synth.gen_frame(geom, rng, hip.INTRA_FRAME if f == 0 else hip.INTER_FRAME, "dense", flimit=2)
        pos = geom.frag_pos[fr["coded_fragis"]]
        tile = pos >> 6
        last = np.r_[tile[1:] != tile[:-1], True]
        fr["levels"][last, 5] = 300 * (1 if f % 2 else -1)
        fr["coeffs"] = synth.dequantise(geom, fr)
        util.oracle_apply(ost, fr)
        util.enqueue_frame(hip, gst, geom, fr, levels=True)
        assert not util.planes_equal(ost, gst), f


def test_one_form_per_frame_for_slot_owning_blocks(hip):
    w, h = 64, 48
    geom = synth.Geometry(w, h)
    gst = hip.State(w, h)
    L = hip._lib.load()
    gst.frame_begin(hip.INTRA_FRAME)
    buf = np.zeros(128, np.int16)
    buf[1] = 5
    gst.frag_recon_levels(0, 0, buf, 3, 20, 0, hip.FRAME_SELF, 0)
    buf[1] = 5
    assert L.thip_state_frag_recon(gst.handle, 1, 0, buf.ctypes.data, 3, 20, hip.FRAME_SELF, 0) == hip._lib.EINVAL
    assert buf[1] == 5                                   # a refused call leaves the block alone
    assert L.thip_state_frag_recon(gst.handle, 1, 0, buf.ctypes.data, 1, 20, hip.FRAME_SELF, 0) == 0     # DC-only: no slot, either entry
    assert L.thip_state_frag_recon_levels(gst.handle, 2, 0, buf.ctypes.data, 3, 20, 3, hip.FRAME_SELF, 0) == hip._lib.EINVAL   # qii 0..2
