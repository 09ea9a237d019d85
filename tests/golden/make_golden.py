"""Generates tests/golden/kernels.npz and qcif_sequence.json.

Provenance: these vectors come from the in-repo CPU oracle (oracle/theora_oracle.c), NOT
from a run of the reference library -- the reference cannot be built in this image (no
libogg headers).  They pin the oracle against accidental change and travel to the GPU box
as fixed inputs/outputs for the HIP path.  Re-run from the repo root:
    python tests/golden/make_golden.py
"""
import json
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from theora_amd import synth  # noqa: E402
from tests import util  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    rng = np.random.default_rng(20250929)
    out = {}
    n = 65 * 8
    x = np.zeros((n, 64), np.int16)
    x[: n // 2] = rng.integers(-32768, 32768, (n // 2, 64))
    x[n // 2:] = rng.integers(-700, 700, (n - n // 2, 64)) * (rng.random((n - n // 2, 64)) < 0.2)
    lz = (np.arange(n) % 65).astype(np.int32)
    out["idct_x"], out["idct_last_zzi"], out["idct_y"] = x, lz, oracle.idct8x8_batch(x, lz)
    fx = rng.integers(-255, 256, (256, 64)).astype(np.int16)
    fx[0] = 0
    fx[1] = 255
    fx[2] = -255
    out["fdct_x"], out["fdct_y"] = fx, oracle.fdct8x8_batch(fx)
    stride, H = 64, 48
    src = rng.integers(0, 256, (H, stride)).astype(np.uint8)
    ref = np.clip(src.astype(np.int32) + rng.integers(-30, 31, (H, stride)), 0, 255).astype(np.uint8)
    m = 200
    so = (rng.integers(0, H - 8, m) * stride + rng.integers(0, stride - 8, m)).astype(np.int32)
    ro = (rng.integers(0, H - 9, m) * stride + rng.integers(0, stride - 9, m)).astype(np.int32)
    r2 = (ro + rng.integers(0, 2, m) * stride + rng.integers(0, 2, m)).astype(np.int32)
    out.update(enc_src=src, enc_ref=ref, enc_stride=np.int32(stride), enc_so=so, enc_ro=ro, enc_r2=r2,
               enc_thresh=np.int32(900))
    for op in ("sad", "satd", "satd2", "intra_satd", "intra_sad", "ssd", "sad2_thresh"):
        v, dc = oracle.enc_metric_batch(op, src, ref, stride, so, ro, r2, 900)
        out["enc_" + op] = v
        if "satd" in op:
            out["enc_" + op + "_dc"] = dc
    st = oracle.State(64, 48, 0)
    st.set_ref_idx(0, 0, 0)
    pix = rng.integers(0, 256, (48, 64)).astype(np.uint8)
    coded = (rng.random(48) < 0.6).astype(np.uint8)
    st.coded[:] = 0
    st.coded[:48] = coded
    st.set_plane(oracle.FRAME_SELF, 0, pix)
    st.loop_filter_rows(9, oracle.FRAME_SELF, 0, 0, 6)
    out.update(lf_in=pix, lf_coded=coded, lf_flimit=np.int32(9), lf_out=st.get_plane(oracle.FRAME_SELF, 0))
    np.savez_compressed(os.path.join(HERE, "kernels.npz"), **out)

    seq = dict(seed=424242, frames=10, kf_interval=4, content="mixed", crc32=[])
    geom = synth.Geometry(176, 144)
    r = np.random.default_rng(seq["seed"])
    s = oracle.State(176, 144)
    for f in range(seq["frames"]):
        fr = synth.gen_frame(geom, r, 0 if f % seq["kf_interval"] == 0 else 1, seq["content"])
        util.oracle_apply(s, fr)
        c = 0
        for pli in range(3):
            c = zlib.crc32(s.get_plane(oracle.FRAME_PREV, pli).tobytes(), c)
        seq["crc32"].append("%08x" % c)
    json.dump(seq, open(os.path.join(HERE, "qcif_sequence.json"), "w"), indent=1)
    print("wrote", os.path.join(HERE, "kernels.npz"), os.path.getsize(os.path.join(HERE, "kernels.npz")))


if __name__ == "__main__":
    main()
