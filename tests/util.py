"""Shared helpers for the parity tests: drive the oracle and the HIP path with the same
synthetic command streams."""
import numpy as np

import oracle
from theora_amd import synth


def oracle_apply(ost, frame):
    """Feed one synth frame to the oracle state (fills refi/mvs like the front end would)."""
    ost.refi[:] = frame["refi"]
    ost.mvs[:] = ((frame["mvx"] & 0xFF) | (frame["mvy"] << 8)).astype(np.int16)
    return ost.decode_frame(frame["frame_type"], frame["coded_fragis"], frame["ncoded"], frame["coeffs"],
                            frame["last_zzi"], frame["dc_quant"], frame["uncoded_fragis"], frame["flimit"])


def planes_equal(ost, gst, slot=oracle.FRAME_PREV):
    """Compare the frame both sides just finished (it is PREV after the ring rotation)."""
    bad = []
    for pli in range(3):
        a = ost.get_plane(slot, pli)
        b = gst.read_plane(gst.ref_idx(slot), pli)
        if not np.array_equal(a, b):
            ys, xs = np.nonzero(a != b)
            bad.append((pli, int((a != b).sum()), int(ys[0]), int(xs[0])))
    return bad


def run_sequence(theora_amd, w, h, fmt, nframes, content, seed, kf_interval=8, enqueue=False, check_every=1, form=None):
    """Decode nframes synthetic frames on both sides; returns list of mismatch reports.  enqueue: False (frame calls),
    True (the enqueue slots) or "alternate" (by turns, on the same state); check_every: compare every n-th frame only,
    so that the frames in between are in flight behind each other; form: the coefficient form of the frame calls
    ("levels", the default, "dequant16", or "alternate": by turns on the same state)."""
    geom = synth.Geometry(w, h, fmt)
    rng = np.random.default_rng(seed)
    ost = oracle.State(w, h, fmt)
    gst = theora_amd.State(w, h, fmt)
    reports = []
    keep = []
    for f in range(nframes):
        ftype = theora_amd.INTRA_FRAME if f % kf_interval == 0 else theora_amd.INTER_FRAME
        fr = synth.gen_frame(geom, rng, ftype, content)
        rc_o = oracle_apply(ost, fr)
        if enqueue is True or (enqueue == "alternate" and f % 2 == 0) or enqueue == "levels" or (enqueue == "levels_alternate" and f % 2 == 0):
            rc_g = enqueue_frame(theora_amd, gst, geom, fr, levels=isinstance(enqueue, str) and enqueue.startswith("levels"))
        else:
            fm = ("levels", "dequant16")[f % 2] if form == "alternate" else form
            desc, ka = synth.upload_frame(synth.pack_frame(geom, fr, fm))
            keep.append(ka)
            rc_g = theora_amd.decode_frames([gst], [desc])[0]
        assert rc_o == rc_g, (f, rc_o, rc_g)
        assert ost.ref_frame_idx == [gst.ref_idx(k) for k in range(3)], f
        if (f + 1) % check_every == 0 or f == nframes - 1:
            bad = planes_equal(ost, gst)
            if bad:
                reports.append((f, bad))
    theora_amd.synchronize()
    return reports


def enqueue_frame(theora_amd, gst, geom, fr, levels=False):
    """Drive the host-enqueue slots exactly as the reference's MCU loop would
    (decode.c:2858-2945): per MCU and plane, frag_recon for the coded fragments, one
    frag_copy_list, then the loop filter with its one-row delays.  levels: through thip_state_frag_recon_levels (the quantised
    levels, the frame's tables handed over with thip_frame_dequant_table) instead of the slot's dequantised coefficients."""
    gst.frame_begin(fr["frame_type"])
    if levels:
        for p_ in range(3):
            for q_ in range(3):
                for t_ in range(2):
                    gst.frame_dequant_table((p_ * 3 + q_) * 2 + t_, fr["dequant"][p_, q_, t_])
    coded = np.zeros(geom.nfrags, bool)
    coded[fr["coded_fragis"]] = True
    base = np.cumsum([0] + fr["ncoded"])
    done = [0, 0, 0]
    buf = np.zeros(128, np.int16)
    mcu = 4 << geom.vdec
    stripe, notstart, notdone = 0, 0, 1
    while notdone:
        notdone = int(stripe + mcu < geom.nv[0])
        for pli in range(3):
            sh = 1 if (pli and geom.vdec) else 0
            y0 = stripe >> sh
            y1 = min(geom.nv[pli], y0 + (mcu >> sh))
            lo = geom.froffset[pli] + y0 * geom.nh[pli]
            hi = geom.froffset[pli] + y1 * geom.nh[pli]
            nc = int(coded[lo:hi].sum())
            for k in range(nc):
                slot = base[pli] + done[pli] + k
                fi = int(fr["coded_fragis"][slot])
                mv = int((int(fr["mvx"][fi]) & 0xFF) | (int(fr["mvy"][fi]) << 8))
                mv = (mv + 0x8000) % 0x10000 - 0x8000
                if levels:
                    buf[:64] = fr["levels"][slot]
                    gst.frag_recon_levels(fi, pli, buf, int(fr["last_zzi"][slot]), int(fr["dc_quant"][slot]), int(fr["qii"][slot]),
                                          int(fr["refi"][fi]), mv)
                else:
                    buf[:64] = fr["coeffs"][slot]
                    gst.frag_recon(fi, pli, buf, int(fr["last_zzi"][slot]), int(fr["dc_quant"][slot]),
                                   int(fr["refi"][fi]), mv)
                assert not buf[:64].any()
            done[pli] += nc
            unc = lo + np.nonzero(~coded[lo:hi])[0]
            if unc.size:
                gst.frag_copy_list(unc)
            if fr["flimit"]:
                gst.loop_filter_frag_rows(fr["flimit"], theora_amd.FRAME_SELF, pli, y0 - notstart, y1 - notdone)
        notstart = 1
        stripe += mcu
    return gst.frame_flush()


import contextlib


def options_snapshot(L):
    """Every run-time option of the library (thip_option_name enumerates the table) except the counters."""
    import ctypes as C
    L.thip_option_name.restype = C.c_char_p
    L.thip_option_name.argtypes = [C.c_int, C.POINTER(C.c_char_p)]
    out, i = {}, 0
    while True:
        help_ = C.c_char_p()
        name = L.thip_option_name(i, C.byref(help_))
        if name is None:
            return out
        i += 1
        if (help_.value or b"").startswith(b"(counter)"):
            continue
        v = C.c_int()
        assert L.thip_get_option(name, C.byref(v)) == 0
        out[name] = v.value


@contextlib.contextmanager
def options(L, **kw):
    """Set options for the body and put back WHAT THEY WERE (not what the test believes the default is)."""
    import ctypes as C
    old = {}
    for k, v in kw.items():
        c = C.c_int()
        assert L.thip_get_option(k.encode(), C.byref(c)) == 0, k
        old[k] = c.value
        assert L.thip_set_option(k.encode(), int(v)) == 0, k
    try:
        yield
    finally:
        for k, v in old.items():
            L.thip_set_option(k.encode(), v)
