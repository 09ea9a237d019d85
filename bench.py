#!/usr/bin/env python3
"""bench.py -- decode throughput of the MI355X fragment-reconstruction path.

A "step" decodes ONE frame of each of the rank's streams (default: 4 concurrent 4K 4:2:0
streams per GPU, keyframe interval 64) from fragment command streams already resident in
HBM: coded-fragment reconstruction (dequantised coefficients -> iDCT -> intra / motion
compensated predictor -> pixels), uncoded-fragment copy and the in-loop deblocking filter,
through the C ABI (thip_decode_frames).  Prints ONE JSON line (rank 0), at most 6 KB: the
contract's keys, `roofline`, `cpu_baseline`, and one {value, ms_per_step, frac} triple per
keyed entry (every BASELINE.json config); everything else -- notes, spreads, per-section
wall times, the encoder kernels' full lines -- goes to bench_detail.json (written next to
gpurun_out/ when that exists, else to the working directory; --detail names it).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--size 4k|1080p|720p]
                  [--content dense|smooth|mixed] [--streams-per-gpu S]
For N>1 launch with  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N
Streams are sharded whole across ranks (no data-path collective); RCCL is used for the
barriers, the max-over-ranks time and the checksum gather.

Secondary modes (one GPU, not the headline metric):
  python bench.py --mode enc    BASELINE.json config 5: encoder block kernels, 1920x1088 4:4:4
  python bench.py --mode e2e    packets in host memory -> th_decode_* -> YUV in host memory

How the run is organised (475 s of wall, 1 % GPU busy): every synthetic command stream is generated, packed and --
for the streams that are compared -- decoded by the oracle in WORKER PROCESSES (numpy + oracle only) that start before torch is
imported; they hand their arrays over as .npz files, which the rocprofv3 --pmc re-executions of this script read too instead of
generating them again.  The CPU baseline is timed behind everything else, on an idle box.
"""
import argparse
import json
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SIZES = {"4k": (3840, 2160), "1080p": (1920, 1088), "720p": (1280, 720), "cif": (352, 288), "qcif": (176, 144)}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec peak
INFINITY_CACHE_MB = 256.0      # same guide: 256 MiB die-level L3
# vector-ALU issue: one wave-instruction per SIMD every 4.3 clocks for the packed / permute / multiply mix these kernels are made of
# (profiles/r04_valu_rate2.txt, two or more waves per SIMD), 1024 SIMDs at 2.4 GHz
VALU_PEAK_GINST = 1024 * 2.4 / 4.3
FUSED = os.environ.get("THIP_FUSE", "3")
# option "fuse": 3 = k_recon_lf (reconstruction + loop filter in one pass), anything else = the two passes
KERNEL_NAMES = ("k_recon_lf", None) if FUSED == "3" else ("k_recon", "k_loopfilter")
KF_INTERVAL = 64


def _kernel_name(name):
    """'void k_recon_lf<true>(BatchK)' -> 'k_recon_lf<true>' (the kernels exist once per coefficient form)."""
    n = name.split("(")[0].strip()
    if n.startswith("void "):
        n = n[5:]
    return n.replace("thip::", "")


def _kernel_base_name(name):
    return _kernel_name(name).split("<")[0]


def _usable_cores():
    """CPU cores this process may really use: the cgroup quota if there is one, else the affinity mask."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


# Frame-39 CRC32s of streams 0..3 of the default workload (4K dense, pool 6, kf 64): the same on every run, every world
# size and every launch shape -- a cheap end-to-end regression check of the timed batch.  Rounds 2 and 3 had
# 78bed3bc a82eb830 ee6d99cc 3ad204b2; round 4 changed the GENERATOR (theora_amd/synth.py: a frame is now quantised levels x
# dequantisation tables, as a real stream is, where it used to be free-form products that no table factors), so the pictures
# are different pictures; the oracle decodes the new ones to these values (checked by this script before it compares).
COMMITTED_CRC_FRAME39 = ["b4858401", "1f7f36d9", "83b0e5ba", "726f206f"]


# =====================================================================================================================
# Worker processes (spawned; numpy + oracle only, never torch / HIP): generation, packing, the oracle's pictures, CPU baseline
# =====================================================================================================================
def seq_frame(i, g, G, pool):
    """A stream's frame sequence: interval n is a key frame followed by KF_INTERVAL - 1 inter frames drawn from the stream's pool
    (shifted by the interval number, so that consecutive intervals are different pictures).  State g of G side-by-side states
    decodes intervals g, g + G, ...: at step i it is at frame i % KF_INTERVAL of interval g + G * (i // KF_INTERVAL).  Returns the
    index into [key, inter_1 .. inter_pool]."""
    j, interval = i % KF_INTERVAL, g + G * (i // KF_INTERVAL)
    return 0 if j == 0 else 1 + ((j + 3 * interval) % pool)


_HOST_FIELDS = ("coded_fragis", "uncoded_fragis", "refi", "mvx", "mvy", "coeffs", "last_zzi", "dc_quant")


def _npz_save(path, d):
    np.savez(path, **{k: np.asarray(v) for k, v in d.items()})


def _npz_load(path):
    with np.load(path) as z:
        return {k: (z[k] if z[k].ndim else z[k].item()) for k in z.files}


def _oracle_decode(ost, fr):
    ost.refi[:] = fr["refi"]
    ost.mvs[:] = ((np.asarray(fr["mvx"]) & 0xFF) | (np.asarray(fr["mvy"]) << 8)).astype(np.int16)
    nc = fr["ncoded"]
    ost.decode_frame(int(fr["frame_type"]), fr["coded_fragis"], [int(x) for x in np.asarray(nc).reshape(-1)], fr["coeffs"], fr["last_zzi"],
                     fr["dc_quant"], fr["uncoded_fragis"], int(fr["flimit"]))


def _gen_job(job):
    """One stream's command streams in a worker process: generated from its seed (theora_amd/synth.py), optionally with a share
    of its tiles widened, packed in every form asked for and written as .npz files into job["dir"]; with job["parity"] the ORACLE
    decodes the stream's sequence that many frames deep and its three planes are written beside them; with job["host"] the
    frames themselves are written too (the CPU-baseline workers read them).  Returns a small dict of paths and numbers."""
    import theora_amd
    from theora_amd import synth
    t0 = time.perf_counter()
    w, h = SIZES[job["size"]]
    geom = synth.Geometry(w, h)
    rng = np.random.default_rng(job["seed"])
    pool = job["pool"]
    frames = [synth.gen_frame(geom, rng, theora_amd.INTRA_FRAME, job["content"], flimit=2)]
    for _ in range(pool):
        frames.append(synth.gen_frame(geom, rng, theora_amd.INTER_FRAME, job["content"], flimit=2))
    nwide = 0
    if job.get("widen"):
        frac, wseed = job["widen"]
        wrng = np.random.default_rng(wseed)
        out = []
        for f in frames:
            fw, nw = synth.widen_tiles(geom, f, frac, wrng)
            nwide += nw
            out.append(fw)
        frames = out
    res = {"tag": job["tag"], "balg": [synth.algorithmic_bytes(geom, f) for f in frames], "files": {}, "wide_tiles": nwide,
           "tiles": geom.ntiles * len(frames), "desc_bytes": {}}
    for form in job.get("forms", ("levels",)):
        paths, nbytes = [], 0
        for k, f in enumerate(frames):
            p = synth.pack_frame(geom, f, form=form)
            nbytes += sum(int(np.asarray(p[x]).nbytes) for x in ("info", "coeffs", "slot0"))
            path = os.path.join(job["dir"], "%s_%s_%d.npz" % (job["tag"], form, k))
            _npz_save(path, p)
            paths.append(path)
        res["files"][form] = paths
        res["desc_bytes"][form] = nbytes
    res["gen_s"] = round(time.perf_counter() - t0, 2)
    if job.get("host"):
        hp = []
        for k, f in enumerate(frames):
            path = os.path.join(job["dir"], "%s_host_%d.npz" % (job["tag"], k))
            d = {x: f[x] for x in _HOST_FIELDS}
            d.update(frame_type=f["frame_type"], flimit=f["flimit"], ncoded=np.asarray(f["ncoded"]))
            _npz_save(path, d)
            hp.append(path)
        res["host"] = hp
    par = job.get("parity")
    if par:
        import oracle
        t1 = time.perf_counter()
        ost = oracle.State(w, h)
        for i in range(par["frames"]):
            _oracle_decode(ost, frames[seq_frame(i, par.get("g", 0), par.get("G", 1), pool)])
        path = os.path.join(job["dir"], "%s_oracle.npz" % job["tag"])
        _npz_save(path, {"p%d" % pli: ost.get_plane(oracle.FRAME_PREV, pli) for pli in range(3)})
        ost.close()
        res["oracle_planes"] = path
        res["oracle_s"] = round(time.perf_counter() - t1, 2)
    return res


def _cpu_job(job):
    """One process of the CPU baseline: the oracle (scalar, or its SSE2 build) decoding `nframes` frames of the stream whose host
    frames a _gen_job wrote (nothing but paths and numbers crosses the process boundary).  Returns the decode time in seconds."""
    paths, size, pool, nframes, simd = job
    import oracle
    w, h = SIZES[size]
    frames = [_npz_load(p) for p in paths]
    ost = oracle.State(w, h, simd=bool(simd))
    t = 0.0
    for i in range(nframes):
        fr = frames[seq_frame(i, 0, 1, pool)]
        ost.refi[:] = fr["refi"]
        ost.mvs[:] = ((fr["mvx"] & 0xFF) | (fr["mvy"] << 8)).astype(np.int16)
        nc = [int(x) for x in fr["ncoded"]]
        t0 = time.perf_counter()
        ost.decode_frame(int(fr["frame_type"]), fr["coded_fragis"], nc, fr["coeffs"], fr["last_zzi"], fr["dc_quant"], fr["uncoded_fragis"],
                         int(fr["flimit"]))
        t += time.perf_counter() - t0
    ost.close()
    return t


def _e2e_prep_job(job):
    """The packets of the keyed entry e2e_720p and the oracle's pictures of them (CRC32 per frame), in a worker process; cached
    under tools/_cache (git-ignored; the Python packet generator takes tens of seconds), keyed by the generator's source."""
    import hashlib
    import pickle
    size, nframes = job["size"], job["nframes"]
    src = open(os.path.join(ROOT, "tests", "streamgen.py"), "rb").read()
    key = hashlib.sha1(src + repr((size, nframes, "dense", 99)).encode()).hexdigest()[:16]
    cdir = os.path.join(ROOT, "tools", "_cache")
    cpath = os.path.join(cdir, "bench_e2e_%s_%d_%s.pkl" % (size, nframes, key))
    if os.path.exists(cpath):
        try:
            with open(cpath, "rb") as f:
                d = pickle.load(f)
            d["cached"] = True
            return d
        except Exception:   # noqa: BLE001 -- a damaged cache file: make the packets again
            pass
    t0 = time.perf_counter()
    from tests import streamgen
    import oracle
    w, h = SIZES[size]
    content = dict(density=0.7, p_dc_only=0.5, p_empty=0.2)
    st = streamgen.Stream(w, h, 0, seed=99, trees="matched", probe_kwargs=content)
    hdr = st.header_packets()
    made = [st.frame(0 if f % 8 == 0 else 1, **content) for f in range(nframes)]
    ost = oracle.State(w, h, 0)
    want = []
    for pkt, truth in made:
        if not truth["dup"]:
            assert ost.decode_frame(**st.oracle_inputs(truth, ost)) == 0
        want.append(zlib.crc32(b"".join(np.ascontiguousarray(ost.get_plane(oracle.FRAME_PREV, pli)[::-1]).tobytes() for pli in range(3))))
    ost.close()
    d = {"hdr": [bytes(x) for x in hdr], "pkts": [bytes(m[0]) for m in made], "want": want, "cached": False,
         "prep_s": round(time.perf_counter() - t0, 1)}
    try:
        os.makedirs(cdir, exist_ok=True)
        with open(cpath + ".tmp", "wb") as f:
            pickle.dump(d, f)
        os.replace(cpath + ".tmp", cpath)
    except OSError:
        pass
    return d


# =====================================================================================================================
# rocprofv3 --pmc re-executions (N = 1): HBM bytes and vector instructions per launch of the timed kernels
# =====================================================================================================================
def run_pmc_passes(manifest_path, want_valu=True):
    """This script re-executed as `--pmc-child MANIFEST` under rocprofv3 --pmc, ONE counter group per pass (FETCH_SIZE and
    WRITE_SIZE do not fit one pass, MI355X_MICROARCH.md), one lane so that every launch has the shape of the instrumented pass.
    The child reads the parent's .npz command streams (no generation), decodes 24 steps of the timed batch in every coefficient
    form the manifest names and calls every encoder kernel of the keyed entry twice.  Returns {kernel name incl. template
    arguments: {work items per dispatch: {counter: average per dispatch, "_n": dispatches}}} or None (no rocprofv3, a failed pass)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    acc = {}
    groups = [("FETCH_SIZE",), ("WRITE_SIZE",)] + ([("SQ_INSTS_VALU", "SQ_WAVES")] if want_valu else [])
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as td:
        for group in groups:
            d = os.path.join(td, group[0])
            cmd = [exe, "--pmc"] + list(group) + ["--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
                                                  "--pmc-child", manifest_path]
            env = dict(os.environ, THIP_LANES="1", TMPDIR=os.environ.get("TMPDIR", "/tmp"))
            try:
                r = subprocess.run(cmd, env=env, cwd=td, capture_output=True, text=True, timeout=240)
            except (OSError, subprocess.TimeoutExpired):
                return None
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                sys.stderr.write("bench: pmc pass %s failed (rc %s): %s\n" % (group[0], r.returncode, (r.stderr or "")[-300:]))
                return None
            for row in csv.DictReader(open(files[0])):
                try:
                    grid = int(row.get("Grid_Size") or 0)
                except ValueError:
                    grid = 0
                k = acc.setdefault(_kernel_name(row["Kernel_Name"]), {}).setdefault(grid, {})
                k.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    out = {}
    for name, by_grid in acc.items():
        for grid, v in by_grid.items():
            d = {c: sum(x) / len(x) for c, x in v.items()}
            d["_n"] = max(len(x) for x in v.values())
            out.setdefault(name, {})[grid] = d
    return out or None


def traffic_of(pmc, kernel):
    """HBM bytes per launch of `kernel` from the L2's fabric-side counters: FETCH_SIZE doubled (gfx950 reports half the bytes of wide
    coalesced reads, MI355X_MICROARCH.md), WRITE_SIZE as reported, both in KiB.  + vector instructions per wave when counted."""
    c = (pmc or {}).get(kernel)
    if not c or "FETCH_SIZE" not in c:
        return None
    f, w = c["FETCH_SIZE"], c.get("WRITE_SIZE", 0.0)
    res = {"fetch_kib_raw": round(f, 1), "write_kib": round(w, 1), "hbm_bytes_per_launch": int(round((2 * f + w) * 1024))}
    if c.get("SQ_INSTS_VALU") and c.get("SQ_WAVES"):
        res["valu_insts_per_wave"] = round(c["SQ_INSTS_VALU"] / c["SQ_WAVES"], 1)
    return res


def pmc_child(manifest_path):
    """The instrumented workload of run_pmc_passes (runs under rocprofv3; prints nothing that matters)."""
    import torch
    import theora_amd
    from theora_amd import synth
    m = json.load(open(manifest_path))
    torch.cuda.set_device(0)
    w, h = SIZES[m["size"]]
    keep = []
    for form, streams in m["forms"].items():
        states = [theora_amd.State(w, h) for _ in streams]
        descs = []
        for paths in streams:
            row = []
            for p in paths:
                d, ka = synth.upload_frame(_npz_load(p))
                keep.append(ka)
                row.append(d)
            descs.append(row)
        pool = len(descs[0]) - 1
        plans = [theora_amd.BatchPlan(states, [descs[s][j] for s in range(len(states))]) for j in range(pool + 1)]
        for i in range(m.get("steps", 24) + 4):
            plans[seq_frame(i, 0, 1, pool)].submit(None)
        theora_amd.synchronize()
        torch.cuda.synchronize()
        for st in states:
            st.close()
    if m.get("enc"):
        main_enc(emit=False, subset=True, calls_only=True)
    torch.cuda.synchronize()


def system_libtheora_baseline(size="720p", nframes=24):
    """if the box has a system libtheoradec, time it on generated packets (one thread) and report it as
    "system libtheora"; otherwise say so.  The packets are tests/streamgen.py's (typical content, matched Huffman trees)."""
    import ctypes as C
    import ctypes.util
    name = None
    for cand in ("libtheoradec.so.1", "libtheoradec.so", ctypes.util.find_library("theoradec")):
        if not cand:
            continue
        try:
            lib = C.CDLL(cand, mode=getattr(os, "RTLD_LOCAL", 0))
            name = cand
            break
        except OSError:
            continue
    if not name:
        return {"kind": "system libtheora", "available": False, "note": "no libtheoradec.so on this box (dlopen failed)"}
    try:
        from tests import streamgen
        from theora_amd import _lib
        from theora_amd._lib import ThImgPlane
        from theora_amd.decoder import _packet
        w, h = SIZES[size]
        content = dict(density=0.35, p_dc_only=0.45, p_empty=0.4)
        st = streamgen.Stream(w, h, 0, seed=99, trees="matched", probe_kwargs=content)
        hdr = st.header_packets()
        pkts = [st.frame(0 if f % 8 == 0 else 1, **content)[0] for f in range(8)]
        info, tc, setup = _lib.ThInfo(), _lib.ThComment(), C.c_void_p()
        lib.th_decode_alloc.restype = C.c_void_p
        lib.th_decode_alloc.argtypes = [C.c_void_p, C.c_void_p]
        lib.th_decode_packetin.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        lib.th_decode_ycbcr_out.argtypes = [C.c_void_p, C.c_void_p]
        lib.th_decode_free.argtypes = [C.c_void_p]
        lib.th_setup_free.argtypes = [C.c_void_p]
        lib.th_info_init(C.byref(info))
        lib.th_comment_init(C.byref(tc))
        for k, hp in enumerate(hdr):
            op, keep = _packet(hp, bos=1 if k == 0 else 0, packetno=k)
            if lib.th_decode_headerin(C.byref(info), C.byref(tc), C.byref(setup), C.byref(op)) <= 0:
                raise RuntimeError("th_decode_headerin refused header %d" % k)
        dec = lib.th_decode_alloc(C.byref(info), setup)
        lib.th_setup_free(setup)
        if not dec:
            raise RuntimeError("th_decode_alloc failed")
        buf = (ThImgPlane * 3)()
        ops = [_packet(p, packetno=3 + k) for k, p in enumerate(pkts)]
        t0 = time.perf_counter()
        n = 0
        while n < nframes:
            for op, keep in ops:
                gp = C.c_int64()
                if lib.th_decode_packetin(dec, C.byref(op), C.byref(gp)) < 0:
                    raise RuntimeError("th_decode_packetin failed")
                lib.th_decode_ycbcr_out(dec, buf)
                n += 1
        dt = time.perf_counter() - t0
        lib.th_decode_free(dec)
        return {"kind": "system libtheora", "available": True, "library": name, "value": round(n / dt, 2), "unit": "frames/s", "cores": 1,
                "sample": "%d packets of a generated %s 4:2:0 stream (typical content), th_decode_packetin + th_decode_ycbcr_out" % (n, size)}
    except Exception as e:   # a library that is there but does not behave: say so, never fail the bench
        return {"kind": "system libtheora", "available": True, "library": name, "error": str(e)[:200]}


def e2e_keyed_entry(prep, size="720p", loops=30, ahead=8):
    """The whole th_decode_* chain on the driver's clock, as a keyed entry of the default line: packets in host memory ->
    th_decode_packetin -> th_decode_ycbcr_out -> pictures in host memory, ONE stream, the plain API loop and with the packets
    announced `ahead` packets ahead (TH_DECCTL_THIP_PREFETCH_PACKET: the entropy decoder on the library's parser threads).  The
    legs must hand out the ORACLE's pictures (`prep`: the packets and the oracle's CRC32 per frame, made by a worker process or
    read from its cache): every frame of a second, untimed pass of each leg over the same packets.  Never fails the bench: an
    exception becomes the entry."""
    from theora_amd import _lib as _l
    try:
        from theora_amd.decoder import Decoder
        hdr, pkts, want = prep["hdr"], prep["pkts"], prep["want"]
        seq = pkts * loops
        res, crcs, bad = {}, {}, {}
        import ctypes as _C
        _v = _C.c_int(0)
        _l.load().thip_get_option(b"fe_pipeline", _C.byref(_v))
        pipe_default = _v.value
        for label, la, pipe in (("plain_loop", 0, 0), ("lookahead_%d" % ahead, ahead, 0), ("lookahead_%d_pipelined" % ahead, ahead, 1)):
            # (the third leg: option fe_pipeline -- th_decode_ycbcr_out hands the next announced frame to the device before it waits)
            _l.load().thip_set_option(b"fe_pipeline", pipe)
            dec = Decoder(hdr)
            for p in pkts:              # warm-up: device buffers, streams, parser threads
                dec.packetin(p)
                dec.ycbcr_out()
            c, nxt = [], 0
            t0 = time.perf_counter()
            for k, p in enumerate(seq):
                while la and nxt < len(seq) and nxt < k + la:
                    nxt = max(nxt, k)
                    if not dec.prefetch(seq[nxt]):
                        break
                    nxt += 1
                dec.packetin(p)
                planes = dec.ycbcr_out()
                if k < len(pkts):
                    c.append(zlib.crc32(b"".join(x.tobytes() for x in planes)))
            dt = time.perf_counter() - t0
            res[label] = round(len(seq) / dt, 1)
            crcs[label] = c
            # ... and once more with the clock off and EVERY frame's checksum taken (half a millisecond a frame in this caller:
            # inside the clock it would be the figure): the same context goes on, so this pass starts where the look-ahead's
            # measured rule stands after the timed one
            nxt, nb = 0, 0
            for k, p in enumerate(seq):
                while la and nxt < len(seq) and nxt < k + la:
                    nxt = max(nxt, k)
                    if not dec.prefetch(seq[nxt]):
                        break
                    nxt += 1
                dec.packetin(p)
                planes = dec.ycbcr_out()
                nb += zlib.crc32(b"".join(x.tobytes() for x in planes)) != want[k % len(pkts)]
            bad[label] = nb + sum(1 for k, v in enumerate(c) if v != want[k])
            dec.close()
        _l.load().thip_set_option(b"fe_pipeline", pipe_default)
        same = len(set(tuple(v) for v in crcs.values())) == 1
        native = _e2e_native_legs(hdr, pkts, loops, ahead)
        if any(bad.values()):
            return {"error": "pictures differ from the oracle's: %r frames of %d" % (bad, len(seq))}
        return {"metric": "end-to-end decode frames/sec, one %s 4:2:0 stream (packets in host memory -> pictures in host memory)" % size,
                "unit": "frames/s", **res, "same_pictures": same, "frames_equal_to_the_oracle": "all %d of each leg's second, untimed pass (and the first %d of the timed one)" % (len(seq), len(pkts)),
                "c_caller": native, "packets_from_cache": bool(prep.get("cached")),
                "avg_packet_bytes": sum(map(len, pkts)) // len(pkts),
                "data": "synthetic packets (tests/streamgen.py), dense content, matched Huffman trees",
                "note": "host-bound (Python caller): the plain loop is one entropy-decode thread per stream; announced packets are "
                        "parsed on up to eight library threads, which also pair tokens and fragments for the device (DESIGN.md 5.1)"}
    except Exception as e:   # noqa: BLE001
        try:
            _l.load().thip_set_option(b"fe_pipeline", 0)
        except Exception:   # noqa: BLE001
            pass
        return {"error": str(e)[:300]}


def _e2e_native_legs(hdr, pkts, loops, ahead):
    """The same packets through examples/decode_bench.c -- the same th_decode_* calls from C, nothing of Python between them: what a
    player or a server pays.  The packets go into an Ogg file (tests/oggmux.py, the library's own demultiplexer reads it back); the
    program is compiled here with the host compiler when it is not there.  The pictures of these legs are the library's, which the
    Python legs above have just compared with the oracle frame by frame; the program itself checks that every call succeeds."""
    import subprocess
    import tempfile
    try:
        from tests import oggmux
        root = os.path.dirname(os.path.abspath(__file__))
        with tempfile.TemporaryDirectory(prefix="thip_e2e_") as td:
            exe = os.path.join(td, "decode_bench")
            cc = subprocess.run(["gcc", "-O2", "-pthread", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "decode_bench.c"),
                                 "-L" + os.path.join(root, "theora_amd"), "-ltheora_hip", "-Wl,-rpath," + os.path.join(root, "theora_amd"), "-o", exe],
                                capture_output=True, text=True)
            if cc.returncode != 0:
                return {"error": "decode_bench.c does not compile: " + cc.stderr[-200:]}
            ls = oggmux.LogicalStream(0x7E0)
            for k, hp in enumerate(hdr):
                ls.add_packet(hp, granulepos=0, flush=(k == 0 or k == len(hdr) - 1))
            reps = 3
            for rep in range(reps):
                for k, p in enumerate(pkts):
                    ls.add_packet(p, granulepos=rep * len(pkts) + k + 1)
            ogv = os.path.join(td, "clip.ogv")
            with open(ogv, "wb") as f:
                f.write(b"".join(ls.finish()))
            out = {}
            for label, extra in (("plain_loop", []), ("lookahead_%d" % ahead, ["--lookahead", str(ahead), "--no-pipeline"]),
                                 ("lookahead_%d_pipelined" % ahead, ["--lookahead", str(ahead), "--pipeline"])):
                best = None
                for _ in range(2):      # (host-bound and short: the better of two runs)
                    r = subprocess.run([exe, ogv, "1", str(max(2, loops // reps))] + extra, capture_output=True, text=True, timeout=300)
                    line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "{}"
                    d = json.loads(line)
                    if r.returncode != 0 or not d.get("ok"):
                        return {"error": "decode_bench %s: rc %d %s" % (label, r.returncode, (r.stderr or line)[-200:])}
                    best = d["frames_per_s"] if best is None else max(best, d["frames_per_s"])
                out[label] = round(best, 1)
            out["program"] = "examples/decode_bench.c, one stream, %d frames a run, better of two runs" % (len(pkts) * reps * max(2, loops // reps))
            return out
    except Exception as e:   # noqa: BLE001
        return {"error": str(e)[:300]}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="decode", choices=["decode", "enc", "e2e"])
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--size", default="4k", choices=sorted(SIZES))
    ap.add_argument("--content", default="dense", choices=["dense", "smooth", "mixed", "static_bg", "static_1pct", "skip", "zeromv_dc", "intra_dense"])
    ap.add_argument("--form", default="levels", choices=["levels", "dequant16"],
                    help="coefficient form of the timed batch: levels (int8 units + tables, dequantised in the kernel) or dequant16 (the "
                         "vtable slot's int16 products, state.h:365-366)")
    ap.add_argument("--streams-per-gpu", type=int, default=4)
    ap.add_argument("--pool", type=int, default=6, help="distinct inter-frame command streams per stream")
    ap.add_argument("--cpu-frames", type=int, default=160, help="frames of stream 0 the scalar CPU oracle decodes for cpu_baseline (~6 s at 4K)")
    ap.add_argument("--parity-frames", type=int, default=40, help="frames every timed stream is decoded and compared with the oracle before timing")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle comparison of the timed batch (profiling runs)")
    ap.add_argument("--repeats", type=int, default=15, help="minimum number of K-step blocks timed")
    ap.add_argument("--min-time", type=float, default=0.3, help="keep timing blocks until this many seconds have been measured")
    ap.add_argument("--second-content", default="smooth", help="content class of the second keyed entry ('' = none)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-late", action="store_true", help="(kept for old command lines: the CPU baseline is always timed behind the GPU's timed regions now)")
    ap.add_argument("--no-cpu-all-cores", action="store_true", help="skip the one-process-per-core leg of the CPU baseline")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--no-1080p", action="store_true", help="skip the 1080p / 720p keyed entries (single stream, four streams, 16 key-frame intervals side by side)")
    ap.add_argument("--no-form16", action="store_true", help="skip the keyed entry form_dequant16 (the timed batch in the vtable slot's int16 form)")
    ap.add_argument("--no-wide", action="store_true", help="skip the keyed entry wide_tiles (the dense batch with 0 %% / 1 %% / 10 %% of the tiles wide)")
    ap.add_argument("--no-enc", action="store_true", help="skip the keyed entry enc_1080p_444 (BASELINE.json config 5: the encoder's block kernels)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the keyed entry e2e_720p (th_decode_* end to end, one stream, with and without the look-ahead)")
    ap.add_argument("--no-pmc", action="store_true", help="do not re-execute under rocprofv3 --pmc for roofline.traffic (N = 1 only)")
    ap.add_argument("--detail", default="", help="where the detail file goes (default: gpurun_out/bench_detail.json if gpurun_out/ exists, else ./bench_detail.json)")
    ap.add_argument("--all-entries", action="store_true", help="the keyed entries at any --size (they belong to the 4K default line; tests/test_bench_flow.py)")
    ap.add_argument("--pmc-child", default="", help=argparse.SUPPRESS)
    return ap.parse_known_args()


def main_enc(emit=True, subset=False, calls_only=False, pmc=None):
    """BASELINE.json config 5: the encoder's block kernels on one 1920x1088 4:4:4 frame.  emit: print one JSON line per kernel
    (--mode enc); else return the entries (the keyed entry enc_1080p_444 of the default line, subset = fDCT, quantiser, both in
    one pass for one and four frames, the motion-search forms of SAD / SATD incl. the half-pel refinement, the cost maps).  Every
    kernel's output is compared with the oracle before its time counts.  calls_only: every kernel called twice and nothing else
    (the workload of the rocprofv3 --pmc passes); pmc: what those passes counted (run_pmc_passes), for the vector-ALU roofline.

    Two rooflines per entry.  Bytes: the per-unit model of SURVEY section 8(d) where every unit brings its own bytes (fDCT,
    quantiser); the UNIQUE bytes -- inputs once + results -- wherever candidates share their input (every SAD / SATD form: nine
    candidates of a block read the same 6 MB planes, so per-pair bytes are not HBM bytes and a fraction above 1 would mean
    nothing).  Vector ALU: SQ_INSTS_VALU of the entry's kernels / its time, against 1024 SIMDs x 2.4 GHz / 4.3 clocks an
    instruction (VALU_PEAK_GINST): the SATD family is bound by that, not by bytes."""
    import torch
    import theora_amd
    import oracle

    torch.cuda.set_device(0)
    W, H, planes = 1920, 1088, 3
    rng = np.random.default_rng(7)
    # frame f-1 and frame f (f = f-1 shifted by (3,1) + noise), three planes stacked vertically
    prev = rng.integers(0, 256, (H * planes + 16, W + 16)).astype(np.uint8)
    cur = np.roll(prev, (1, 3), (0, 1))
    cur = np.clip(cur.astype(np.int32) + rng.integers(-6, 7, cur.shape), 0, 255).astype(np.uint8)
    stride = prev.shape[1]
    by, bx = np.mgrid[0:H * planes // 8, 0:W // 8]
    base = ((by * 8 + 8) * stride + bx * 8 + 8).reshape(-1).astype(np.int32)      # 97 920 blocks, 8-px margin
    sites = [(0, 0), (-1, -1), (0, -1), (1, -1), (-1, 0), (1, 0), (-1, 1), (0, 1), (1, 1)]
    src_offs = np.tile(base, len(sites))
    ref_offs = np.concatenate([base + dy * stride + dx for dx, dy in sites]).astype(np.int32)
    ref2_offs = (ref_offs + 1).astype(np.int32)
    nblk = base.size
    npairs = src_offs.size
    d_prev, d_cur = torch.from_numpy(prev).cuda(), torch.from_numpy(cur).cuda()
    d_so, d_ro, d_r2 = (torch.from_numpy(a).cuda() for a in (src_offs, ref_offs, ref2_offs))
    resid = rng.integers(-255, 256, (nblk, 64)).astype(np.int16)
    d_res = torch.from_numpy(resid).cuda()
    frame_bytes = 2 * nblk * 64          # both pictures once

    from theora_amd import _lib
    L = _lib.load()
    timing_modes = set()

    def timed(fn, reps=50):
        """Average time of one call when the calls run back to back on one stream: kernel time, not launch-and-wait time.  The
        calls are captured ONCE into a HIP graph (thip_set_batch_stream(capture stream, synchronous=0): the library only launches
        kernels) and the graph is replayed between two events -- these kernels take 8-40 us and the Python wrapper around the C call
        (output tensors, argument marshalling) takes about as long, so a plain loop times the interpreter on a slow host.  A call
        that cannot be captured falls back to the plain loop; "timing" in the entries says which were used."""
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if not os.environ.get("THIP_BENCH_NO_GRAPH"):
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    L.thip_set_batch_stream(torch.cuda.current_stream().cuda_stream, 0)
                    for _ in range(reps):
                        fn()
                L.thip_set_batch_stream(None, 1)
                best = None
                for _ in range(4):
                    e0.record()
                    g.replay()
                    e1.record()
                    torch.cuda.synchronize()
                    t = e0.elapsed_time(e1) * 1e-3 / reps
                    best = t if best is None else min(best, t)
                del g
                timing_modes.add("hipGraph replay of %d calls, best of 4" % reps)
                return best
            except Exception as ex:   # noqa: BLE001 -- capture refused: time the plain loop instead
                L.thip_set_batch_stream(None, 1)
                torch.cuda.synchronize()
                sys.stderr.write("bench enc: graph capture failed (%s), timing the plain loop\n" % (str(ex).splitlines() or [""])[0][:200])
        s = torch.cuda.current_stream()
        L.thip_set_batch_stream(s.cuda_stream, 0)
        e0.record(s)
        for _ in range(reps):
            fn()
        e1.record(s)
        torch.cuda.synchronize()
        L.thip_set_batch_stream(None, 1)
        timing_modes.add("plain loop of %d calls" % reps)
        return e0.elapsed_time(e1) * 1e-3 / reps

    # Every case: key (the compact line's name), label, the call, what one call processes, its byte models, the kernels it
    # launches (for the counters), and a check that compares the call's output with the oracle and returns the oracle's rate.
    cases = []

    def case(key, label, call, units, unit, bytes_per_unit, kernels, check, unique_bytes=None, frames=1):
        cases.append(dict(key=key, label=label, call=call, units=units, unit=unit, bytes_per_unit=bytes_per_unit, kernels=kernels,
                          check=check, unique_bytes=unique_bytes, frames=frames))

    ncpu_blk = 20000
    memo = {}

    # --- fDCT ---------------------------------------------------------------------------------
    def chk_fdct():
        got = theora_amd.fdct8x8_batch(d_res).cpu().numpy().reshape(-1, 64)
        t0 = time.perf_counter()
        want = oracle.fdct8x8_batch(resid[:ncpu_blk])
        tc = time.perf_counter() - t0
        assert np.array_equal(got[:ncpu_blk], want)
        memo["fdct_out"], memo["tc_fdct"] = got, tc
        return ncpu_blk / tc
    case("fdct", "oc_enc_fdct8x8", lambda: theora_amd.fdct8x8_batch(d_res), nblk, "blocks", 256, ["k_enc_fdct4", "k_enc_fdct"], chk_fdct)
    # --- quantiser (enquant.c:219) on the fDCT output ------------------------------------------
    d_dct = theora_amd.fdct8x8_batch(d_res)
    dq = np.clip(np.arange(64) * 3 + 16, 8, 4096).astype(np.uint16)
    d_dq = torch.from_numpy(dq).cuda()

    def chk_quant():
        gq, gnz = theora_amd.enc_quantize_batch(d_dct, d_dq)
        t0 = time.perf_counter()
        wq, wnz = oracle.quantize_batch(memo["fdct_out"][:ncpu_blk], dq)
        tc = time.perf_counter() - t0
        assert np.array_equal(gq.cpu().numpy().reshape(-1, 64)[:ncpu_blk], wq) and np.array_equal(gnz.cpu().numpy()[:ncpu_blk], wnz)
        memo["wq"], memo["wnz"], memo["tc_quant"] = wq, wnz, tc
        return ncpu_blk / tc
    case("quantize", "oc_enc_quantize", lambda: theora_amd.enc_quantize_batch(d_dct, d_dq), nblk, "blocks", 260, ["k_enc_quantize"], chk_quant)
    # --- both in one pass (thip_enc_fdct_quantize_batch), one frame and four frames of residuals per call ------------------------
    for F in (1, 4):
        resF = np.tile(resid, (F, 1)) if F > 1 else resid
        d_resF = torch.from_numpy(np.ascontiguousarray(resF)).cuda()

        def chk_fq(d_resF=d_resF):
            fq, fnz = theora_amd.enc_fdct_quantize_batch(d_resF, d_dq)
            assert np.array_equal(fq.cpu().numpy().reshape(-1, 64)[:ncpu_blk], memo["wq"]) and np.array_equal(fnz.cpu().numpy()[:ncpu_blk], memo["wnz"])
            return ncpu_blk / (memo["tc_fdct"] + memo["tc_quant"])      # (the oracle's two calls)
        case("fdct_quantize" + ("_x%d" % F if F > 1 else ""),
             "oc_enc_fdct8x8 + oc_enc_quantize in one pass (thip_enc_fdct_quantize_batch)%s" % (", %d frames per call" % F if F > 1 else ""),
             lambda d_resF=d_resF: theora_amd.enc_fdct_quantize_batch(d_resF, d_dq), nblk * F, "blocks", 260,
             ["k_enc_fdct_quantize4", "k_enc_fdct_quantize"], chk_fq, frames=F)
    # --- SAD / SATD / SATD2 as lists of (block, candidate) pairs: the single slot, batched -------------------------------------------
    for op, bpu in (() if subset else (("sad", 132), ("satd", 136), ("satd2", 136 + 64), ("intra_satd", 72))):
        def call_pairs(op=op):
            return theora_amd.enc_metric_batch(op, d_cur, d_prev, stride, d_so, d_ro, d_r2, 0)

        def chk_pairs(op=op, call=call_pairs):
            v, dc = call()
            ncpu = 30000
            t0 = time.perf_counter()
            wv, wdc = oracle.enc_metric_batch(op, cur, prev, stride, src_offs[:ncpu], ref_offs[:ncpu], ref2_offs[:ncpu], 0)
            tc = time.perf_counter() - t0
            assert np.array_equal(v.cpu().numpy()[:ncpu].view(np.uint32), wv)
            return ncpu / tc
        # unique bytes: the pictures once (the source only for intra_satd), the offset lists the call reads, the results
        uniq = (frame_bytes if op != "intra_satd" else frame_bytes // 2) + npairs * 4 * (1 if op == "intra_satd" else (3 if op == "satd2" else 2)) + npairs * 8
        case("pairs_" + op, "oc_enc_frag_" + op + ", list of pairs (thip_enc_frag_metric_batch)", call_pairs, npairs, "(block,candidate)", bpu,
             ["k_enc_metric<%d>" % _lib.ENC_OPS[op]], chk_pairs, unique_bytes=uniq)
    # --- the same candidates through the motion-search form (one reference position + the 9 sites per block) -----
    d_base = torch.from_numpy(base).cuda()
    for op, bpu in (("sad", 132), ("satd", 136)):
        def call_sites(op=op):
            return theora_amd.enc_metric_sites_batch(op, d_cur, d_prev, stride, d_base, d_base, sites)

        def chk_sites(op=op, call=call_sites):
            v, dc = call()
            want_v, _ = theora_amd.enc_metric_batch(op, d_cur, d_prev, stride, d_so, d_ro, d_r2, 0)
            assert torch.equal(v.reshape(-1), want_v)      # candidate-major = the order of the pair list, which is compared with the oracle:
            ncpu = 30000
            t0 = time.perf_counter()
            wv, _ = oracle.enc_metric_batch(op, cur, prev, stride, src_offs[:ncpu], ref_offs[:ncpu], ref2_offs[:ncpu], 0)
            tc = time.perf_counter() - t0
            assert np.array_equal(want_v.cpu().numpy()[:ncpu].view(np.uint32), wv)
            return ncpu / tc
        case(op + "_search", "oc_enc_frag_%s, motion-search form (thip_enc_frag_metric_sites_batch)" % op, call_sites, npairs, "(block,candidate)", bpu,
             (["k_enc_sites_satd"] if op == "satd" else []) + ["k_enc_sites<%d>" % _lib.ENC_OPS[op]], chk_sites, unique_bytes=frame_bytes + nblk * 8 + npairs * (4 if op == "sad" else 8))
    # --- the half-pel refinement around each block's whole-pel vector: eight sites (thip_enc_frag_metric_halfpel_batch; what the
    #     reference does with eight oc_enc_frag_satd2 / oc_enc_frag_sad2_thresh calls per block, mcenc.c:551-657) -----------------------
    hp_sites = [(-1, -1), (0, -1), (1, -1), (-1, 0), (1, 0), (-1, 1), (0, 1), (1, 1)]
    vecs = ((rng.integers(-2, 3, nblk) & 0xFF) | (rng.integers(-2, 3, nblk) << 8)).astype(np.int16)
    d_vecs = torch.from_numpy(vecs).cuda()
    for op, bpu in (("satd2", 136 + 64), ("sad2_thresh", 132 + 64)):
        def call_hp(op=op):
            return theora_amd.enc_metric_halfpel_batch(op, d_cur, d_prev, stride, d_base, d_base, d_vecs, hp_sites)

        def chk_hp(op=op, call=call_hp):
            v, dc = call()
            ncpu = 6000
            sel = np.random.default_rng(11).integers(0, nblk, ncpu)
            t0 = time.perf_counter()
            wv, wdc = oracle.enc_halfpel_sites(op, cur, prev, stride, base[sel], base[sel], vecs[sel], hp_sites)
            tc = time.perf_counter() - t0
            assert np.array_equal(wv, v.cpu().numpy().view(np.uint32)[:, sel])
            if dc is not None:
                assert np.array_equal(wdc, dc.cpu().numpy()[:, sel])
            return ncpu * len(hp_sites) / tc
        case(op.replace("_thresh", "") + "_halfpel", "oc_enc_frag_%s, half-pel refinement form (thip_enc_frag_metric_halfpel_batch)" % op, call_hp,
             nblk * len(hp_sites), "(block,candidate)", bpu, ["k_enc_halfpel<%d, 2>" % _lib.ENC_OPS[op], "k_enc_halfpel<%d, 3>" % _lib.ENC_OPS[op]], chk_hp,
             unique_bytes=frame_bytes + nblk * 10 + nblk * len(hp_sites) * (4 if op != "satd2" else 8))
    # --- ... and over several frames in one call (a single frame is one round of waves: its launch ramp, first loads and tail
    #     are a third of the call; an encoder with more than one stream to search hands them over together) -------------------
    for F in ((4,) if subset else (2, 4)):
        prevF = rng.integers(0, 256, (H * planes * F + 16, W + 16)).astype(np.uint8)
        curF = np.clip(np.roll(prevF, (1, 3), (0, 1)).astype(np.int32) + rng.integers(-6, 7, prevF.shape), 0, 255).astype(np.uint8)
        byF, bxF = np.mgrid[0:H * planes * F // 8, 0:W // 8]
        baseF = ((byF * 8 + 8) * stride + bxF * 8 + 8).reshape(-1).astype(np.int32)
        d_prevF, d_curF, d_baseF = torch.from_numpy(prevF).cuda(), torch.from_numpy(curF).cuda(), torch.from_numpy(baseF).cuda()
        for op, bpu in (("sad", 132), ("satd", 136)):
            def call_sitesF(op=op, d_curF=d_curF, d_prevF=d_prevF, d_baseF=d_baseF):
                return theora_amd.enc_metric_sites_batch(op, d_curF, d_prevF, stride, d_baseF, d_baseF, sites)

            def chk_sitesF(op=op, call=call_sitesF, curF=curF, prevF=prevF, baseF=baseF):
                v, dc = call()
                ncpu = 20000
                sel = np.random.default_rng(12).integers(0, baseF.size, ncpu)
                tc = 0.0
                for si, (dx, dy) in enumerate(sites[:3]):
                    t0 = time.perf_counter()
                    wv, _ = oracle.enc_metric_batch(op, curF, prevF, stride, baseF[sel], (baseF[sel] + dy * stride + dx).astype(np.int32), baseF[sel], 0)
                    tc += time.perf_counter() - t0
                    assert np.array_equal(v.reshape(len(sites), -1)[si].cpu().numpy()[sel].view(np.uint32), wv)
                return 3 * ncpu / tc
            case("%s_search_x%d" % (op, F), "oc_enc_frag_%s, motion-search form, %d frames per call" % (op, F), call_sitesF, baseF.size * len(sites),
                 "(block,candidate)", bpu, (["k_enc_sites_satd"] if op == "satd" else []) + ["k_enc_sites<%d>" % _lib.ENC_OPS[op]], chk_sitesF,
                 unique_bytes=2 * baseF.size * 64 + baseF.size * 8 + baseF.size * len(sites) * (4 if op == "sad" else 8), frames=F)
        if F == 4:
            vecsF = ((rng.integers(-2, 3, baseF.size) & 0xFF) | (rng.integers(-2, 3, baseF.size) << 8)).astype(np.int16)
            d_vecsF = torch.from_numpy(vecsF).cuda()
            for op, bpu in (("satd2", 136 + 64), ("sad2_thresh", 132 + 64)):
                def call_hpF(op=op, d_curF=d_curF, d_prevF=d_prevF, d_baseF=d_baseF, d_vecsF=d_vecsF):
                    return theora_amd.enc_metric_halfpel_batch(op, d_curF, d_prevF, stride, d_baseF, d_baseF, d_vecsF, hp_sites)

                def chk_hpF(op=op, call=call_hpF, curF=curF, prevF=prevF, baseF=baseF, vecsF=vecsF):
                    v, dc = call()
                    ncpu = 6000
                    sel = np.random.default_rng(13).integers(0, baseF.size, ncpu)
                    t0 = time.perf_counter()
                    wv, wdc = oracle.enc_halfpel_sites(op, curF, prevF, stride, baseF[sel], baseF[sel], vecsF[sel], hp_sites)
                    tc = time.perf_counter() - t0
                    assert np.array_equal(wv, v.cpu().numpy().view(np.uint32)[:, sel])
                    if dc is not None:
                        assert np.array_equal(wdc, dc.cpu().numpy()[:, sel])
                    return ncpu * len(hp_sites) / tc
                case("%s_halfpel_x%d" % (op.replace("_thresh", ""), F), "oc_enc_frag_%s, half-pel refinement form, %d frames per call" % (op, F), call_hpF,
                     baseF.size * len(hp_sites), "(block,candidate)", bpu, ["k_enc_halfpel<%d, 2>" % _lib.ENC_OPS[op], "k_enc_halfpel<%d, 3>" % _lib.ENC_OPS[op]], chk_hpF,
                     unique_bytes=2 * baseF.size * 64 + baseF.size * 10 + baseF.size * len(hp_sites) * (4 if op != "satd2" else 8), frames=F)
    # --- the per-macro-block cost maps of a whole frame (thip_enc_mb_cost_maps: oc_mb_intra_satd, oc_mb_activity, _fast) ---------
    Wc, Hc = 1920, 1088
    cplanes = [rng.integers(0, 256, (Hc, Wc)).astype(np.uint8) for _ in range(3)]
    yy, xx = np.mgrid[0:Hc, 0:Wc]
    cplanes[0] = np.where(((xx // 16 + yy // 16) % 3) == 0, 90, np.where(((xx // 16 + yy // 16) % 3) == 1, cplanes[0],
                          np.where((xx + yy) % 16 < 8, 20, 230))).astype(np.uint8)       # flat / texture / edge macro blocks by turns
    d_cpl = [torch.from_numpy(p).cuda() for p in cplanes]
    nmb = (Wc // 16) * (Hc // 16)

    def call_maps():
        return theora_amd.enc_mb_cost_maps(d_cpl, Wc, Hc, 3)

    def chk_maps():
        got_maps = [g.cpu().numpy().view(np.uint32) for g in call_maps()]
        t0 = time.perf_counter()
        want_maps = oracle.mb_cost_maps(cplanes, Wc, Hc, 3)
        tc = time.perf_counter() - t0
        assert all(np.array_equal(g, wv) for g, wv in zip(got_maps, want_maps))
        return nmb / tc
    case("cost_maps", "oc_mb_intra_satd + oc_mb_activity + oc_mb_activity_fast, whole frame (thip_enc_mb_cost_maps)", call_maps, nmb, "macro blocks",
         12 * 64 + 21 * 4, ["k_enc_cost_maps"], chk_maps, unique_bytes=3 * Wc * Hc + nmb * 21 * 4)

    if calls_only:
        for c in cases:
            c["call"]()
            c["call"]()
        torch.cuda.synchronize()
        return []
    lines = []
    for c in cases:
        t = timed(c["call"])
        cpu_rate = c["check"]()
        # bytes moved: see the docstring
        nbytes = c["unique_bytes"] if c["unique_bytes"] is not None else c["units"] * c["bytes_per_unit"]
        gbs = nbytes / t / 1e9
        line = {
            "key": c["key"], "metric": c["label"] + " throughput", "value": round(c["units"] / t / 1e6, 1), "unit": "M%s/s" % c["unit"],
            "config": {"workload": "1920x1088 4:4:4%s, %d %s per call, 9-site square pattern" % (" x %d frames" % c["frames"] if c["frames"] > 1 else "", c["units"], c["unit"])},
            "ms_per_call": round(1e3 * t, 4), "timing": "; ".join(sorted(timing_modes)),
            "dtype": "u8/i16", "data": "synthetic", "bit_exact_vs_oracle": True,
            "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(gbs / HBM_PEAK_GBS, 4), "alg_bytes_per_unit": c["bytes_per_unit"],
                         "bytes_per_call": int(nbytes), "byte_model": "unique bytes (inputs once + results)" if c["unique_bytes"] is not None
                         else "per-unit bytes of SURVEY section 8(d)"},
            "cpu_baseline": {"value": round(cpu_rate / 1e6, 3), "unit": "M%s/s" % c["unit"], "cores": 1, "kind": "port"}}
        # vector-ALU roofline from the counter pass (pmc: {kernel name incl. template arguments: {work items per dispatch: counters}}):
        # instructions per wave x the call's waves / its time, against the chip's issue rate.  The same kernel serves the one-frame
        # and the four-frame call: the smaller grid is the former's, the larger the latter's.
        for kn in c["kernels"]:
            grids = sorted(g for g, cv in (pmc or {}).get(kn, {}).items() if cv.get("SQ_INSTS_VALU") and cv.get("SQ_WAVES") and g > 0)
            if not grids:
                continue
            g = grids[-1] if c["frames"] > 1 else grids[0]
            cv = pmc[kn][g]
            per_wave = cv["SQ_INSTS_VALU"] / cv["SQ_WAVES"]
            insts = per_wave * g / 64.0
            line["valu"] = {"kernel": kn, "insts_per_wave": round(per_wave, 1), "waves": int(g // 64), "peak_Ginst_s": round(VALU_PEAK_GINST, 1),
                            "achieved_Ginst_s": round(insts / t / 1e9, 1), "frac": round(insts / t / 1e9 / VALU_PEAK_GINST, 4)}
            break
        lines.append(line)
    if emit:
        for ln in lines:
            print(json.dumps(ln))
    return lines


def main_e2e(main_args, argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=12)
    ap.add_argument("--loops", type=int, default=5)
    ap.add_argument("--no-output", action="store_true", help="skip th_decode_ycbcr_out (no D2H)")
    ap.add_argument("--threads", type=int, default=1, help="host threads, one independent stream each")
    ap.add_argument("--e2e-size", default="", choices=["", "720p", "1080p", "4k", "cif", "qcif"], help="picture size of the generated packets")
    ap.add_argument("--no-native", action="store_true", help="skip the dump_video_hip / decode_bench legs")
    ap.add_argument("--trees", choices=["matched", "random"], default="matched")
    ap.add_argument("--lookahead", type=int, default=0,
                    help="announce every packet this many packets ahead of its th_decode_packetin (TH_DECCTL_THIP_PREFETCH_PACKET: "
                         "parsed on the library's own threads, up to option fe_lookahead = 4 at a time); 0: the plain API loop")
    ap.add_argument("--packets", choices=["dense", "typical"], default="dense",
                    help="dense: 70 %% of the super blocks coded, 30 %% of the coded blocks with AC coefficients "
                         "(~80 KB per 720p frame); typical: 35 %% / 15 %% (closer to SURVEY section 6's statistics, ~30 KB)")
    args = ap.parse_args(argv)
    # --size as given when it is given; the mode's default is 720p (bench.py's own default, 4K, costs the Python packet
    # generator about 15 s of set-up and 5 s a frame: ask for it with --e2e-size 4k)
    args.size = args.e2e_size or (main_args.size if main_args.size != "4k" else "720p")
    if args.size == "4k":
        args.frames = min(args.frames, 6)
    elif args.size == "1080p":
        args.frames = min(args.frames, 8)
    import torch
    from tests import streamgen
    from theora_amd.decoder import Decoder
    torch.cuda.set_device(0)
    w, h = SIZES[args.size]
    # Huffman trees built from the content's own token statistics, as an encoder's are (--trees random:
    # trees unrelated to the content, a fifth of the tokens with codes of 10+ bits)
    content = dict(density=0.7, p_dc_only=0.5, p_empty=0.2) if args.packets == "dense" else \
        dict(density=0.35, p_dc_only=0.45, p_empty=0.4)
    st = streamgen.Stream(w, h, 0, seed=99, trees=args.trees, probe_kwargs=content)
    hdr = st.header_packets()
    pkts = []
    for f in range(args.frames):
        pkt, truth = st.frame(0 if f % 8 == 0 else 1, **content)
        pkts.append(pkt)
    nbytes = sum(len(p) for p in pkts)
    import threading
    T = max(1, args.threads)
    decs = [Decoder(hdr) for _ in range(T)]      # one decoder context (one stream) per host thread
    for dec in decs:
        for p in pkts:                  # warm-up pass
            dec.packetin(p)
            dec.ycbcr_out()
    counts = [0] * T

    def worker(i):
        dec = decs[i]
        seq = pkts * args.loops
        nxt = 0
        for k, p in enumerate(seq):
            while args.lookahead and nxt < len(seq) and nxt < k + args.lookahead:
                nxt = max(nxt, k)
                if not dec.prefetch(seq[nxt]) and len(seq[nxt]):
                    break                    # no slot free
                nxt += 1
            dec.packetin(p)
            if not args.no_output:
                dec.ycbcr_out()
            counts[i] += 1
        if args.no_output:
            dec.ycbcr_out()

    t0 = time.perf_counter()
    if T == 1:
        worker(0)
    else:
        ths = [threading.Thread(target=worker, args=(i,)) for i in range(T)]
        for th in ths:
            th.start()
        for th in ths:
            th.join()
    el = time.perf_counter() - t0
    n = sum(counts)
    print(json.dumps({"metric": "end-to-end decode frames/sec (%s 4:2:0, packets in host memory -> YUV in host memory)" % args.size,
                      "value": round(n / el, 2), "unit": "frames/s", "frames": n, "host_threads": T, "streams": T,
                      "avg_packet_bytes": nbytes // len(pkts), "with_ycbcr_out": not args.no_output, "lookahead": args.lookahead,
                      "data": "synthetic packets (tests/streamgen.py), %s content, %s Huffman trees" % (args.packets, args.trees),
                      "note": "host-bound: one entropy-decode thread per stream + PCIe; th_decode_* contexts are independent"
                              if not args.lookahead else
                              "packets announced ahead: the entropy decoder runs on the library's parser threads (up to 4 per stream), "
                              "the caller's thread hands frames to the device"}))
    for dec in decs:
        dec.close()
    # the same packets in an Ogg file through the C program (examples/dump_video_hip.c): no Python between the calls
    exe = os.path.join(ROOT, "examples", "dump_video_hip")
    if os.path.exists(exe) and not args.no_native:
        import subprocess
        import tempfile
        from tests import oggmux
        ls = oggmux.LogicalStream(0x7E0)
        for k, hp in enumerate(hdr):
            ls.add_packet(hp, granulepos=0, flush=(k == 0 or k == len(hdr) - 1))
        for rep in range(4 * args.loops):
            for k, pk in enumerate(pkts):
                ls.add_packet(pk, granulepos=rep * len(pkts) + k + 1)
        with tempfile.TemporaryDirectory() as td:
            ogv = os.path.join(td, "clip.ogv")
            with open(ogv, "wb") as f:
                f.write(b"".join(ls.finish()))
            for label, extra in (("decode only", ["--fps-only"]), ("with YUV4MPEG2 output", ["-o", "/dev/null"])):
                r = subprocess.run([exe] + extra + [ogv], capture_output=True, text=True, timeout=600)
                print(json.dumps({"metric": "dump_video_hip on an Ogg file (%s 4:2:0, %s)" % (args.size, label),
                                  "stderr": r.stderr.strip().splitlines()[-1] if r.stderr.strip() else "", "rc": r.returncode}))
            # many streams, one native host thread each (examples/decode_bench.c)
            nb = os.path.join(ROOT, "examples", "decode_bench")
            if os.path.exists(nb):
                for nt, la in [(n, 0) for n in (1, 2, 4, 8, 16, 32, 64)] + [(1, 4), (1, 8), (4, 4), (4, 8)]:
                    r = subprocess.run([nb, ogv, str(nt), "2"] + (["--lookahead", str(la)] if la else []),
                                       capture_output=True, text=True, timeout=900)
                    line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "{}"
                    try:
                        d = json.loads(line)
                    except ValueError:
                        d = {"raw": line}
                    d["metric"] = "decode_bench: concurrent %s streams, native threads, packets -> host YUV" % args.size
                    d["rc"] = r.returncode
                    print(json.dumps(d))



class Wall:
    """Wall-clock seconds by section of the run (detail file: where a bench run's minutes go)."""

    def __init__(self):
        self.t0 = time.perf_counter()
        self.last = self.t0
        self.sections = {}

    def mark(self, name):
        now = time.perf_counter()
        self.sections[name] = round(self.sections.get(name, 0.0) + now - self.last, 2)
        self.last = now

    def total(self):
        return round(time.perf_counter() - self.t0, 1)


def main():
    args, rest = parse_args()
    if args.pmc_child:
        return pmc_child(args.pmc_child)
    if args.mode == "enc":
        return main_enc()
    if args.mode == "e2e":
        return main_e2e(args, rest)
    wall = Wall()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world
    from theora_amd import shard       # (torch only: no HIP yet)

    S = args.streams_per_gpu
    nparity = 0 if args.no_parity else max(1, args.parity_frames)
    main_4k = args.size == "4k" or args.all_entries
    do_second = bool(args.second_content) and args.second_content != args.content
    do_1080p = main_4k and not args.no_1080p
    do_form16 = main_4k and args.form == "levels" and not args.no_form16
    do_wide = main_4k and args.content == "dense" and args.form == "levels" and not args.no_wide and world == 1
    do_enc = rank == 0 and world == 1 and main_4k and not args.no_enc
    do_e2e = rank == 0 and world == 1 and main_4k and not args.no_e2e
    do_cpu = rank == 0 and not args.no_cpu_baseline and nparity > 0
    # At every world size rank 0's CPU baseline (tens of seconds, one process per core) runs BEHIND the timed regions, so that no
    # rank's clock starts after a long wait in a collective and nothing competes with the generation workers
    cpu_late = True

    # ---- worker processes: every command stream of the run is asked for NOW ----------------------------------------------
    import multiprocessing as mp
    import tempfile
    tmp = tempfile.TemporaryDirectory(prefix="thip_bench_", dir=os.environ.get("TMPDIR", "/tmp"))
    ncores = _usable_cores()
    nworkers = max(2, min(24, ncores // max(1, world)))
    workers = mp.get_context("spawn").Pool(nworkers)
    gids = shard.stream_ids(rank, world, S)
    jobs = {}

    def ask(tag, **kw):
        job = dict(dict(tag=tag, dir=tmp.name, pool=args.pool, forms=("levels",)), **kw)
        jobs[tag] = workers.apply_async(_gen_job, (job,))

    for s, gid in enumerate(gids):
        forms = (args.form,) + (("dequant16",) if do_form16 else ())
        ask("main%d" % s, size=args.size, content=args.content, seed=shard.stream_seed(12345, gid), forms=forms,
            parity={"frames": nparity} if nparity else None, host=(do_cpu and s == 0))
    if do_1080p:
        # (1080p: seed base 777, stream gid; the single stream is rank's stream `rank`, the four streams 4 rank .. 4 rank + 3)
        for gid in sorted(set(shard.stream_ids(rank, world, 1) + shard.stream_ids(rank, world, 4))):
            ask("p1080_%d" % gid, size="1080p", content=args.content, seed=shard.stream_seed(777, gid))
        ask("p720", size="720p", content=args.content, seed=shard.stream_seed(779, rank))
    # (the second class's command streams are a third of the dense ones: twice the pool, so that its working set exceeds the Infinity Cache too)
    pool2 = max(args.pool, 12) if main_4k else args.pool
    if do_second:
        for s, gid in enumerate(gids):
            ask("second%d" % s, size=args.size, content=args.second_content, seed=shard.stream_seed(12345, gid), pool=pool2)
    G3 = 16
    if do_1080p:
        # config 3's roofline form: 16 key-frame intervals of ONE 1080p stream side by side -- 16 intervals of a real stream are 16
        # different sets of coefficients, so every state gets its OWN pool (one shared 24.7 MB pool stayed in cache)
        for g in range(G3):
            ask("gop%d" % g, size="1080p", content=args.content, seed=shard.stream_seed(778, G3 * rank + g),
                parity={"frames": 6, "g": g, "G": G3} if (nparity and g in (0, G3 - 1)) else None)
    if do_wide:
        for label, frac in (("1pct", 0.01), ("10pct", 0.10)):
            for s, gid in enumerate(gids):
                ask("wide_%s_%d" % (label, s), size=args.size, content=args.content, seed=shard.stream_seed(12345, gid),
                    widen=(frac, 4242 + 17 * s), parity={"frames": 3} if (nparity and s == 0) else None)
    e2e_prep = workers.apply_async(_e2e_prep_job, (dict(size="720p", nframes=8),)) if do_e2e else None
    wall.mark("start_workers")

    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    # under torch.distributed.run (RANK set) the process group is created at any world size, one included:
    # barriers, the MAX of the block times and the checksum gather then really go through RCCL
    under_launcher = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if world > 1 or under_launcher:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    import theora_amd
    from theora_amd import synth
    dev = torch.device("cuda", local_rank)
    theora_amd.version()
    wall.mark("import_torch_and_library")

    def sync():
        theora_amd.synchronize()
        torch.cuda.synchronize()

    def barrier():
        shard.barrier(world)

    keep = []

    def upload(paths):
        row = []
        for p in paths:
            d, ka = synth.upload_frame(_npz_load(p))
            keep.append(ka)
            row.append(d)
        return row

    def timed_blocks(submit, K, first, reps):
        """`reps` blocks of K steps, each bracketed by synchronize + barrier on both sides; the block times (MAX over ranks)."""
        b = []
        for rep in range(reps):
            sync()
            barrier()
            t0 = time.perf_counter()
            for i in range(first + rep * K, first + (rep + 1) * K):
                submit(i)
            sync()
            b.append(time.perf_counter() - t0)
            barrier()
        return shard.reduce_max(b, dev)

    def planes_equal(st, path):
        want = _npz_load(path)
        bad = []
        c = 0
        for pli in range(3):
            got = st.read_plane(st.ref_idx(theora_amd.FRAME_PREV), pli)
            c = zlib.crc32(got.tobytes(), c)
            if not np.array_equal(want["p%d" % pli], got):
                bad.append((pli, int((want["p%d" % pli] != got).sum())))
        return bad, c

    def ws_mb(desc_bytes, states_):
        """Working set of a pool-cycled entry: every descriptor of the pool + the states' three frames each, in MB."""
        return round((desc_bytes + sum(3 * st.frame_bytes for st in states_)) / 1e6, 1)

    reps_small = max(5, min(args.repeats, 15))
    entries, detail = {}, {"entries": {}}
    w, h = SIZES[args.size]
    geom = synth.Geometry(w, h)

    # ---- the timed batch: S streams of this rank, one thip_decode_frames call per step ----------------------------------------
    main_res = [jobs["main%d" % s].get() for s in range(S)]
    wall.mark("wait_for_main_streams")
    descs = [upload(r["files"][args.form]) for r in main_res]
    balg = [r["balg"] for r in main_res]
    states = [theora_amd.State(w, h) for _ in range(S)]
    plans = [theora_amd.BatchPlan(states, [descs[s][j] for s in range(S)]) for j in range(args.pool + 1)]

    def frame_of_step(i):
        return seq_frame(i, 0, 1, args.pool)

    def run(nsteps, first=0, stream=None, plans_=None):
        for i in range(first, first + nsteps):
            (plans_ or plans)[frame_of_step(i)].submit(stream)

    def alg_of_step(i, which, balg_=None):
        return sum((balg_ or balg)[s][frame_of_step(i)][which] for s in range(S))

    # Parity of the TIMED batch: the states, plans and launch shape that are timed below first decode the first `parity_frames`
    # frames of the sequence; every plane of every stream is then compared with the oracle's decoding of the same frames (done by
    # the worker that generated the stream; a stream's content depends on its global id only, so streams [0, S) are the same
    # pictures at every world size).  No number is printed on a mismatch.
    parity, parity_crcs = None, []
    if nparity:
        run(nparity)
        sync()
        ok, bad = True, []
        for s in range(S):
            b, c = planes_equal(states[s], main_res[s]["oracle_planes"])
            parity_crcs.append(c)
            if b:
                ok = False
                bad.append((gids[s], b))
        if not shard.reduce_min(1 if ok else 0, dev):
            raise SystemExit("bench: GPU output differs from the oracle %s -- refusing to report a number" % bad[:4])
        parity = {"frames": nparity, "bit_exact": True,
                  "checked": "the timed batch itself: all %d streams of every rank, decoded %d frames deep by the timed states in the timed launch "
                             "shape (one thip_decode_frames call per step), every plane of every stream against the oracle" % (S, nparity)}
    wall.mark("upload_and_parity")

    # ---- timed region ---------------------------------------------------------------------
    # Pass A: blocks of exactly K steps, each bracketed by synchronize + barrier on both sides, no
    # instrumentation inside.  One block at the driver's --steps 20 is 0.8 ms, too short to be a stable
    # number, so the block is repeated (at least `repeats` times and until ~0.3 s of GPU time has been
    # measured); `value` comes from the MEDIAN block, min and max are reported beside it.
    step0 = nparity
    run(args.warmup, first=step0)
    step0 += args.warmup
    sync()
    elapsed_blocks, submit_times, total_t = [], [], 0.0
    while len(elapsed_blocks) < args.repeats or (total_t < args.min_time and len(elapsed_blocks) < 4096):
        sync()
        barrier()
        t0 = time.perf_counter()
        run(args.steps, first=step0)
        t_sub = time.perf_counter() - t0   # (diagnostic: the host's part, all K steps submitted)
        sync()
        dt = time.perf_counter() - t0      # this rank's K steps, start aligned by the barrier; the MAX over ranks is taken below
        submit_times.append(t_sub)
        barrier()                          # (outside the clock: a trailing collective would add its own latency to every 0.8 ms block)
        elapsed_blocks.append(dt)
        total_t += dt
        step0 += args.steps
        if dist.is_initialized():   # every rank must run the same number of blocks
            total_t = shard.reduce_max([total_t], dev)[0]
    # (diagnostic: what the bracket itself costs -- sync() on an idle device, part of every block's time)
    idle_sync = []
    for _ in range(20):
        t0 = time.perf_counter()
        sync()
        idle_sync.append(time.perf_counter() - t0)
    idle_sync_us = 1e6 * sorted(idle_sync)[len(idle_sync) // 2]
    submit_us = 1e6 * sorted(submit_times)[len(submit_times) // 2]
    # Pass B: steps with every kernel bracketed by HIP events on the stream it runs on -> per-kernel durations for the
    # roofline.  Kept out of pass A because four event records per step cost ~15 % of the step.  Two shapes:
    #   (1) ONE stream, one launch per step carrying all S streams: the kernel has the chip to itself -- the roofline of the
    #       kernel proper (`roofline`), what `rocprofv3 --kernel-trace --stats` of a THIP_LANES=1 run shows;
    #   (2) the TIMED shape: the library's lanes, S/lanes streams per launch, launches of different lanes overlapping
    #       (detail: a launch takes longer there because it shares the chip with the other lane's).
    profiling = not args.no_profile
    launches, kms, elapsed_b = [0, 0], [0.0, 0.0], 0.0
    launches_t, kms_t = [0, 0], [0.0, 0.0]
    prof_steps = max(args.steps, 256)
    first_prof = step0
    if profiling:
        theora_amd.profile_reset()
        theora_amd.profile_enable(True)
        sync()
        t0 = time.perf_counter()
        pstream = torch.cuda.Stream()
        run(prof_steps, first=step0, stream=pstream.cuda_stream)   # one launch per kernel per step
        pstream.synchronize()
        sync()
        elapsed_b = time.perf_counter() - t0
        launches, kms = theora_amd.profile_read()
        step0 += prof_steps
        theora_amd.profile_reset()
        run(prof_steps, first=step0)                               # the timed shape
        sync()
        theora_amd.profile_enable(False)
        launches_t, kms_t = theora_amd.profile_read()
        step0 += prof_steps
    # checksum of each stream's final frame (all planes): gathered over ranks, printed
    crcs = []
    for st in states:
        c = 0
        for pli in range(3):
            c = zlib.crc32(st.read_plane(st.ref_idx(theora_amd.FRAME_PREV), pli).tobytes(), c)
        crcs.append(c)
    blocks = shard.reduce_max(elapsed_blocks, dev)          # per block: the slowest rank
    per_rank_ms = [round(1e3 * v / args.steps, 5) for v in shard.gather_floats(float(np.median(elapsed_blocks)), dev)]   # every rank's own median
    pg = {"initialized": bool(dist.is_initialized()), "world_size": dist.get_world_size() if dist.is_initialized() else 1,
          "backend": dist.get_backend() if dist.is_initialized() else None}
    _, crcs = shard.reduce_results(0.0, crcs, dev)
    if nparity:
        _, parity_crcs = shard.reduce_results(0.0, parity_crcs, dev)
    kms = shard.reduce_max(kms, dev)
    kms_t = shard.reduce_max(kms_t, dev)
    elapsed = float(np.median(blocks))
    main_desc_bytes = sum(r["desc_bytes"][args.form] for r in main_res)
    main_ws = ws_mb(main_desc_bytes, states)
    wall.mark("timed_region_and_profile_pass")

    def keyed(name, value, e, K, read_bytes, ws, note=None, **extra):
        """One keyed entry: the compact triple (+ working set) on the line, the rest in the detail file."""
        frac = round(read_bytes / e / 1e9 / HBM_PEAK_GBS, 4) if read_bytes else None
        entries[name] = {"value": round(value, 1), "ms_per_step": round(1e3 * e / K, 5), "frac": frac, "ws_MB": ws, "gt_IC": bool(ws > INFINITY_CACHE_MB)}
        detail["entries"][name] = dict(entries[name], unit="frames/s", steps_per_block=K, frac_is="B_read of SURVEY 8(d) / time / 8 TB/s (HBM-read roofline)",
                                       working_set="every descriptor of the pool + three frames per state; %s the 256 MB Infinity Cache"
                                                   % ("exceeds" if ws > INFINITY_CACHE_MB else "FITS IN (the entry is latency-bound, its frac is not an HBM measurement)"),
                                       **({"note": note} if note else {}), **extra)

    # ---- the same measurement on the content class SURVEY section 8d defines from the reference's own
    #      statistics (66 % coded, 80 % of the coded blocks DC-only): a second keyed entry on the line ------
    if do_second:
        res2 = [jobs["second%d" % s].get() for s in range(S)]
        descs2 = [upload(r["files"]["levels"]) for r in res2]
        balg2 = [r["balg"] for r in res2]
        plans2 = [theora_amd.BatchPlan(states, [descs2[s][j] for s in range(S)]) for j in range(pool2 + 1)]
        K2 = max(args.steps, 64)

        def fos2(i):
            return seq_frame(i, 0, 1, pool2)
        for i in range(KF_INTERVAL):     # starts with a key frame; warm
            plans2[fos2(i)].submit(None)
        sync()
        b2 = timed_blocks(lambda i: plans2[fos2(i)].submit(None), K2, KF_INTERVAL, reps_small)
        e2 = float(np.median(b2))
        read2 = sum(balg2[s][fos2(KF_INTERVAL + i)][1] for i in range(K2) for s in range(S))
        keyed("4k_" + args.second_content, K2 * S * world / e2, e2, K2, read2, ws_mb(sum(r["desc_bytes"]["levels"] for r in res2), states),
              note="same streams, states and launch shape as the headline, content class '%s', a pool of %d inter frames per stream; parity of this class: "
                   "tests/test_gpu_frames.py" % (args.second_content, pool2),
              ms_per_step_min_max=[round(1e3 * min(b2) / K2, 5), round(1e3 * max(b2) / K2, 5)])
        del plans2, descs2
        wall.mark("second_content")

    # ---- the timed batch in the vtable slot's form: dequantised int16 coefficients (state.h:365-366, decode.c:1573-1581), what
    #      INTEGRATION.md's oc_state_accel_init_hip binding feeds; k_recon_lf<false>.  Same pictures: compared with the same oracle planes.
    form16 = None
    if do_form16:
        descs16 = [upload(r["files"]["dequant16"]) for r in main_res]
        plans16 = [theora_amd.BatchPlan(states, [descs16[s][j] for s in range(S)]) for j in range(args.pool + 1)]
        if nparity:
            run(nparity, 0, plans_=plans16)
            sync()
            for s in range(S):
                b, c = planes_equal(states[s], main_res[s]["oracle_planes"])
                if b:
                    raise SystemExit("bench: the int16-form batch differs from the oracle (stream %d: %s)" % (gids[s], b))
        run(KF_INTERVAL, 0, plans_=plans16)
        sync()
        K16 = max(args.steps, 64)
        b16 = timed_blocks(lambda i: plans16[frame_of_step(i)].submit(None), K16, KF_INTERVAL, reps_small)
        e16 = float(np.median(b16))
        read16 = sum(alg_of_step(KF_INTERVAL + i, 1) for i in range(K16))
        form16 = {"desc_bytes": sum(r["desc_bytes"]["dequant16"] for r in main_res)}
        keyed("form_dequant16", K16 * S * world / e16, e16, K16, read16, ws_mb(form16["desc_bytes"], states),
              note="the headline's batch packed as THIP_COEFFS_DEQUANT16 (128-byte int16 slots: what oc_state_frag_recon hands over), k_recon_lf<false>; "
                   "parity: all streams %d frames deep against the oracle" % nparity, bit_exact=bool(nparity),
              ms_per_step_min_max=[round(1e3 * min(b16) / K16, 5), round(1e3 * max(b16) / K16, 5)])
        del plans16
        wall.mark("form_dequant16")

    # ---- wide tiles: the same dense step with 0 % (control), 1 % and 10 % of the tiles holding a level beyond eight bits (the levels
    #      form's escape, include/theora_hip.h THIP_SLOT_WIDE: such a tile's blocks own two units of int16 levels).  All three cycle the
    #      headline's own 7-frame pool (a 3-frame pool read 6 % FASTER than the headline for cache reasons) and are timed in
    #      one loop, block by block in turn. ---------------------------------------------------------------------------------------
    if do_wide:
        try:
            variants = [("0pct", plans, balg, main_desc_bytes, 0, sum(r["tiles"] for r in main_res))]
            for label in ("1pct", "10pct"):
                resw = [jobs["wide_%s_%d" % (label, s)].get() for s in range(S)]
                descsw = [upload(r["files"]["levels"]) for r in resw]
                plansw = [theora_amd.BatchPlan(states, [descsw[s][j] for s in range(S)]) for j in range(args.pool + 1)]
                if nparity:        # stream 0 against the oracle, three frames deep
                    run(3, 0, plans_=plansw)
                    sync()
                    b, c = planes_equal(states[0], resw[0]["oracle_planes"])
                    if b:
                        raise SystemExit("bench: the wide-tile batch differs from the oracle (%s)" % b)
                variants.append((label, plansw, [r["balg"] for r in resw], sum(r["desc_bytes"]["levels"] for r in resw),
                                 sum(r["wide_tiles"] for r in resw), sum(r["tiles"] for r in resw)))
            K4 = max(args.steps, 64)
            times = {v[0]: [] for v in variants}
            for v in variants:
                run(KF_INTERVAL, 0, plans_=v[1])
            sync()
            for rep in range(reps_small):
                for v in variants:
                    times[v[0]] += timed_blocks(lambda i, pl=v[1]: pl[frame_of_step(i)].submit(None), K4, KF_INTERVAL + rep * K4, 1)
            for label, pl, bl, db, nw, nt in variants:
                e4 = float(np.median(times[label]))
                read4 = sum(alg_of_step(KF_INTERVAL + i, 1, bl) for i in range(K4))
                keyed("wide_" + label, K4 * S / e4, e4, K4, read4, ws_mb(db, states), wide_tiles=nw, tiles=nt, bit_exact=bool(nparity),
                      note="the timed dense batch with one level of 300 planted in that share of the tiles (every block of such a tile then travels as "
                           "two units of int16 levels instead of one of int8); 0pct = the headline's own descriptors, timed in the same loop")
            del variants
        except SystemExit:
            raise
        except Exception as e:   # noqa: BLE001 -- never fails the bench
            detail["entries"]["wide_tiles_error"] = str(e)[:300]
        wall.mark("wide_tiles")

    # ---- the other sizes the metric names: 1080p (BASELINE.json config 3: ONE 1080p stream, kf 64 -- launches that leave the
    #      chip half empty -- and four streams in one call) and one 720p stream (config 2), same content class, same clock ---------
    if do_1080p:
        def small(name, size, tags, note):
            w2, h2 = SIZES[size]
            res3 = [jobs[t].get() for t in tags]
            descs3 = [upload(r["files"]["levels"]) for r in res3]
            states3 = [theora_amd.State(w2, h2) for _ in tags]
            plans3 = [theora_amd.BatchPlan(states3, [descs3[s][j] for s in range(len(tags))]) for j in range(args.pool + 1)]
            K3 = max(args.steps, 128)
            for i in range(KF_INTERVAL):
                plans3[frame_of_step(i)].submit(None)
            sync()
            b3 = timed_blocks(lambda i: plans3[frame_of_step(i)].submit(None), K3, KF_INTERVAL, reps_small)
            e3 = float(np.median(b3))
            read3 = sum(res3[s]["balg"][frame_of_step(KF_INTERVAL + i)][1] for i in range(K3) for s in range(len(tags)))
            keyed(name, K3 * len(tags) * world / e3, e3, K3, read3, ws_mb(sum(r["desc_bytes"]["levels"] for r in res3), states3), note=note,
                  streams_per_gpu=len(tags))
            for st3 in states3:
                st3.close()
        small("1080p_single_stream", "1080p", ["p1080_%d" % g for g in shard.stream_ids(rank, world, 1)],
              "BASELINE.json config 3 as written: ONE 1080p (1920x1088 coded) stream, kf 64, frames one after the other; parity of this shape: "
              "tests/test_gpu_frames.py::test_config3_as_written")
        small("1080p_four_streams", "1080p", ["p1080_%d" % g for g in shard.stream_ids(rank, world, 4)], "four 1080p streams in one thip_decode_frames call per step")
        small("720p_single_stream", "720p", ["p720"], "BASELINE.json config 2's shape: one 720p stream (k_recon_lf_sb: one super block per wave)")
        wall.mark("1080p_720p")
        # ---- config 3's roofline form: ONE 1080p stream whose key-frame intervals are decoded side by side (a key frame resets
        #      both references, decode.c:2947-2955, so intervals are independent): 16 states, state g takes intervals g, g + 16, ...,
        #      every state with its own 7-frame pool (16 intervals of a stream are 16 different sets of coefficients) ----------------
        try:
            w2, h2 = SIZES["1080p"]
            resg = [jobs["gop%d" % g].get() for g in range(G3)]
            descsg = [upload(r["files"]["levels"]) for r in resg]
            states3 = [theora_amd.State(w2, h2) for _ in range(G3)]

            def fos3(g, i):
                return seq_frame(i, g, G3, args.pool)
            cache3 = {}

            def plan3(i):
                key = tuple(fos3(g, i) for g in range(G3))
                if key not in cache3:
                    cache3[key] = theora_amd.BatchPlan(states3, [descsg[g][key[g]] for g in range(G3)])
                return cache3[key]
            NP3 = 6     # parity of this shape before its clock: the first six steps, states 0 and 15 against the oracle
            for i in range(NP3):
                plan3(i).submit(None)
            sync()
            if nparity:
                for g in (0, G3 - 1):
                    b, c = planes_equal(states3[g], resg[g]["oracle_planes"])
                    if b:
                        raise SystemExit("bench: single_gop16 differs from the oracle (state %d: %s)" % (g, b))
            K3 = max(args.steps, 64)
            for i in range(NP3, KF_INTERVAL + K3 * reps_small):     # every plan exists before the clock starts
                plan3(i)
            for i in range(NP3, KF_INTERVAL):
                plan3(i).submit(None)
            sync()
            b3 = timed_blocks(lambda i: plan3(i).submit(None), K3, KF_INTERVAL, reps_small)
            e3 = float(np.median(b3))
            read3 = sum(resg[g]["balg"][fos3(g, KF_INTERVAL + i)][1] for i in range(K3) for g in range(G3))
            keyed("1080p_single_gop16", K3 * G3 * world / e3, e3, K3, read3, ws_mb(sum(r["desc_bytes"]["levels"] for r in resg), states3),
                  note="BASELINE.json config 3 in the form that fills the chip: the caller decodes 16 key-frame intervals of the one stream side by side "
                       "(16 states in one thip_decode_frames call per step, each with its OWN pool of 7 command streams: %d distinct descriptors)" % (G3 * (args.pool + 1)),
                  key_frame_intervals_side_by_side=G3, distinct_descriptors=G3 * (args.pool + 1), bit_exact=bool(nparity))
            for st3 in states3:
                st3.close()
            del descsg, cache3
        except SystemExit:
            raise
        except Exception as e:   # noqa: BLE001 -- never fails the bench
            detail["entries"]["1080p_single_gop16_error"] = str(e)[:300]
        wall.mark("1080p_gop16")

    # ---- HBM bytes and vector instructions per launch from the PMC counters of THIS workload: this script again under rocprofv3,
    #      one pass per counter group, reading the command streams the workers wrote ---------------------------------------------------
    pmc = None
    if rank == 0 and world == 1 and not args.no_pmc and profiling:
        manifest = {"size": args.size, "steps": 24, "enc": bool(do_enc),
                    "forms": {args.form: [r["files"][args.form] for r in main_res]}}
        if do_form16:
            manifest["forms"]["dequant16"] = [r["files"]["dequant16"] for r in main_res]
        mpath = os.path.join(tmp.name, "pmc_manifest.json")
        json.dump(manifest, open(mpath, "w"))
        sync()
        pmc = run_pmc_passes(mpath)
        wall.mark("pmc_passes")

    # ---- BASELINE.json config 5: the encoder's block kernels on one 1920x1088 4:4:4 frame (each checked against the oracle first) ----
    if do_enc:
        try:
            enc_lines = main_enc(emit=False, subset=True, pmc=_pmc_by_grid(pmc))
            detail["enc_1080p_444"] = {"entries": enc_lines,
                                       "note": "BASELINE.json config 5 (1920x1088 4:4:4): the SAD / SATD forms' and the cost maps' byte rooflines are on UNIQUE "
                                               "bytes (their candidates share their input); valu = vector instructions counted by rocprofv3 / time against "
                                               "1024 SIMDs x 2.4 GHz / 4.3 clocks; cpu_baseline = the oracle's scalar C on one core, same run"}
            for ln in enc_lines:
                entries["enc_" + ln["key"]] = {"value": ln["value"], "unit": ln["unit"], "us": round(1e3 * ln["ms_per_call"], 2),
                                               "frac": ln["roofline"]["frac"], "valu_frac": (ln.get("valu") or {}).get("frac")}
        except Exception as e:   # noqa: BLE001
            import traceback
            traceback.print_exc()
            detail["enc_1080p_444"] = {"error": repr(e)[:300], "traceback": traceback.format_exc()[-1500:]}
            entries["enc_error"] = repr(e)[:120]
        wall.mark("enc_entry")

    if do_e2e:
        try:
            prep = e2e_prep.get(timeout=900)
            wall.mark("wait_for_e2e_packets")
            e2e = e2e_keyed_entry(prep)
        except Exception as e:   # noqa: BLE001
            e2e = {"error": str(e)[:300]}
        detail["e2e_720p"] = e2e
        if "error" in e2e:
            entries["e2e_720p"] = {"error": e2e["error"][:120]}
        else:
            cc = e2e.get("c_caller") or {}
            entries["e2e_720p"] = {"unit": "frames/s", "py": [e2e.get("plain_loop"), e2e.get("lookahead_8"), e2e.get("lookahead_8_pipelined")],
                                   "c": [cc.get("plain_loop"), cc.get("lookahead_8"), cc.get("lookahead_8_pipelined")],
                                   "legs": "plain loop / 8 packets ahead, fe_pipeline off / 8 ahead, fe_pipeline on (the default)", "bit_exact": True}
        wall.mark("e2e_entry")

    # ---- CPU baseline, behind everything the GPU did: the oracle on the box's cores (scalar, its SSE2 build, every core) -----------
    cpu_baseline = None
    if do_cpu:
        host = main_res[0]["host"]
        t1 = workers.apply(_cpu_job, ((host, args.size, args.pool, args.cpu_frames, False),))
        cpu_baseline = {"value": round(args.cpu_frames / t1, 3), "unit": "frames/s", "cores": 1, "kind": "port",
                        "sample": "%d frames of one %s %s stream through oracle/theora_oracle.c (scalar C restatement of the reference's C path, gcc -O2), "
                                  "timed behind the GPU's timed regions on an otherwise idle box" % (args.cpu_frames, args.size, args.content)}
        cb_detail = {"note": "a scalar-C port, not libtheora's x86 SIMD path (which cannot be built here: no libogg); the vectorised build of the same "
                             "port is timed beside it (simd)"}
        # the vectorised build of the same decoder (oracle/Makefile -DORC_SIMD: inverse transform, reconstruction and loop-filter
        # edges as SSE2 intrinsics, equal to the scalar build value for value, tests/test_oracle.py), one core
        nsimd = max(16, args.cpu_frames // 2)
        ts = workers.apply(_cpu_job, ((host, args.size, args.pool, nsimd, True),))
        cpu_baseline["simd"] = {"value": round(nsimd / ts, 3), "cores": 1, "kind": "port, SSE2"}
        cb_detail["simd"] = {"sample": "%d frames of the same stream through oracle/_build/libtheora_oracle_simd.so" % nsimd,
                             "note": "own SSE2 code for the three hot loops, the rest scalar; the reference's x86 path (which cannot be built here: no "
                                     "libogg) also vectorises its loop filter's row order and runs MMX/SSE2 assembly throughout"}
        # the same decoder on every core the box gives us, one process per core (the reference is single-threaded per stream;
        # many streams are many processes)
        nc = min(ncores, 64)
        if nc > 1 and not args.no_cpu_all_cores:
            per = max(12, args.cpu_frames // 8)
            with mp.get_context("spawn").Pool(nc) as pool_:
                times = pool_.map(_cpu_job, [(host, args.size, args.pool, per, False)] * nc)
                times_simd = pool_.map(_cpu_job, [(host, args.size, args.pool, per, True)] * nc)
            cpu_baseline["all_cores"] = {"value": round(nc * per / max(times), 2), "simd": round(nc * per / max(times_simd), 2), "cores": nc}
            cb_detail["all_cores"] = {"sample": "%d processes x %d frames, decode time of the slowest" % (nc, per)}
        cb_detail["system_libtheora"] = system_libtheora_baseline()
        detail["cpu_baseline"] = cb_detail
        wall.mark("cpu_baseline")
    workers.close()
    workers.join()

    if rank == 0:
        first_timed = nparity + args.warmup
        med_i = int(np.argsort(blocks)[len(blocks) // 2])
        rng_steps = range(first_timed + med_i * args.steps, first_timed + (med_i + 1) * args.steps)
        steps_b_alg = sum(alg_of_step(i, 0) for i in rng_steps)
        steps_b_read = sum(alg_of_step(i, 1) for i in rng_steps)
        prof_b_alg = sum(alg_of_step(i, 0) for i in range(first_prof, first_prof + prof_steps))
        prof_t_alg = sum(alg_of_step(i, 0) for i in range(first_prof + prof_steps, first_prof + 2 * prof_steps))
        total_frames = args.steps * S * world
        fps = total_frames / elapsed
        kname = KERNEL_NAMES[0] + ("<true>" if args.form == "levels" else "<false>")
        out = {
            "metric": "decode frames/sec (%s 4:2:0, bit-exact)" % args.size,
            "value": round(fps, 2),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 5),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "i16",
            "data": "synthetic",
            "config": {"workload": "%s (%dx%d coded) 4:2:0, %d concurrent streams per GPU, keyframe interval %d, content class '%s', %s form "
                                   "(seeded fragment command streams resident in HBM), loop filter on (flimit 2)"
                                   % (args.size, w, h, S, KF_INTERVAL, args.content, args.form),
                       "streams_per_gpu": S, "frame_pool": args.pool, "ws_MB": main_ws, "gt_IC": bool(main_ws > INFINITY_CACHE_MB),
                       "parallelism": "stream-sharded x%d" % world,
                       "process_group": ("nccl (RCCL %s), world %d" % (".".join(map(str, torch.cuda.nccl.version())), world))
                       if dist.is_initialized() else None},
            "timing": {"blocks": len(blocks), "statistic": "median block", "ms_per_step_min": round(1e3 * min(blocks) / args.steps, 5),
                       "ms_per_step_max": round(1e3 * max(blocks) / args.steps, 5), "ms_per_step_by_rank": per_rank_ms},
        }
        detail["timing"] = {"blocks": len(blocks), "steps_per_block": args.steps, "statistic": "median block",
                            "ms_per_step_first_block": round(1e3 * blocks[0] / args.steps, 5),
                            "host_submit_us_per_block": round(submit_us, 1), "idle_sync_us": round(idle_sync_us, 1)}
        detail["process_group"] = pg
        td_ = traffic_of(_pmc_flat(pmc), kname) or (traffic_of(_pmc_flat(pmc), KERNEL_NAMES[0]) if pmc else None)
        if profiling and kms[0] > 0:
            gbs = prof_b_alg / (kms[0] * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "kernel": kname, "achieved": round(gbs, 1),
                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                               # HBM bytes per launch of this kernel: FETCH_SIZE x 2 + WRITE_SIZE from two rocprofv3 --pmc passes of
                               # this very workload, in this run (null when rocprofv3 is missing, at N > 1, or with --no-pmc)
                               "traffic": td_["hbm_bytes_per_launch"] if td_ else None,
                               "traffic_source": "in-run rocprofv3 --pmc passes of this workload (FETCH_SIZE x 2 + WRITE_SIZE, KiB, one lane)" if td_ else None,
                               "avg_launch_us": round(1e3 * kms[0] / max(launches[0], 1), 3),
                               "alg_bytes_per_launch": int(prof_b_alg / max(launches[0], 1)),
                               "shape": "one launch per step carrying all %d streams, alone on the chip; HIP events around every launch of %d steps" % (S, prof_steps)}
            detail["roofline"] = {"traffic_detail": td_, "second_kernel": KERNEL_NAMES[1] if launches[1] else None,
                                  "second_kernel_avg_launch_us": round(1e3 * kms[1] / max(launches[1], 1), 3) if launches[1] else None,
                                  "instrumented_pass_ms_per_step": round(1e3 * elapsed_b / prof_steps, 5)}
            if td_ and td_.get("valu_insts_per_wave"):
                # the arithmetic floor of a step: waves x vector instructions per wave x 4.3 clocks an instruction (measured issue rate
                # of the packed / permute / multiply instructions the kernel is made of, profiles/r04_valu_rate2.txt) / 1024 SIMDs
                waves = geom.ntiles * S
                floor_us = waves * td_["valu_insts_per_wave"] * 4.3 / 1024.0 / 2400.0
                out["roofline"]["valu"] = {"insts_per_wave": td_["valu_insts_per_wave"], "issue_floor_us_per_step": round(floor_us, 2),
                                           "step_us": round(1e6 * elapsed / args.steps, 2)}
            if launches_t[0]:
                # the timed shape: launches of the library's lanes overlap, so a launch is longer than its share of a step;
                # sum of the durations / (steps x ms_per_step) = how many launches are in flight on average
                lt_us = 1e3 * kms_t[0] / launches_t[0]
                per_step = launches_t[0] / prof_steps
                detail["roofline"]["timed_shape"] = {
                    "launches_per_step": round(per_step, 2), "avg_launch_us": round(lt_us, 3),
                    "alg_bytes_per_launch": int(prof_t_alg / launches_t[0]),
                    "per_launch_GBps": round(prof_t_alg / launches_t[0] / (lt_us * 1e-6) / 1e9, 1),
                    "launches_in_flight": round(per_step * lt_us * 1e-3 / (1e3 * elapsed / args.steps), 2),
                    "note": "per_launch_GBps x launches_in_flight = pipeline.alg_GBps_per_gpu (the driver-clocked figure)"}
        # whole pipeline (recon + loop filter + launch gaps) against the HBM-read roofline of BASELINE.md section 3
        out["pipeline"] = {"read_roofline_frac": round((steps_b_read / elapsed) / 1e9 / HBM_PEAK_GBS, 4),
                           "alg_GBps_per_gpu": round(steps_b_alg / elapsed / 1e9, 1)}
        if form16 is not None and pmc:
            t16 = traffic_of(_pmc_flat(pmc), KERNEL_NAMES[0] + "<false>")
            if t16:
                entries["form_dequant16"]["traffic_MB"] = round(t16["hbm_bytes_per_launch"] / 1e6, 1)
                detail["entries"]["form_dequant16"]["traffic_detail"] = t16
        if entries:
            out["entries"] = entries
        if cpu_baseline:
            out["cpu_baseline"] = cpu_baseline
        if parity:
            detail["parity"] = dict(parity)
            detail["parity"]["stream_crc32_at_frame_%d" % (nparity - 1)] = ["%08x" % c for c in parity_crcs]
            out["parity"] = {"bit_exact": True, "frames": nparity, "streams": S * world}
            if (args.size, args.content, args.pool, nparity) == ("4k", "dense", 6, 40) and S >= 4:
                # streams [0, 4) are the same pictures at every world size, under every launch shape and in both coefficient forms
                got39 = ["%08x" % c for c in parity_crcs[:4]]
                if got39 != COMMITTED_CRC_FRAME39:
                    raise SystemExit("bench: frame-39 CRCs of streams 0..3 %s differ from the committed %s" % (got39, COMMITTED_CRC_FRAME39))
                out["parity"]["matches_committed_crc32"] = True
        detail["stream_crc32"] = ["%08x" % c for c in crcs]
        wall.mark("assemble")
        detail["wall_s"] = dict(wall.sections, total=wall.total(), workers=nworkers,
                                gen_s_by_job={t: j.get()["gen_s"] for t, j in jobs.items() if j.ready() and j.successful()})
        out["wall_s"] = wall.total()
        dpath = args.detail or (os.path.join(ROOT, "gpurun_out", "bench_detail.json") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else "bench_detail.json")
        try:
            detail["line"] = out
            with open(dpath, "w") as f:
                json.dump(detail, f, indent=1)
            out["detail"] = os.path.relpath(dpath, ROOT) if dpath.startswith(ROOT) else dpath
        except OSError as e:
            out["detail"] = "not written: %s" % e
        line = json.dumps(out)
        if len(line) > 6000:       # the driver keeps an 8 KB tail of the output: the line must fit with room to spare
            for k in ("timing", "pipeline"):
                detail.setdefault("moved_off_the_line", {})[k] = out.pop(k, None)
            line = json.dumps(out)
        print(line)
    try:
        tmp.cleanup()
    except OSError:
        pass
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def _pmc_flat(pmc):
    """run_pmc_passes' {kernel: {grid: counters}} -> {kernel: counters of its most frequent grid} (the decode kernels have one)."""
    if not pmc:
        return None
    out = {}
    for k, by_grid in pmc.items():
        g = max(by_grid, key=lambda x: by_grid[x].get("_n", 0))
        out[k] = by_grid[g]
    return out


def _pmc_by_grid(pmc):
    return pmc


if __name__ == "__main__":
    main()
