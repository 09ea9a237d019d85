#!/usr/bin/env python3
"""bench.py -- decode throughput of the MI355X fragment-reconstruction path.

A "step" decodes ONE frame of each of the rank's streams (default: 4 concurrent 4K 4:2:0
streams per GPU, keyframe interval 64) from fragment command streams already resident in
HBM: coded-fragment reconstruction (dequantised coefficients -> iDCT -> intra / motion
compensated predictor -> pixels), uncoded-fragment copy and the in-loop deblocking filter,
through the C ABI (thip_decode_frames).  Prints ONE JSON line (rank 0).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--size 4k|1080p|720p]
                  [--content dense|smooth|mixed] [--streams-per-gpu S]
For N>1 launch with  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N
Streams are sharded whole across ranks (no data-path collective); RCCL is used for the
barriers, the max-over-ranks time and the checksum gather.

Secondary modes (one GPU, not the headline metric):
  python bench.py --mode enc    BASELINE.json config 5: encoder block kernels, 1920x1088 4:4:4
  python bench.py --mode e2e    packets in host memory -> th_decode_* -> YUV in host memory
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
FUSED = os.environ.get("THIP_FUSE", "3")
# option "fuse": 3 = k_recon_lf (reconstruction + loop filter in one pass), anything else = the two passes
KERNEL_NAMES = ("k_recon_lf", None) if FUSED == "3" else ("k_recon", "k_loopfilter")
TRAFFIC_PROFILE = "profiles/r04_pmc_traffic.json"


def _kernel_base_name(name):
    """'void k_recon_lf<true>(BatchK)' -> 'k_recon_lf' (the kernels exist once per coefficient form)."""
    n = name.split("(")[0].strip()
    if n.startswith("void "):
        n = n[5:]
    return n.split("<")[0]
KF_INTERVAL = 64


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


def _cpu_worker(job):
    """One process of the all-cores CPU baseline: the oracle decoding `nframes` frames of stream 0
    (regenerated here from its seed: nothing but numbers crosses the process boundary)."""
    size, content, pool, nframes = job[:4]
    simd = len(job) > 4 and bool(job[4])
    import oracle
    import theora_amd
    from theora_amd import shard, synth
    w, h = SIZES[size]
    geom = synth.Geometry(w, h)
    rng = np.random.default_rng(shard.stream_seed(12345, 0))
    frames = [synth.gen_frame(geom, rng, theora_amd.INTRA_FRAME, content, flimit=2)]
    for _ in range(pool):
        frames.append(synth.gen_frame(geom, rng, theora_amd.INTER_FRAME, content, flimit=2))
    ost = oracle.State(w, h, simd=simd)
    t = 0.0
    for i in range(nframes):
        fr = frames[0 if i % KF_INTERVAL == 0 else 1 + (i % pool)]
        ost.refi[:] = fr["refi"]
        ost.mvs[:] = ((fr["mvx"] & 0xFF) | (fr["mvy"] << 8)).astype(np.int16)
        t0 = time.perf_counter()
        ost.decode_frame(fr["frame_type"], fr["coded_fragis"], fr["ncoded"], fr["coeffs"], fr["last_zzi"],
                         fr["dc_quant"], fr["uncoded_fragis"], fr["flimit"])
        t += time.perf_counter() - t0
    return t


# Frame-39 CRC32s of streams 0..3 of the default workload (4K dense, pool 6, kf 64): the same on every run, every world
# size and every launch shape -- a cheap end-to-end regression check of the timed batch.  Rounds 2 and 3 had
# 78bed3bc a82eb830 ee6d99cc 3ad204b2; round 4 changed the GENERATOR (theora_amd/synth.py: a frame is now quantised levels x
# dequantisation tables, as a real stream is, where it used to be free-form products that no table factors), so the pictures
# are different pictures; the oracle decodes the new ones to these values (checked by this script before it compares).
COMMITTED_CRC_FRAME39 = ["b4858401", "1f7f36d9", "83b0e5ba", "726f206f"]


def measure_pmc_traffic(args, kernels):
    """HBM bytes per launch of the timed kernels from the L2's fabric-side counters: this script re-executed under
    rocprofv3 --pmc, ONE counter per pass (FETCH_SIZE and WRITE_SIZE do not fit one pass, MI355X_MICROARCH.md), one lane so
    that every launch has the shape of the instrumented pass; FETCH_SIZE doubled (gfx950 reports half the bytes of wide
    coalesced reads, same guide), WRITE_SIZE as reported, both in KiB.  Returns a dict or None (no rocprofv3, a failed pass)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    out = {}
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as td:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(td, counter)
            cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
                   "--steps", "24", "--warmup", "4", "--repeats", "1", "--min-time", "0", "--no-cpu-baseline", "--no-parity",
                   "--no-profile", "--no-pmc", "--no-1080p", "--no-e2e", "--second-content", "", "--content", args.content, "--size", args.size,
                   "--streams-per-gpu", str(args.streams_per_gpu), "--pool", str(args.pool)]
            env = dict(os.environ, THIP_LANES="1", TMPDIR=os.environ.get("TMPDIR", "/tmp"))
            try:
                r = subprocess.run(cmd, env=env, cwd=td, capture_output=True, text=True, timeout=240)
            except (OSError, subprocess.TimeoutExpired):
                return None
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None
            acc = {}
            for row in csv.DictReader(open(files[0])):
                if row["Counter_Name"] == counter:
                    acc.setdefault(_kernel_base_name(row["Kernel_Name"]), []).append(float(row["Counter_Value"]))
            out[counter] = {k: sum(v) / len(v) for k, v in acc.items()}
    # ... and, in a third pass, the vector ALU: instructions issued and the share of the kernel's time the VALUs were busy (the
    # kernel is bound by them since round 4: DESIGN.md section 5.1)
    valu = {}
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as td:
        for group in (("VALUBusy",), ("SQ_INSTS_VALU", "SQ_WAVES")):
            d = os.path.join(td, group[0])
            cmd = [exe, "--pmc"] + list(group) + ["--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
                   "--steps", "24", "--warmup", "4", "--repeats", "1", "--min-time", "0", "--no-cpu-baseline", "--no-parity",
                   "--no-profile", "--no-pmc", "--no-1080p", "--no-e2e", "--second-content", "", "--content", args.content, "--size", args.size,
                   "--streams-per-gpu", str(args.streams_per_gpu), "--pool", str(args.pool)]
            try:
                r = subprocess.run(cmd, env=dict(os.environ, THIP_LANES="1", TMPDIR=os.environ.get("TMPDIR", "/tmp")), cwd=td,
                                   capture_output=True, text=True, timeout=240)
                files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
                if r.returncode == 0 and files:
                    for row in csv.DictReader(open(files[0])):
                        valu.setdefault((_kernel_base_name(row["Kernel_Name"]), row["Counter_Name"]), []).append(float(row["Counter_Value"]))
            except (OSError, subprocess.TimeoutExpired):
                pass
    res = {}
    for k in kernels:
        if k and k in out["FETCH_SIZE"]:
            f, w = out["FETCH_SIZE"][k], out["WRITE_SIZE"].get(k, 0.0)
            res[k] = {"fetch_kib_raw": round(f, 1), "write_kib": round(w, 1), "hbm_bytes_per_launch": int(round((2 * f + w) * 1024))}

            def avg(c):
                v = valu.get((k, c))
                return sum(v) / len(v) if v else None
            if avg("VALUBusy") is not None:
                res[k]["valu_busy_pct_alone"] = round(avg("VALUBusy"), 1)      # of a launch that has the chip to itself (THIP_LANES=1)
            if avg("SQ_INSTS_VALU") and avg("SQ_WAVES"):
                res[k]["valu_insts_per_wave"] = round(avg("SQ_INSTS_VALU") / avg("SQ_WAVES"), 1)
    return res or None


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


def e2e_keyed_entry(size="720p", nframes=8, loops=30, ahead=8):
    """The whole th_decode_* chain on the driver's clock, as a keyed entry of the default line: packets in host memory ->
    th_decode_packetin -> th_decode_ycbcr_out -> pictures in host memory, ONE stream, the plain API loop and with the packets
    announced `ahead` packets ahead (TH_DECCTL_THIP_PREFETCH_PACKET: the entropy decoder on the library's parser threads).  The two
    legs must hand out the ORACLE's pictures: every frame of a second, untimed pass of each leg over the same packets (CRC32 per frame).
    Never fails the bench: an exception becomes the entry."""
    try:
        import zlib
        from tests import streamgen
        from theora_amd.decoder import Decoder
        w, h = SIZES[size]
        content = dict(density=0.7, p_dc_only=0.5, p_empty=0.2)
        st = streamgen.Stream(w, h, 0, seed=99, trees="matched", probe_kwargs=content)
        hdr = st.header_packets()
        made = [st.frame(0 if f % 8 == 0 else 1, **content) for f in range(nframes)]
        pkts = [m[0] for m in made]
        # the oracle's pictures of these packets (packet 0 is a key frame, so every loop over the packets gives the same pictures):
        # EVERY frame of both legs is compared with them, the look-ahead's changes of sides (option fe_assign = 2) included
        import oracle
        ost = oracle.State(w, h, 0)
        want = []
        for pkt, truth in made:
            if not truth["dup"]:
                assert ost.decode_frame(**st.oracle_inputs(truth, ost)) == 0
            want.append(zlib.crc32(b"".join(np.ascontiguousarray(ost.get_plane(oracle.FRAME_PREV, pli)[::-1]).tobytes() for pli in range(3))))
        ost.close()
        seq = pkts * loops
        res, crcs, bad = {}, {}, {}
        from theora_amd import _lib as _l
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
        _l.load().thip_set_option(b"fe_pipeline", 0)
        same = len(set(tuple(v) for v in crcs.values())) == 1
        native = _e2e_native_legs(hdr, pkts, loops, ahead)
        if any(bad.values()):
            return {"error": "pictures differ from the oracle's: %r frames of %d" % (bad, len(seq))}
        return {"metric": "end-to-end decode frames/sec, one %s 4:2:0 stream (packets in host memory -> pictures in host memory)" % size,
                "unit": "frames/s", **res, "same_pictures": same, "frames_equal_to_the_oracle": "all %d of each leg's second, untimed pass (and the first %d of the timed one)" % (len(seq), len(pkts)),
                "c_caller": native,
                "avg_packet_bytes": sum(map(len, pkts)) // len(pkts),
                "data": "synthetic packets (tests/streamgen.py), dense content, matched Huffman trees",
                "note": "host-bound (Python caller): the plain loop is one entropy-decode thread per stream; announced packets are "
                        "parsed on up to eight library threads, which also pair tokens and fragments for the device (DESIGN.md 5.1)"}
    except Exception as e:
        try:
            from theora_amd import _lib as _l2
            _l2.load().thip_set_option(b"fe_pipeline", 0)
        except Exception:
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
            for label, extra in (("plain_loop", []), ("lookahead_%d" % ahead, ["--lookahead", str(ahead)]),
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
    ap.add_argument("--streams-per-gpu", type=int, default=4)
    ap.add_argument("--gop-parallel", type=int, default=1,
                    help="key-frame intervals of ONE stream decoded side by side (they are independent: a key frame resets "
                         "both references, decode.c:2947-2955): G decoder states per stream, state g takes intervals g, g+G, ...")
    ap.add_argument("--pool", type=int, default=6, help="distinct inter-frame command streams per stream")
    ap.add_argument("--cpu-frames", type=int, default=288, help="frames of stream 0 the CPU oracle decodes (~10 s at 4K)")
    ap.add_argument("--parity-frames", type=int, default=40, help="frames every timed stream is decoded and compared with the oracle before timing")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle comparison of the timed batch (profiling runs)")
    ap.add_argument("--repeats", type=int, default=15, help="minimum number of K-step blocks timed")
    ap.add_argument("--min-time", type=float, default=0.3, help="keep timing blocks until this many seconds have been measured")
    ap.add_argument("--second-content", default="smooth", help="content class of the second keyed entry ('' = none)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-late", action="store_true", help="time the CPU baseline behind the GPU's timed region (always so at N > 1)")
    ap.add_argument("--no-cpu-all-cores", action="store_true", help="skip the one-process-per-core leg of the CPU baseline")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--no-1080p", action="store_true", help="skip the 1080p keyed entries (single stream, four streams)")
    ap.add_argument("--no-wide", action="store_true", help="skip the keyed entry wide_tiles (the dense batch with 1 %% / 10 %% of the tiles wide)")
    ap.add_argument("--no-enc", action="store_true", help="skip the keyed entry enc_1080p_444 (BASELINE.json config 5: the encoder's block kernels)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the keyed entry e2e_720p (th_decode_* end to end, one stream, with and without the look-ahead)")
    ap.add_argument("--no-pmc", action="store_true", help="do not re-execute under rocprofv3 --pmc for roofline.traffic (N = 1 only)")
    return ap.parse_known_args()


def main_enc(emit=True, subset=False):
    """BASELINE.json config 5: the encoder's block kernels on one 1920x1088 4:4:4 frame.  emit: print one JSON line per kernel
    (--mode enc); else return the entries (the keyed entry enc_1080p_444 of the default line, subset = the kernels VERDICT r04
    asks for: fDCT, quantiser, both in one pass for one and four frames, the motion-search forms of SAD / SATD incl. the half-pel
    refinement, the cost maps).  Every kernel's output is compared with the oracle before its time counts."""
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
    d_prev, d_cur = torch.from_numpy(prev).cuda(), torch.from_numpy(cur).cuda()
    d_so, d_ro, d_r2 = (torch.from_numpy(a).cuda() for a in (src_offs, ref_offs, ref2_offs))
    resid = rng.integers(-255, 256, (nblk, 64)).astype(np.int16)
    d_res = torch.from_numpy(resid).cuda()

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

    results = []
    # --- fDCT ---------------------------------------------------------------------------------
    t = timed(lambda: theora_amd.fdct8x8_batch(d_res))
    got = theora_amd.fdct8x8_batch(d_res).cpu().numpy().reshape(-1, 64)
    ncpu = 20000
    t0 = time.perf_counter()
    want = oracle.fdct8x8_batch(resid[:ncpu])
    tc = time.perf_counter() - t0
    assert np.array_equal(got[:ncpu], want)
    tc_fdct = tc
    results.append(dict(kernel="oc_enc_fdct8x8", units=nblk, unit="blocks", seconds=t, bytes_per_unit=256, cpu_rate=ncpu / tc))
    # --- quantiser (enquant.c:219) on the fDCT output ------------------------------------------
    d_dct = theora_amd.fdct8x8_batch(d_res)
    dq = np.clip(np.arange(64) * 3 + 16, 8, 4096).astype(np.uint16)
    d_dq = torch.from_numpy(dq).cuda()
    t = timed(lambda: theora_amd.enc_quantize_batch(d_dct, d_dq))
    gq, gnz = theora_amd.enc_quantize_batch(d_dct, d_dq)
    t0 = time.perf_counter()
    wq, wnz = oracle.quantize_batch(got[:ncpu], dq)
    tc = time.perf_counter() - t0
    assert np.array_equal(gq.cpu().numpy().reshape(-1, 64)[:ncpu], wq) and np.array_equal(gnz.cpu().numpy()[:ncpu], wnz)
    tc_quant = tc
    results.append(dict(kernel="oc_enc_quantize", units=nblk, unit="blocks", seconds=t, bytes_per_unit=260, cpu_rate=ncpu / tc))
    # --- both in one pass (thip_enc_fdct_quantize_batch), one frame and four frames of residuals per call ------------------------
    for F in (1, 4):
        resF = np.tile(resid, (F, 1)) if F > 1 else resid
        d_resF = torch.from_numpy(np.ascontiguousarray(resF)).cuda()
        t = timed(lambda: theora_amd.enc_fdct_quantize_batch(d_resF, d_dq))
        fq, fnz = theora_amd.enc_fdct_quantize_batch(d_resF, d_dq)
        assert np.array_equal(fq.cpu().numpy().reshape(-1, 64)[:ncpu], wq) and np.array_equal(fnz.cpu().numpy()[:ncpu], wnz)
        results.append(dict(kernel="oc_enc_fdct8x8 + oc_enc_quantize in one pass (thip_enc_fdct_quantize_batch)%s" % (", %d frames per call" % F if F > 1 else ""),
                            units=nblk * F, unit="blocks", seconds=t, bytes_per_unit=260, cpu_rate=ncpu / (tc_fdct + tc_quant)))   # (the oracle's two calls)
    # --- SAD / SATD / SATD2 ---------------------------------------------------------------------
    for op, bpu in (() if subset else (("sad", 132), ("satd", 136), ("satd2", 136 + 64), ("intra_satd", 72))):
        call = lambda: theora_amd.enc_metric_batch(op, d_cur, d_prev, stride, d_so, d_ro, d_r2, 0)   # noqa: E731
        t = timed(call)
        v, dc = call()
        ncpu = 30000
        t0 = time.perf_counter()
        wv, wdc = oracle.enc_metric_batch(op, cur, prev, stride, src_offs[:ncpu], ref_offs[:ncpu], ref2_offs[:ncpu], 0)
        tc = time.perf_counter() - t0
        assert np.array_equal(v.cpu().numpy()[:ncpu].view(np.uint32), wv)
        results.append(dict(kernel="oc_enc_frag_" + op, units=src_offs.size, unit="(block,candidate)", seconds=t,
                            bytes_per_unit=bpu, cpu_rate=ncpu / tc))
    # --- the same candidates through the motion-search form (one reference position + the 9 sites per block) -----
    d_base = torch.from_numpy(base).cuda()
    for op, bpu in (("sad", 132), ("satd", 136)):
        call = lambda: theora_amd.enc_metric_sites_batch(op, d_cur, d_prev, stride, d_base, d_base, sites)   # noqa: E731
        t = timed(call)
        v, dc = call()
        want_v, _ = theora_amd.enc_metric_batch(op, d_cur, d_prev, stride, d_so, d_ro, d_r2, 0)
        assert torch.equal(v.reshape(-1), want_v)      # candidate-major = the order of the pair list (checked against the oracle above, or here:)
        ncpu = 30000
        t0 = time.perf_counter()
        wv, _ = oracle.enc_metric_batch(op, cur, prev, stride, src_offs[:ncpu], ref_offs[:ncpu], ref2_offs[:ncpu], 0)
        tc_pairs = time.perf_counter() - t0
        assert np.array_equal(want_v.cpu().numpy()[:ncpu].view(np.uint32), wv)
        # (the nine candidates of a block share its source block and all but a rim of the reference window: the traffic model is
        #  the UNIQUE bytes -- both frames once, the results -- not 132 / 136 bytes a pair)
        uniq = 2 * nblk * 64 + src_offs.size * (4 if op == "sad" else 8)
        results.append(dict(kernel="oc_enc_frag_%s, motion-search form (thip_enc_frag_metric_sites_batch)" % op, units=src_offs.size,
                            unit="(block,candidate)", seconds=t, bytes_per_unit=bpu, cpu_rate=ncpu / tc_pairs, unique_bytes=uniq))
    # --- the half-pel refinement around each block's whole-pel vector: eight sites (thip_enc_frag_metric_halfpel_batch; what the
    #     reference does with eight oc_enc_frag_satd2 / oc_enc_frag_sad2_thresh calls per block, mcenc.c:551-657) -----------------------
    hp_sites = [(-1, -1), (0, -1), (1, -1), (-1, 0), (1, 0), (-1, 1), (0, 1), (1, 1)]
    vecs = ((rng.integers(-2, 3, nblk) & 0xFF) | (rng.integers(-2, 3, nblk) << 8)).astype(np.int16)
    d_vecs = torch.from_numpy(vecs).cuda()
    for op, bpu in (("satd2", 136 + 64), ("sad2_thresh", 132 + 64)):
        call = lambda: theora_amd.enc_metric_halfpel_batch(op, d_cur, d_prev, stride, d_base, d_base, d_vecs, hp_sites)   # noqa: E731
        t = timed(call)
        v, dc = call()
        ncpu = 6000
        sel = rng.integers(0, nblk, ncpu)
        t0 = time.perf_counter()
        wv, wdc = oracle.enc_halfpel_sites(op, cur, prev, stride, base[sel], base[sel], vecs[sel], hp_sites)
        tc_pairs = time.perf_counter() - t0
        assert np.array_equal(wv, v.cpu().numpy().view(np.uint32)[:, sel])
        if dc is not None:
            assert np.array_equal(wdc, dc.cpu().numpy()[:, sel])
        results.append(dict(kernel="oc_enc_frag_%s, half-pel refinement form (thip_enc_frag_metric_halfpel_batch)" % op, units=nblk * len(hp_sites),
                            unit="(block,candidate)", seconds=t, bytes_per_unit=bpu, cpu_rate=ncpu * len(hp_sites) / tc_pairs,
                            unique_bytes=2 * nblk * 64 + nblk * 2 + nblk * len(hp_sites) * (4 if op != "satd2" else 8)))
    # --- ... and over several frames in one call (a single frame is one round of waves: its launch ramp, first loads and tail
    #     are a third of the call; an encoder with more than one stream to search hands them over together) -------------------
    for F in ((4,) if subset else (2, 4)):
        prevF = rng.integers(0, 256, (H * planes * F + 16, W + 16)).astype(np.uint8)
        curF = np.clip(np.roll(prevF, (1, 3), (0, 1)).astype(np.int32) + rng.integers(-6, 7, prevF.shape), 0, 255).astype(np.uint8)
        byF, bxF = np.mgrid[0:H * planes * F // 8, 0:W // 8]
        baseF = ((byF * 8 + 8) * stride + bxF * 8 + 8).reshape(-1).astype(np.int32)
        d_prevF, d_curF, d_baseF = torch.from_numpy(prevF).cuda(), torch.from_numpy(curF).cuda(), torch.from_numpy(baseF).cuda()
        for op, bpu in (("sad", 132), ("satd", 136)):
            call = lambda: theora_amd.enc_metric_sites_batch(op, d_curF, d_prevF, stride, d_baseF, d_baseF, sites)   # noqa: E731
            t = timed(call)
            v, dc = call()
            ncpu = 20000
            sel = rng.integers(0, baseF.size, ncpu)
            tc_pairs = 0.0
            for si, (dx, dy) in enumerate(sites[:3]):
                t0 = time.perf_counter()
                wv, _ = oracle.enc_metric_batch(op, curF, prevF, stride, baseF[sel], (baseF[sel] + dy * stride + dx).astype(np.int32),
                                                baseF[sel], 0)
                tc_pairs += time.perf_counter() - t0
                assert np.array_equal(v.reshape(len(sites), -1)[si].cpu().numpy()[sel].view(np.uint32), wv)
            results.append(dict(kernel="oc_enc_frag_%s, motion-search form, %d frames per call" % (op, F), units=baseF.size * len(sites),
                                unit="(block,candidate)", seconds=t, bytes_per_unit=bpu, cpu_rate=3 * ncpu / tc_pairs,
                                unique_bytes=2 * baseF.size * 64 + baseF.size * len(sites) * (4 if op == "sad" else 8)))
        if F == 4:
            vecsF = ((rng.integers(-2, 3, baseF.size) & 0xFF) | (rng.integers(-2, 3, baseF.size) << 8)).astype(np.int16)
            d_vecsF = torch.from_numpy(vecsF).cuda()
            for op, bpu in (("satd2", 136 + 64), ("sad2_thresh", 132 + 64)):
                call = lambda: theora_amd.enc_metric_halfpel_batch(op, d_curF, d_prevF, stride, d_baseF, d_baseF, d_vecsF, hp_sites)   # noqa: E731
                t = timed(call)
                v, dc = call()
                ncpu = 6000
                sel = rng.integers(0, baseF.size, ncpu)
                t0 = time.perf_counter()
                wv, wdc = oracle.enc_halfpel_sites(op, curF, prevF, stride, baseF[sel], baseF[sel], vecsF[sel], hp_sites)
                tc_pairs = time.perf_counter() - t0
                assert np.array_equal(wv, v.cpu().numpy().view(np.uint32)[:, sel])
                if dc is not None:
                    assert np.array_equal(wdc, dc.cpu().numpy()[:, sel])
                results.append(dict(kernel="oc_enc_frag_%s, half-pel refinement form, %d frames per call" % (op, F), units=baseF.size * len(hp_sites),
                                    unit="(block,candidate)", seconds=t, bytes_per_unit=bpu, cpu_rate=ncpu * len(hp_sites) / tc_pairs,
                                    unique_bytes=2 * baseF.size * 64 + baseF.size * 2 + baseF.size * len(hp_sites) * (4 if op != "satd2" else 8)))
    # --- the per-macro-block cost maps of a whole frame (thip_enc_mb_cost_maps: oc_mb_intra_satd, oc_mb_activity, _fast) ---------
    Wc, Hc = 1920, 1088
    cplanes = [rng.integers(0, 256, (Hc, Wc)).astype(np.uint8) for _ in range(3)]
    yy, xx = np.mgrid[0:Hc, 0:Wc]
    cplanes[0] = np.where(((xx // 16 + yy // 16) % 3) == 0, 90, np.where(((xx // 16 + yy // 16) % 3) == 1, cplanes[0],
                          np.where((xx + yy) % 16 < 8, 20, 230))).astype(np.uint8)       # flat / texture / edge macro blocks by turns
    d_cpl = [torch.from_numpy(p).cuda() for p in cplanes]
    call = lambda: theora_amd.enc_mb_cost_maps(d_cpl, Wc, Hc, 3)   # noqa: E731
    t = timed(call)
    got_maps = [g.cpu().numpy().view(np.uint32) for g in call()]
    t0 = time.perf_counter()
    want_maps = oracle.mb_cost_maps(cplanes, Wc, Hc, 3)
    tc = time.perf_counter() - t0
    assert all(np.array_equal(g, wv) for g, wv in zip(got_maps, want_maps))
    nmb = (Wc // 16) * (Hc // 16)
    results.append(dict(kernel="oc_mb_intra_satd + oc_mb_activity + oc_mb_activity_fast, whole frame (thip_enc_mb_cost_maps)", units=nmb,
                        unit="macro blocks", seconds=t, bytes_per_unit=12 * 64 + 21 * 4, cpu_rate=nmb / tc,
                        unique_bytes=3 * Wc * Hc + want_maps[0].shape[0] * 21 * 4 * 2))
    lines = []
    for r in results:
        # bytes moved: the per-unit model of SURVEY section 8(d) for the pair lists (every pair fetches its own blocks); the
        # unique bytes where units share their input (a frac above 1 against bytes that are not moved is not evidence)
        nbytes = r.get("unique_bytes", r["units"] * r["bytes_per_unit"])
        gbs = nbytes / r["seconds"] / 1e9
        lines.append(({
            "metric": r["kernel"] + " throughput", "value": round(r["units"] / r["seconds"] / 1e6, 1), "unit": "M%s/s" % r["unit"],
            "config": {"workload": "1920x1088 4:4:4%s, %d %s per call, 9-site square pattern" % (" x %s frames" % r["kernel"].split(", ")[-1].split()[0] if "frames per call" in r["kernel"] else "", r["units"], r["unit"])},
            "ms_per_call": round(1e3 * r["seconds"], 4), "timing": "; ".join(sorted(timing_modes)),
            "dtype": "u8/i16", "data": "synthetic", "bit_exact_vs_oracle": True,
            "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(gbs / HBM_PEAK_GBS, 4), "alg_bytes_per_unit": r["bytes_per_unit"],
                         "bytes_per_call": int(nbytes), "byte_model": "unique bytes (inputs once + results)" if "unique_bytes" in r
                         else "per-unit bytes of SURVEY section 8(d)"},
            "cpu_baseline": ({"value": round(r["cpu_rate"] / 1e6, 3), "unit": "M%s/s" % r["unit"], "cores": 1, "kind": "port"}
                             if r["cpu_rate"] == r["cpu_rate"] else None)}))
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



def main():
    args, rest = parse_args()
    if args.mode == "enc":
        return main_enc()
    if args.mode == "e2e":
        return main_e2e(args, rest)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world
    torch.cuda.set_device(local_rank)
    # under torch.distributed.run (RANK set) the process group is created at any world size, one included:
    # barriers, the MAX of the block times and the checksum gather then really go through RCCL
    under_launcher = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if world > 1 or under_launcher:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    import theora_amd
    from theora_amd import shard, synth

    w, h = SIZES[args.size]
    S = args.streams_per_gpu
    geom = synth.Geometry(w, h)
    # ---- synthetic command streams -> HBM ------------------------------------------------
    # Every stream has its own seeded content (seed = f(global stream id), so a stream is
    # the same pictures on whichever GPU it lands) and its own copy in HBM: one keyframe
    # command stream + `pool` inter-frame command streams, cycled.
    t_gen = time.time()
    keep, descs, balg = [], [], []            # descs[stream][frame], balg[stream][frame]
    host_frames = []                          # host_frames[stream][frame]: kept for the parity check / CPU baseline
    for gid in shard.stream_ids(rank, world, S):
        rng = np.random.default_rng(shard.stream_seed(12345, gid))
        frames = [synth.gen_frame(geom, rng, theora_amd.INTRA_FRAME, args.content, flimit=2)]
        for _ in range(args.pool):
            frames.append(synth.gen_frame(geom, rng, theora_amd.INTER_FRAME, args.content, flimit=2))
        host_frames.append(frames)
        row = []
        for f in frames:
            d, ka = synth.upload_frame(synth.pack_frame(geom, f))
            keep.append(ka)
            row.append(d)
        descs.append(row)
        balg.append([synth.algorithmic_bytes(geom, f) for f in frames])
    t_gen = time.time() - t_gen
    # A stream's frame sequence: interval n is a key frame followed by KF_INTERVAL-1 inter frames drawn from the pool
    # (shifted by the interval number, so that consecutive intervals are different pictures).  With
    # --gop-parallel G every stream has G states; state g decodes intervals g, g+G, g+2G, ...: at step i
    # it is at frame i % KF_INTERVAL of interval g + G * (i // KF_INTERVAL).
    G = max(1, args.gop_parallel)
    states = [theora_amd.State(w, h) for _ in range(S * G)]     # state q = s * G + g

    def frame_in_interval(j, interval):
        return 0 if j == 0 else 1 + ((j + 3 * interval) % args.pool)

    def frame_of_state(g, i):
        return frame_in_interval(i % KF_INTERVAL, g + G * (i // KF_INTERVAL))

    def frame_of_step(i):       # G == 1: the sequence of a stream
        return frame_of_state(0, i)

    plan_cache = {}

    def plan_for(i):
        key = tuple(frame_of_state(g, i) for g in range(G))
        if key not in plan_cache:
            plan_cache[key] = theora_amd.BatchPlan(states, [descs[s][key[g]] for s in range(S) for g in range(G)])
        return plan_cache[key]

    def run(nsteps, first=0, stream=None):
        for i in range(first, first + nsteps):
            plan_for(i).submit(stream)

    def alg_of_step(i, which):
        return sum(balg[s][frame_of_state(g, i)][which] for s in range(S) for g in range(G))

    def sync():
        theora_amd.synchronize()
        torch.cuda.synchronize()

    def barrier():
        shard.barrier(world)

    # ---- parity of the TIMED batch + CPU baseline ----------------------------------------
    # The states, plans and launch shape that are timed below (all S streams of this rank in one
    # thip_decode_frames call, the library's two lanes) first decode the first `parity_frames` frames of
    # the sequence; every plane of every stream is then compared with the oracle decoding the same
    # frames (every rank checks its own streams; a stream's content depends on its global id only, so
    # streams [0, S) are the same pictures at every world size).  No number is printed on a mismatch.
    cpu_baseline, parity = None, None
    # At N > 1 rank 0's CPU baseline (tens of seconds, up to 64 processes) runs BEHIND the timed region, so that no rank's clock
    # starts after a long wait in a collective (--cpu-baseline-late forces the same order at N = 1)
    cpu_late = world > 1 or args.cpu_baseline_late
    nparity = 0 if args.no_parity else max(1, args.parity_frames)
    if nparity:
        import oracle
        run(nparity)
        sync()
        t_cpu, n_cpu, ok, bad = 0.0, 0, True, []
        parity_crcs = []     # of every stream's frame nparity-1: the same at every world size and on every run
        for q, (s_local, gid, g) in enumerate((sl, gi, gg) for sl, gi in enumerate(shard.stream_ids(rank, world, S)) for gg in range(G)):
            ost = oracle.State(w, h)
            # stream 0 of rank 0 goes on to `cpu_frames` frames for the CPU baseline; the comparison is at frame nparity-1
            nf = max(nparity, args.cpu_frames) if (rank == 0 and q == 0 and not args.no_cpu_baseline and not cpu_late) else nparity
            for i in range(nf):
                fr = host_frames[s_local][frame_of_state(g, i)]
                ost.refi[:] = fr["refi"]
                ost.mvs[:] = ((fr["mvx"] & 0xFF) | (fr["mvy"] << 8)).astype(np.int16)
                t0 = time.perf_counter()
                ost.decode_frame(fr["frame_type"], fr["coded_fragis"], fr["ncoded"], fr["coeffs"], fr["last_zzi"],
                                 fr["dc_quant"], fr["uncoded_fragis"], fr["flimit"])
                if rank == 0 and q == 0:
                    t_cpu += time.perf_counter() - t0
                    n_cpu += 1
                if i == nparity - 1:
                    c = 0
                    for pli in range(3):
                        a = ost.get_plane(oracle.FRAME_PREV, pli)
                        b = states[q].read_plane(states[q].ref_idx(theora_amd.FRAME_PREV), pli)
                        c = zlib.crc32(b.tobytes(), c)
                        if not np.array_equal(a, b):
                            ok = False
                            bad.append((gid, pli, int((a != b).sum())))
                    parity_crcs.append(c)
            ost.close()
        ok_all = shard.reduce_min(1 if ok else 0, torch.device("cuda", local_rank))
        if not ok_all:
            raise SystemExit("bench: GPU output differs from the oracle %s -- refusing to report a number" % bad[:4])
        parity = {"frames": nparity, "bit_exact": True,
                  "checked": "the timed batch itself: all %d streams of every rank%s, decoded %d frames deep by the timed "
                             "states in the timed launch shape (one thip_decode_frames call per step), every plane of "
                             "every stream against the oracle" % (S, " (x %d key-frame intervals side by side)" % G if G > 1 else "", nparity)}
        def cpu_all_cores(cb):
            # the vectorised build of the same decoder (oracle/Makefile -DORC_SIMD: inverse transform, reconstruction and loop-filter
            # edges as SSE2 intrinsics, equal to the scalar build value for value, tests/test_oracle.py), one core
            nsimd = max(16, args.cpu_frames // 2)
            ts = _cpu_worker((args.size, args.content, args.pool, nsimd, True))
            cb["simd"] = {"value": round(nsimd / ts, 3), "unit": "frames/s", "cores": 1, "kind": "port, simd",
                          "sample": "%d frames of the same stream through oracle/_build/libtheora_oracle_simd.so" % nsimd,
                          "note": "own SSE2 code for the three hot loops, the rest scalar; the reference's x86 path (which cannot be built "
                                  "here: no libogg) also vectorises its loop filter's row order and runs MMX/SSE2 assembly throughout"}
            # the same decoder on every core the box gives us, one process per core (the reference is
            # single-threaded per stream; many streams are many processes)
            ncores = min(_usable_cores(), 64)
            if ncores > 1 and not args.no_cpu_all_cores:
                import multiprocessing as mp
                per = max(16, args.cpu_frames // 4)
                with mp.get_context("spawn").Pool(ncores) as pool_:
                    times = pool_.map(_cpu_worker, [(args.size, args.content, args.pool, per)] * ncores)
                    times_simd = pool_.map(_cpu_worker, [(args.size, args.content, args.pool, per, True)] * ncores)
                cb["all_cores"] = {"value": round(ncores * per / max(times), 2), "unit": "frames/s", "cores": ncores,
                                   "sample": "%d processes x %d frames, decode time of the slowest" % (ncores, per),
                                   "simd": round(ncores * per / max(times_simd), 2)}
        if rank == 0 and not args.no_cpu_baseline and not cpu_late:
            cpu_baseline = {"value": round(n_cpu / t_cpu, 3), "unit": "frames/s", "cores": 1, "kind": "port",
                            "sample": "%d frames of one %s %s stream through oracle/theora_oracle.c "
                                      "(scalar C restatement of the reference's C path, gcc -O2)" % (n_cpu, args.size, args.content),
                            "note": "a scalar-C port, not libtheora's x86 SIMD path (which cannot be built here: no libogg); the "
                                    "vectorised build of the same port is timed beside it (simd)"}
            cpu_all_cores(cpu_baseline)

    # ---- timed region ---------------------------------------------------------------------
    # Pass A: blocks of exactly K steps, each bracketed by synchronize + barrier on both sides, no
    # instrumentation inside.  One block at the driver's --steps 20 is 1.3 ms, too short to be a stable
    # number, so the block is repeated (at least `repeats` times and until ~0.3 s of GPU time has been
    # measured); `value` comes from the MEDIAN block, min and max are reported beside it.
    step0 = nparity
    for i in range(step0, step0 + 3 * KF_INTERVAL * 2):   # every combination the timed steps can ask for exists before the clock starts
        plan_for(i)
    run(args.warmup, first=step0)
    step0 += args.warmup
    sync()
    elapsed_blocks = []
    submit_times = []
    total_t = 0.0
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
            total_t = shard.reduce_max([total_t], torch.device("cuda", local_rank))[0]
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
    #       (`roofline.timed_shape`: a launch takes longer there because it shares the chip with the other lane's).
    profiling = not args.no_profile
    launches, kms, elapsed_b = [0, 0], [0.0, 0.0], 0.0
    launches_t, kms_t = [0, 0], [0.0, 0.0]
    prof_steps = max(args.steps, 256)
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
    dev = torch.device("cuda", local_rank)
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

    # ---- the same measurement on the content class SURVEY section 8d defines from the reference's own
    #      statistics (66 % coded, 80 % of the coded blocks DC-only): a second keyed entry on the line ------
    second = None
    if args.second_content and args.second_content != args.content and G == 1:
        descs2, balg2, keep2 = [], [], []
        for gid in shard.stream_ids(rank, world, S):
            rng = np.random.default_rng(shard.stream_seed(12345, gid))
            frames = [synth.gen_frame(geom, rng, theora_amd.INTRA_FRAME, args.second_content, flimit=2)]
            for _ in range(args.pool):
                frames.append(synth.gen_frame(geom, rng, theora_amd.INTER_FRAME, args.second_content, flimit=2))
            row = []
            for f in frames:
                d, ka = synth.upload_frame(synth.pack_frame(geom, f))
                keep2.append(ka)
                row.append(d)
            descs2.append(row)
            balg2.append([synth.algorithmic_bytes(geom, f) for f in frames])
        plans2 = [theora_amd.BatchPlan(states, [descs2[s][j] for s in range(S)]) for j in range(args.pool + 1)]
        K2 = max(args.steps, 64)

        def run2(n, first):
            for i in range(first, first + n):
                plans2[frame_of_step(i)].submit(None)
        run2(KF_INTERVAL, 0)     # starts with a key frame; warm
        sync()
        b2 = []
        for rep in range(max(5, min(args.repeats, 15))):
            sync()
            barrier()
            t0 = time.perf_counter()
            run2(K2, KF_INTERVAL + rep * K2)
            sync()
            b2.append(time.perf_counter() - t0)
            barrier()
        b2 = shard.reduce_max(b2, dev)
        e2 = float(np.median(b2))
        steps2 = [KF_INTERVAL + i for i in range(K2)]
        alg2 = sum(balg2[s][frame_of_step(i)][0] for i in steps2 for s in range(S))
        read2 = sum(balg2[s][frame_of_step(i)][1] for i in steps2 for s in range(S))
        second = {"content": args.second_content, "value": round(K2 * S * world / e2, 2), "unit": "frames/s",
                  "steps": K2, "ms_per_step": round(1e3 * e2 / K2, 5),
                  "ms_per_step_min_max": [round(1e3 * min(b2) / K2, 5), round(1e3 * max(b2) / K2, 5)],
                  "pipeline_read_roofline_frac": round(read2 / e2 / 1e9 / HBM_PEAK_GBS, 4),
                  "alg_GBps_per_gpu": round(alg2 / e2 / 1e9, 1),
                  "note": "same streams, states and launch shape; parity of this class: tests/test_gpu_frames.py"}

    # ---- the other size the metric names: 1080p (BASELINE.json config 3: ONE 1080p stream, kf 64 -- launches that leave the
    #      chip half empty -- and four streams in one call), same content class, same clock ------------------------------------
    other_size = None
    if args.size == "4k" and not args.no_1080p and G == 1:
        w2, h2 = SIZES["1080p"]
        geom2 = synth.Geometry(w2, h2)
        other_size = {}
        for label, S2 in (("single_stream", 1), ("four_streams", 4)):
            descs3, balg3, keep3 = [], [], []
            for gid in shard.stream_ids(rank, world, S2):
                rng = np.random.default_rng(shard.stream_seed(777, gid))
                frames = [synth.gen_frame(geom2, rng, theora_amd.INTRA_FRAME, args.content, flimit=2)]
                for _ in range(args.pool):
                    frames.append(synth.gen_frame(geom2, rng, theora_amd.INTER_FRAME, args.content, flimit=2))
                row = []
                for f in frames:
                    d, ka = synth.upload_frame(synth.pack_frame(geom2, f))
                    keep3.append(ka)
                    row.append(d)
                descs3.append(row)
                balg3.append([synth.algorithmic_bytes(geom2, f) for f in frames])
            states3 = [theora_amd.State(w2, h2) for _ in range(S2)]
            plans3 = [theora_amd.BatchPlan(states3, [descs3[s][j] for s in range(S2)]) for j in range(args.pool + 1)]
            K3 = max(args.steps, 128)

            def run3(n, first):
                for i in range(first, first + n):
                    plans3[frame_of_step(i)].submit(None)
            run3(KF_INTERVAL, 0)
            sync()
            b3 = []
            for rep in range(max(5, min(args.repeats, 15))):
                sync()
                barrier()
                t0 = time.perf_counter()
                run3(K3, KF_INTERVAL + rep * K3)
                sync()
                b3.append(time.perf_counter() - t0)
                barrier()
            b3 = shard.reduce_max(b3, dev)
            e3 = float(np.median(b3))
            steps3 = [KF_INTERVAL + i for i in range(K3)]
            read3 = sum(balg3[s][frame_of_step(i)][1] for i in steps3 for s in range(S2))
            other_size[label] = {"value": round(K3 * S2 * world / e3, 2), "unit": "frames/s", "streams_per_gpu": S2, "steps": K3,
                                 "ms_per_step": round(1e3 * e3 / K3, 5),
                                 "pipeline_read_roofline_frac": round(read3 / e3 / 1e9 / HBM_PEAK_GBS, 4)}
            for st3 in states3:
                st3.close()
            del keep3, descs3
        # ---- config 3's roofline form: ONE 1080p stream whose key-frame intervals are decoded side by side (a key frame resets
        #      both references, decode.c:2947-2955, so intervals are independent): 16 states, state g takes intervals g, g + 16, ... ----
        try:
            G3 = 16
            rng = np.random.default_rng(shard.stream_seed(777, shard.stream_ids(rank, world, 1)[0]))
            frames3 = [synth.gen_frame(geom2, rng, theora_amd.INTRA_FRAME, args.content, flimit=2)]
            for _ in range(args.pool):
                frames3.append(synth.gen_frame(geom2, rng, theora_amd.INTER_FRAME, args.content, flimit=2))
            keep3, descs3 = [], []
            for f in frames3:
                d, ka = synth.upload_frame(synth.pack_frame(geom2, f))
                keep3.append(ka)
                descs3.append(d)
            balg3 = [synth.algorithmic_bytes(geom2, f) for f in frames3]
            states3 = [theora_amd.State(w2, h2) for _ in range(G3)]

            def fos3(g, i):
                return frame_in_interval(i % KF_INTERVAL, g + G3 * (i // KF_INTERVAL))
            cache3 = {}

            def plan3(i):
                key = tuple(fos3(g, i) for g in range(G3))
                if key not in cache3:
                    cache3[key] = theora_amd.BatchPlan(states3, [descs3[k] for k in key])
                return cache3[key]
            # parity of this shape before its clock: the first six steps, states 0 and 15 against the oracle
            NP3 = 6
            for i in range(NP3):
                plan3(i).submit(None)
            sync()
            if nparity:
                for g in (0, G3 - 1):
                    ost = oracle.State(w2, h2)
                    for i in range(NP3):
                        fr = frames3[fos3(g, i)]
                        ost.refi[:] = fr["refi"]
                        ost.mvs[:] = ((fr["mvx"] & 0xFF) | (fr["mvy"] << 8)).astype(np.int16)
                        ost.decode_frame(fr["frame_type"], fr["coded_fragis"], fr["ncoded"], fr["coeffs"], fr["last_zzi"],
                                         fr["dc_quant"], fr["uncoded_fragis"], fr["flimit"])
                    for pli in range(3):
                        if not np.array_equal(ost.get_plane(oracle.FRAME_PREV, pli), states3[g].read_plane(states3[g].ref_idx(theora_amd.FRAME_PREV), pli)):
                            raise SystemExit("bench: single_gop16 differs from the oracle (state %d, plane %d)" % (g, pli))
                    ost.close()
            K3 = max(args.steps, 64)
            for i in range(NP3, KF_INTERVAL):
                plan3(i).submit(None)
            sync()
            b3 = []
            for rep in range(max(5, min(args.repeats, 15))):
                sync()
                barrier()
                t0 = time.perf_counter()
                for i in range(KF_INTERVAL + rep * K3, KF_INTERVAL + (rep + 1) * K3):
                    plan3(i).submit(None)
                sync()
                b3.append(time.perf_counter() - t0)
                barrier()
            b3 = shard.reduce_max(b3, dev)
            e3 = float(np.median(b3))
            read3 = sum(balg3[fos3(g, KF_INTERVAL + i)][1] for i in range(K3) for g in range(G3))
            other_size["single_gop16"] = {"value": round(K3 * G3 * world / e3, 2), "unit": "frames/s", "streams_per_gpu": 1,
                                          "key_frame_intervals_side_by_side": G3, "steps": K3, "ms_per_step": round(1e3 * e3 / K3, 5),
                                          "pipeline_read_roofline_frac": round(read3 / e3 / 1e9 / HBM_PEAK_GBS, 4),
                                          "bit_exact": bool(nparity),
                                          "note": "BASELINE.json config 3 in the form that fills the chip: the caller decodes 16 key-frame "
                                                  "intervals of the one stream side by side (16 states in one thip_decode_frames call per step)"}
            for st3 in states3:
                st3.close()
            del keep3, descs3
        except SystemExit:
            raise
        except Exception as e:   # (never fails the bench)
            other_size["single_gop16"] = {"error": str(e)[:300]}
        other_size["note"] = ("1080p (1920x1088 coded) 4:2:0, kf %d, content class '%s', frames decoded one after the other through "
                              "thip_decode_frames; parity of these shapes: tests/test_gpu_frames.py::test_config3_as_written, "
                              "::test_full_size_sequences" % (KF_INTERVAL, args.content))

    # ---- rank 0's CPU baseline behind the timed region (N > 1, or --cpu-baseline-late): the other ranks go on to the final gather ----
    if nparity and rank == 0 and not args.no_cpu_baseline and cpu_late:
        t1 = _cpu_worker((args.size, args.content, args.pool, args.cpu_frames))
        cpu_baseline = {"value": round(args.cpu_frames / t1, 3), "unit": "frames/s", "cores": 1, "kind": "port",
                        "sample": "%d frames of one %s %s stream through oracle/theora_oracle.c (scalar C restatement of the reference's "
                                  "C path, gcc -O2), timed behind the GPU's timed region" % (args.cpu_frames, args.size, args.content)}
        cpu_all_cores(cpu_baseline)

    # ---- wide tiles: the same dense step with 1 % and 10 % of the tiles holding a level beyond eight bits (the levels form's
    #      escape, include/theora_hip.h THIP_SLOT_WIDE: such a tile's blocks own two units of int16 levels) ---------------------------
    wide_tiles = None
    if args.size == "4k" and args.content == "dense" and G == 1 and not args.no_wide and world == 1:
        wide_tiles = {}
        try:
            for label, frac in (("1pct", 0.01), ("10pct", 0.10)):
                descs4, keep4, balg4, frames4 = [], [], [], []
                wrng = np.random.default_rng(4242)
                nwide, ntl = 0, 0
                for s_ in range(S):
                    row, rowf = [], []
                    for f in host_frames[s_][:3]:      # the key frame and two inter frames of the timed pool, some tiles widened
                        fw, nw = synth.widen_tiles(geom, f, frac, wrng)
                        nwide += nw
                        ntl += geom.ntiles
                        d, ka = synth.upload_frame(synth.pack_frame(geom, fw))
                        keep4.append(ka)
                        row.append(d)
                        rowf.append(fw)
                    descs4.append(row)
                    frames4.append(rowf)
                    balg4.append([synth.algorithmic_bytes(geom, f) for f in rowf])
                plans4 = [theora_amd.BatchPlan(states, [descs4[s_][j] for s_ in range(S)]) for j in range(3)]

                def f4(i):
                    return 0 if i % KF_INTERVAL == 0 else 1 + (i % 2)
                for i in range(3):
                    plans4[f4(i)].submit(None)
                sync()
                if nparity:        # stream 0 against the oracle, three frames deep
                    ost = oracle.State(w, h)
                    for i in range(3):
                        fr = frames4[0][f4(i)]
                        ost.refi[:] = fr["refi"]
                        ost.mvs[:] = ((fr["mvx"] & 0xFF) | (fr["mvy"] << 8)).astype(np.int16)
                        ost.decode_frame(fr["frame_type"], fr["coded_fragis"], fr["ncoded"], fr["coeffs"], fr["last_zzi"],
                                         fr["dc_quant"], fr["uncoded_fragis"], fr["flimit"])
                    for pli in range(3):
                        if not np.array_equal(ost.get_plane(oracle.FRAME_PREV, pli), states[0].read_plane(states[0].ref_idx(theora_amd.FRAME_PREV), pli)):
                            raise SystemExit("bench: the wide-tile batch differs from the oracle (plane %d)" % pli)
                    ost.close()
                K4 = max(args.steps, 64)
                for i in range(3, KF_INTERVAL):
                    plans4[f4(i)].submit(None)
                sync()
                b4 = []
                for rep in range(max(5, min(args.repeats, 15))):
                    sync()
                    t0 = time.perf_counter()
                    for i in range(KF_INTERVAL + rep * K4, KF_INTERVAL + (rep + 1) * K4):
                        plans4[f4(i)].submit(None)
                    sync()
                    b4.append(time.perf_counter() - t0)
                e4 = float(np.median(b4))
                read4 = sum(balg4[s_][f4(KF_INTERVAL + i)][1] for i in range(K4) for s_ in range(S))
                wide_tiles[label] = {"value": round(K4 * S / e4, 2), "unit": "frames/s", "steps": K4, "ms_per_step": round(1e3 * e4 / K4, 5),
                                     "pipeline_read_roofline_frac": round(read4 / e4 / 1e9 / HBM_PEAK_GBS, 4),
                                     "wide_tiles": nwide, "tiles": ntl, "bit_exact": bool(nparity)}
                del keep4, descs4, plans4
            wide_tiles["note"] = ("the timed dense batch with one level of 300 planted in that share of the tiles (every block of such a tile "
                                  "then travels as two units of int16 levels instead of one of int8); the headline's content has no wide tile")
        except SystemExit:
            raise
        except Exception as e:
            wide_tiles["error"] = str(e)[:300]

    # ---- BASELINE.json config 5: the encoder's block kernels on one 1920x1088 4:4:4 frame (each checked against the oracle first) ----
    enc_entry = None
    if rank == 0 and world == 1 and args.size == "4k" and G == 1 and not args.no_enc:
        try:
            enc_entry = {"entries": main_enc(emit=False, subset=True),
                         "note": "BASELINE.json config 5 (1920x1088 4:4:4): rooflines of the motion-search and cost-map entries are on UNIQUE "
                                 "bytes (their candidates share their input); cpu_baseline = the oracle's scalar C on one core, same run"}
        except Exception as e:
            enc_entry = {"error": str(e)[:300]}

    if rank == 0:
        first_timed = nparity + args.warmup
        med_i = int(np.argsort(blocks)[len(blocks) // 2])
        rng_steps = range(first_timed + med_i * args.steps, first_timed + (med_i + 1) * args.steps)
        steps_b_alg = sum(alg_of_step(i, 0) for i in rng_steps)
        steps_b_read = sum(alg_of_step(i, 1) for i in rng_steps)
        first_prof = first_timed + len(blocks) * args.steps
        prof_b_alg = sum(alg_of_step(i, 0) for i in range(first_prof, first_prof + prof_steps))
        prof_t_alg = sum(alg_of_step(i, 0) for i in range(first_prof + prof_steps, first_prof + 2 * prof_steps))
        total_frames = args.steps * S * G * world
        fps = total_frames / elapsed
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
            "config": {"workload": "%s (%dx%d coded) 4:2:0, %d concurrent streams per GPU%s, keyframe interval %d, "
                                   "content class '%s' (seeded fragment command streams resident in HBM), "
                                   "loop filter on (flimit 2)" % (args.size, w, h, S, (", %d key-frame intervals of each stream decoded "
                                                                  "side by side" % G) if G > 1 else "", KF_INTERVAL, args.content),
                       "gop_parallel": G,
                       "streams_per_gpu": S, "frame_pool": args.pool, "parallelism": "stream-sharded x%d" % world,
                       "process_group": ("nccl (RCCL %s), world %d" % (".".join(map(str, torch.cuda.nccl.version())), world))
                       if dist.is_initialized() else None},
            "timing": {"blocks": len(blocks), "steps_per_block": args.steps, "statistic": "median block",
                       "ms_per_step_min": round(1e3 * min(blocks) / args.steps, 5),
                       "ms_per_step_max": round(1e3 * max(blocks) / args.steps, 5),
                       "ms_per_step_first_block": round(1e3 * blocks[0] / args.steps, 5),
                       "host_submit_us_per_block": round(submit_us, 1), "idle_sync_us": round(idle_sync_us, 1),
                       "ms_per_step_by_rank": per_rank_ms},
            "process_group": pg,
        }
        # HBM bytes per launch from the PMC counters of THIS workload: two more passes of this script under rocprofv3
        traffic, traffic_bytes = None, None
        if world == 1 and not args.no_pmc and profiling and G == 1:
            traffic = measure_pmc_traffic(args, KERNEL_NAMES)
            if traffic and KERNEL_NAMES[0] in traffic:
                traffic_bytes = traffic[KERNEL_NAMES[0]]["hbm_bytes_per_launch"]
        if profiling and kms[0] > 0:
            gbs = prof_b_alg / (kms[0] * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "kernel": KERNEL_NAMES[0], "achieved": round(gbs, 1),
                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                               # HBM bytes per launch of this kernel: FETCH_SIZE x 2 + WRITE_SIZE from two rocprofv3 --pmc passes
                               # of this very workload (measure_pmc_traffic; null when rocprofv3 is missing, at N > 1, or with
                               # --no-pmc); the committed passes of the round are under profiles/
                               "traffic": traffic_bytes, "traffic_detail": traffic, "traffic_profile": TRAFFIC_PROFILE,
                               "avg_launch_us": round(1e3 * kms[0] / max(launches[0], 1), 3),
                               "second_kernel": KERNEL_NAMES[1] if launches[1] else None,
                               "second_kernel_avg_launch_us": round(1e3 * kms[1] / max(launches[1], 1), 3) if launches[1] else None,
                               "alg_bytes_per_launch": int(prof_b_alg / max(launches[0], 1)),
                               "shape": "one launch per step carrying all %d streams, alone on the chip" % (S * G),
                               "measured": "HIP events around every launch, separate instrumented pass of %d steps on one "
                                           "stream (ms_per_step there: %.5f)" % (prof_steps, 1e3 * elapsed_b / prof_steps)}
            td_ = (traffic or {}).get(KERNEL_NAMES[0]) or {}
            if td_.get("valu_insts_per_wave"):
                # the arithmetic floor of a step: waves x vector instructions per wave x 4.3 clocks an instruction (measured issue rate
                # of the packed / permute / multiply instructions the kernel is made of, profiles/r04_valu_rate2.txt) / 1024 SIMDs
                waves = geom.ntiles * S * G
                floor_us = waves * td_["valu_insts_per_wave"] * 4.3 / 1024.0 / 2400.0
                out["roofline"]["valu"] = {"insts_per_wave": td_["valu_insts_per_wave"], "busy_pct_of_a_lone_launch": td_.get("valu_busy_pct_alone"),
                                           "issue_floor_us_per_step": round(floor_us, 2),
                                           "step_us": round(1e6 * elapsed / args.steps, 2),
                                           "note": "the kernel is bound by vector-ALU issue since round 4 (DESIGN.md section 5.1): the floor is "
                                                   "waves x instructions x 4.3 clocks / 1024 SIMDs at 2.4 GHz"}
            if launches_t[0]:
                # the timed shape: launches of the library's lanes overlap, so a launch is longer than its share of a step;
                # sum of the durations / (steps x ms_per_step) = how many launches are in flight on average
                lt_us = 1e3 * kms_t[0] / launches_t[0]
                per_step = launches_t[0] / prof_steps
                out["roofline"]["timed_shape"] = {
                    "launches_per_step": round(per_step, 2), "avg_launch_us": round(lt_us, 3),
                    "alg_bytes_per_launch": int(prof_t_alg / launches_t[0]),
                    "per_launch_GBps": round(prof_t_alg / launches_t[0] / (lt_us * 1e-6) / 1e9, 1),
                    "launches_in_flight": round(per_step * lt_us * 1e-3 / (1e3 * elapsed / args.steps), 2),
                    "note": "per_launch_GBps x launches_in_flight = pipeline.alg_GBps_per_gpu (the driver-clocked figure)"}
        # whole pipeline (recon + loop filter + launch gaps) against the HBM-read roofline of BASELINE.md section 3
        out["pipeline"] = {"read_roofline_frac": round((steps_b_read / elapsed) / 1e9 / HBM_PEAK_GBS, 4),
                           "alg_GBps_per_gpu": round(steps_b_alg / elapsed / 1e9, 1)}
        if second:
            out["second_content"] = second
        if other_size:
            out["size_1080p"] = other_size
        if wide_tiles:
            out["wide_tiles"] = wide_tiles
        if enc_entry:
            out["enc_1080p_444"] = enc_entry
        if world == 1 and G == 1 and not args.no_e2e and args.size == "4k":
            out["e2e_720p"] = e2e_keyed_entry()
        if cpu_baseline:
            cpu_baseline["system_libtheora"] = system_libtheora_baseline()
            out["cpu_baseline"] = cpu_baseline
        if parity:
            parity["stream_crc32_at_frame_%d" % (nparity - 1)] = ["%08x" % c for c in parity_crcs]
            if (args.size, args.content, args.pool, nparity, G) == ("4k", "dense", 6, 40, 1) and S >= 4:
                # streams [0, 4) are the same pictures at every world size and under every launch shape
                got39 = ["%08x" % c for c in parity_crcs[:4]]
                if got39 != COMMITTED_CRC_FRAME39:
                    raise SystemExit("bench: frame-39 CRCs of streams 0..3 %s differ from the committed %s" % (got39, COMMITTED_CRC_FRAME39))
                parity["matches_committed_crc32"] = True
            out["parity"] = parity
        out["stream_crc32"] = ["%08x" % c for c in crcs]
        out["setup_s"] = round(t_gen, 1)
        print(json.dumps(out))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
