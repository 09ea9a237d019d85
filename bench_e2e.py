#!/usr/bin/env python3
"""bench_e2e.py -- the SECOND throughput number of SURVEY.md section 8(d): end to end, Theora
packets in host memory -> th_decode_packetin (own front end: Huffman, modes, vectors, tokens,
DC un-prediction, dequantisation, staging, upload, HIP reconstruction + loop filter) ->
th_decode_ycbcr_out planes in host memory.  Host-bound by construction (one CPU thread
parses, PCIe carries coefficients in and pictures out); bench.py carries the device-pipeline
number.  Streams come from tests/streamgen.py: random coefficients at a far higher bit rate
than real content (hundreds of KB per frame), i.e. a pessimistic entropy-decode load.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="720p", choices=["qcif", "cif", "720p", "1080p"])
    ap.add_argument("--frames", type=int, default=12)
    ap.add_argument("--loops", type=int, default=5)
    ap.add_argument("--no-output", action="store_true", help="skip th_decode_ycbcr_out (no D2H)")
    args = ap.parse_args()
    import torch
    from tests import streamgen
    from theora_amd.decoder import Decoder
    torch.cuda.set_device(0)
    w, h = {"qcif": (176, 144), "cif": (352, 288), "720p": (1280, 720), "1080p": (1920, 1088)}[args.size]
    st = streamgen.Stream(w, h, 0, seed=99)
    hdr = st.header_packets()
    pkts = []
    for f in range(args.frames):
        pkt, truth = st.frame(0 if f % 8 == 0 else 1, density=0.7, p_dc_only=0.5, p_empty=0.2)
        pkts.append(pkt)
    nbytes = sum(len(p) for p in pkts)
    dec = Decoder(hdr)
    for p in pkts:                      # warm-up pass
        dec.packetin(p)
        dec.ycbcr_out()
    t0 = time.perf_counter()
    n = 0
    for _ in range(args.loops):
        for p in pkts:
            dec.packetin(p)
            if not args.no_output:
                dec.ycbcr_out()
            n += 1
    if args.no_output:
        dec.ycbcr_out()
    el = time.perf_counter() - t0
    print(json.dumps({"metric": "end-to-end decode frames/sec (%s 4:2:0, packets in host memory -> YUV in host memory)" % args.size,
                      "value": round(n / el, 2), "unit": "frames/s", "frames": n, "host_threads": 1,
                      "avg_packet_bytes": nbytes // len(pkts), "with_ycbcr_out": not args.no_output,
                      "data": "synthetic packets (tests/streamgen.py)", "note": "host-bound: single-thread entropy decode + PCIe"}))


if __name__ == "__main__":
    main()
