"""Stage times of th_decode_packetin on the CPU alone: a context in slot-trace mode (option fe_trace_backend) parses the packets
completely and records the slot calls instead of making them, so no device is needed; option fe_prof prints the wall time per
stage when the context is freed.  The packets are bench.py --mode e2e's (tests/streamgen.py), cached under tools/_cache/.
  python tools/fe_stage_cpu.py [720p|1080p|4k] [dense|typical] [loops] [lookahead]
With a look-ahead of K the packets are announced K ahead (TH_DECCTL_THIP_PREFETCH_PACKET) and parsed on K threads: the stage
table is then the CALLER's thread -- the wait for the parser, the adoption and what follows the packet's last bit."""
import os
import pickle
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
SIZES = {"qcif": (176, 144), "cif": (352, 288), "720p": (1280, 720), "1080p": (1920, 1088), "4k": (3840, 2160)}


def packets(size, kind, frames):
    cache = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_cache")   # (git-ignored; travels with gpurun)
    os.makedirs(cache, exist_ok=True)
    path = os.path.join(cache, "fe_pkts_%s_%s_%d.pkl" % (size, kind, frames))
    if os.path.exists(path):
        return pickle.load(open(path, "rb"))
    from tests import streamgen
    w, h = SIZES[size]
    content = dict(density=0.7, p_dc_only=0.5, p_empty=0.2) if kind == "dense" else dict(density=0.35, p_dc_only=0.45, p_empty=0.4)
    st = streamgen.Stream(w, h, 0, seed=99, trees="matched", probe_kwargs=content)
    hdr = st.header_packets()
    pk = [st.frame(0 if f % 8 == 0 else 1, **content)[0] for f in range(frames)]
    pickle.dump((hdr, pk), open(path, "wb"))
    return hdr, pk


def main():
    size = sys.argv[1] if len(sys.argv) > 1 else "720p"
    kind = sys.argv[2] if len(sys.argv) > 2 else "dense"
    loops = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    ahead = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    frames = {"4k": 4, "1080p": 8}.get(size, 12)
    hdr, pk = packets(size, kind, frames)
    from theora_amd import _lib
    from theora_amd.decoder import Decoder
    L = _lib.load()
    assert L.thip_set_option(b"fe_trace_backend", 1) == 0
    assert L.thip_set_option(b"fe_prof", 1) == 0
    dec = Decoder(hdr)
    for p in pk:
        dec.packetin(p)
    t0 = time.perf_counter()
    seq = pk * loops
    nxt = 0
    for i, p in enumerate(seq):
        while ahead and nxt < len(seq) and nxt < i + ahead and dec.prefetch(seq[nxt]):
            nxt += 1
        nxt = max(nxt, i + 1)
        dec.packetin(p)
    el = time.perf_counter() - t0
    print("%s %s look-ahead %d: %.3f ms per packet (%d bytes avg)" % (size, kind, ahead, el / (loops * len(pk)) * 1e3, sum(map(len, pk)) // len(pk)))
    sys.stdout.flush()
    dec.close()


if __name__ == "__main__":
    main()
