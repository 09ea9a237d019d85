"""End to end with and without the look-ahead (TH_DECCTL_THIP_PREFETCH_PACKET): packets in host memory -> th_decode_packetin ->
th_decode_ycbcr_out, one context per host thread, the packets of bench.py --mode e2e (cached by tools/fe_stage_cpu.py under
tools/_cache/ so that the GPU box does not spend its minutes in the Python packet generator).  One JSON line per cell.
  python tools/e2e_lookahead.py [sizes=720p,1080p,4k] [kinds=dense,typical] [threads=1,4] [lookaheads=0,2,4]"""
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tools.fe_stage_cpu import packets  # noqa: E402


def main():
    arg = lambda i, d: (sys.argv[i] if len(sys.argv) > i else d).split(",")
    sizes, kinds = arg(1, "720p,1080p,4k"), arg(2, "dense")
    threads, aheads = [int(x) for x in arg(3, "1,4")], [int(x) for x in arg(4, "0,2,4")]
    import torch
    torch.cuda.set_device(0)
    from theora_amd.decoder import Decoder
    for size in sizes:
        for kind in kinds:
            hdr, pk = packets(size, kind, {"4k": 4, "1080p": 8}.get(size, 12))
            loops = {"4k": 6, "1080p": 6}.get(size, 8) * int(os.environ.get("E2E_LOOPS", "1"))   # (option fe_assign = 2 takes ~70 frames to settle)
            for T in threads:
                for la in aheads:
                    decs = [Decoder(hdr) for _ in range(T)]
                    for dec in decs:
                        for p in pk:
                            dec.packetin(p)
                            dec.ycbcr_out()
                    counts = [0] * T

                    def worker(i):
                        dec, seq, nxt = decs[i], pk * loops, 0
                        for k, p in enumerate(seq):
                            while la and nxt < len(seq) and nxt < k + la:
                                nxt = max(nxt, k)
                                if not dec.prefetch(seq[nxt]):
                                    break
                                nxt += 1
                            dec.packetin(p)
                            dec.ycbcr_out()
                            counts[i] += 1

                    t0 = time.perf_counter()
                    ths = [threading.Thread(target=worker, args=(i,)) for i in range(T)]
                    for th in ths:
                        th.start()
                    for th in ths:
                        th.join()
                    el = time.perf_counter() - t0
                    for dec in decs:
                        dec.close()
                    print(json.dumps({"size": size, "packets": kind, "avg_packet_bytes": sum(map(len, pk)) // len(pk), "streams": T,
                                      "lookahead": la, "frames_per_s": round(sum(counts) / el, 1), "frames": sum(counts)}), flush=True)


if __name__ == "__main__":
    main()
