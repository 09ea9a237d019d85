"""examples/decode_bench (no Python between the calls) on the cached e2e packets, with and without --lookahead: writes the packets
into an Ogg file (tests/oggmux.py) and runs the C program.  python tools/native_lookahead.py [sizes] [kinds] [threads] [lookaheads] [pipeline: 0,1]"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from tools.fe_stage_cpu import packets  # noqa: E402
from tests import oggmux  # noqa: E402


def main():
    arg = lambda i, d: (sys.argv[i] if len(sys.argv) > i else d).split(",")
    sizes, kinds = arg(1, "720p,1080p,4k"), arg(2, "dense")
    threads, aheads = [int(x) for x in arg(3, "1,4")], [int(x) for x in arg(4, "0,4")]
    pipes = [int(x) for x in arg(5, "0")]      # 1: decode_bench --pipeline (option fe_pipeline)
    exe = os.path.join(ROOT, "examples", "decode_bench")
    with tempfile.TemporaryDirectory() as td:
        for size in sizes:
            for kind in kinds:
                hdr, pk = packets(size, kind, {"4k": 4, "1080p": 8}.get(size, 12))
                ls = oggmux.LogicalStream(0x7E0)
                for k, hp in enumerate(hdr):
                    ls.add_packet(hp, granulepos=0, flush=(k == 0 or k == len(hdr) - 1))
                reps = {"4k": 8, "1080p": 16}.get(size, 24)
                for rep in range(reps):
                    for k, p in enumerate(pk):
                        ls.add_packet(p, granulepos=rep * len(pk) + k + 1)
                ogv = os.path.join(td, "clip_%s_%s.ogv" % (size, kind))
                with open(ogv, "wb") as f:
                    f.write(b"".join(ls.finish()))
                for T in threads:
                    for la, pipe in [(a_, p_) for a_ in aheads for p_ in pipes if a_ or not p_]:
                        r = subprocess.run([exe, ogv, str(T), os.environ.get("E2E_LOOPS", "2")] + (["--lookahead", str(la)] if la else []) + (["--pipeline"] if pipe else ["--no-pipeline"]),
                                           capture_output=True, text=True, timeout=300)
                        line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "{}"
                        try:
                            d = json.loads(line)
                        except ValueError:
                            d = {"raw": line[:200]}
                        d.update(size_name=size, packets=kind, rc=r.returncode)
                        print(json.dumps(d), flush=True)


if __name__ == "__main__":
    main()
