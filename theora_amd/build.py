"""Builds libtheora_hip.so (the C-ABI product library) for gfx950 with hipcc, in-tree."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libtheora_hip.so")
SOURCES = ["thip_decode.hip", "thip_slots.hip", "thip_frontend.cpp", "thip_ogg.cpp"]
# every header under csrc/ and include/ is a dependency (a list kept by hand went stale once: thip_fused.h was edited and the
# library was not rebuilt)
def _headers():
    import glob
    return sorted(glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h")))

def _stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in SOURCES] + _headers() + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def compile_library(out, extra_flags=(), verbose=False):
    """hipcc for gfx950 -> `out`.  The .hip files are device + host code; the .cpp files are host code only (the th_decode_* front
    end, the Ogg demuxer): compiled as plain C++ first -- no device pass over them, and host-only builtins (the front end asks the
    CPU for BMI2) are legal --, the objects linked in with the .hip files' compilation."""
    import tempfile
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = ["-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"] + os.environ.get("THIP_EXTRA_CFLAGS", "").split() + list(extra_flags)
    with tempfile.TemporaryDirectory(prefix="thip_build_") as td:
        objs = []
        for s in SOURCES:
            if s.endswith(".hip"):
                continue
            o = os.path.join(td, os.path.splitext(s)[0] + ".o")
            subprocess.check_call([hipcc, "-x", "c++", "-c"] + flags + ["-o", o, os.path.join(CSRC, s)])
            objs.append(o)
        # (objects first: hipcc's "-x hip" for the .hip files sticks to what follows them)
        cmd = [hipcc, "--offload-arch=gfx950", "-shared"] + flags + ["-o", out] + objs + [os.path.join(CSRC, s) for s in SOURCES if s.endswith(".hip")]
        if verbose:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        subprocess.check_call(cmd)
    return out


def build(force=False, verbose=False):
    if not force and not _stale():
        return OUT
    return compile_library(OUT, verbose=verbose)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
