"""Builds libtheora_hip.so (the C-ABI product library) for gfx950 with hipcc, in-tree."""
import os
import re
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
        # the resource remark costs nothing and is the build's own check: every kernel's registers / LDS / scratch go into
        # <out>.resources.json, and tests/test_abi.py fails on any kernel that owns scratch (k_recon_lf<false>
        # picked up 16 bytes a lane unnoticed)
        cmd = [hipcc, "-Rpass-analysis=kernel-resource-usage", "--offload-arch=gfx950", "-shared"] + flags + ["-o", out] + objs + \
              [os.path.join(CSRC, s) for s in SOURCES if s.endswith(".hip")]
        r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
        rows = parse_resource_remarks(r.stderr)
        other = [l for l in r.stderr.splitlines() if "kernel-resource-usage" not in l and not re.match(r"^\s*(\d+ \||\| )", l)]
        if r.returncode or verbose:
            sys.stderr.write(r.stderr if verbose else "\n".join(other) + "\n")
        elif any("warning" in l or "error" in l for l in other):
            sys.stderr.write("\n".join(other) + "\n")
        if r.returncode:
            raise subprocess.CalledProcessError(r.returncode, cmd)
        import json
        with open(out + ".resources.json", "w") as f:
            json.dump(rows, f, indent=0)
    return out


def parse_resource_remarks(text):
    """hipcc -Rpass-analysis=kernel-resource-usage -> [{name, VGPRs, AGPRs, TotalSGPRs, ScratchSize, Occupancy, LDS}] (demangled)."""
    cur, rows = None, []
    for line in text.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"mangled": m.group(1)}
            rows.append(cur)
            continue
        m = re.search(r"\s(VGPRs|AGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).split(" ")[0]] = int(m.group(2))
    if rows:
        try:
            names = subprocess.run(["c++filt"], input="\n".join(c["mangled"] for c in rows), capture_output=True, text=True).stdout.splitlines()
        except OSError:
            names = [c["mangled"] for c in rows]
        for c, n in zip(rows, names):
            c["name"] = n.split("(")[0].replace("void ", "")
    return rows


def resources():
    """The table written by the last build of the in-tree library (built now if there is none)."""
    import json
    if not os.path.exists(OUT + ".resources.json"):
        build(force=True)
    return json.load(open(OUT + ".resources.json"))


def build(force=False, verbose=False):
    if not force and not _stale():
        return OUT
    return compile_library(OUT, verbose=verbose)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
