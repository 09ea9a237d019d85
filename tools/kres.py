#!/usr/bin/env python3
"""Per-kernel register / LDS / occupancy table from hipcc's -Rpass-analysis=kernel-resource-usage.
usage: python tools/kres.py [substring ...]   (rebuilds the library with the remark switched on)"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from theora_amd import build as B
hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
cmd = [hipcc, "-Rpass-analysis=kernel-resource-usage", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall",
       "-Wno-unused-function"] + os.environ.get("THIP_EXTRA_CFLAGS", "").split() + ["-o", B.OUT] + [os.path.join(B.CSRC, s) for s in B.SOURCES]
r = subprocess.run(cmd, capture_output=True, text=True)
if r.returncode:
    sys.stderr.write(r.stderr[-6000:])
    sys.exit(1)
cur, rows = None, []
for line in r.stderr.splitlines():
    m = re.search(r"remark: [^:]*:\d+:\d+: (?:Function )?Name: (\S+)", line) or re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    m = re.search(r"\s(VGPRs|AGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).split(" ")[0]] = int(m.group(2))
for w in [l for l in r.stderr.splitlines() if "warning" in l]:
    print(w)
pat = sys.argv[1:]
print("%-44s %5s %5s %5s %7s %6s %4s" % ("kernel", "VGPR", "AGPR", "SGPR", "scratch", "LDS", "occ"))
for c in rows:
    n = subprocess.run(["c++filt", c["name"]], capture_output=True, text=True).stdout.strip().split("(")[0]
    if pat and not any(p in n for p in pat):
        continue
    print("%-44s %5d %5d %5d %7d %6d %4d" % (n[:44], c.get("VGPRs", -1), c.get("AGPRs", -1), c.get("TotalSGPRs", -1), c.get("ScratchSize", -1),
                                            c.get("LDS", -1), c.get("Occupancy", -1)))
