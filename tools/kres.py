#!/usr/bin/env python3
"""Per-kernel register / LDS / occupancy table from hipcc's -Rpass-analysis=kernel-resource-usage (theora_amd/build.py records it at
every build).  usage: python tools/kres.py [-f] [substring ...]   (-f: rebuild first)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from theora_amd import build as B
args = [a for a in sys.argv[1:] if a != "-f"]
B.build(force="-f" in sys.argv[1:])
print("%-52s %5s %5s %5s %7s %6s %4s" % ("kernel", "VGPR", "AGPR", "SGPR", "scratch", "LDS", "occ"))
for c in B.resources():
    if args and not any(p in c["name"] for p in args):
        continue
    print("%-52s %5d %5d %5d %7d %6d %4d" % (c["name"][:52], c.get("VGPRs", -1), c.get("AGPRs", -1), c.get("TotalSGPRs", -1),
                                            c.get("ScratchSize", -1), c.get("LDS", -1), c.get("Occupancy", -1)))
