"""Packets for tools/fe_tokbench.cpp: python tools/fe_tokbench.py 720p 6 out.bin [dense|typical]"""
import os
import struct
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import streamgen  # noqa: E402

SIZES = {"qcif": (176, 144), "cif": (352, 288), "720p": (1280, 720), "1080p": (1920, 1088), "4k": (3840, 2160)}
w, h = SIZES[sys.argv[1]]
n = int(sys.argv[2])
content = dict(density=0.7, p_dc_only=0.5, p_empty=0.2) if (len(sys.argv) < 5 or sys.argv[4] == "dense") else dict(density=0.35, p_dc_only=0.45, p_empty=0.4)
st = streamgen.Stream(w, h, 0, seed=99, trees="matched", probe_kwargs=content)
hdr = st.header_packets()
pkts = [st.frame(0 if f == 0 else 1, **content)[0] for f in range(n)]
with open(sys.argv[3], "wb") as f:
    f.write(struct.pack("<II", len(hdr), len(pkts)))
    for p in list(hdr) + list(pkts):
        f.write(struct.pack("<I", len(p)))
        f.write(bytes(p))
print("wrote", sys.argv[3], [len(p) for p in pkts])
