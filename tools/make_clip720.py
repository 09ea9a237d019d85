import sys, os
sys.path.insert(0, ".")
from tests import streamgen, oggmux
st = streamgen.Stream(1280, 720, 0, seed=99)
hdr = st.header_packets()
ls = oggmux.LogicalStream(0x7E0)
for k, hp in enumerate(hdr):
    ls.add_packet(hp, granulepos=0, flush=(k == 0 or k == len(hdr) - 1))
for f in range(12):
    pk, _ = st.frame(0 if f % 8 == 0 else 1, density=0.7, p_dc_only=0.5, p_empty=0.2)
    ls.add_packet(pk, granulepos=f + 1)
os.makedirs("gpurun_out", exist_ok=True)
open("gpurun_out/clip720.ogv", "wb").write(b"".join(ls.finish()))
