"""Python face of the th_decode_* API exported by libtheora_hip.so (include/theoradec_hip.h):
the calls a libtheoradec user makes, in the order they make them."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import OggPacket, ThComment, ThImgPlane, ThInfo, TheoraHipError

TH_DUPFRAME = 1


def _packet(data, bos=0, packetno=0):
    buf = (C.c_ubyte * max(len(data), 1)).from_buffer_copy(bytes(data) if len(data) else b"\0")
    op = OggPacket(C.cast(buf, C.c_void_p), len(data), bos, 0, -1, packetno)
    return op, buf


class Decoder:
    """th_decode_headerin x3 -> th_decode_alloc -> {th_decode_packetin, th_decode_ycbcr_out}*."""

    def __init__(self, header_packets):
        L = self._L = _lib.load()
        self.info = ThInfo()
        self.comment = ThComment()
        L.th_info_init(C.byref(self.info))
        L.th_comment_init(C.byref(self.comment))
        setup = C.c_void_p()
        for k, pkt in enumerate(header_packets):
            op, keep = _packet(pkt, bos=1 if k == 0 else 0, packetno=k)
            rc = L.th_decode_headerin(C.byref(self.info), C.byref(self.comment), C.byref(setup), C.byref(op))
            if rc <= 0:
                raise TheoraHipError("th_decode_headerin(packet %d) returned %d" % (k, rc))
        self._dec = L.th_decode_alloc(C.byref(self.info), setup)
        L.th_setup_free(setup)
        if not self._dec:
            raise TheoraHipError("th_decode_alloc failed")
        self._npackets = len(header_packets)

    def packetin(self, data):
        """Returns (rc, granulepos); rc 0 = new frame, TH_DUPFRAME = repeat of the last one."""
        op, keep = _packet(data, packetno=self._npackets)
        self._npackets += 1
        gp = C.c_int64(-1)
        rc = self._L.th_decode_packetin(self._dec, C.byref(op), C.byref(gp))
        if rc < 0:
            raise TheoraHipError("th_decode_packetin returned %d" % rc)
        return rc, gp.value

    def ycbcr_out(self):
        """Three numpy planes, display order (top row first), the full coded frame."""
        buf = (ThImgPlane * 3)()
        rc = self._L.th_decode_ycbcr_out(self._dec, buf)
        if rc < 0:
            raise TheoraHipError("th_decode_ycbcr_out returned %d" % rc)
        out = []
        for p in buf:
            a = np.ctypeslib.as_array(p.data, (p.height, p.stride))[:, :p.width]
            out.append(a.copy())
        return out

    def close(self):
        if getattr(self, "_dec", None):
            self._L.th_decode_free(self._dec)
            self._dec = None
            self._L.th_comment_clear(C.byref(self.comment))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
