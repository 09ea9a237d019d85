"""Python face of the th_decode_* API exported by libtheora_hip.so (include/theoradec_hip.h):
the calls a libtheoradec user makes, in the order they make them."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import OggPacket, ThComment, ThImgPlane, ThInfo, TheoraHipError

TH_DUPFRAME = 1
TH_DECCTL_THIP_GET_SLOT_TRACE = 0x7101
TH_DECCTL_THIP_PREFETCH_PACKET = 0x7105


class SlotTrace(C.Structure):
    """thip_slot_trace (include/theoradec_hip.h)."""
    _fields_ = [("ncoded", C.c_int64), ("fragi", C.POINTER(C.c_int32)), ("pli", C.POINTER(C.c_uint8)),
                ("last_zzi", C.POINTER(C.c_uint8)), ("refi", C.POINTER(C.c_uint8)),
                ("dc_quant", C.POINTER(C.c_uint16)), ("mv", C.POINTER(C.c_int16)),
                ("coeffs", C.POINTER(C.c_int16)), ("nuncoded", C.c_int64), ("uncoded", C.POINTER(C.c_int64)),
                ("flimit", C.c_int32), ("frame_type", C.c_int32)]


def _packet(data, bos=0, packetno=0):
    buf = (C.c_ubyte * max(len(data), 1)).from_buffer_copy(bytes(data) if len(data) else b"\0")
    op = OggPacket(C.cast(buf, C.c_void_p), len(data), bos, 0, -1, packetno)
    return op, buf


class Decoder:
    """th_decode_headerin x3 -> th_decode_alloc -> {th_decode_packetin, th_decode_ycbcr_out}*."""

    def __init__(self, header_packets, device=None):
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
        self._dec = (L.th_decode_alloc(C.byref(self.info), setup) if device is None
                     else L.th_decode_alloc_on(C.byref(self.info), setup, int(device)))
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

    def prefetch(self, data):
        """TH_DECCTL_THIP_PREFETCH_PACKET: announce a packet that a later packetin() will bring (decode order).  True when
        it was taken; the pictures are the same either way."""
        op, keep = _packet(data)
        rc = self._L.th_decode_ctl(self._dec, TH_DECCTL_THIP_PREFETCH_PACKET, C.byref(op), C.sizeof(op))
        if rc < 0:
            raise TheoraHipError("TH_DECCTL_THIP_PREFETCH_PACKET returned %d" % rc)
        return rc == 0

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

    def slot_trace(self):
        """The accel-vtable slot calls of the last frame as numpy arrays; only on a context
        allocated with THIP_FE_TRACE_BACKEND=1 in the environment (no device needed)."""
        t = SlotTrace()
        rc = self._L.th_decode_ctl(self._dec, TH_DECCTL_THIP_GET_SLOT_TRACE, C.byref(t), C.sizeof(t))
        if rc < 0:
            raise TheoraHipError("TH_DECCTL_THIP_GET_SLOT_TRACE returned %d" % rc)
        n, u = int(t.ncoded), int(t.nuncoded)

        def arr(ptr, count, shape=None):
            if count == 0:
                return np.zeros(shape or (0,), np.ctypeslib.as_array(ptr, (1,)).dtype if ptr else np.int64)
            a = np.ctypeslib.as_array(ptr, (count,)).copy()
            return a.reshape(shape) if shape else a

        return dict(fragi=arr(t.fragi, n), pli=arr(t.pli, n), last_zzi=arr(t.last_zzi, n), refi=arr(t.refi, n),
                    dc_quant=arr(t.dc_quant, n), mv=arr(t.mv, n), coeffs=arr(t.coeffs, n * 64, (n, 64)),
                    uncoded=arr(t.uncoded, u), flimit=int(t.flimit), frame_type=int(t.frame_type))

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


def ogg_packets(data):
    """Packets of a physical Ogg bitstream held in `data` (bytes), in page order, through the
    library's demultiplexer (include/thip_ogg.h): a list of (serialno, payload, b_o_s, e_o_s,
    granulepos, packetno) and the (bad_pages, gaps) counters."""
    L = _lib.load()
    buf = (C.c_ubyte * max(len(data), 1)).from_buffer_copy(bytes(data) if len(data) else b"\0")
    r = L.thip_ogg_open_memory(C.cast(buf, C.c_void_p), len(data))
    if not r:
        raise TheoraHipError("thip_ogg_open_memory failed")
    out = []
    op, serial = OggPacket(), C.c_uint32()
    while L.thip_ogg_next_packet(r, C.byref(op), C.byref(serial)) == 1:
        payload = C.string_at(op.packet, op.bytes) if op.bytes else b""
        out.append((serial.value, payload, int(op.b_o_s), int(op.e_o_s), int(op.granulepos), int(op.packetno)))
    bad, gaps = C.c_int64(), C.c_int64()
    L.thip_ogg_stats(r, C.byref(bad), C.byref(gaps))
    L.thip_ogg_close(r)
    return out, (bad.value, gaps.value)
