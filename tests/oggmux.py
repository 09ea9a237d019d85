"""A small Ogg multiplexer (RFC 3533) for the tests: wraps packets of one or more logical
streams into pages -- lacing, packets spanning pages, continued/first/last flags, granule
positions, page sequence numbers, CRC-32 (polynomial 0x04c11db7) -- so that the library's own
demultiplexer (include/thip_ogg.h) and examples/dump_video_hip.c can be exercised without libogg."""
import struct

_CRC = []
for _i in range(256):
    _r = _i << 24
    for _ in range(8):
        _r = ((_r << 1) ^ 0x04C11DB7) & 0xFFFFFFFF if _r & 0x80000000 else (_r << 1) & 0xFFFFFFFF
    _CRC.append(_r)


def crc32_ogg(data):
    crc = 0
    for b in data:
        crc = ((crc << 8) & 0xFFFFFFFF) ^ _CRC[((crc >> 24) ^ b) & 0xFF]
    return crc


def make_page(serial, seq, flags, granulepos, lacing, body):
    hdr = b"OggS" + bytes([0, flags]) + struct.pack("<q", granulepos) + struct.pack("<III", serial, seq, 0)
    hdr += bytes([len(lacing)]) + bytes(lacing)
    page = bytearray(hdr + body)
    struct.pack_into("<I", page, 22, crc32_ogg(page))
    return bytes(page)


class LogicalStream:
    """Pages of one logical stream.  `max_segs` caps the segments per page (small values force
    packets to span pages); add_packet(flush=True) ends the page after the packet (libogg's
    ogg_stream_flush, used after the header packets)."""

    def __init__(self, serial, max_segs=255):
        self.serial, self.max_segs = serial, max_segs
        self.pages = []
        self.seq = 0
        self._lacing, self._body = [], bytearray()
        self._gp = -1                 # granulepos of the last packet completed on the page being built
        self._continued = False       # the page being built starts in the middle of a packet
        self._first = True

    def _emit(self, eos=False):
        if not self._lacing:
            return
        flags = (1 if self._continued else 0) | (2 if self._first else 0) | (4 if eos else 0)
        self.pages.append(make_page(self.serial, self.seq, flags, self._gp, self._lacing, bytes(self._body)))
        self.seq += 1
        self._first = False
        self._lacing, self._body, self._gp = [], bytearray(), -1

    def add_packet(self, data, granulepos=-1, flush=False, eos=False):
        data = bytes(data)
        segs = [255] * (len(data) // 255) + [len(data) % 255]
        off = 0
        for i, s in enumerate(segs):
            if len(self._lacing) >= self.max_segs:
                self._emit()
                self._continued = i > 0
            elif not self._lacing:
                self._continued = i > 0
            self._lacing.append(s)
            self._body += data[off:off + s]
            off += s
        self._gp = granulepos
        if flush or eos:
            self._emit(eos=eos)

    def finish(self):
        self._emit(eos=True)
        return self.pages


def interleave(*page_lists):
    """Round-robin interleaving of the pages of several logical streams; all first pages come
    first, as RFC 3533 section 4 requires of grouped streams."""
    out = [pl[0] for pl in page_lists if pl]
    rest = [list(pl[1:]) for pl in page_lists]
    while any(rest):
        for r in rest:
            if r:
                out.append(r.pop(0))
    return b"".join(out)
