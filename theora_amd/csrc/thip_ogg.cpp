// thip_ogg.cpp -- Ogg page capture, checksum and packet reassembly (include/thip_ogg.h), from
// RFC 3533.  Host only.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <vector>

#include "../../include/thip_ogg.h"

namespace {

uint32_t g_crc[256];
bool g_crc_ready = false;
void crc_init() {   // RFC 3533 section 6, field 22: generator 0x04c11db7, direct (MSB first)
  for (uint32_t i = 0; i < 256; i++) {
    uint32_t r = i << 24;
    for (int b = 0; b < 8; b++) r = (r & 0x80000000u) ? (r << 1) ^ 0x04c11db7u : r << 1;
    g_crc[i] = r;
  }
  g_crc_ready = true;
}
uint32_t crc_update(uint32_t crc, const uint8_t *p, size_t n) {
  for (size_t i = 0; i < n; i++) crc = (crc << 8) ^ g_crc[((crc >> 24) ^ p[i]) & 0xFF];
  return crc;
}

struct Logical {
  std::vector<uint8_t> partial;   // packet being assembled across pages
  bool assembling = false;
  bool discard = false;           // the packet in progress lost its head or a middle page: drop it whole
  bool seen_page = false;
  uint32_t next_seq = 0;
  int64_t packetno = 0;
};

struct Pending {   // a completed packet waiting to be handed out
  std::vector<uint8_t> data;
  uint32_t serial;
  long bos, eos;
  int64_t granulepos, packetno;
};

}  // namespace

struct thip_ogg_reader {
  const uint8_t *data;
  size_t size, pos;
  std::vector<uint8_t> owned;     // open_file
  std::map<uint32_t, Logical> streams;
  std::vector<Pending> queue;     // packets of the current page, in order
  size_t qpos;
  std::vector<uint8_t> current;   // storage behind the packet last returned
  int64_t bad_pages, gaps;
};

namespace {

// Parses the page at r->pos if a whole valid one is there.  Returns its length, 0 if the data ends
// inside it, -1 if this is not a valid page (caller re-synchronises one byte further).
long parse_page(thip_ogg_reader *r) {
  const uint8_t *p = r->data + r->pos;
  const size_t left = r->size - r->pos;
  if (left < 27) return 0;
  if (memcmp(p, "OggS", 4) != 0 || p[4] != 0) return -1;
  const int nsegs = p[26];
  if (left < (size_t)27 + nsegs) return 0;
  size_t body = 0;
  for (int i = 0; i < nsegs; i++) body += p[27 + i];
  const size_t total = 27 + (size_t)nsegs + body;
  if (left < total) return 0;
  // checksum with the CRC field taken as zero
  uint32_t crc = crc_update(0, p, 22);
  const uint8_t zero[4] = {0, 0, 0, 0};
  crc = crc_update(crc, zero, 4);
  crc = crc_update(crc, p + 26, total - 26);
  const uint32_t want = (uint32_t)p[22] | (uint32_t)p[23] << 8 | (uint32_t)p[24] << 16 | (uint32_t)p[25] << 24;
  if (crc != want) return -1;

  const int flags = p[5];
  int64_t gp = 0;
  for (int i = 7; i >= 0; i--) gp = (int64_t)(((uint64_t)gp << 8) | p[6 + i]);
  const uint32_t serial = (uint32_t)p[14] | (uint32_t)p[15] << 8 | (uint32_t)p[16] << 16 | (uint32_t)p[17] << 24;
  const uint32_t seq = (uint32_t)p[18] | (uint32_t)p[19] << 8 | (uint32_t)p[20] << 16 | (uint32_t)p[21] << 24;
  Logical &L = r->streams[serial];
  if (L.seen_page && seq != L.next_seq) {   // lost page(s): whatever was being assembled is incomplete
    r->gaps++;
    L.partial.clear();
    L.assembling = false;
  }
  L.seen_page = true;
  L.next_seq = seq + 1;
  const bool continued = (flags & 1) != 0;
  if (continued) {
    if (!L.assembling) L.discard = true;                  // the head of this packet was lost
  } else {
    if (L.assembling) L.partial.clear();                  // the tail of the previous one never came
    L.assembling = false;
    L.discard = false;
  }
  const uint8_t *bp = p + 27 + nsegs;
  const size_t first_in_queue = r->queue.size();
  for (int i = 0; i < nsegs; i++) {
    const int len = p[27 + i];
    if (!L.discard) L.partial.insert(L.partial.end(), bp, bp + len);
    L.assembling = true;
    bp += len;
    if (len < 255) {   // packet boundary
      if (!L.discard) {
        Pending q;
        q.data.swap(L.partial);
        q.serial = serial;
        q.bos = 0;
        q.eos = 0;
        q.granulepos = -1;
        q.packetno = L.packetno++;
        r->queue.push_back(std::move(q));
      }
      L.partial.clear();
      L.assembling = false;
      L.discard = false;
    }
  }
  // libogg's conventions: b_o_s on the packets of a first page, e_o_s and the page's granule
  // position on the last packet that ENDS on the page
  if (r->queue.size() > first_in_queue) {
    if (flags & 2)
      for (size_t i = first_in_queue; i < r->queue.size(); i++) r->queue[i].bos = 1;
    Pending &last = r->queue.back();
    last.granulepos = gp;
    if (flags & 4) last.eos = 1;
  }
  return (long)total;
}

}  // namespace

extern "C" {

thip_ogg_reader *thip_ogg_open_memory(const uint8_t *data, size_t size) {
  if (!data && size) return nullptr;
  if (!g_crc_ready) crc_init();
  thip_ogg_reader *r = new thip_ogg_reader();
  r->data = data;
  r->size = size;
  r->pos = 0;
  r->qpos = 0;
  r->bad_pages = r->gaps = 0;
  return r;
}

thip_ogg_reader *thip_ogg_open_file(const char *path) {
  if (!path) return nullptr;
  FILE *f = fopen(path, "rb");
  if (!f) return nullptr;
  std::vector<uint8_t> buf;
  uint8_t tmp[65536];
  size_t n;
  while ((n = fread(tmp, 1, sizeof(tmp), f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
  fclose(f);
  thip_ogg_reader *r = thip_ogg_open_memory(buf.empty() ? (const uint8_t *)"" : buf.data(), buf.size());
  if (!r) return nullptr;
  r->owned.swap(buf);
  r->data = r->owned.data();
  return r;
}

int thip_ogg_next_packet(thip_ogg_reader *r, ogg_packet *op, uint32_t *serialno) {
  if (!r || !op) return -1;
  while (r->qpos >= r->queue.size()) {
    r->queue.clear();
    r->qpos = 0;
    // capture the next page
    bool lost = false;
    for (;;) {
      if (r->pos >= r->size) return 0;
      const long n = parse_page(r);
      if (n > 0) {
        r->pos += (size_t)n;
        break;
      }
      if (n == 0) return 0;   // truncated tail
      if (!lost) {
        r->bad_pages++;
        lost = true;
      }
      // re-synchronise: next "OggS" after this byte
      const uint8_t *q = (const uint8_t *)memchr(r->data + r->pos + 1, 'O', r->size - r->pos - 1);
      r->pos = q ? (size_t)(q - r->data) : r->size;
    }
  }
  Pending &q = r->queue[r->qpos++];
  r->current.swap(q.data);
  memset(op, 0, sizeof(*op));
  op->packet = r->current.empty() ? nullptr : r->current.data();
  op->bytes = (long)r->current.size();
  op->b_o_s = q.bos;
  op->e_o_s = q.eos;
  op->granulepos = q.granulepos;
  op->packetno = q.packetno;
  if (serialno) *serialno = q.serial;
  return 1;
}

void thip_ogg_stats(const thip_ogg_reader *r, int64_t *bad_pages, int64_t *gaps) {
  if (!r) return;
  if (bad_pages) *bad_pages = r->bad_pages;
  if (gaps) *gaps = r->gaps;
}

void thip_ogg_close(thip_ogg_reader *r) { delete r; }

}  // extern "C"
