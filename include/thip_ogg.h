/* thip_ogg.h -- the part of an Ogg demultiplexer a Theora player needs, written from RFC 3533
 * (the reference leans on libogg for this -- ogg_sync_* / ogg_stream_* in
 * examples/dump_video.c:365-470 -- and libogg is not part of this image).  It turns the bytes of
 * a physical Ogg bitstream into the packets of its logical streams:
 *   - page capture on "OggS", stream structure version 0, header-type flags (continued packet,
 *     first page, last page), 64-bit granule position, serial number, page sequence number;
 *   - CRC-32 of every page (polynomial 0x04c11db7, initial value 0, no reflection, no final xor);
 *     pages that fail it are skipped and capture resumes at the next "OggS";
 *   - lacing: segments of 255 continue a packet, a shorter one ends it; packets may span any
 *     number of pages; a page-sequence gap drops the packet that was being assembled;
 *   - multiplexed (interleaved) and chained logical streams: every packet comes with its serial
 *     number, the caller keeps the ones it wants.
 * Not a general libogg replacement: no seeking, no encoder side.
 */
#ifndef THIP_OGG_H
#define THIP_OGG_H
#include <stddef.h>
#include <stdint.h>

#include "theoradec_hip.h" /* ogg_packet */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct thip_ogg_reader thip_ogg_reader;

/* The reader borrows [data, data+size): it must stay valid until thip_ogg_close. */
thip_ogg_reader *thip_ogg_open_memory(const uint8_t *data, size_t size);
/* Reads the whole file into memory first (players stream; a test harness does not need to). */
thip_ogg_reader *thip_ogg_open_file(const char *path);
/* Next packet of any logical stream, in page order.  Returns 1 and fills *op (op->packet points
   into storage owned by the reader, valid until the next call) and *serialno; 0 at the end of
   the data; the fields b_o_s / e_o_s / granulepos / packetno follow libogg's conventions
   (granulepos is -1 except on the last packet that ends on a page). */
int thip_ogg_next_packet(thip_ogg_reader *r, ogg_packet *op, uint32_t *serialno);
/* Pages dropped so far because of a bad checksum or lost capture, and sequence gaps seen. */
void thip_ogg_stats(const thip_ogg_reader *r, int64_t *bad_pages, int64_t *gaps);
void thip_ogg_close(thip_ogg_reader *r);

#ifdef __cplusplus
}
#endif
#endif
