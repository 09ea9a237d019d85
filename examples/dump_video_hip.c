/* dump_video_hip -- what the reference's examples/dump_video.c does (Ogg file -> decoded frames as
 * YUV4MPEG2 or raw planes), written against this repository's library only: thip_ogg.h for the
 * container (the reference uses libogg), theoradec_hip.h for th_decode_*.
 *
 *   cc -Iinclude examples/dump_video_hip.c -Ltheora_amd -ltheora_hip -o dump_video_hip
 *   dump_video_hip [-o out.y4m] [-c|--crop] [-r|--raw] [-f|--fps-only] [--lookahead K] in.ogv
 *
 * The tool reads K (default 8, 0: none) data packets ahead of the one it decodes and announces each to the library as it
 * comes off the demultiplexer (th_decode_ctl TH_DECCTL_THIP_PREFETCH_PACKET): their entropy decoding runs on the library's
 * parser threads while th_decode_packetin / th_decode_ycbcr_out of the current frame run here.  The frames are the same.
 *
 * Same observable behaviour as the reference tool for one Theora stream: the first logical stream
 * whose first packet is a Theora identification header is decoded, other streams are skipped;
 * the YUV4MPEG2 header line has the reference's form; a frame is written for every data packet,
 * duplicate frames included; without --crop the full coded frame is written.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "thip_ogg.h"
#include "theoradec_hip.h"

/* One frame: every plane's rectangle, row by row.  The rectangle is the coded frame, or the picture
   region with --crop; chroma rectangles follow from the luma one by the format's decimation, rounding
   the far edge up (dump_video.c:206-241 writes the same bytes). */
static void write_frame(FILE *out, const th_info *ti, th_ycbcr_buffer yb, int crop, int raw) {
  const int left = crop ? (int)ti->pic_x : 0, top = crop ? (int)ti->pic_y : 0;
  const int right = crop ? left + (int)ti->pic_width : (int)ti->frame_width;
  const int bottom = crop ? top + (int)ti->pic_height : (int)ti->frame_height;
  int pli;
  if (!raw) fputs("FRAME\n", out);
  for (pli = 0; pli < 3; pli++) {
    const int sx = pli && !(ti->pixel_fmt & 1), sy = pli && !(ti->pixel_fmt & 2); /* log2 decimation of this plane */
    const int col0 = left >> sx, ncols = ((right + sx) >> sx) - col0;
    const int row1 = (bottom + sy) >> sy;
    int row;
    for (row = top >> sy; row < row1; row++)
      fwrite(yb[pli].data + (size_t)row * (size_t)yb[pli].stride + (size_t)col0, 1, (size_t)ncols, out);
  }
}

/* the data packets read ahead of the one being decoded (copies: the demultiplexer's buffer is its own) */
#define QMAX 17
typedef struct {
  unsigned char *data;
  long bytes, cap;
} queued_packet;
static queued_packet g_q[QMAX];
static int g_qhead, g_qcount;

static void queue_push(th_dec_ctx *td, const ogg_packet *op, int announce) {
  queued_packet *q = &g_q[(g_qhead + g_qcount) % QMAX];
  if (op->bytes > q->cap) {
    q->cap = op->bytes + 4096;
    q->data = (unsigned char *)realloc(q->data, (size_t)q->cap);
    if (!q->data) exit(1);
  }
  q->bytes = op->bytes > 0 ? op->bytes : 0;
  if (q->bytes) memcpy(q->data, op->packet, (size_t)q->bytes);
  g_qcount++;
  if (announce && q->bytes) {
    ogg_packet a;
    memset(&a, 0, sizeof(a));
    a.packet = q->data;
    a.bytes = q->bytes;
    (void)th_decode_ctl(td, TH_DECCTL_THIP_PREFETCH_PACKET, &a, sizeof(a));   /* 1 = not taken: parsed in its turn, as ever */
  }
}

/* the oldest queued packet through th_decode_packetin (+ th_decode_ycbcr_out and the output file) */
static int decode_head(th_dec_ctx *td, FILE *out, const th_info *ti, int crop, int raw, long *frames) {
  int64_t gp = -1;
  ogg_packet cur;
  memset(&cur, 0, sizeof(cur));
  cur.packet = g_q[g_qhead].data;
  cur.bytes = g_q[g_qhead].bytes;
  g_qhead = (g_qhead + 1) % QMAX;
  g_qcount--;
  if (th_decode_packetin(td, &cur, &gp) < 0) return 0;   /* undecodable packet: no frame, like the reference */
  ++*frames;
  if (out) {
    th_ycbcr_buffer yb;
    if (th_decode_ycbcr_out(td, yb) < 0) return -1;
    write_frame(out, ti, yb, crop, raw);
  }
  return 0;
}

int main(int argc, char **argv) {
  const char *in = NULL, *outname = NULL;
  int crop = 0, raw = 0, fps_only = 0, lookahead = 8, i;
  for (i = 1; i < argc; i++) {
    if (!strcmp(argv[i], "-o") && i + 1 < argc) outname = argv[++i];
    else if (!strcmp(argv[i], "-c") || !strcmp(argv[i], "--crop")) crop = 1;
    else if (!strcmp(argv[i], "-r") || !strcmp(argv[i], "--raw")) raw = 1;
    else if (!strcmp(argv[i], "-f") || !strcmp(argv[i], "--fps-only")) fps_only = 1;
    else if (!strcmp(argv[i], "--lookahead") && i + 1 < argc) lookahead = atoi(argv[++i]);
    else if (argv[i][0] != '-') in = argv[i];
    else {
      fprintf(stderr, "usage: %s [-o out.y4m] [--crop] [--raw] [--fps-only] [--lookahead K] in.ogv\n", argv[0]);
      return 1;
    }
  }
  if (!in) {
    fprintf(stderr, "usage: %s [-o out.y4m] [--crop] [--raw] [--fps-only] [--lookahead K] in.ogv\n", argv[0]);
    return 1;
  }
  if (lookahead < 0) lookahead = 0;
  if (lookahead > QMAX - 1) lookahead = QMAX - 1;
  thip_ogg_reader *og = thip_ogg_open_file(in);
  if (!og) {
    fprintf(stderr, "cannot read %s\n", in);
    return 1;
  }
  FILE *out = NULL;
  if (!fps_only) {
    out = outname ? fopen(outname, "wb") : stdout;
    if (!out) {
      fprintf(stderr, "cannot write %s\n", outname);
      return 1;
    }
  }
  th_info ti;
  th_comment tc;
  th_setup_info *ts = NULL;
  th_dec_ctx *td = NULL;
  th_info_init(&ti);
  th_comment_init(&tc);
  int have_stream = 0, headers_done = 0;
  uint32_t theora_serial = 0;
  long frames = 0;
  ogg_packet op;
  uint32_t serial;
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  while (thip_ogg_next_packet(og, &op, &serial) == 1) {
    if (!have_stream) {
      /* the first Theora identification header picks the stream (dump_video.c:383-407) */
      if (!op.b_o_s) continue;
      th_info probe_i;
      th_comment probe_c;
      th_setup_info *probe_s = NULL;
      th_info_init(&probe_i);
      th_comment_init(&probe_c);
      if (th_decode_headerin(&probe_i, &probe_c, &probe_s, &op) > 0) {
        ti = probe_i;
        tc = probe_c;
        ts = probe_s;
        theora_serial = serial;
        have_stream = 1;
      } else {
        th_comment_clear(&probe_c);
        th_info_clear(&probe_i);
      }
      continue;
    }
    if (serial != theora_serial) continue;
    if (!headers_done) {
      const int rc = th_decode_headerin(&ti, &tc, &ts, &op);
      if (rc > 0) continue;   /* another header packet consumed */
      if (rc < 0) {
        fprintf(stderr, "error parsing Theora stream headers (%d)\n", rc);
        return 1;
      }
      /* rc == 0: first data packet */
      td = th_decode_alloc(&ti, ts);
      th_setup_free(ts);
      ts = NULL;
      if (!td) {
        fprintf(stderr, "th_decode_alloc failed\n");
        return 1;
      }
      headers_done = 1;
      clock_gettime(CLOCK_MONOTONIC, &t0); /* the rate below is of the decode loop, not of device start-up */
      if (out && !raw) {
        static const char *const chroma[4] = {"420jpeg", NULL, "422jpeg", "444"};
        int w = (int)ti.frame_width, h = (int)ti.frame_height;
        if (crop) {
          const int hdec = !(ti.pixel_fmt & 1), vdec = !(ti.pixel_fmt & 2);
          if ((hdec && ((ti.pic_x & 1) || (ti.pic_width & 1))) || (vdec && ((ti.pic_y & 1) || (ti.pic_height & 1)))) {
            fprintf(stderr, "cropped images with odd offsets/sizes and chroma subsampling cannot be output to YUV4MPEG2\n");
            return 1;
          }
          w = (int)ti.pic_width;
          h = (int)ti.pic_height;
        }
        fprintf(out, "YUV4MPEG2 C%s W%d H%d F%d:%d I%c A%d:%d\n", chroma[ti.pixel_fmt], w, h, (int)ti.fps_numerator,
                (int)ti.fps_denominator, 'p', (int)ti.aspect_numerator, (int)ti.aspect_denominator);
      }
    }
    queue_push(td, &op, lookahead > 0);
    /* (while fewer than `lookahead` packets wait behind the oldest, keep reading: they are being parsed meanwhile) */
    if (g_qcount > lookahead && decode_head(td, out, &ti, crop, raw, &frames) < 0) return 1;
  }
  while (td && g_qcount)   /* the file has ended: what was read ahead */
    if (decode_head(td, out, &ti, crop, raw, &frames) < 0) return 1;
  clock_gettime(CLOCK_MONOTONIC, &t1);
  {
    int64_t bad = 0, gaps = 0;
    const double el = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    thip_ogg_stats(og, &bad, &gaps);
    fprintf(stderr, "%ld frames in %.3f s (%.1f fps); %lld damaged pages, %lld gaps\n", frames, el, el > 0 ? frames / el : 0.0,
            (long long)bad, (long long)gaps);
  }
  if (td) th_decode_free(td);
  if (ts) th_setup_free(ts);
  th_comment_clear(&tc);
  th_info_clear(&ti);
  thip_ogg_close(og);
  if (out && out != stdout) fclose(out);
  return have_stream ? 0 : 1;
}
