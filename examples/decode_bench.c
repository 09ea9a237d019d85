/* decode_bench -- many independent Theora streams decoded concurrently through th_decode_*, one
 * host thread and one decoder context per stream (the multi-stream server shape of BASELINE.json's
 * batch configuration, end to end: packets in host memory -> frames in host memory).
 *
 *   cc -O2 -pthread -Iinclude examples/decode_bench.c -Ltheora_amd -ltheora_hip -o decode_bench
 *   decode_bench in.ogv <threads> <loops> [--no-output] [--lookahead K] [--pipeline|--no-pipeline] [--devices]
 *
 * --lookahead K: every stream announces its packets K ahead (th_decode_ctl TH_DECCTL_THIP_PREFETCH_PACKET, what a player does
 * with the packets its demultiplexer has queued): the library parses them on threads of its own and th_decode_packetin only
 * hands the frame to the GPU -- one stream is no longer bound by one host thread.
 *
 * The contexts take the node's GPUs in turn (context i on device i mod thip_device_count()): a single
 * stream stays on one GPU, the batch is spread over all of them.
 *
 * Every thread decodes the same packets (read once from the Ogg file) `loops` times; the clock
 * runs from a common start line, after every context exists, to the last thread's finish.
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "thip_ogg.h"
#include "theoradec_hip.h"
#include "theora_hip.h"

typedef struct {
  unsigned char *data;
  long bytes;
} packet;

static packet *g_pkts;
static int g_npkts, g_nhdr, g_loops, g_output, g_ahead;
static pthread_barrier_t g_start;

typedef struct {
  th_dec_ctx *dec;
  long frames;
  int rc;
} worker;

static void *run(void *arg) {
  worker *w = (worker *)arg;
  int l, i;
  const long ndata = g_npkts - g_nhdr, total = ndata * g_loops;
  long cur = 0, nxt = 0;   /* packets decoded / announced so far */
  pthread_barrier_wait(&g_start);
  for (l = 0; l < g_loops && w->rc == 0; l++)
    for (i = g_nhdr; i < g_npkts; i++, cur++) {
      ogg_packet op;
      int64_t gp;
      while (g_ahead && nxt < total && nxt < cur + g_ahead) {
        if (nxt < cur) nxt = cur;
        memset(&op, 0, sizeof(op));
        op.packet = g_pkts[g_nhdr + nxt % ndata].data;
        op.bytes = g_pkts[g_nhdr + nxt % ndata].bytes;
        if (th_decode_ctl(w->dec, TH_DECCTL_THIP_PREFETCH_PACKET, &op, sizeof(op)) != 0 && op.bytes > 0) break;   /* no slot free */
        nxt++;
      }
      memset(&op, 0, sizeof(op));
      op.packet = g_pkts[i].data;
      op.bytes = g_pkts[i].bytes;
      if (th_decode_packetin(w->dec, &op, &gp) < 0) { w->rc = 1; break; }
      if (g_output) {
        th_ycbcr_buffer yb;
        if (th_decode_ycbcr_out(w->dec, yb) < 0) { w->rc = 1; break; }
      }
      w->frames++;
    }
  return NULL;
}

int main(int argc, char **argv) {
  if (argc < 4) {
    fprintf(stderr, "usage: %s in.ogv <threads> <loops> [--no-output] [--lookahead K] [--pipeline|--no-pipeline] [--devices]\n", argv[0]);
    return 1;
  }
  const int nthreads = atoi(argv[2]);
  int pipeline = -1;       /* -1: the library's default (on since round 6) */
  int check_devices = 0;   /* --devices: every context must sit on the GPU its turn gives it (context i on device i mod the node's GPUs) */
  g_loops = atoi(argv[3]);
  g_output = 1;
  for (int a = 4; a < argc; a++) {
    if (!strcmp(argv[a], "--no-output")) g_output = 0;
    else if (!strcmp(argv[a], "--lookahead") && a + 1 < argc) g_ahead = atoi(argv[++a]);
    else if (!strcmp(argv[a], "--devices")) check_devices = 1;
    else if (!strcmp(argv[a], "--pipeline")) pipeline = 1;   /* option fe_pipeline: the next announced frame goes to the device inside th_decode_ycbcr_out */
    else if (!strcmp(argv[a], "--no-pipeline")) pipeline = 0;
  }
  if (pipeline >= 0) thip_set_option("fe_pipeline", pipeline);
  if (g_ahead > 8) thip_set_option("fe_lookahead", g_ahead > 16 ? 16 : g_ahead);   /* (the default takes eight announcements at a time) */
  thip_ogg_reader *og = thip_ogg_open_file(argv[1]);
  if (!og || nthreads < 1 || g_loops < 1) return 1;
  /* all packets of the first logical stream */
  int cap = 64;
  uint32_t serial0 = 0, serial;
  ogg_packet op;
  g_pkts = (packet *)malloc(sizeof(packet) * (size_t)cap);
  while (thip_ogg_next_packet(og, &op, &serial) == 1) {
    if (g_npkts == 0) serial0 = serial;
    if (serial != serial0) continue;
    if (g_npkts == cap) g_pkts = (packet *)realloc(g_pkts, sizeof(packet) * (size_t)(cap *= 2));
    g_pkts[g_npkts].data = (unsigned char *)malloc((size_t)op.bytes + 1);
    memcpy(g_pkts[g_npkts].data, op.packet, (size_t)op.bytes);
    g_pkts[g_npkts].bytes = op.bytes;
    g_npkts++;
  }
  thip_ogg_close(og);
  th_info ti;
  th_comment tc;
  th_setup_info *ts = NULL;
  th_info_init(&ti);
  th_comment_init(&tc);
  for (g_nhdr = 0; g_nhdr < g_npkts; g_nhdr++) {
    memset(&op, 0, sizeof(op));
    op.packet = g_pkts[g_nhdr].data;
    op.bytes = g_pkts[g_nhdr].bytes;
    op.b_o_s = g_nhdr == 0;
    if (th_decode_headerin(&ti, &tc, &ts, &op) <= 0) break;
  }
  if (g_nhdr < 3 || g_nhdr >= g_npkts) {
    fprintf(stderr, "not a Theora stream\n");
    return 1;
  }
  worker *w = (worker *)calloc((size_t)nthreads, sizeof(worker));
  pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
  int i;
  const int ndev = thip_device_count();
  for (i = 0; i < nthreads; i++) {
    w[i].dec = th_decode_alloc_on(&ti, ts, ndev > 0 ? i % ndev : -1);
    if (!w[i].dec) {
      fprintf(stderr, "th_decode_alloc failed for stream %d\n", i);
      return 1;
    }
    if (check_devices) {
      int dev = -1;
      if (th_decode_ctl(w[i].dec, TH_DECCTL_THIP_GET_DEVICE, &dev, sizeof(dev)) != 0 || dev != (ndev > 0 ? i % ndev : dev)) {
        fprintf(stderr, "--devices: context %d sits on device %d, its turn is device %d of %d\n", i, dev, ndev > 0 ? i % ndev : -1, ndev);
        return 2;
      }
      fprintf(stderr, "context %d -> device %d\n", i, dev);
    }
    /* one untimed frame per context: device buffers, streams and staging come into being here */
    memset(&op, 0, sizeof(op));
    op.packet = g_pkts[g_nhdr].data;
    op.bytes = g_pkts[g_nhdr].bytes;
    th_decode_packetin(w[i].dec, &op, NULL);
  }
  th_setup_free(ts);
  pthread_barrier_init(&g_start, NULL, (unsigned)nthreads + 1);
  for (i = 0; i < nthreads; i++) pthread_create(&th[i], NULL, run, &w[i]);
  struct timespec t0, t1;
  pthread_barrier_wait(&g_start);
  clock_gettime(CLOCK_MONOTONIC, &t0);
  long frames = 0;
  int bad = 0;
  for (i = 0; i < nthreads; i++) {
    pthread_join(th[i], NULL);
    frames += w[i].frames;
    bad |= w[i].rc;
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  const double el = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
  printf("{\"streams\": %d, \"host_threads\": %d, \"frames\": %ld, \"seconds\": %.4f, \"frames_per_s\": %.1f, "
         "\"size\": \"%ux%u\", \"gpus\": %d, \"with_ycbcr_out\": %s, \"lookahead\": %d, \"pipeline\": %d, \"ok\": %s}\n",
         nthreads, nthreads, frames, el, el > 0 ? (double)frames / el : 0.0, (unsigned)ti.frame_width, (unsigned)ti.frame_height,
         ndev < nthreads ? ndev : nthreads,
         g_output ? "true" : "false", g_ahead, pipeline, bad ? "false" : "true");
  for (i = 0; i < nthreads; i++) th_decode_free(w[i].dec);
  th_comment_clear(&tc);
  th_info_clear(&ti);
  return bad;
}
