// Fuzz driver for the th_decode_* front end (tests/test_frontend_fuzz.py builds it with
// -fsanitize=address,undefined and links it against theora_amd/csrc/thip_frontend.cpp).
// The HIP backend is not linked: contexts run in slot-trace mode (option fe_trace_backend, which the stub of thip_option below
// answers with 1), so
// the whole host path -- headers, flags, modes, vectors, tokens, DC prediction, dequantisation --
// is exercised on mutated packets.  Any sanitizer report or crash fails the test.
//   fe_fuzz <packet file> <iterations> <seed>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/theora_hip.h"
#include "../../include/theoradec_hip.h"

extern "C" {   // never reached in trace mode; present for the linker only
int thip_state_create(thip_state **, int, int, int) { return -1; }
void thip_state_free(thip_state *) {}
int thip_frame_begin(thip_state *, int) { return -1; }
int thip_state_frag_recon(thip_state *, ptrdiff_t, int, int16_t *, int, uint16_t, int, int16_t) { return -1; }
int thip_frag_copy_list(thip_state *, const ptrdiff_t *, ptrdiff_t) { return -1; }
int thip_state_loop_filter_frag_rows(thip_state *, int, int, int, int, int) { return -1; }
int thip_frame_flush(thip_state *) { return -1; }
int thip_state_ycbcr_out(thip_state *, uint8_t *const *, const int32_t *) { return -1; }
int thip_state_ycbcr_map(thip_state *, const uint8_t **, int32_t *) { return -1; }
int thip_state_ycbcr_map_begin(thip_state *) { return -1; }
int thip_state_ycbcr_map_end(thip_state *, const uint8_t **, int32_t *) { return -1; }
int thip_state_set_eager_output(thip_state *, int) { return -1; }
int thip_state_create_on(thip_state **, int, int, int, int) { return -1; }
int thip_state_postprocess(thip_state *, int, const uint8_t *, const uint8_t *, const int32_t *, const int32_t *) { return -1; }
int thip_state_decode_token_lists(thip_state *, const thip_token_lists *) { return -1; }
int thip_state_token_lists_begin(thip_state *, const thip_token_lists *) { return -1; }
int thip_state_token_lists_begin_assigned(thip_state *, const thip_token_lists *, const uint32_t *, const uint8_t *) { return -1; }
int thip_state_token_lists_finish(thip_state *, const int16_t *) { return -1; }
int thip_state_token_lists_open(thip_state *, const thip_token_lists *) { return -1; }
int thip_state_token_lists_append(thip_state *, int, int, const uint32_t *, int64_t, const uint32_t (*)[64], const uint32_t (*)[64],
                                  const uint32_t (*)[64], const uint32_t (*)[64]) { return -1; }
int thip_state_token_lists_abort(thip_state *) { return -1; }
int thip_state_ring_mark(thip_state *, int64_t *) { return -1; }
int thip_state_ring_rewind(thip_state *, const int64_t *) { return -1; }
int thip_state_token_lists_staging(thip_state *, thip_token_staging *) { return -1; }
int thip_device_count(void) { return 0; }
int thip_option(const char *name) {   // the library's option table is not linked: trace mode on, everything else at its default
  if (name) {   // (THIP_<NAME> in the environment first, as the library's own table has it: tools/fe_tokbench.cpp sets options that way)
    char env[64] = "THIP_";
    size_t k = 5;
    for (const char *c = name; *c && k + 1 < sizeof(env); c++) env[k++] = (char)(*c >= 'a' && *c <= 'z' ? *c - 32 : *c);
    env[k] = 0;
    const char *v = getenv(env);
    if (v && *v) return atoi(v);
  }
  if (name && !strcmp(name, "device")) return -1;
  if (name && !strcmp(name, "fe_trace_backend")) return 1;
  if (name && !strcmp(name, "fe_device_lists")) return 0;
  if (name && !strcmp(name, "fe_lookahead")) return 4;
  if (name && !strcmp(name, "fe_worker_pin")) return 1;
  if (name && !strcmp(name, "fe_assign")) return 2;
  if (name && !strcmp(name, "fe_assign_settle")) return 16;   // (short, so that the fuzzed streams cross the rule's switch points)
  return 0;
}
void thip_option_add(const char *, int) {}   // (the counters live in the table that is not linked)
int thip_state_device(const thip_state *) { return -1; }
int thip_state_set_device_dc(thip_state *, int) { return -1; }
int thip_frame_dequant_table(thip_state *, int, const uint16_t *) { return -1; }
int thip_state_frag_recon_levels(thip_state *, ptrdiff_t, int, int16_t *, int, uint16_t, int, int, int16_t) { return -1; }
int thip_state_frag_recon_tokens(thip_state *, ptrdiff_t, int, const uint32_t *, int, int16_t, int, uint16_t, int, int, int16_t) {
  return -1;
}
}

static uint32_t g_rng;
static uint32_t rnd() { return g_rng = g_rng * 1664525u + 1013904223u; }

typedef std::vector<unsigned char> Pkt;

static void mutate(Pkt &p) {
  if (p.empty()) return;
  switch (rnd() % 6) {
    case 0: for (int i = 0, n = 1 + rnd() % 4; i < n; i++) p[rnd() % p.size()] ^= (unsigned char)(1u << (rnd() % 8)); break;
    case 1: p.resize(rnd() % (p.size() + 1)); break;                                    // truncate
    case 2: for (int i = 0, n = 1 + rnd() % 16; i < n; i++) p[rnd() % p.size()] = (unsigned char)rnd(); break;
    case 3: { size_t a = rnd() % p.size(), n = rnd() % 64; for (size_t i = a; i < p.size() && i < a + n; i++) p[i] = 0xFF; break; }
    case 4: { size_t a = rnd() % p.size(), n = rnd() % 64; for (size_t i = a; i < p.size() && i < a + n; i++) p[i] = 0x00; break; }
    default: p.insert(p.begin() + (long)(rnd() % p.size()), (unsigned char)rnd()); break;  // shift the rest by a byte
  }
}

static ogg_packet as_packet(Pkt &p, int bos) {
  ogg_packet op;
  memset(&op, 0, sizeof(op));
  op.packet = p.empty() ? nullptr : p.data();
  op.bytes = (long)p.size();
  op.b_o_s = bos;
  return op;
}

int main(int argc, char **argv) {
  if (argc < 4) return 2;
  FILE *f = fopen(argv[1], "rb");
  if (!f) return 2;
  unsigned nh = 0, np = 0;
  if (fread(&nh, 4, 1, f) != 1 || fread(&np, 4, 1, f) != 1) return 2;
  std::vector<Pkt> P(nh + np);
  for (auto &p : P) {
    unsigned n = 0;
    if (fread(&n, 4, 1, f) != 1) return 2;
    p.resize(n);
    if (n && fread(p.data(), 1, n, f) != n) return 2;
  }
  fclose(f);
  const int iters = atoi(argv[2]);
  g_rng = (uint32_t)atoi(argv[3]);
  long decoded = 0, rejected = 0, hdr_rejected = 0, compared = 0;
  for (int it = 0; it < iters; it++) {
    const bool fuzz_headers = it % 4 == 3;
    th_info info;
    th_comment tc;
    th_setup_info *setup = nullptr;
    th_info_init(&info);
    th_comment_init(&tc);
    bool ok = true;
    for (unsigned i = 0; i < nh && ok; i++) {
      Pkt h = P[i];
      if (fuzz_headers && rnd() % 2) mutate(h);
      ogg_packet op = as_packet(h, i == 0);
      ok = th_decode_headerin(&info, &tc, &setup, &op) > 0;
    }
    th_dec_ctx *d = ok ? th_decode_alloc(&info, setup) : nullptr;
    th_setup_info *setup_keep = setup;
    if (!d) {
      if (setup) th_setup_free(setup);
      hdr_rejected++;
      th_comment_clear(&tc);
      th_info_clear(&info);
      continue;
    }
    // the context's packets, mutated, up front: on every other iteration a second context decodes the same packets with a
    // look-ahead (TH_DECCTL_THIP_PREFETCH_PACKET, now and then announcing something else than what comes) and must answer the same
    std::vector<Pkt> seq;
    for (unsigned i = nh; i < nh + np; i++) {
      Pkt p = P[i];
      if (!fuzz_headers || rnd() % 2) mutate(p);
      if (rnd() % 16 == 0) p.clear();   // a dropped frame
      seq.push_back(p);
    }
    th_dec_ctx *d2 = (it & 1) ? th_decode_alloc(&info, setup_keep) : nullptr;
    const unsigned ahead = 1 + rnd() % 5;
    size_t announced = 0;
    for (size_t i = 0; i < seq.size(); i++) {
      Pkt &p = seq[i];
      ogg_packet op = as_packet(p, 0);
      int64_t gp = 0;
      const int rc = th_decode_packetin(d, &op, &gp);
      unsigned long sum = 0;
      if (rc < 0) rejected++;
      else {
        decoded++;
        thip_slot_trace t;
        if (th_decode_ctl(d, TH_DECCTL_THIP_GET_SLOT_TRACE, &t, sizeof(t)) != 0) return 3;
        // touch what the trace points to (ASan checks the bounds)
        for (int64_t k = 0; k < t.ncoded; k++)
          sum = sum * 31 + (unsigned long)(t.fragi[k] + t.last_zzi[k] * 3 + t.coeffs[k * 64 + 63] * 5 + t.coeffs[k * 64] * 7 + t.mv[k] * 11 + t.refi[k] * 13);
        for (int64_t k = 0; k < t.nuncoded; k++) sum = sum * 31 + (unsigned long)t.uncoded[k];
      }
      if (d2) {
        while (announced < seq.size() && announced < i + ahead) {
          if (announced < i) announced = i;
          Pkt q = seq[announced];
          if (rnd() % 10 == 0) mutate(q);   // (an announcement that does not match: dropped, the packet parsed the ordinary way)
          ogg_packet oq = as_packet(q, 0);
          const int prc = th_decode_ctl(d2, TH_DECCTL_THIP_PREFETCH_PACKET, &oq, sizeof(oq));
          if (prc < 0) return 5;
          if (prc != 0 && !q.empty() && announced < i + 4) return 6;   // (four slots: refused only when they are full)
          if (prc != 0 && !q.empty()) break;
          announced++;
        }
        int64_t gp2 = 0;
        const int rc2 = th_decode_packetin(d2, &op, &gp2);
        unsigned long sum2 = 0;
        if (rc2 >= 0) {
          thip_slot_trace t;
          if (th_decode_ctl(d2, TH_DECCTL_THIP_GET_SLOT_TRACE, &t, sizeof(t)) != 0) return 3;
          for (int64_t k = 0; k < t.ncoded; k++)
            sum2 = sum2 * 31 + (unsigned long)(t.fragi[k] + t.last_zzi[k] * 3 + t.coeffs[k * 64 + 63] * 5 + t.coeffs[k * 64] * 7 + t.mv[k] * 11 + t.refi[k] * 13);
          for (int64_t k = 0; k < t.nuncoded; k++) sum2 = sum2 * 31 + (unsigned long)t.uncoded[k];
        }
        if (rc2 != rc || (rc >= 0 && (gp2 != gp || sum2 != sum))) {
          printf("look-ahead differs: iteration %d packet %zu rc %d / %d granpos %ld / %ld\n", it, i, rc, rc2, (long)gp, (long)gp2);
          return 4;
        }
        compared++;
      }
    }
    if (d2) th_decode_free(d2);
    th_ycbcr_buffer yb;
    th_decode_ycbcr_out(d, yb);
    th_decode_free(d);
    if (setup_keep) th_setup_free(setup_keep);
    th_comment_clear(&tc);
    th_info_clear(&info);
  }
  printf("fe_fuzz: %d iterations, %ld packets decoded, %ld rejected, %ld header sets rejected\n", iters, decoded, rejected,
         hdr_rejected);
  printf("fe_fuzz: %ld packets compared with a look-ahead context\n", compared);
  return 0;
}
