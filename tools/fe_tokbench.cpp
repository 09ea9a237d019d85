// Token-loop timing without a GPU: the th_decode_* front end in slot-trace mode (the stubs of tests/native/fe_fuzz.cpp) over a
// packet file written by tools/fe_tokbench.py, the same packets decoded `loops` times; THIP_FE_PROF=1 prints the front end's own
// stage table (ms per frame and tokens per frame of the token stage) at th_decode_free.
//   g++ -O2 -std=c++17 -Iinclude tools/fe_tokbench.cpp theora_amd/csrc/thip_frontend.cpp -lpthread -o tools/_build/tok/bench
//   THIP_FE_PROF=1 tools/_build/tok/bench pkts.bin 20 [packets announced ahead; THIP_FE_LOOKAHEAD=8 THIP_FE_ASSIGN=1: eight parsers that pair]
#define main fe_fuzz_main
#include "../tests/native/fe_fuzz.cpp"
#undef main
#include <chrono>

int main(int argc, char **argv) {
  if (argc < 3) return 2;
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
  const int loops = atoi(argv[2]);
  const int ahead = argc > 3 ? atoi(argv[3]) : 0;   // packets announced ahead (TH_DECCTL_THIP_PREFETCH_PACKET; THIP_FE_LOOKAHEAD parsers)
  th_info info;
  th_comment tc;
  th_setup_info *setup = nullptr;
  th_info_init(&info);
  th_comment_init(&tc);
  for (unsigned i = 0; i < nh; i++) {
    ogg_packet op = as_packet(P[i], i == 0);
    if (th_decode_headerin(&info, &tc, &setup, &op) <= 0) return 3;
  }
  th_dec_ctx *d = th_decode_alloc(&info, setup);
  if (!d) return 4;
  unsigned long sum = 0;
  long frames = 0;
  const auto t0 = std::chrono::steady_clock::now();
  const long total = (long)loops * np;
  long announced = 0;
  for (int l = 0; l < loops; l++)
    for (unsigned i = nh; i < nh + np; i++) {
      while (ahead && announced < total && announced < frames + ahead) {
        if (announced < frames) announced = frames;
        ogg_packet oq = as_packet(P[nh + announced % np], 0);
        if (th_decode_ctl(d, TH_DECCTL_THIP_PREFETCH_PACKET, &oq, sizeof(oq)) != 0) break;
        announced++;
      }
      ogg_packet op = as_packet(P[i], 0);
      int64_t gp = 0;
      const int rc = th_decode_packetin(d, &op, &gp);
      if (rc < 0) return 5;
      thip_slot_trace t;
      if (th_decode_ctl(d, TH_DECCTL_THIP_GET_SLOT_TRACE, &t, sizeof(t)) != 0) return 6;
      sum = sum * 31 + (unsigned long)t.ncoded;
      frames++;
    }
  const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  printf("fe_tokbench: %ld frames, %.3f ms/frame (whole packetin in slot-trace mode), checksum %lu\n", frames, 1e3 * s / frames, sum);
  th_decode_free(d);
  th_setup_free(setup);
  return 0;
}
