// Which part of k_recon's memory behaviour costs time?  Copies 4 "frames" (3840x3232 bytes,
// one 128x32-pixel tile per wave, one 8x8 block per lane in Hilbert order) under variations:
//   HOP   0/1  block address depends on a per-lane command word loaded first
//   WPE   waves/SIMD cap (amdgpu_waves_per_eu)
//   ROWB  8 or 12 bytes loaded per source row
//   COEF  0/1  also stream 128 B/lane of "coefficients" (8 x 16 B, slot layout) and xor them in
// hipcc --offload-arch=gfx950 -O3 tools/tile_pattern.hip -o tools/_build/tile_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

constexpr int W = 3840, H = 3232, TX = W / 128, TY = H / 32, NT = TX * TY, NF = 4;
__constant__ uint8_t kHx[16] = {0, 1, 1, 0, 0, 0, 1, 1, 2, 2, 3, 3, 3, 2, 2, 3};
__constant__ uint8_t kHy[16] = {0, 0, 1, 1, 2, 3, 3, 2, 2, 3, 3, 2, 1, 1, 0, 0};

template <int HOP, int WPE, int ROWB, int COEF, int TILED = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
void k_tile(uint8_t *dst, const uint8_t *src, const uint32_t *cmd, const uint4 *coef) {
  const int lane = threadIdx.x & 63, tile = blockIdx.x * 4 + (threadIdx.x >> 6), f = blockIdx.y;
  if (tile >= NT) return;
  const int tx = tile % TX, ty = tile / TX;
  int fx = tx * 16 + (lane >> 4) * 4 + kHx[lane & 15], fy = ty * 4 + kHy[lane & 15];
  size_t off = (size_t)f * W * H + (size_t)fy * 8 * W + fx * 8;
  const size_t RS = TILED ? 128 : W;   // TILED: a tile's 128 x 32 pixels are 4 KB in one piece
  if (TILED) off = (size_t)f * W * H + (size_t)tile * 4096 + (size_t)(kHy[lane & 15] * 8) * 128 + ((lane >> 4) * 4 + kHx[lane & 15]) * 8;
  if (HOP) {
    uint32_t c = cmd[((size_t)f * NT + tile) * 64 + lane];
    off += (c & 1) * 4;                       // always 0 in the data, but the compiler cannot know
  }
  uint4 acc[8];
  if (COEF) {
    const uint4 *cp = coef + ((size_t)f * NT + tile) * 512 + lane;
#pragma unroll
    for (int q = 0; q < 8; q++) acc[q] = cp[q * 64];
  }
  uint32_t r[8][3];
#pragma unroll
  for (int y = 0; y < 8; y++) {
    const uint32_t *p = (const uint32_t *)(src + off + (size_t)y * RS);
    r[y][0] = p[0]; r[y][1] = p[1];
    if (ROWB == 12) r[y][2] = p[2]; else r[y][2] = 0;
  }
#pragma unroll
  for (int y = 0; y < 8; y++) {
    uint32_t a = r[y][0] ^ (r[y][2] & 0x100), b = r[y][1];
    if (COEF) { a ^= acc[y].x & acc[y].z; b ^= acc[y].y & acc[y].w; }
    *(uint2 *)(dst + off + (size_t)y * RS) = uint2{a, b};
  }
}

constexpr int NSETS = 6;   // rotate through 6 disjoint buffer sets (~1.8 GB) so the 256 MB MALL cannot help
template <int HOP, int WPE, int ROWB, int COEF, int TILED = 0> static void run(const char *name, uint8_t *d, uint8_t *s, uint32_t *c, uint4 *k) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int it = 0;
  auto go = [&] {
    const size_t o = (size_t)(it++ % NSETS);
    hipLaunchKernelGGL((k_tile<HOP, WPE, ROWB, COEF, TILED>), dim3((NT + 3) / 4, NF), dim3(256), 0, 0, d + o * ((size_t)NF * W * H + 4096),
                       s + o * ((size_t)NF * W * H + 4096), c + o * ((size_t)NF * NT * 64), k + o * ((size_t)NF * NT * 512));
  };
  for (int i = 0; i < 5; i++) go();
  hipEventRecord(e0, 0);
  const int reps = 50;
  for (int i = 0; i < reps; i++) go();
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double bytes = (double)NF * W * H * 2 + (HOP ? (double)NF * NT * 256 : 0) + (COEF ? (double)NF * NT * 8192 : 0);
  printf("%-34s %7.1f MB  %7.2f us  %7.1f GB/s\n", name, bytes / 1e6, ms * 1e3 / reps, bytes * reps / (ms * 1e-3) / 1e9);
}

int main() {
  uint8_t *d, *s; uint32_t *c; uint4 *k;
  hipMalloc(&d, NSETS * ((size_t)NF * W * H + 4096)); hipMalloc(&s, NSETS * ((size_t)NF * W * H + 4096));
  hipMalloc(&c, NSETS * (size_t)NF * NT * 256); hipMalloc(&k, NSETS * (size_t)NF * NT * 8192);
  hipMemset(s, 3, NSETS * ((size_t)NF * W * H + 4096)); hipMemset(c, 0, NSETS * (size_t)NF * NT * 256);
  hipMemset(k, 5, NSETS * (size_t)NF * NT * 8192);
  printf("tiles/frame %d, waves %d\n", NT, NT * NF);
  run<0, 8, 8, 0>("copy nohop wpe8 row8", d, s, c, k);
  run<1, 8, 8, 0>("copy hop   wpe8 row8", d, s, c, k);
  run<1, 4, 8, 0>("copy hop   wpe4 row8", d, s, c, k);
  run<1, 2, 8, 0>("copy hop   wpe2 row8", d, s, c, k);
  run<1, 8, 12, 0>("copy hop   wpe8 row12", d, s, c, k);
  run<1, 4, 12, 0>("copy hop   wpe4 row12", d, s, c, k);
  run<0, 8, 8, 1>("coef nohop wpe8 row8", d, s, c, k);
  run<1, 8, 8, 1>("coef hop   wpe8 row8", d, s, c, k);
  run<1, 4, 8, 1>("coef hop   wpe4 row8", d, s, c, k);
  run<1, 4, 12, 1>("coef hop   wpe4 row12", d, s, c, k);
  run<1, 2, 12, 1>("coef hop   wpe2 row12", d, s, c, k);
  run<1, 8, 8, 0, 1>("copy hop   wpe8 row8 TILED", d, s, c, k);
  run<1, 8, 8, 1, 1>("coef hop   wpe8 row8 TILED", d, s, c, k);
  run<1, 4, 8, 1, 1>("coef hop   wpe4 row8 TILED", d, s, c, k);
  run<1, 8, 8, 1>("coef hop   wpe8 row8 (again)", d, s, c, k);
  return 0;
}
