// Micro-benchmark: what a kernel launch costs on this box as a function of kernarg size and
// where the per-stream table lives (kernarg by value vs a device buffer).  Not part of the
// product; used to decide how thip_decode_frames passes its per-stream table.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
struct Big { unsigned long long p[8][36]; };   // ~2.3 KB, like BatchK
__global__ void k_small(unsigned *out) { if (out && threadIdx.x == 0 && blockIdx.x == 1u << 30) out[0] = 1; }
__global__ void k_big(const Big b, unsigned *out) {
  const unsigned long long v = b.p[blockIdx.y][blockIdx.x % 36];
  if (v == 0x1234567812345678ull && threadIdx.x == 0) out[0] = 1;
}
__global__ void k_ptr(const Big *b, unsigned *out) {
  const unsigned long long v = b->p[blockIdx.y][blockIdx.x % 36];
  if (v == 0x1234567812345678ull && threadIdx.x == 0) out[0] = 1;
}
int main() {
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  unsigned *out; hipMalloc(&out, 4);
  Big hb; memset(&hb, 1, sizeof(hb));
  Big *db; hipMalloc(&db, sizeof(Big)); hipMemcpy(db, &hb, sizeof(Big), hipMemcpyHostToDevice);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int N = 200;
  for (int grid : {1, 256, 2176}) {
    for (int mode = 0; mode < 3; mode++) {
      for (int rep = 0; rep < 2; rep++) {
        hipStreamSynchronize(s);
        hipEventRecord(a, s);
        for (int i = 0; i < N; i++) {
          if (mode == 0) hipLaunchKernelGGL(k_small, dim3(grid, 4), dim3(256), 0, s, out);
          if (mode == 1) hipLaunchKernelGGL(k_big, dim3(grid, 4), dim3(256), 0, s, hb, out);
          if (mode == 2) hipLaunchKernelGGL(k_ptr, dim3(grid, 4), dim3(256), 0, s, db, out);
        }
        hipEventRecord(b, s);
        hipStreamSynchronize(s);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (rep) printf("grid=%4dx4 mode=%s  %.2f us per launch (back-to-back, same stream)\n", grid,
                        mode == 0 ? "small-arg" : mode == 1 ? "2.3KB-by-value" : "table-in-HBM", 1e3 * ms / N);
      }
    }
  }
  return 0;
}
