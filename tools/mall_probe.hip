// Does a read of what the previous kernel just wrote come from the 256 MB Infinity Cache?
// Writes a buffer of N MB with one kernel, then reads (a) the same buffer, (b) a cold one.
// hipcc --offload-arch=gfx950 -O3 tools/mall_probe.hip -o tools/_build/mall_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ __launch_bounds__(256) void k_read(const uint4 *a, uint4 *sink, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint4 v = a[i];
  if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345678u) sink[0] = v;
}
__global__ __launch_bounds__(256) void k_write(uint4 *a, size_t n, unsigned s) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) a[i] = uint4{1, 2, s, (unsigned)i};
}
__global__ __launch_bounds__(256) void k_rmw(uint4 *a, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint4 v = a[i]; v.x += 1; a[i] = v;
}
int main(int argc, char **argv) {
  const int NB = 24;
  for (int mb : {12, 50, 100, 200}) {
    size_t n = (size_t)mb * 1000000 / 16;
    uint4 *buf[NB], *sink;
    (void)hipMalloc(&sink, 64);
    for (int i = 0; i < NB; i++) { (void)hipMalloc(&buf[i], n * 16); (void)hipMemset(buf[i], 1, n * 16); }
    hipEvent_t e[4]; for (auto &x : e) (void)hipEventCreate(&x);
    dim3 g((unsigned)((n + 255) / 256)), b(256);
    double t_w = 0, t_hot = 0, t_cold = 0, t_rmw_hot = 0, t_rmw_cold = 0;
    const int reps = 20;
    for (int r = 0; r < reps + 2; r++) {
      uint4 *w = buf[(2 * r) % NB], *c = buf[(2 * r + 1) % NB];
      (void)hipEventRecord(e[0], 0);
      hipLaunchKernelGGL(k_write, g, b, 0, 0, w, n, (unsigned)r);
      (void)hipEventRecord(e[1], 0);
      hipLaunchKernelGGL(k_read, g, b, 0, 0, w, sink, n);       // just written
      (void)hipEventRecord(e[2], 0);
      hipLaunchKernelGGL(k_read, g, b, 0, 0, c, sink, n);       // not touched for a long time
      (void)hipEventRecord(e[3], 0);
      (void)hipEventSynchronize(e[3]);
      float a, bb, cc; (void)hipEventElapsedTime(&a, e[0], e[1]); (void)hipEventElapsedTime(&bb, e[1], e[2]); (void)hipEventElapsedTime(&cc, e[2], e[3]);
      if (r >= 2) { t_w += a; t_hot += bb; t_cold += cc; }
      // read-modify-write of a just-written buffer vs a cold one (the loop filter's pattern)
      uint4 *w2 = buf[(2 * r + 7) % NB], *c2 = buf[(2 * r + 12) % NB];
      hipLaunchKernelGGL(k_write, g, b, 0, 0, w2, n, (unsigned)r);
      (void)hipEventRecord(e[0], 0);
      hipLaunchKernelGGL(k_rmw, g, b, 0, 0, w2, n);
      (void)hipEventRecord(e[1], 0);
      hipLaunchKernelGGL(k_rmw, g, b, 0, 0, c2, n);
      (void)hipEventRecord(e[2], 0);
      (void)hipEventSynchronize(e[2]);
      (void)hipEventElapsedTime(&a, e[0], e[1]); (void)hipEventElapsedTime(&bb, e[1], e[2]);
      if (r >= 2) { t_rmw_hot += a; t_rmw_cold += bb; }
    }
    auto gbs = [&](double ms, double mult) { return mult * n * 16 / (ms / reps * 1e-3) / 1e9; };
    printf("%4d MB: write %6.1f us (%5.0f GB/s) | read just-written %6.1f us (%5.0f) | read cold %6.1f us (%5.0f) | rmw just-written %6.1f us (%5.0f) | rmw cold %6.1f us (%5.0f)\n",
           mb, t_w / reps * 1e3, gbs(t_w, 1), t_hot / reps * 1e3, gbs(t_hot, 1), t_cold / reps * 1e3, gbs(t_cold, 1),
           t_rmw_hot / reps * 1e3, gbs(t_rmw_hot, 2), t_rmw_cold / reps * 1e3, gbs(t_rmw_cold, 2));
    for (int i = 0; i < NB; i++) (void)hipFree(buf[i]);
  }
  return 0;
}
