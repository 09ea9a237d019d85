// Practical HBM ceilings on this box for the access mixes the decode kernels have:
// read-only, write-only, copy (1R:1W) and 3R:1W, at the per-launch footprint of the bench
// (≈200 MB) and with 16-byte accesses per lane.  hipcc --offload-arch=gfx950 -O3 tools/hbm_ceiling.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

__global__ __launch_bounds__(256) void k_read(const uint4 *a, uint4 *sink, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, step = (size_t)gridDim.x * 256;
  uint4 acc = {0, 0, 0, 0};
  for (; i < n; i += step) { uint4 v = a[i]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void k_write(uint4 *a, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, step = (size_t)gridDim.x * 256;
  for (; i < n; i += step) a[i] = uint4{1, 2, 3, (unsigned)i};
}
template <int R> __global__ __launch_bounds__(256) void k_mix(const uint4 *a, uint4 *b, size_t n) {
  // R reads (from R disjoint regions of a) per write
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, step = (size_t)gridDim.x * 256;
  for (; i < n; i += step) {
    uint4 acc = a[i];
#pragma unroll
    for (int r = 1; r < R; r++) { uint4 v = a[i + r * n]; acc.x += v.x; acc.y ^= v.y; acc.z += v.z; acc.w ^= v.w; }
    b[i] = acc;
  }
}

int main(int argc, char **argv) {
  size_t mb = argc > 1 ? atoi(argv[1]) : 50;     // bytes of the WRITTEN stream, MB
  int grid = argc > 2 ? atoi(argv[2]) : 0;
  size_t n = mb * 1000000 / 16;
  uint4 *a, *b;
  hipMalloc(&a, n * 16 * 4); hipMalloc(&b, n * 16);
  hipMemset(a, 1, n * 16 * 4); hipMemset(b, 0, n * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 50;
  auto run = [&](const char *name, double bytes, auto fn) {
    for (int i = 0; i < 5; i++) fn();
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; i++) fn();
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-10s %7.1f MB/launch  %8.2f us  %7.1f GB/s\n", name, bytes / 1e6, ms * 1e3 / reps, bytes * reps / (ms * 1e-3) / 1e9);
  };
  int g = grid ? grid : (int)((n + 255) / 256);
  printf("written stream %zu MB, grid %d\n", mb, g);
  run("read", n * 16.0 * 4, [&] { hipLaunchKernelGGL(k_read, dim3(grid ? grid : (int)((4 * n + 255) / 256)), dim3(256), 0, 0, a, b, 4 * n); });
  run("write", n * 16.0, [&] { hipLaunchKernelGGL(k_write, dim3(g), dim3(256), 0, 0, b, n); });
  run("copy 1R:1W", n * 32.0, [&] { hipLaunchKernelGGL(k_mix<1>, dim3(g), dim3(256), 0, 0, a, b, n); });
  run("2R:1W", n * 48.0, [&] { hipLaunchKernelGGL(k_mix<2>, dim3(g), dim3(256), 0, 0, a, b, n); });
  run("3R:1W", n * 64.0, [&] { hipLaunchKernelGGL(k_mix<3>, dim3(g), dim3(256), 0, 0, a, b, n); });
  run("4R:1W", n * 80.0, [&] { hipLaunchKernelGGL(k_mix<4>, dim3(g), dim3(256), 0, 0, a, b, n); });
  return 0;
}
