// Can consecutive frames of ONE stream overlap their launch overhead and their reference-independent part?
// Frame i+1's kernel goes down a second HIP stream and spins on a word that the first stream writes (hipStreamWriteValue32)
// once frame i's kernel has completed; before the spin it does the work that needs no reference pixels.
//   baseline: N kernels of `pre + post` microseconds of busy work each, one stream.
//   chained : the same kernels on two streams alternately, `pre` before the wait, `post` after it.
// hipcc --offload-arch=gfx950 -O3 tools/chain_probe.hip -o tools/_build/chain_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

__device__ __forceinline__ uint64_t now() { return __builtin_readcyclecounter(); }   // s_memtime
__device__ __forceinline__ uint64_t rt() { uint64_t t; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)); return t; }   // 100 MHz

__global__ __launch_bounds__(64) void k_frame(uint32_t *flag, uint32_t want, int pre_ticks, int post_ticks, uint32_t *fault, uint32_t *sink) {
  uint64_t t0 = rt();
  while (rt() - t0 < (uint64_t)pre_ticks) {}
  if (want) {
    int spins = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < want) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1 << 22)) { *fault = 1; break; }
    }
  }
  t0 = rt();
  while (rt() - t0 < (uint64_t)post_ticks) {}
  if (t0 == 0x1234567) sink[0] = 1;
}

int main(int argc, char **argv) {
  const int waves = argc > 1 ? atoi(argv[1]) : 816;
  const int pre = argc > 2 ? atoi(argv[2]) : 300, post = argc > 3 ? atoi(argv[3]) : 400;   // 10 ns ticks
  const int N = 400;
  uint32_t *flag, *fault, *sink;
  hipMalloc(&flag, 4); hipMalloc(&fault, 4); hipMalloc(&sink, 4);
  hipMemset(flag, 0, 4); hipMemset(fault, 0, 4);
  hipStream_t s[2]; hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking); hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms;
  for (int rep = 0; rep < 2; rep++) {
    hipEventRecord(e0, s[0]);
    for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_frame, dim3(waves), dim3(64), 0, s[0], flag, 0u, pre, post, fault, sink);
    hipEventRecord(e1, s[0]); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("baseline  one stream          : %6.2f us per frame (busy %5.2f)\n", ms * 1e3 / N, (pre + post) * 0.01);
  }
  for (int rep = 0; rep < 2; rep++) {
    hipMemset(flag, 0, 4);
    hipDeviceSynchronize();
    uint32_t base = 0;
    hipEventRecord(e0, s[0]);
    hipStreamWaitEvent(s[1], e0, 0);
    for (int i = 0; i < N; i++) {
      hipStream_t st = s[i & 1];
      hipLaunchKernelGGL(k_frame, dim3(waves), dim3(64), 0, st, flag, i == 0 ? 0u : base + (uint32_t)i, pre, post, fault, sink);
      hipStreamWriteValue32(st, flag, base + (uint32_t)i + 1, 0);
    }
    hipEventRecord(e1, s[(N - 1) & 1]); hipEventSynchronize(e1);
    hipDeviceSynchronize();
    hipEventElapsedTime(&ms, e0, e1);
    uint32_t f; hipMemcpy(&f, fault, 4, hipMemcpyDeviceToHost);
    printf("chained   two streams + flag  : %6.2f us per frame (post %5.2f) fault %u\n", ms * 1e3 / N, post * 0.01, f);
  }
  return 0;
}
