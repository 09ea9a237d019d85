// Micro-benchmark: issue rate of the VALU instructions the reconstruction kernels are made of
// (v_pk_add_i16, v_perm_b32, v_mul_i32_i24, v_pk_ashrrev_i16, v_lerp_u8, v_sat_pk_u8_i16 ...) and of the
// whole packed 8x8 inverse DCT, at 1 / 2 / 4 waves per SIMD.  Answers: how many clocks does a wave64
// instruction of each kind occupy a SIMD, i.e. what is the arithmetic floor of k_recon / k_recon_walk.
//   hipcc --offload-arch=gfx950 -O3 -I. tools/valu_rate.hip -o tools/_build/valu_rate && tools/_build/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "theora_amd/csrc/thip_device.h"
using namespace thip;

#define REP16(x) x x x x x x x x x x x x x x x x
template <int OP>
__global__ void k_op(uint32_t *out, int iters, uint32_t seed) {
  uint32_t a[8];
#pragma unroll
  for (int i = 0; i < 8; i++) a[i] = seed * (threadIdx.x + 1 + i);
  const uint32_t c = seed | 1u;
  for (int it = 0; it < iters; it++) {
    // eight independent chains, 16 rounds: 128 instructions per iteration, no dependency stall
#pragma unroll
    for (int r = 0; r < 16; r++)
#pragma unroll
      for (int i = 0; i < 8; i++) {
        if (OP == 0) asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        if (OP == 1) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(0x07060302u));
        if (OP == 2) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        if (OP == 3) asm volatile("v_pk_ashrrev_i16 %0, 3, %0" : "+v"(a[i]));
        if (OP == 4) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        if (OP == 5) asm volatile("v_lerp_u8 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(0u));
        if (OP == 6) asm volatile("v_sat_pk_u8_i16 %0, %0" : "+v"(a[i]));
        if (OP == 7) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        if (OP == 8) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        if (OP == 9) asm volatile("v_alignbyte_b32 %0, %0, %1, 1" : "+v"(a[i]) : "v"(c));
        if (OP == 10) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        if (OP == 11) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        if (OP == 12) asm volatile("v_mad_i32_i24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(c));
        if (OP == 13) asm volatile("v_pk_add_i16 %0, %0, %1 clamp" : "+v"(a[i]) : "v"(c));
        if (OP == 14) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c));
        if (OP == 15) asm volatile("v_pk_sub_i16 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      }
  }
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) s ^= a[i];
  if (s == 0x12345u) out[0] = s;
}

__global__ void k_idct(uint32_t *out, int iters, uint32_t seed) {
  uint32_t P[32], Y[32];
#pragma unroll
  for (int i = 0; i < 32; i++) P[i] = seed * (threadIdx.x + 3 + i);
  for (int it = 0; it < iters; it++) {
    pk_idct8x8(P, Y, false);
#pragma unroll
    for (int i = 0; i < 32; i++) P[i] = Y[i] ^ (uint32_t)it;
  }
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < 32; i++) s ^= P[i];
  if (s == 0x12345u) out[0] = s;
}

template <typename F>
static float timed(F launch) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(a, 0);
  launch();
  hipEventRecord(b, 0);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms;
}

int main() {
  uint32_t *out;
  hipMalloc(&out, 4);
  int cus = 256, clk = 2400000;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
  printf("CUs %d, clock %d kHz\n", cus, clk);
  const char *names[] = {"v_pk_add_i16", "v_perm_b32", "v_mul_i32_i24", "v_pk_ashrrev_i16", "v_add_u32", "v_lerp_u8",
                         "v_sat_pk_u8_i16", "v_pk_max_i16", "v_and_b32", "v_alignbyte_b32", "v_pk_mul_lo_u16", "v_mul_lo_u32",
                         "v_mad_i32_i24", "v_pk_add_i16 clamp", "v_cndmask_b32", "v_pk_sub_i16"};
  const int iters = 2000;
  for (int wps : {1, 2, 4}) {
    const int blocks = cus, threads = 256 * wps;   // wps waves on each of the 4 SIMDs of every CU
    printf("-- %d wave(s) per SIMD\n", wps);
#define RUN(OP)                                                                                                   \
  {                                                                                                               \
    float ms = timed([&] { hipLaunchKernelGGL(k_op<OP>, dim3(blocks), dim3(threads), 0, 0, out, iters, 12345u); }); \
    double inst = (double)iters * 128 * wps;   /* wave-instructions per SIMD */                                   \
    printf("   %-20s %7.3f ms  %.2f clocks per wave-instruction per SIMD (at %.2f GHz)\n", names[OP], ms,            \
           ms * 1e-3 * clk * 1e3 / inst, clk * 1e-6);                                                              \
  }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13) RUN(14) RUN(15)
    {
      const int it2 = 4000;
      float ms = timed([&] { hipLaunchKernelGGL(k_idct, dim3(blocks), dim3(threads), 0, 0, out, it2, 12345u); });
      printf("   %-20s %7.3f ms  %.0f clocks per 8x8 block-wave per SIMD = %.2f G blocks/s on the chip\n", "pk_idct8x8", ms,
             ms * 1e-3 * clk * 1e3 / ((double)it2 * wps), (double)it2 * wps * 64 * 4 * cus / (ms * 1e-3) * 1e-9);
    }
  }
  return 0;
}
