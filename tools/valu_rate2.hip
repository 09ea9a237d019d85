// Micro-benchmark, round 4: issue rates of the VALU instructions NOT covered by tools/valu_rate.hip -- conditional moves (the
// round-2 probe measured v_cndmask_b32 at 22.8 clocks: with VCC or with an SGPR pair?), bit-field and 32-bit shifts, the
// three-operand integer ops, SDWA forms with a destination half, high multiplies, dot products.  One wave64 instruction of each
// kind: how many clocks of a SIMD?   hipcc --offload-arch=gfx950 -O3 tools/valu_rate2.hip -o tools/_build/valu_rate2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define OPS(X) \
  X(0, "v_cndmask_b32 vcc", "v_cndmask_b32 %[a], %[a], %[c], vcc") \
  X(1, "v_cndmask_b32 s[pair]", "v_cndmask_b32_e64 %[a], %[a], %[c], %[m]") \
  X(2, "v_bfe_i32", "v_bfe_i32 %[a], %[a], 3, 9") \
  X(3, "v_mul_hi_i32", "v_mul_hi_i32 %[a], %[a], %[c]") \
  X(4, "v_lshlrev_b32", "v_lshlrev_b32 %[a], 3, %[a]") \
  X(5, "v_and_or_b32", "v_and_or_b32 %[a], %[a], %[c], %[c]") \
  X(6, "v_lshl_or_b32", "v_lshl_or_b32 %[a], %[a], 16, %[c]") \
  X(7, "v_add3_u32", "v_add3_u32 %[a], %[a], %[c], %[c]") \
  X(8, "v_pk_mad_i16", "v_pk_mad_i16 %[a], %[a], %[c], %[c]") \
  X(9, "v_pk_lshlrev_b16", "v_pk_lshlrev_b16 %[a], 8, %[a]") \
  X(10, "v_mul_i32_i24 sdwa W1", "v_mul_i32_i24_sdwa %[a], sext(%[a]), %[c] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD") \
  X(11, "v_add_u16 sdwa ->W1", "v_add_u16_sdwa %[a], %[a], %[c] dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_0") \
  X(12, "v_dot2_i32_i16", "v_dot2_i32_i16 %[a], %[a], %[c], %[c]") \
  X(13, "v_dot4_i32_i8", "v_dot4_i32_i8 %[a], %[a], %[c], %[c]") \
  X(14, "v_mad_u32_u24", "v_mad_u32_u24 %[a], %[a], %[c], %[c]") \
  X(15, "v_sad_u8", "v_sad_u8 %[a], %[a], %[c], %[c]") \
  X(16, "v_med3_i32", "v_med3_i32 %[a], %[a], %[c], %[c]") \
  X(17, "v_min_i32", "v_min_i32 %[a], %[a], %[c]") \
  X(18, "v_mov_b32", "v_mov_b32 %[a], %[c]") \
  X(19, "v_xor_b32", "v_xor_b32 %[a], %[a], %[c]") \
  X(20, "v_sub_u16 sdwa W1,W1->W0", "v_sub_u16_sdwa %[a], %[a], %[c] dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1") \
  X(21, "v_ashrrev_i32", "v_ashrrev_i32 %[a], 16, %[a]") \
  X(22, "v_pk_min_i16", "v_pk_min_i16 %[a], %[a], %[c]") \
  X(23, "v_alignbit_b32", "v_alignbit_b32 %[a], %[a], %[c], 8") \
  X(24, "v_mul_u32_u24", "v_mul_u32_u24 %[a], %[a], %[c]") \
  X(25, "v_cmp+v_cndmask", "v_cmp_lt_u32 vcc, %[a], %[c]\n\tv_cndmask_b32 %[a], %[a], %[c], vcc") \
  X(26, "v_bfi_b32", "v_bfi_b32 %[a], %[c], %[a], %[c]") \
  X(27, "v_mul_hi_u32_u24", "v_mul_hi_u32_u24 %[a], %[a], %[c]") \
  X(28, "v_mul_hi_i32_i24", "v_mul_hi_i32_i24 %[a], %[a], %[c]") \
  X(29, "v_pk_add_u16 op_sel", "v_pk_add_u16 %[a], %[a], %[c] op_sel:[1,0] op_sel_hi:[0,1]") \
  X(30, "v_cvt_pk_u8_f32", "v_cvt_pk_u8_f32 %[a], %[a], 1, %[c]") \
  X(31, "v_mad_i32_i16", "v_mad_i32_i16 %[a], %[a], %[c], %[c]") \
  X(32, "4x: cndmask vcc + 3 pk_add", "v_cndmask_b32 %[a], %[a], %[c], vcc\n\tv_pk_add_i16 %[a], %[a], %[c]\n\tv_pk_add_i16 %[a], %[a], %[c]\n\tv_pk_add_i16 %[a], %[a], %[c]") \
  X(33, "4x: cndmask s[] + 3 pk_add", "v_cndmask_b32_e64 %[a], %[a], %[c], %[m]\n\tv_pk_add_i16 %[a], %[a], %[c]\n\tv_pk_add_i16 %[a], %[a], %[c]\n\tv_pk_add_i16 %[a], %[a], %[c]") \
  X(34, "8x: cndmask vcc + 7 pk_add", "v_cndmask_b32 %[a], %[a], %[c], vcc\n\tv_pk_add_i16 %[a], %[a], %[c]\n\tv_pk_add_i16 %[a], %[a], %[c]\n\tv_pk_add_i16 %[a], %[a], %[c]\n\tv_pk_add_i16 %[a], %[a], %[c]\n\tv_pk_add_i16 %[a], %[a], %[c]\n\tv_pk_add_i16 %[a], %[a], %[c]\n\tv_pk_add_i16 %[a], %[a], %[c]") \
  X(35, "2x: v_cmp s[] + cndmask s[]", "v_cmp_lt_u32_e64 s[20:21], %[a], %[c]\n\tv_cndmask_b32_e64 %[a], %[a], %[c], s[20:21]") \
  X(36, "v_addc_co vcc", "v_addc_co_u32 %[a], vcc, %[a], %[c], vcc") \
  X(37, "v_add_co_u32 ->vcc", "v_add_co_u32 %[a], vcc, %[a], %[c]") \
  X(38, "v_mad_u64_u32", "v_mad_u64_u32 %[w], s[20:21], %[a], %[c], %[w]") \
  X(39, "v_lshl_add_u64", "v_lshl_add_u64 %[w], %[w], 1, %[w]") \
  X(40, "v_mul_lo_u32", "v_mul_lo_u32 %[a], %[a], %[c]") \
  X(41, "v_lshrrev_b32", "v_lshrrev_b32 %[a], 3, %[a]") \
  X(42, "v_sub_u32", "v_sub_u32 %[a], %[a], %[c]") \
  X(43, "v_or_b32", "v_or_b32 %[a], %[a], %[c]") \
  X(44, "v_max_i32", "v_max_i32 %[a], %[a], %[c]") \
  X(45, "v_med3_i32", "v_med3_i32 %[a], %[a], %[c], %[c]") \
  X(46, "v_cmp_lt_i32 e64 s[]", "v_cmp_lt_i32_e64 s[20:21], %[a], %[c]") \
  X(47, "v_bfe_u32", "v_bfe_u32 %[a], %[a], 3, 9")

template <int OP>
__global__ void k_op(uint32_t *out, int iters, uint32_t seed) {
  uint32_t a[8];
  unsigned long long w[8];
#pragma unroll
  for (int i = 0; i < 8; i++) a[i] = seed * (threadIdx.x + 1 + i);
#pragma unroll
  for (int i = 0; i < 8; i++) w[i] = (unsigned long long)a[i] * 77u;
  const uint32_t c = seed | 1u;
  const unsigned long long m = 0x5555aaaa3333ccccull ^ seed;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 16; r++)
#pragma unroll
      for (int i = 0; i < 8; i++) {
#define X(N, NAME, ASM) if (OP == N) asm volatile(ASM : [a] "+v"(a[i]), [w] "+v"(w[i]) : [c] "v"(c), [m] "s"(m) : "vcc", "s20", "s21");
        OPS(X)
#undef X
      }
  }
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) s ^= a[i] ^ (uint32_t)w[i] ^ (uint32_t)(w[i] >> 32);
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
  const int iters = 2000;
  for (int wps : {1, 4}) {
    const int blocks = cus, threads = 256 * wps;
    printf("-- %d wave(s) per SIMD\n", wps);
#define X(N, NAME, ASM)                                                                                            \
  {                                                                                                               \
    float ms = timed([&] { hipLaunchKernelGGL(k_op<N>, dim3(blocks), dim3(threads), 0, 0, out, iters, 12345u); }); \
    double inst = (double)iters * 128 * wps;                                                                      \
    printf("   %-26s %7.3f ms  %.2f clocks per wave-instruction per SIMD\n", NAME, ms, ms * 1e-3 * clk * 1e3 / inst); \
  }
    OPS(X)
#undef X
  }
  return 0;
}
