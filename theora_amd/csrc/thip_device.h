// thip_device.h -- device-side building blocks shared by the gfx950 kernels.
//
// One 8x8 block per lane: a block's 64 values live in that lane's VGPRs, every loop
// below is fully unrolled with compile-time indices so nothing spills to scratch.
// Integer semantics follow the reference exactly (file:line cites into /root/reference).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace thip {

// cos(k*pi/16) in Q16 -- lib/dct.h:23-29
constexpr int kC1 = 64277, kC2 = 60547, kC3 = 54491, kC4 = 46341, kC5 = 36410, kC6 = 25080,
              kC7 = 12785;

__device__ __forceinline__ int sx16(int v) { return (int)(short)v; }
// C*(int16)x>>16 on int32, lib/idct.c:35-48 (x must already be an int16-range value)
__device__ __forceinline__ int q16(int c, int x) { return (c * x) >> 16; }
__device__ __forceinline__ int clamp255(int v) { return min(max(v, 0), 255); }  // ocintrin.h:59

// 1-D inverse DCT, in place on eight int16-range values -- lib/idct.c:30-81.
// Values are truncated to 16 bits exactly where the reference casts.
__device__ __forceinline__ void idct8(int &x0, int &x1, int &x2, int &x3, int &x4, int &x5,
                                      int &x6, int &x7) {
  int t0 = q16(kC4, sx16(x0 + x4));
  int t1 = q16(kC4, sx16(x0 - x4));
  int t2 = q16(kC6, x2) - q16(kC2, x6);
  int t3 = q16(kC2, x2) + q16(kC6, x6);
  int t4 = q16(kC7, x1) - q16(kC1, x7);
  int t5 = q16(kC3, x5) - q16(kC5, x3);
  int t6 = q16(kC5, x5) + q16(kC3, x3);
  int t7 = q16(kC1, x1) + q16(kC7, x7);
  int r;
  r = t4 + t5; t5 = q16(kC4, sx16(t4 - t5)); t4 = r;
  r = t7 + t6; t6 = q16(kC4, sx16(t7 - t6)); t7 = r;
  r = t0 + t3; t3 = t0 - t3; t0 = r;
  r = t1 + t2; t2 = t1 - t2; t1 = r;
  r = t6 + t5; t5 = t6 - t5; t6 = r;
  x0 = sx16(t0 + t7);
  x1 = sx16(t1 + t6);
  x2 = sx16(t2 + t5);
  x3 = sx16(t3 + t4);
  x4 = sx16(t3 - t4);
  x5 = sx16(t2 - t5);
  x6 = sx16(t1 - t6);
  x7 = sx16(t0 - t7);
}

// 2-D inverse DCT on v[64] (natural order, row-major), in place: rows, then columns,
// then (y+8)>>4 -- lib/idct.c:286-296.  The reference's _3/_10 variants (idct.c:234-277)
// equal this transform once the coefficients they ignore are zeroed, which
// idct_mask_by_last_zzi() does.
__device__ __forceinline__ void idct8x8(int v[64]) {
#pragma unroll
  for (int r = 0; r < 8; r++)
    idct8(v[r * 8 + 0], v[r * 8 + 1], v[r * 8 + 2], v[r * 8 + 3], v[r * 8 + 4], v[r * 8 + 5],
          v[r * 8 + 6], v[r * 8 + 7]);
#pragma unroll
  for (int c = 0; c < 8; c++)
    idct8(v[0 * 8 + c], v[1 * 8 + c], v[2 * 8 + c], v[3 * 8 + c], v[4 * 8 + c], v[5 * 8 + c],
          v[6 * 8 + c], v[7 * 8 + c]);
#pragma unroll
  for (int i = 0; i < 64; i++) v[i] = sx16((v[i] + 8) >> 4);
}

// Same transform when only the top-left 4x4 can be non-zero (last_zzi<=10 pattern is a
// subset of it): rows 4..7 are zero so their row pass is skipped.  Bit-identical to
// idct8x8() on such input because idct8 of eight zeros is eight zeros.
__device__ __forceinline__ void idct8x8_rows4(int v[64]) {
#pragma unroll
  for (int r = 0; r < 4; r++)
    idct8(v[r * 8 + 0], v[r * 8 + 1], v[r * 8 + 2], v[r * 8 + 3], v[r * 8 + 4], v[r * 8 + 5],
          v[r * 8 + 6], v[r * 8 + 7]);
#pragma unroll
  for (int c = 0; c < 8; c++)
    idct8(v[0 * 8 + c], v[1 * 8 + c], v[2 * 8 + c], v[3 * 8 + c], v[4 * 8 + c], v[5 * 8 + c],
          v[6 * 8 + c], v[7 * 8 + c]);
#pragma unroll
  for (int i = 0; i < 64; i++) v[i] = sx16((v[i] + 8) >> 4);
}

// oc_idct8x8_c dispatches on last_zzi (idct.c:327-329): <=3 reads only x[0],x[1],x[8];
// <=10 reads only the ten coefficients listed at idct.c:276.  Zero everything else so
// the full transform reproduces the selected variant for ANY input block.
__device__ __forceinline__ void idct_mask_by_last_zzi(int v[64], int last_zzi) {
  const bool c3 = last_zzi <= 3;
  const bool c10 = last_zzi <= 10;
#pragma unroll
  for (int i = 0; i < 64; i++) {
    const int r = i >> 3, c = i & 7;
    const bool in3 = (i == 0 || i == 1 || i == 8);
    const bool in10 = (r + c <= 3);  // {0,1,2,3,8,9,10,16,17,24}
    if (!in10) v[i] = c10 ? 0 : v[i];
    else if (!in3) v[i] = c3 ? 0 : v[i];
  }
}

// ---- byte helpers -----------------------------------------------------------------
__device__ __forceinline__ uint32_t pack4(int a, int b, int c, int d) {
  return (uint32_t)a | ((uint32_t)b << 8) | ((uint32_t)c << 16) | ((uint32_t)d << 24);
}
__device__ __forceinline__ int byte_of(uint32_t w, int k) { return (int)((w >> (8 * k)) & 0xFFu); }

// (a+b)>>1 per byte, truncating -- the half-pel predictor of fragment.c:76.
// v_lerp_u8 computes (a+b+(c&1))>>1 per byte; c=0 gives the truncating average.
__device__ __forceinline__ uint32_t avg4_trunc(uint32_t a, uint32_t b) {
  return __builtin_amdgcn_lerp(a, b, 0u);
}

// unaligned 8-byte load / aligned 8-byte store of one block row
__device__ __forceinline__ uint2 load_row8(const uint8_t *p) {
  uint2 v;
  __builtin_memcpy(&v, p, 8);
  return v;
}
__device__ __forceinline__ void store_row8(uint8_t *p, uint2 v) {
#if defined(THIP_NT_STORES)
  // non-temporal: the pixels are not read again by this kernel
  __builtin_nontemporal_store(v.x, reinterpret_cast<uint32_t *>(p));
  __builtin_nontemporal_store(v.y, reinterpret_cast<uint32_t *>(p) + 1);
#else
  *reinterpret_cast<uint2 *>(p) = v;
#endif
}

// 12 bytes of a reference row starting at a 4-byte-aligned address
struct Row12 {
  uint32_t a, b, c;
};
__device__ __forceinline__ Row12 load_row12(const uint8_t *p) {
  const uint32_t *q = reinterpret_cast<const uint32_t *>(p);   // p is 4-byte aligned
  Row12 r;
  r.a = q[0];
  r.b = q[1];
  r.c = q[2];
  return r;
}
// bytes off..off+7 of the window, off in 0..4: one v_perm_b32 per dword with the selector extract_sel(off) -- the lane's
// selector is computed once for all its rows (v_alignbyte_b32 looks at two bits of the shift only: off = 4 cost two selects a row)
__device__ __forceinline__ uint32_t extract_sel(int off) { return 0x03020100u + (uint32_t)off * 0x01010101u; }
__device__ __forceinline__ uint2 extract8s(Row12 w, uint32_t sel) {
  return make_uint2(__builtin_amdgcn_perm(w.b, w.a, sel), __builtin_amdgcn_perm(w.c, w.b, sel));
}
__device__ __forceinline__ uint2 extract8(Row12 w, int off) { return extract8s(w, extract_sel(off)); }

// clamp255(res[k] + pred byte k) for the eight pixels of a row
__device__ __forceinline__ uint2 recon_row(const int *res, uint2 pred) {
  uint2 o;
  o.x = pack4(clamp255(res[0] + byte_of(pred.x, 0)), clamp255(res[1] + byte_of(pred.x, 1)),
              clamp255(res[2] + byte_of(pred.x, 2)), clamp255(res[3] + byte_of(pred.x, 3)));
  o.y = pack4(clamp255(res[4] + byte_of(pred.y, 0)), clamp255(res[5] + byte_of(pred.y, 1)),
              clamp255(res[6] + byte_of(pred.y, 2)), clamp255(res[7] + byte_of(pred.y, 3)));
  return o;
}

// Truncating division/modulo by a runtime divisor using its float reciprocal; exact
// for n < 2^24 (enforced at state creation).
__device__ __forceinline__ void divmod_u24(uint32_t n, uint32_t d, float rcp, uint32_t &q,
                                           uint32_t &r) {
  q = (uint32_t)(__uint2float_rz(n) * rcp);
  int rem = (int)(n - q * d);
  if (rem < 0) { q--; rem += (int)d; }
  else if (rem >= (int)d) { q++; rem -= (int)d; }
  r = (uint32_t)rem;
}

// Motion vector component -> whole-pel offset and direction of the second (half-pel)
// sample -- lib/state.c:846-957 (the tables OC_MVMAP/OC_MVMAP2 tabulate exactly this:
// divide by 2, or 4 on a decimated axis, truncating toward zero; the second offset
// steps one pel further from zero whenever a fractional bit is set).
__device__ __forceinline__ void mv_axis(int v, bool quarter, int &whole, int &frac) {
  const int sh = quarter ? 2 : 1;
  const int m = (1 << sh) - 1;
  whole = (v + ((v >> 31) & m)) >> sh;
  frac = (v & m) ? ((v >> 31) | 1) : 0;
}

// lflim(R,L): the function tabulated by oc_loop_filter_init_c (state.c:1036-1045),
// spec.tex:7140-7148: sign(R)*min(|R|, max(2L-|R|,0)).
__device__ __forceinline__ int lflim(int R, int L2) {
  const int a = abs(R);
  const int m = min(a, max(L2 - a, 0));
  return R < 0 ? -m : m;
}


// =====================================================================================
// Packed 16-bit formulation: every 32-bit register carries TWO independent int16 values
// (two rows in the row pass, two columns in the column pass), the way the reference's
// SSE2 backend carries eight (lib/x86/sse2idct.c).  All additions may wrap at 16 bits:
// between two of the reference's (ogg_int16_t) casts only +/- happen, so arithmetic
// mod 2^16 reaches the same 16-bit results as idct.c's int32 arithmetic.  The only
// non-linear step, C*x>>16, is done on each half in 32 bits and repacked.
// =====================================================================================
typedef short pk16 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ pk16 as_pk(uint32_t u) { return __builtin_bit_cast(pk16, u); }
__device__ __forceinline__ uint32_t as_u32(pk16 p) { return __builtin_bit_cast(uint32_t, p); }

// {C*lo>>16, C*hi>>16}: two 24-bit multiplies and one byte permute that keeps the upper
// halves of both products (idct.c:35-48 for each half).
__device__ __forceinline__ pk16 pk_q16(int c, pk16 x) {
  const int pl = c * (int)x.x;
  const int ph = c * (int)x.y;
  return as_pk(__builtin_amdgcn_perm((uint32_t)ph, (uint32_t)pl, 0x07060302u));
}

// {(ca*a >> 16) -+ (cb*b >> 16)} on both halves: the four 32-bit products stay where the multiplies leave them and the
// subtraction (addition) reads their upper words and writes one half of the result each (SDWA): two instructions where
// packing each product pair first (pk_q16) and a packed subtraction take three.  Exact: the same 16-bit differences mod 2^16.
#ifndef THIP_NO_SDWA_ROT
template <bool ADD>
__device__ __forceinline__ pk16 pk_q16_rot(int ca, pk16 a, int cb, pk16 b) {
  const int pla = ca * (int)a.x, pha = ca * (int)a.y, plb = cb * (int)b.x, phb = cb * (int)b.y;
  uint32_t t;
  if (ADD) {
    asm("v_add_u16_sdwa %0, %1, %2 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1" : "=v"(t) : "v"(pla), "v"(plb));
    asm("v_add_u16_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1" : "+v"(t) : "v"(pha), "v"(phb));
  } else {
    asm("v_sub_u16_sdwa %0, %1, %2 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1" : "=v"(t) : "v"(pla), "v"(plb));
    asm("v_sub_u16_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1" : "+v"(t) : "v"(pha), "v"(phb));
  }
  return as_pk(t);
}
#else
template <bool ADD>
__device__ __forceinline__ pk16 pk_q16_rot(int ca, pk16 a, int cb, pk16 b) { return ADD ? pk_q16(ca, a) + pk_q16(cb, b) : pk_q16(ca, a) - pk_q16(cb, b); }
#endif

// 1-D inverse DCT on eight packed registers, in place -- lib/idct.c:30-81 on both halves.
__device__ __forceinline__ void pk_idct8(pk16 &x0, pk16 &x1, pk16 &x2, pk16 &x3, pk16 &x4, pk16 &x5,
                                         pk16 &x6, pk16 &x7) {
  pk16 t0 = pk_q16(kC4, x0 + x4);
  pk16 t1 = pk_q16(kC4, x0 - x4);
  pk16 t2 = pk_q16_rot<false>(kC6, x2, kC2, x6);
  pk16 t3 = pk_q16_rot<true>(kC2, x2, kC6, x6);
  pk16 t4 = pk_q16_rot<false>(kC7, x1, kC1, x7);
  pk16 t5 = pk_q16_rot<false>(kC3, x5, kC5, x3);
  pk16 t6 = pk_q16_rot<true>(kC5, x5, kC3, x3);
  pk16 t7 = pk_q16_rot<true>(kC1, x1, kC7, x7);
  pk16 r;
  r = t4 + t5; t5 = pk_q16(kC4, t4 - t5); t4 = r;
  r = t7 + t6; t6 = pk_q16(kC4, t7 - t6); t7 = r;
  r = t0 + t3; t3 = t0 - t3; t0 = r;
  r = t1 + t2; t2 = t1 - t2; t1 = r;
  r = t6 + t5; t5 = t6 - t5; t6 = r;
  x0 = t0 + t7;
  x1 = t1 + t6;
  x2 = t2 + t5;
  x3 = t3 + t4;
  x4 = t3 - t4;
  x5 = t2 - t5;
  x6 = t1 - t6;
  x7 = t0 - t7;
}

// The column pass straight off the row pass's registers, without the 2x2 transposes between them.  The row pass leaves
// A[j] = { r[2j][c], r[2j+1][c] } and B[j] = the same for column c'; the column pass wants x_r = { r[r][c], r[r][c'] }.  Its
// multiplications read ONE 16-bit half each (v_mul_i32_i24 with an SDWA word select), so they can take r[r][c] from wherever it
// lies -- half r & 1 of A[r >> 1] -- and the SDWA additions behind them write each product pair's sum into the half of the
// result it belongs to; only x0 +- x4 need a packed addition first, done per column on { r[0], r[1] } +- { r[4], r[5] } (the upper
// halves compute something nobody reads; THIP_COLS_ADD4) or two of the old transposes (the default: fewer registers in flight).  24 v_perm_b32 fewer per block; the same values in the same
// 16-bit arithmetic as pk_idct8 (lib/idct.c:30-81 on both halves).
#ifndef THIP_NO_SDWA_ROT
template <bool ADD>
__device__ __forceinline__ pk16 q16_rot4(int ca, int a0, int a1, int cb, int b0, int b1) {
  const int pla = ca * a0, pha = ca * a1, plb = cb * b0, phb = cb * b1;
  uint32_t t;
  if (ADD) {
    asm("v_add_u16_sdwa %0, %1, %2 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1" : "=v"(t) : "v"(pla), "v"(plb));
    asm("v_add_u16_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1" : "+v"(t) : "v"(pha), "v"(phb));
  } else {
    asm("v_sub_u16_sdwa %0, %1, %2 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1" : "=v"(t) : "v"(pla), "v"(plb));
    asm("v_sub_u16_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1" : "+v"(t) : "v"(pha), "v"(phb));
  }
  return as_pk(t);
}
__device__ __forceinline__ pk16 q16_pair(int c, int a0, int a1) {
  return as_pk(__builtin_amdgcn_perm((uint32_t)(c * a1), (uint32_t)(c * a0), 0x07060302u));
}
__device__ __forceinline__ void pk_idct8_cols(const pk16 A[4], const pk16 B[4], pk16 &x0, pk16 &x1, pk16 &x2, pk16 &x3, pk16 &x4,
                                              pk16 &x5, pk16 &x6, pk16 &x7) {
#ifdef THIP_COLS_ADD4
  const pk16 sa = A[0] + A[2], sb = B[0] + B[2], da = A[0] - A[2], db = B[0] - B[2];   // .x: x0 +- x4 of column c / c'
  pk16 t0 = q16_pair(kC4, (int)sa.x, (int)sb.x);
  pk16 t1 = q16_pair(kC4, (int)da.x, (int)db.x);
#else
  const pk16 q0 = as_pk(__builtin_amdgcn_perm(as_u32(B[0]), as_u32(A[0]), 0x05040100u)), q4 = as_pk(__builtin_amdgcn_perm(as_u32(B[2]), as_u32(A[2]), 0x05040100u));
  pk16 t0 = pk_q16(kC4, q0 + q4);
  pk16 t1 = pk_q16(kC4, q0 - q4);
#endif
  pk16 t2 = q16_rot4<false>(kC6, (int)A[1].x, (int)B[1].x, kC2, (int)A[3].x, (int)B[3].x);   // x2, x6
  pk16 t3 = q16_rot4<true>(kC2, (int)A[1].x, (int)B[1].x, kC6, (int)A[3].x, (int)B[3].x);
  pk16 t4 = q16_rot4<false>(kC7, (int)A[0].y, (int)B[0].y, kC1, (int)A[3].y, (int)B[3].y);   // x1, x7
  pk16 t5 = q16_rot4<false>(kC3, (int)A[2].y, (int)B[2].y, kC5, (int)A[1].y, (int)B[1].y);   // x5, x3
  pk16 t6 = q16_rot4<true>(kC5, (int)A[2].y, (int)B[2].y, kC3, (int)A[1].y, (int)B[1].y);
  pk16 t7 = q16_rot4<true>(kC1, (int)A[0].y, (int)B[0].y, kC7, (int)A[3].y, (int)B[3].y);
  pk16 r;
  r = t4 + t5; t5 = pk_q16(kC4, t4 - t5); t4 = r;
  r = t7 + t6; t6 = pk_q16(kC4, t7 - t6); t7 = r;
  r = t0 + t3; t3 = t0 - t3; t0 = r;
  r = t1 + t2; t2 = t1 - t2; t1 = r;
  r = t6 + t5; t5 = t6 - t5; t6 = r;
  x0 = t0 + t7;
  x1 = t1 + t6;
  x2 = t2 + t5;
  x3 = t3 + t4;
  x4 = t3 - t4;
  x5 = t2 - t5;
  x6 = t1 - t6;
  x7 = t0 - t7;
}
#define THIP_HAVE_IDCT8_COLS 1
#endif

// Same transform when inputs 4..7 are zero (lib/idct.c:92-130 on both halves).
__device__ __forceinline__ void pk_idct8_first4(pk16 &x0, pk16 &x1, pk16 &x2, pk16 &x3, pk16 &x4,
                                                pk16 &x5, pk16 &x6, pk16 &x7) {
  pk16 t0 = pk_q16(kC4, x0);
  pk16 t2 = pk_q16(kC6, x2);
  pk16 t3 = pk_q16(kC2, x2);
  pk16 t4 = pk_q16(kC7, x1);
  pk16 t6 = pk_q16(kC3, x3);
  pk16 t5 = -pk_q16(kC5, x3);
  pk16 t7 = pk_q16(kC1, x1);
  pk16 r;
  r = t4 + t5; t5 = pk_q16(kC4, t4 - t5); t4 = r;
  r = t7 + t6; t6 = pk_q16(kC4, t7 - t6); t7 = r;
  pk16 t1 = t0 + t2;
  t2 = t0 - t2;
  r = t0 + t3; t3 = t0 - t3; t0 = r;
  r = t6 + t5; t5 = t6 - t5; t6 = r;
  x0 = t0 + t7;
  x1 = t1 + t6;
  x2 = t2 + t5;
  x3 = t3 + t4;
  x4 = t3 - t4;
  x5 = t2 - t5;
  x6 = t1 - t6;
  x7 = t0 - t7;
}

// (y+8)>>4 (idct.c:243) for a residual that goes on to `clamp255(residual + predictor)` (fragment.c:34-80) and nowhere else: y + 8
// with SIGNED SATURATION (v_pk_add_i16 clamp), then the shift -- two operations.  Where y + 8 fits 16 bits this is the reference's
// value; for y in 32760..32767 it is 2047 where the reference has 2048, and with a predictor of 0..255 (or 128) both end as pixel
// 255.  (The exact residual in three operations, for whoever needs the number itself: ((y >> 3) + 1) >> 1.)
#ifndef THIP_DESCALE3
__device__ __forceinline__ pk16 pk_descale(pk16 y) {
  const pk16 eight = {(short)8, (short)8};
  return __builtin_elementwise_add_sat(y, eight) >> 4;
}
#else
__device__ __forceinline__ pk16 pk_descale(pk16 y) { return ((y >> 3) + (short)1) >> 1; }
#endif

// two predictor bytes (k, k+1 of a packed word) -> two int16
__device__ __forceinline__ pk16 pk_bytes01(uint32_t w) { return as_pk(__builtin_amdgcn_perm(0u, w, 0x0c010c00u)); }
__device__ __forceinline__ pk16 pk_bytes23(uint32_t w) { return as_pk(__builtin_amdgcn_perm(0u, w, 0x0c030c02u)); }

// clamp255 of two int16 -> two bytes in bits 0..15 (v_sat_pk_u8_i16)
__device__ __forceinline__ uint32_t sat_pk_u8(pk16 v) {
  uint32_t r;
  asm("v_sat_pk_u8_i16 %0, %1" : "=v"(r) : "v"(as_u32(v)));
  return r;
}
// four bytes: clamp255 of lo's two values, then of hi's two.  The second conversion writes the upper half of the register the first
// one filled (SDWA, the lower half preserved): two operations where shifting and merging the second result takes three.
#ifndef THIP_NO_SDWA_SAT
__device__ __forceinline__ uint32_t sat_pk_u8x4(pk16 lo, pk16 hi) {
  uint32_t r;
  asm("v_sat_pk_u8_i16 %0, %1" : "=v"(r) : "v"(as_u32(lo)));
  asm("v_sat_pk_u8_i16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(r) : "v"(as_u32(hi)));
  return r;
}
#else
__device__ __forceinline__ uint32_t sat_pk_u8x4(pk16 lo, pk16 hi) { return sat_pk_u8(lo) | (sat_pk_u8(hi) << 16); }
#endif

// One row of 8 pixels: OC_CLAMP255(residue + predictor), fragment.c:54,64,76.  The add
// saturates at 16 bits so a residue near +32767 still clamps to 255 like the int sum.
__device__ __forceinline__ uint2 pk_recon_row(pk16 r01, pk16 r23, pk16 r45, pk16 r67, uint2 pred) {
  const pk16 s01 = __builtin_elementwise_add_sat(r01, pk_bytes01(pred.x));
  const pk16 s23 = __builtin_elementwise_add_sat(r23, pk_bytes23(pred.x));
  const pk16 s45 = __builtin_elementwise_add_sat(r45, pk_bytes01(pred.y));
  const pk16 s67 = __builtin_elementwise_add_sat(r67, pk_bytes23(pred.y));
  uint2 o;
  o.x = sat_pk_u8x4(s01, s23);
  o.y = sat_pk_u8x4(s45, s67);
  return o;
}

// Block held as P[j][c] = { x[2j][c], x[2j+1][c] } (row pairs): zero what the variant
// selected by last_zzi does not read (idct.c:327-329, :245, :276).
__device__ __forceinline__ void pk_mask_by_last_zzi(uint32_t P[32], int last_zzi) {
  const bool c3 = last_zzi <= 3, c10 = last_zzi <= 10;
#pragma unroll
  for (int j = 0; j < 4; j++)
#pragma unroll
    for (int c = 0; c < 8; c++) {
      const int rlo = 2 * j, rhi = 2 * j + 1;
      const uint32_t m10 = ((rlo + c <= 3) ? 0x0000FFFFu : 0u) | ((rhi + c <= 3) ? 0xFFFF0000u : 0u);
      const bool lo3 = (rlo == 0 && c <= 1), hi3 = (rhi == 1 && c == 0);
      const uint32_t m3 = (lo3 ? 0x0000FFFFu : 0u) | (hi3 ? 0xFFFF0000u : 0u);
      const uint32_t m = c3 ? m3 : (c10 ? m10 : 0xFFFFFFFFu);
      if (m10 != 0xFFFFFFFFu || m3 != 0xFFFFFFFFu) P[j * 8 + c] &= m;
    }
}

// Full 2-D transform.  In: P[j*8+c] = {x[2j][c], x[2j+1][c]}.  Out: Y[r*4+k] =
// {y[r][2k], y[r][2k+1]} (residue pairs along a row), descaled.  rows4: rows 4..7 of the
// input are known to be zero for every lane of the wave.
template <bool FUSED_COLS = true>
__device__ __forceinline__ void pk_idct8x8(const uint32_t P[32], uint32_t Y[32], bool rows4) {
  pk16 R[32];
#pragma unroll
  for (int i = 0; i < 32; i++) R[i] = as_pk(P[i]);
  pk_idct8(R[0], R[1], R[2], R[3], R[4], R[5], R[6], R[7]);
  pk_idct8(R[8], R[9], R[10], R[11], R[12], R[13], R[14], R[15]);
  if (!rows4) {
    pk_idct8(R[16], R[17], R[18], R[19], R[20], R[21], R[22], R[23]);
    pk_idct8(R[24], R[25], R[26], R[27], R[28], R[29], R[30], R[31]);
  }
  pk16 Q[32];
#if defined(THIP_HAVE_IDCT8_COLS) && !defined(THIP_NO_FUSED_TRANSPOSE)
  if (FUSED_COLS && !rows4) {   // (wave-uniform)
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const pk16 A[4] = {R[0 * 8 + 2 * k], R[1 * 8 + 2 * k], R[2 * 8 + 2 * k], R[3 * 8 + 2 * k]};
      const pk16 B[4] = {R[0 * 8 + 2 * k + 1], R[1 * 8 + 2 * k + 1], R[2 * 8 + 2 * k + 1], R[3 * 8 + 2 * k + 1]};
      pk_idct8_cols(A, B, Q[0 * 4 + k], Q[1 * 4 + k], Q[2 * 4 + k], Q[3 * 4 + k], Q[4 * 4 + k], Q[5 * 4 + k], Q[6 * 4 + k], Q[7 * 4 + k]);
#ifdef THIP_COLS_FENCE
      __builtin_amdgcn_sched_barrier(0);   // one column pair after the other: interleaved they need registers the wave does not have
#endif
    }
#pragma unroll
    for (int i = 0; i < 32; i++) Y[i] = as_u32(pk_descale(Q[i]));
    return;
  }
#endif
  // 2x2 transposes: row pairs -> column pairs.  Q[r*4+k] = {R[r][2k], R[r][2k+1]}
#pragma unroll
  for (int j = 0; j < 4; j++)
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t a = as_u32(R[j * 8 + 2 * k]), b = as_u32(R[j * 8 + 2 * k + 1]);
      Q[(2 * j) * 4 + k] = as_pk(__builtin_amdgcn_perm(b, a, 0x05040100u));
      Q[(2 * j + 1) * 4 + k] = as_pk(__builtin_amdgcn_perm(b, a, 0x07060302u));
    }
#pragma unroll
  for (int k = 0; k < 4; k++) {
    if (rows4)
      pk_idct8_first4(Q[0 * 4 + k], Q[1 * 4 + k], Q[2 * 4 + k], Q[3 * 4 + k], Q[4 * 4 + k], Q[5 * 4 + k],
                      Q[6 * 4 + k], Q[7 * 4 + k]);
    else
      pk_idct8(Q[0 * 4 + k], Q[1 * 4 + k], Q[2 * 4 + k], Q[3 * 4 + k], Q[4 * 4 + k], Q[5 * 4 + k],
               Q[6 * 4 + k], Q[7 * 4 + k]);
  }
#pragma unroll
  for (int i = 0; i < 32; i++) Y[i] = as_u32(pk_descale(Q[i]));
}

}  // namespace thip
