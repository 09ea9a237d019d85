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
  *reinterpret_cast<uint2 *>(p) = v;
}

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

}  // namespace thip
