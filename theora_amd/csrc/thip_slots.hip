// thip_slots.hip -- batched forms of the individual accel-vtable slots.
//
// Decoder side: oc_idct8x8, oc_frag_recon_intra/_inter/_inter2, oc_frag_copy_list,
// oc_state_loop_filter_frag_rows on one plane (lib/state.h:352-370) -- each exposed on its
// own so it can be parity-tested against the oracle; the frame path (thip_decode.hip)
// fuses them.
// Encoder side: oc_enc_opt_vtable's block kernels (lib/encint.h:292-326): SAD family,
// SATD family, SSD, sub, copy2 and the forward DCT -- one 8x8 block per lane.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <stdio.h>

#include "../../include/theora_hip.h"
#include "thip_device.h"
#include "thip_enc.h"
#include "thip_costmaps.h"

using namespace thip;

#define HIP_TRY(expr)                                                                    \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess) {                                                              \
      fprintf(stderr, "theora_hip: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), \
              __FILE__, __LINE__);                                                       \
      return THIP_EFAULT;                                                                \
    }                                                                                    \
  } while (0)

static inline dim3 grid_for(int64_t n) { return dim3((unsigned)((n + 255) / 256)); }

// ---------------------------------------------------------------------------------------
// decoder slots
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void load_block16(int v[64], const int16_t *p) {
  const int4 *q = reinterpret_cast<const int4 *>(p);
#pragma unroll
  for (int r = 0; r < 8; r++) {
    const int4 w = q[r];
    const int w4[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int k = 0; k < 4; k++) {
      v[r * 8 + 2 * k] = sx16(w4[k]);
      v[r * 8 + 2 * k + 1] = w4[k] >> 16;
    }
  }
}

__device__ __forceinline__ void store_block16(int16_t *p, const int v[64]) {
  int4 *q = reinterpret_cast<int4 *>(p);
#pragma unroll
  for (int r = 0; r < 8; r++) {
    int4 w;
    w.x = (v[r * 8 + 0] & 0xFFFF) | (v[r * 8 + 1] << 16);
    w.y = (v[r * 8 + 2] & 0xFFFF) | (v[r * 8 + 3] << 16);
    w.z = (v[r * 8 + 4] & 0xFFFF) | (v[r * 8 + 5] << 16);
    w.w = (v[r * 8 + 6] & 0xFFFF) | (v[r * 8 + 7] << 16);
    q[r] = w;
  }
}

// Wave-cooperative forms of the two functions above for arrays of consecutive blocks (block i of
// the array belongs to thread i).  A thread that fetches its own 128-byte block makes every load
// instruction of the wave touch 64 different cache lines; instead the wave moves its 64 blocks
// (8 KB contiguous) with eight fully coalesced 1-KB instructions and redistributes through LDS.
// Piece p of block b is parked at int4 index b*8 + ((p + b) & 7): the rotation keeps both the
// linear side and the per-block side free of bank conflicts.  All 64 lanes must call; `i < n`
// says whether this lane's block exists (the others move nothing).  lds: 512 int4 per wave.
__device__ __forceinline__ void load_block16_wave(int v[64], const int16_t *array, int64_t i, int64_t n, int4 *lds) {
  const int lane = (int)threadIdx.x & 63;
  const int64_t b0 = i - lane;                                     // first block of the wave
  const int4 *g = reinterpret_cast<const int4 *>(array) + b0 * 8;
#pragma unroll
  for (int q = 0; q < 8; q++) {
    const int idx = q * 64 + lane, b = idx >> 3, pc = idx & 7;
    if (b0 + b < n) lds[b * 8 + ((pc + b) & 7)] = g[idx];
  }
#pragma unroll
  for (int r = 0; r < 8; r++) {
    const int4 w = lds[lane * 8 + ((r + lane) & 7)];
    const int w4[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int k = 0; k < 4; k++) {
      v[r * 8 + 2 * k] = sx16(w4[k]);
      v[r * 8 + 2 * k + 1] = w4[k] >> 16;
    }
  }
}
__device__ __forceinline__ void store_block16_wave(int16_t *array, int64_t i, int64_t n, const int v[64], int4 *lds) {
  const int lane = (int)threadIdx.x & 63;
  const int64_t b0 = i - lane;
#pragma unroll
  for (int r = 0; r < 8; r++) {
    int4 w;
    w.x = (v[r * 8 + 0] & 0xFFFF) | (v[r * 8 + 1] << 16);
    w.y = (v[r * 8 + 2] & 0xFFFF) | (v[r * 8 + 3] << 16);
    w.z = (v[r * 8 + 4] & 0xFFFF) | (v[r * 8 + 5] << 16);
    w.w = (v[r * 8 + 6] & 0xFFFF) | (v[r * 8 + 7] << 16);
    lds[lane * 8 + ((r + lane) & 7)] = w;
  }
  int4 *g = reinterpret_cast<int4 *>(array) + b0 * 8;
#pragma unroll
  for (int q = 0; q < 8; q++) {
    const int idx = q * 64 + lane, b = idx >> 3, pc = idx & 7;
    if (b0 + b < n) g[idx] = lds[b * 8 + ((pc + b) & 7)];
  }
}

__global__ __launch_bounds__(256) void k_idct_batch(int16_t *y, const int16_t *x, const int32_t *last_zzi,
                                                   int64_t n) {
  __shared__ int4 s_x[4 * 512];
  int4 *lds = s_x + (threadIdx.x >> 6) * 512;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int v[64];
  load_block16_wave(v, x, i, n, lds);
  idct_mask_by_last_zzi(v, (last_zzi && i < n) ? last_zzi[i] : 64);
  idct8x8(v);
  store_block16_wave(y, i, n, v, lds);
}

template <int NSRC>
__global__ __launch_bounds__(256) void k_frag_recon_batch(uint8_t *dst_frame, const uint8_t *src_frame,
                                                         int ystride, const int32_t *dst_offs,
                                                         const int32_t *src1_offs, const int32_t *src2_offs,
                                                         const int16_t *residue, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int v[64];
  load_block16(v, residue + i * 64);
  uint8_t *dst = dst_frame + dst_offs[i];
#pragma unroll
  for (int r = 0; r < 8; r++) {
    uint2 pred = make_uint2(0x80808080u, 0x80808080u);   // fragment.c:54
    if (NSRC >= 1) pred = load_row8(src_frame + src1_offs[i] + (ptrdiff_t)r * ystride);   // fragment.c:64
    if (NSRC == 2) {                                                                      // fragment.c:76
      const uint2 b = load_row8(src_frame + src2_offs[i] + (ptrdiff_t)r * ystride);
      pred.x = avg4_trunc(pred.x, b.x);
      pred.y = avg4_trunc(pred.y, b.y);
    }
    const uint2 o = recon_row(v + r * 8, pred);
    __builtin_memcpy(dst + (ptrdiff_t)r * ystride, &o, 8);
  }
}

__global__ __launch_bounds__(256) void k_frag_copy_list(uint8_t *dst_frame, const uint8_t *src_frame,
                                                       int ystride, const int32_t *fragis, int64_t n,
                                                       const int32_t *frag_buf_offs) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const ptrdiff_t off = frag_buf_offs[fragis[i]];
#pragma unroll
  for (int r = 0; r < 8; r++) {
    const uint2 t = load_row8(src_frame + off + (ptrdiff_t)r * ystride);
    __builtin_memcpy(dst_frame + off + (ptrdiff_t)r * ystride, &t, 8);
  }
}

// ---------------------------------------------------------------------------------------
// encoder block kernels
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void load_pixels(int p[64], const uint8_t *s, int ystride) {
#pragma unroll
  for (int r = 0; r < 8; r++) {
    const uint2 w = load_row8(s + (ptrdiff_t)r * ystride);
#pragma unroll
    for (int q = 0; q < 4; q++) {
      p[r * 8 + q] = byte_of(w.x, q);
      p[r * 8 + 4 + q] = byte_of(w.y, q);
    }
  }
}

__global__ __launch_bounds__(256) void k_enc_border_ssd(uint32_t *out, const uint8_t *src_plane,
                                                       const uint8_t *ref_plane, int ystride,
                                                       const int32_t *src_offs, const int32_t *ref_offs,
                                                       const int64_t *masks, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int s[64], p[64];
  load_pixels(s, src_plane + src_offs[i], ystride);
  load_pixels(p, ref_plane + ref_offs[i], ystride);
  const uint64_t m = (uint64_t)masks[i];
  unsigned v = 0;
#pragma unroll
  for (int k = 0; k < 64; k++)
    if ((m >> k) & 1) v += (unsigned)((s[k] - p[k]) * (s[k] - p[k]));   // encfrag.c:352-366
  out[i] = v;
}

__global__ __launch_bounds__(256) void k_enc_sub(int16_t *diff, const uint8_t *src_plane,
                                                const uint8_t *ref_plane, int ystride,
                                                const int32_t *src_offs, const int32_t *ref_offs, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int s[64];
  load_pixels(s, src_plane + src_offs[i], ystride);
  if (ref_offs) {   // encfrag.c:21-30
    int p[64];
    load_pixels(p, ref_plane + ref_offs[i], ystride);
#pragma unroll
    for (int k = 0; k < 64; k++) s[k] -= p[k];
  } else {          // encfrag.c:32-40
#pragma unroll
    for (int k = 0; k < 64; k++) s[k] -= 128;
  }
  store_block16(diff + i * 64, s);
}

__global__ __launch_bounds__(256) void k_enc_copy2(uint8_t *dst_plane, const uint8_t *src_plane, int ystride,
                                                  const int32_t *dst_offs, const int32_t *src1_offs,
                                                  const int32_t *src2_offs, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
#pragma unroll
  for (int r = 0; r < 8; r++) {   // encfrag.c:368-378
    const uint2 a = load_row8(src_plane + src1_offs[i] + (ptrdiff_t)r * ystride);
    const uint2 b = load_row8(src_plane + src2_offs[i] + (ptrdiff_t)r * ystride);
    const uint2 o = make_uint2(avg4_trunc(a.x, b.x), avg4_trunc(a.y, b.y));
    __builtin_memcpy(dst_plane + dst_offs[i] + (ptrdiff_t)r * ystride, &o, 8);
  }
}

// 1-D forward DCT, in place (the arithmetic of lib/fdct.c:28-120, whose rounding constants make the
// transform the exact inverse partner of the decoder's; outputs truncate to int16 where the reference stores
// into ogg_int16_t).  Written as the three kinds of step it consists of:
//   fd_scale(v, bias)        v * (1 + 27146/65536) rounded with `bias`, nudged away from zero: the sqrt(2)-ish
//                            scalings of the even part and of the two middle odd terms;
//   fd_rot(a, b, ca, cb, k)  the first output of a plane rotation, (ca*a + cb*b + k) >> 16, nudged by b != 0;
//   fd_back(u, c, x, m, k, s) the second output recovered from the first: r = +-(c*u >> 16 - x), then
//                            r * (1 + m / 2^s) rounded with k, nudged away from zero.
__device__ __forceinline__ int fd_nz(int v) { return v != 0 ? 1 : 0; }
__device__ __forceinline__ int fd_scale(int v, int bias) { return ((27146 * v + bias) >> 16) + v + fd_nz(v); }
__device__ __forceinline__ int fd_rot(int a, int b, int ca, int cb, int k) { return ((ca * a + cb * b + k) >> 16) + fd_nz(b); }
__device__ __forceinline__ int fd_back(int r, int m, int k, int s) { return ((r * m + k) >> s) + r + fd_nz(r); }
__device__ __forceinline__ void fdct8(int &x0, int &x1, int &x2, int &x3, int &x4, int &x5, int &x6,
                                      int &x7) {
  // stage 1: mirror sums and differences; stage 2: the even half folds once more
  const int s07 = x0 + x7, d07 = x0 - x7, s16 = x1 + x6, d16 = x1 - x6;
  const int s25 = x2 + x5, d25 = x2 - x5, s34 = x3 + x4, d34 = x3 - x4;
  const int e0 = s07 + s34, e3 = s07 - s34, e1 = s16 + s25, e2 = s16 - s25;
  // even outputs 0 and 4 (fdct.c:96-100), 2 and 6 (fdct.c:102-106)
  const int p = ((27146 * e0 + 0x4000) >> 16) + e0 + fd_nz(e0), q = fd_scale(e1, 0xB500);
  const int y0 = (p + q) >> 1, y4 = p - y0;
  const int y2 = fd_rot(e2, e3, kC6, kC2, 0x6CB7);
  const int y6 = fd_back(((kC6 * y2) >> 16) - e2, 21600, 0x2800, 18);
  // odd half: the two middle differences are rotated by pi/4 first (fdct.c:87-93)
  const int ms = d16 + d25, md = d16 - d25;
  const int h5 = fd_scale(md, 0xB500) >> 1, h6 = fd_scale(ms, 0xB500) >> 1;
  const int o4 = d34 + h5, o5 = d34 - h5, o7 = d07 + h6, o6 = d07 - h6;
  // odd outputs 5 and 3 (fdct.c:108-112), 1 and 7 (fdct.c:114-118)
  const int y5 = fd_rot(o6, o5, kC5, kC3, 0x0E3D);
  const int y3 = fd_back(o6 - ((kC5 * y5) >> 16), 26568, 0x3400, 17);
  const int y1 = fd_rot(o4, o7, kC7, kC1, 0x7B1B);
  const int y7 = fd_back(((kC7 * y1) >> 16) - o4, 20539, 0x3000, 20);
  x0 = sx16(y0); x1 = sx16(y1); x2 = sx16(y2); x3 = sx16(y3);
  x4 = sx16(y4); x5 = sx16(y5); x6 = sx16(y6); x7 = sx16(y7);
}

// natural position of zig-zag index i -- lib/internal.c:27 (first 64 entries)
__device__ constexpr int kFZigZag[64] = {
    0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
    41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
    30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

__global__ __launch_bounds__(256) void k_enc_fdct(int16_t *y, const int16_t *x, int64_t n) {
  __shared__ int4 s_x[4 * 512];
  int4 *lds = s_x + (threadIdx.x >> 6) * 512;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int w[64];
  load_block16_wave(w, x, i, n, lds);
#pragma unroll
  for (int k = 0; k < 64; k++) w[k] = sx16(w[k] << 2);        // fdct.c:136
  w[0] = sx16(w[0] + (w[0] != 0) + 1);                        // fdct.c:139-141
  w[1] = sx16(w[1] + 1);
  w[8] = sx16(w[8] - 1);
  // columns of w -> rows of z (fdct.c:143), then columns of z -> rows of w (fdct.c:145).
  // In registers: transform each column in place (result element k of column c sits at
  // [k][c], i.e. z transposed), then each row in place; the final element (r,c) holds
  // what the reference leaves at w[r*8+c] transposed twice == natural position.
#pragma unroll
  for (int c = 0; c < 8; c++)
    fdct8(w[0 * 8 + c], w[1 * 8 + c], w[2 * 8 + c], w[3 * 8 + c], w[4 * 8 + c], w[5 * 8 + c], w[6 * 8 + c],
          w[7 * 8 + c]);
#pragma unroll
  for (int r = 0; r < 8; r++)
    fdct8(w[r * 8 + 0], w[r * 8 + 1], w[r * 8 + 2], w[r * 8 + 3], w[r * 8 + 4], w[r * 8 + 5], w[r * 8 + 6],
          w[r * 8 + 7]);
  int o[64];
#pragma unroll
  for (int k = 0; k < 64; k++) o[k] = sx16((w[kFZigZag[k]] + 2) >> 2);   // fdct.c:149
  store_block16_wave(y, i, n, o, lds);
}

// zig-zag index of natural position (the inverse of kFZigZag), a byte each: k_enc_fdct4's lanes read the eight of a row at once
__device__ __attribute__((aligned(8), unused)) constexpr uint8_t kIZigZag8[64] = {
    0,  1,  5,  6,  14, 15, 27, 28, 2,  4,  7,  13, 16, 26, 29, 42, 3,  8,  12, 17, 25, 30, 41, 43, 9,  11, 18, 24, 31, 40, 44, 53,
    10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38, 46, 51, 55, 60, 21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63};

// oc_enc_fdct8x8 with FOUR lanes per block (round 6; the layout of k_enc_fdct_quantize4 below, without its second half): one block per
// lane is 700 dependent instructions a wave on 1.5 waves per SIMD behind one exposed round trip, 105 registers and 32 KB of LDS a
// work group -- 10.4 us for the 25 MB of a 1080p 4:4:4 frame.  Lane 4b + j takes columns 2j, 2j + 1 of block b for the first pass
// (fdct.c:143), the block is transposed through the wave's 2 KB of LDS as int16 pairs, the lane takes rows 2j, 2j + 1 for the
// second (fdct.c:145); the zig-zag order (fdct.c:149) happens on the way out through the same 2 KB, so that loads and stores are
// whole 16-byte pieces.

__global__ __launch_bounds__(256) void k_enc_fdct4(int16_t *y, const int16_t *x, int64_t n) {
  __shared__ int4 s_x[4 * 128];                          // 2 KB a wave: 16 blocks of eight 16-byte pieces (piece r = row r)
  int4 *lds = s_x + (threadIdx.x >> 6) * 128;
  const int lane = (int)threadIdx.x & 63, b = lane >> 2, j = lane & 3;
  const int64_t b0 = ((int64_t)blockIdx.x * 256 + (threadIdx.x & ~63u)) >> 2;   // the wave's first block
  // piece pc of block bb lives at lds[bb * 8 + ((pc + bb) & 7)]: rotated, so that sixteen blocks' equal rows spread over the banks
  {
    const int4 *g = reinterpret_cast<const int4 *>(x) + b0 * 8;
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int idx = q * 64 + lane, bb = idx >> 3, pc = idx & 7;
      if (b0 + bb < n) lds[bb * 8 + ((pc + bb) & 7)] = g[idx];
    }
  }
  // (the wave's own 2 KB: no other wave reads them, the wave's LDS operations are issued in order -- no barrier; the zig-zag
  //  indices of the lane's two rows come out of a 64-byte table in memory, asked for now)
  const uint2 zz0 = reinterpret_cast<const uint2 *>(kIZigZag8)[2 * j], zz1 = reinterpret_cast<const uint2 *>(kIZigZag8)[2 * j + 1];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const int *ldw = reinterpret_cast<const int *>(lds);
  int c0[8], c1[8];   // columns 2j and 2j + 1
#pragma unroll
  for (int r = 0; r < 8; r++) {
    const int w = ldw[(b * 8 + ((r + b) & 7)) * 4 + j];
    c0[r] = sx16(sx16(w) << 2);                          // fdct.c:136
    c1[r] = sx16((w >> 16) << 2);
  }
  if (j == 0) {                                          // fdct.c:139-141: positions 0, 1 and 8
    c0[0] = sx16(c0[0] + (c0[0] != 0) + 1);
    c1[0] = sx16(c1[0] + 1);
    c0[1] = sx16(c0[1] - 1);
  }
  fdct8(c0[0], c0[1], c0[2], c0[3], c0[4], c0[5], c0[6], c0[7]);
  fdct8(c1[0], c1[1], c1[2], c1[3], c1[4], c1[5], c1[6], c1[7]);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // every lane of the wave has read the input
  int *ldww = reinterpret_cast<int *>(lds);
#pragma unroll
  for (int k = 0; k < 8; k++) ldww[(b * 8 + ((k + b) & 7)) * 4 + j] = (c0[k] & 0xFFFF) | (c1[k] << 16);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  int o[16];           // rows 2j and 2j + 1, natural position (2j + h) * 8 + c at o[h * 8 + c]
  uint2 zz[2];         // ... and their zig-zag indices, a byte each
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const int r = 2 * j + h;
    const int4 w = lds[b * 8 + ((r + b) & 7)];
    zz[h] = h ? zz1 : zz0;
    int v[8] = {sx16(w.x), w.x >> 16, sx16(w.y), w.y >> 16, sx16(w.z), w.z >> 16, sx16(w.w), w.w >> 16};
    fdct8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
#pragma unroll
    for (int c = 0; c < 8; c++) o[h * 8 + c] = sx16((v[c] + 2) >> 2);   // fdct.c:149
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  int16_t *lds16 = reinterpret_cast<int16_t *>(lds);
  // where zig-zag index z of block b lies in the wave's area (the same rotation of 16-byte pieces)
  auto at = [&](int z) { return (b * 8 + (((z >> 3) + b) & 7)) * 8 + (z & 7); };
#pragma unroll
  for (int h = 0; h < 2; h++)
#pragma unroll
    for (int c = 0; c < 8; c++) {
      const uint32_t word = c < 4 ? zz[h].x : zz[h].y;
      lds16[at((int)((word >> (8 * (c & 3))) & 0xFFu))] = (int16_t)o[h * 8 + c];
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  int4 *g = reinterpret_cast<int4 *>(y) + b0 * 8;
#pragma unroll
  for (int q = 0; q < 2; q++) {
    const int idx = q * 64 + lane, bb = idx >> 3, pc = idx & 7;
    if (b0 + bb < n) g[idx] = lds[bb * 8 + ((pc + bb) & 7)];
  }
}

// oc_enc_quantize_c (enquant.c:219-248); the {m,l} reciprocal of each step is derived in
// place exactly as oc_iquant_init does (enquant.c:183-191).
__global__ __launch_bounds__(256) void k_enc_quantize(int16_t *qdct, int32_t *nonzero, const int16_t *dct,
                                                     const uint16_t *dequant, int64_t n) {
  __shared__ int s_d[64], s_m[64], s_l[64];
  if (threadIdx.x < 64) {
    const uint32_t d = (uint32_t)dequant[threadIdx.x] << 1;
    const int l = 31 - __builtin_clz(d);                       // OC_ILOGNZ_32(d)-1
    const uint32_t t = 1u + ((1u << (16 + l)) / d);
    s_d[threadIdx.x] = (int)dequant[threadIdx.x];
    s_m[threadIdx.x] = (int)(int16_t)(t - 0x10000u);
    s_l[threadIdx.x] = l;
  }
  __syncthreads();
  __shared__ int4 s_x[4 * 512];
  int4 *lds = s_x + (threadIdx.x >> 6) * 512;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int v[64];
  load_block16_wave(v, dct, i, n, lds);
  int nz = 0;
#pragma unroll
  for (int z = 0; z < 64; z++) {
    int val = v[z] << 1;
    const int d = s_d[z];
    int q = 0;
    if (abs(val) >= d) {
      const int s = val >> 31;
      val += (d + s) ^ s;
      q = sx16(((((s_m[z] * val) >> 16) + val) >> s_l[z]) - s);
      nz = z;
    }
    v[z] = q;
  }
  store_block16_wave(qdct, i, n, v, lds);
  if (i < n) nonzero[i] = nz;
}

// oc_enc_fdct8x8 followed by oc_enc_quantize on the same block (what oc_enc_block_transform_quantize does with every residual,
// tokenize.c / analyze.c: fdct.c:128 then enquant.c:219) in ONE pass: the coefficients never travel to memory and back, one
// launch instead of two.  dct_out (may be null) still receives the unquantised coefficients for callers that want both.
__global__ __launch_bounds__(256) void k_enc_fdct_quantize(int16_t *qdct, int32_t *nonzero, int16_t *dct_out, const int16_t *x,
                                                          const uint16_t *dequant, const int16_t *enquant, int64_t n) {
  __shared__ int s_d[64], s_m[64], s_l[64];
  if (threadIdx.x < 64) {
    s_d[threadIdx.x] = (int)dequant[threadIdx.x];
    if (enquant) {
      s_m[threadIdx.x] = (int)enquant[2 * threadIdx.x];
      s_l[threadIdx.x] = (int)enquant[2 * threadIdx.x + 1];
    } else {   // oc_iquant_init (enquant.c:183-191)
      const uint32_t d = (uint32_t)dequant[threadIdx.x] << 1;
      const int l = 31 - __builtin_clz(d);
      const uint32_t t = 1u + ((1u << (16 + l)) / d);
      s_m[threadIdx.x] = (int)(int16_t)(t - 0x10000u);
      s_l[threadIdx.x] = l;
    }
  }
  __shared__ int4 s_x[4 * 512];
  int4 *lds = s_x + (threadIdx.x >> 6) * 512;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int w[64];
  load_block16_wave(w, x, i, n, lds);
#pragma unroll
  for (int k = 0; k < 64; k++) w[k] = sx16(w[k] << 2);        // fdct.c:136
  w[0] = sx16(w[0] + (w[0] != 0) + 1);                        // fdct.c:139-141
  w[1] = sx16(w[1] + 1);
  w[8] = sx16(w[8] - 1);
#pragma unroll
  for (int c = 0; c < 8; c++)
    fdct8(w[0 * 8 + c], w[1 * 8 + c], w[2 * 8 + c], w[3 * 8 + c], w[4 * 8 + c], w[5 * 8 + c], w[6 * 8 + c], w[7 * 8 + c]);
#pragma unroll
  for (int r = 0; r < 8; r++)
    fdct8(w[r * 8 + 0], w[r * 8 + 1], w[r * 8 + 2], w[r * 8 + 3], w[r * 8 + 4], w[r * 8 + 5], w[r * 8 + 6], w[r * 8 + 7]);
  int o[64];
#pragma unroll
  for (int k = 0; k < 64; k++) o[k] = sx16((w[kFZigZag[k]] + 2) >> 2);   // fdct.c:149
  if (dct_out) store_block16_wave(dct_out, i, n, o, lds);
  __syncthreads();   // the tables; and every wave is done with its LDS area's first use
  int nz = 0;
#pragma unroll
  for (int z = 0; z < 64; z++) {   // enquant.c:228-245
    int val = o[z] << 1;
    const int d = s_d[z];
    int q = 0;
    if (abs(val) >= d) {
      const int sg = val >> 31;
      val += (d + sg) ^ sg;
      q = sx16(((((s_m[z] * val) >> 16) + val) >> s_l[z]) - sg);
      nz = z;
    }
    o[z] = q;
  }
  store_block16_wave(qdct, i, n, o, lds);
  if (i < n) nonzero[i] = nz;
}

// The same pass with FOUR lanes per block (the default of thip_enc_fdct_quantize_batch).  One block per lane is 1 500 dependent
// vector instructions a wave, and a 1080p 4:4:4 frame is 1 530 such waves -- one and a half per SIMD, issuing at the single-wave
// rate (8.3 clocks an instruction, profiles/r04_valu_rate2.txt) behind one exposed round trip: 17.3 us, slower per block than
// either half alone.  Here lane 4b + j takes columns 2j, 2j + 1 of block b for the first pass (fdct.c:143), the block is
// transposed through the wave's 2 KB of LDS as int16 pairs, the lane takes rows 2j, 2j + 1 for the second (fdct.c:145), and it
// quantises its sixteen coefficients with the tables kept by NATURAL position; the zig-zag order (fdct.c:149) happens on the way
// out, through the same 2 KB, so that the stores are whole 16-byte pieces.  Four times the waves, a quarter of the chain each.
__global__ __launch_bounds__(256) void k_enc_fdct_quantize4(int16_t *qdct, int32_t *nonzero, int16_t *dct_out, const int16_t *x,
                                                           const uint16_t *dequant, const int16_t *enquant, int64_t n) {
  // by natural position, ONE 8-byte entry: step | reciprocal m << 16, shift l | zig-zag index << 8 -- a lane's sixteen positions are
  // two runs of eight entries, eight 16-byte LDS reads instead of the 64 four-byte ones of four separate tables (the LDS pipe is
  // one per compute unit: with four lanes a block it is as busy as the vector units)
  __shared__ __attribute__((aligned(16))) uint2 s_t[64];
  if (threadIdx.x < 64) {
    const int z = (int)threadIdx.x, pos = kFZigZag[z];
    int m, l;
    if (enquant) {
      m = (int)enquant[2 * z];
      l = (int)enquant[2 * z + 1];
    } else {   // oc_iquant_init (enquant.c:183-191)
      const uint32_t d = (uint32_t)dequant[z] << 1;
      l = 31 - __builtin_clz(d);
      const uint32_t t = 1u + ((1u << (16 + l)) / d);
      m = (int)(int16_t)(t - 0x10000u);
    }
    s_t[pos] = make_uint2((uint32_t)dequant[z] | (uint32_t)(uint16_t)m << 16, (uint32_t)(l & 0xFF) | (uint32_t)z << 8);
  }
  __shared__ int4 s_x[4 * 128];                          // 2 KB a wave: 16 blocks of eight 16-byte pieces (piece r = row r)
  int4 *lds = s_x + (threadIdx.x >> 6) * 128;
  const int lane = (int)threadIdx.x & 63, b = lane >> 2, j = lane & 3;
  const int64_t b0 = ((int64_t)blockIdx.x * 256 + (threadIdx.x & ~63u)) >> 2;   // the wave's first block
  const int64_t i = b0 + b;
  // piece pc of block bb lives at lds[bb * 8 + ((pc + bb) & 7)]: rotated, so that sixteen blocks' equal rows spread over the banks
  {
    const int4 *g = reinterpret_cast<const int4 *>(x) + b0 * 8;
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int idx = q * 64 + lane, bb = idx >> 3, pc = idx & 7;
      if (b0 + bb < n) lds[bb * 8 + ((pc + bb) & 7)] = g[idx];
    }
  }
  __syncthreads();   // (the tables too)
  const int *ldw = reinterpret_cast<const int *>(lds);
  int c0[8], c1[8];   // columns 2j and 2j + 1
#pragma unroll
  for (int r = 0; r < 8; r++) {
    const int w = ldw[(b * 8 + ((r + b) & 7)) * 4 + j];
    c0[r] = sx16(sx16(w) << 2);                          // fdct.c:136
    c1[r] = sx16((w >> 16) << 2);
  }
  if (j == 0) {                                          // fdct.c:139-141: positions 0, 1 and 8
    c0[0] = sx16(c0[0] + (c0[0] != 0) + 1);
    c1[0] = sx16(c1[0] + 1);
    c0[1] = sx16(c0[1] - 1);
  }
  fdct8(c0[0], c0[1], c0[2], c0[3], c0[4], c0[5], c0[6], c0[7]);
  fdct8(c1[0], c1[1], c1[2], c1[3], c1[4], c1[5], c1[6], c1[7]);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // every lane of the wave has read the input
  int *ldww = reinterpret_cast<int *>(lds);
#pragma unroll
  for (int k = 0; k < 8; k++) ldww[(b * 8 + ((k + b) & 7)) * 4 + j] = (c0[k] & 0xFFFF) | (c1[k] << 16);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  int o[16];           // rows 2j and 2j + 1, natural position (2j + h) * 8 + c at o[h * 8 + c]
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const int r = 2 * j + h;
    const int4 w = lds[b * 8 + ((r + b) & 7)];
    int v[8] = {sx16(w.x), w.x >> 16, sx16(w.y), w.y >> 16, sx16(w.z), w.z >> 16, sx16(w.w), w.w >> 16};
    fdct8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
#pragma unroll
    for (int c = 0; c < 8; c++) o[h * 8 + c] = sx16((v[c] + 2) >> 2);   // fdct.c:149
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  int16_t *lds16 = reinterpret_cast<int16_t *>(lds);
  // where zig-zag index z of block b lies in the wave's area (the same rotation of 16-byte pieces)
  auto at = [&](int z) { return (b * 8 + (((z >> 3) + b) & 7)) * 8 + (z & 7); };
  // (the table entries of positions k, k + 1: one 16-byte read)
  auto entries = [&](int k) { return *reinterpret_cast<const uint4 *>(&s_t[(2 * j + (k >> 3)) * 8 + (k & 7)]); };
  if (dct_out) {
#pragma unroll
    for (int k = 0; k < 16; k += 2) {
      const uint4 e = entries(k);
      lds16[at((int)(e.y >> 8))] = (int16_t)o[k];
      lds16[at((int)(e.w >> 8))] = (int16_t)o[k + 1];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    int4 *g = reinterpret_cast<int4 *>(dct_out) + b0 * 8;
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int idx = q * 64 + lane, bb = idx >> 3, pc = idx & 7;
      if (b0 + bb < n) g[idx] = lds[bb * 8 + ((pc + bb) & 7)];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  int nz = 0;
#pragma unroll
  for (int k = 0; k < 16; k += 2) {   // enquant.c:228-245
    const uint4 e = entries(k);
#pragma unroll
    for (int h2 = 0; h2 < 2; h2++) {
      const uint32_t ex = h2 ? e.z : e.x, ey = h2 ? e.w : e.y;
      const int z = (int)(ey >> 8), d = (int)(ex & 0xFFFFu), m = (int)ex >> 16, l = (int)(ey & 0xFFu);
      int val = o[k + h2] << 1, q = 0;
      if (abs(val) >= d) {
        const int sg = val >> 31;
        val += (d + sg) ^ sg;
        q = sx16(((((m * val) >> 16) + val) >> l) - sg);
        nz = max(nz, z);            // (the reference's loop runs up the zig-zag order: the last index that passes is the largest)
      }
      lds16[at(z)] = (int16_t)q;
    }
  }
  nz = max(nz, __shfl_xor(nz, 1));
  nz = max(nz, __shfl_xor(nz, 2));
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  {
    int4 *g = reinterpret_cast<int4 *>(qdct) + b0 * 8;
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int idx = q * 64 + lane, bb = idx >> 3, pc = idx & 7;
      if (b0 + bb < n) g[idx] = lds[bb * 8 + ((pc + bb) & 7)];
    }
  }
  if (j == 0 && i < n) nonzero[i] = nz;
}

// The same with the reciprocals handed in: `enquant` is the 64-entry {m, l} table
// thip_enc_enquant_table_init built once (oc_enc_enquant_table_init, enquant.c:194), as the
// reference's quantize slot receives it (encint.h:319-320), instead of being re-derived per launch.
__global__ __launch_bounds__(256) void k_enc_quantize_tab(int16_t *qdct, int32_t *nonzero, const int16_t *dct,
                                                         const uint16_t *dequant, const int16_t *enquant, int64_t n) {
  __shared__ int s_d[64], s_m[64], s_l[64];
  if (threadIdx.x < 64) {
    s_d[threadIdx.x] = (int)dequant[threadIdx.x];
    s_m[threadIdx.x] = (int)enquant[2 * threadIdx.x];
    s_l[threadIdx.x] = (int)enquant[2 * threadIdx.x + 1];
  }
  __syncthreads();
  __shared__ int4 s_x[4 * 512];
  int4 *lds = s_x + (threadIdx.x >> 6) * 512;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int v[64];
  load_block16_wave(v, dct, i, n, lds);
  int nz = 0;
#pragma unroll
  for (int z = 0; z < 64; z++) {
    int val = v[z] << 1;
    const int d = s_d[z];
    int q = 0;
    if (abs(val) >= d) {
      const int s = val >> 31;
      val += (d + s) ^ s;
      q = sx16(((((s_m[z] * val) >> 16) + val) >> s_l[z]) - s);
      nz = z;
    }
    v[z] = q;
  }
  store_block16_wave(qdct, i, n, v, lds);
  if (i < n) nonzero[i] = nz;
}

// Stream and completion policy of the batched entry points below (thip_set_batch_stream).
static hipStream_t g_batch_stream = nullptr;
static int g_batch_sync = 1;

// ---------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------
extern "C" {

int thip_set_batch_stream(void *stream, int synchronous) {
  g_batch_stream = (hipStream_t)stream;
  g_batch_sync = synchronous ? 1 : 0;
  return THIP_OK;
}

int thip_enc_quantize_batch(int16_t *qdct, int32_t *nonzero, const int16_t *dct, const uint16_t *dequant,
                            int64_t n) {
  if (!qdct || !nonzero || !dct || !dequant) return THIP_EFAULT;
  if (n < 0) return THIP_EINVAL;
  if (n == 0) return THIP_OK;
  hipLaunchKernelGGL(k_enc_quantize, grid_for(n), dim3(256), 0, g_batch_stream, qdct, nonzero, dct, dequant, n);
  HIP_TRY(hipGetLastError());
  if (g_batch_sync) HIP_TRY(hipStreamSynchronize(g_batch_stream));
  return THIP_OK;
}

// ---- the enquant_table_* slots of oc_enc_opt_vtable (encint.h:316-318): host functions, like the
//      reference's (the tables are built when the quantisation parameters change, not per block) -------
void thip_enc_enquant_table_init(void *enquant, const uint16_t dequant[64]) {
  int16_t *t = (int16_t *)enquant;   // oc_iquant {ogg_int16_t m, l} (enquant.h), 64 entries
  for (int zzi = 0; zzi < 64; zzi++) {
    // oc_iquant_init, enquant.c:183-191
    const uint32_t d = (uint32_t)dequant[zzi] << 1;
    const int l = 31 - __builtin_clz(d | 1u);
    const uint32_t tt = 1u + ((1u << (16 + l)) / (d ? d : 1u));
    t[2 * zzi] = (int16_t)(tt - 0x10000u);
    t[2 * zzi + 1] = (int16_t)l;
  }
}

void thip_enc_enquant_table_fixup(void *enquant[3][3][2], int nqis) {   // enquant.c:210-217
  for (int pli = 0; pli < 3; pli++)
    for (int qii = 1; qii < nqis; qii++)
      for (int qti = 0; qti < 2; qti++) memcpy(enquant[pli][qii][qti], enquant[pli][0][qti], 4);
}

void thip_enc_opt_data(size_t *enquant_table_size, int *enquant_table_alignment) {   // encint.h:331-338
  if (enquant_table_size) *enquant_table_size = THIP_ENQUANT_TABLE_SIZE;
  if (enquant_table_alignment) *enquant_table_alignment = 16;
}

int thip_enc_fdct_quantize_batch(int16_t *qdct, int32_t *nonzero, int16_t *dct, const int16_t *x, const uint16_t *dequant,
                                 const void *enquant, int64_t n) {
  if (n < 0) return THIP_EINVAL;
  if (n == 0) return THIP_OK;
  if (!qdct || !nonzero || !x || !dequant) return THIP_EFAULT;
  // four lanes per block (option "enc_fq_lanes" = 1: the round-4 kernel, one block per lane)
  const bool lanes1 = thip_option("enc_fq_lanes") == 1;
  if (lanes1)
    hipLaunchKernelGGL(k_enc_fdct_quantize, grid_for(n), dim3(256), 0, g_batch_stream, qdct, nonzero, dct, x, dequant, (const int16_t *)enquant, n);
  else
    hipLaunchKernelGGL(k_enc_fdct_quantize4, grid_for(4 * n), dim3(256), 0, g_batch_stream, qdct, nonzero, dct, x, dequant, (const int16_t *)enquant, n);
  HIP_TRY(hipGetLastError());
  if (g_batch_sync) HIP_TRY(hipStreamSynchronize(g_batch_stream));
  return THIP_OK;
}

int thip_enc_quantize_tab_batch(int16_t *qdct, int32_t *nonzero, const int16_t *dct, const uint16_t *dequant,
                                const void *enquant, int64_t n) {
  if (!qdct || !nonzero || !dct || !dequant || !enquant) return THIP_EFAULT;
  if (n < 0) return THIP_EINVAL;
  if (n == 0) return THIP_OK;
  hipLaunchKernelGGL(k_enc_quantize_tab, grid_for(n), dim3(256), 0, g_batch_stream, qdct, nonzero, dct, dequant,
                     (const int16_t *)enquant, n);
  HIP_TRY(hipGetLastError());
  if (g_batch_sync) HIP_TRY(hipStreamSynchronize(g_batch_stream));
  return THIP_OK;
}

int thip_idct8x8_batch(int16_t *y, const int16_t *x, const int32_t *last_zzi, int64_t n) {
  if (!y || !x) return THIP_EFAULT;
  if (n < 0) return THIP_EINVAL;
  if (n == 0) return THIP_OK;
  hipLaunchKernelGGL(k_idct_batch, grid_for(n), dim3(256), 0, g_batch_stream, y, x, last_zzi, n);
  HIP_TRY(hipGetLastError());
  if (g_batch_sync) HIP_TRY(hipStreamSynchronize(g_batch_stream));
  return THIP_OK;
}

int thip_frag_recon_batch(uint8_t *dst_frame, const uint8_t *src_frame, int ystride, int nsrc,
                          const int32_t *dst_offs, const int32_t *src1_offs, const int32_t *src2_offs,
                          const int16_t *residue, int64_t n) {
  if (!dst_frame || !dst_offs || !residue) return THIP_EFAULT;
  if (nsrc < 0 || nsrc > 2 || n < 0) return THIP_EINVAL;
  if ((nsrc >= 1 && (!src_frame || !src1_offs)) || (nsrc == 2 && !src2_offs)) return THIP_EFAULT;
  if (n == 0) return THIP_OK;
  if (nsrc == 0)
    hipLaunchKernelGGL(k_frag_recon_batch<0>, grid_for(n), dim3(256), 0, g_batch_stream, dst_frame, src_frame, ystride,
                       dst_offs, src1_offs, src2_offs, residue, n);
  else if (nsrc == 1)
    hipLaunchKernelGGL(k_frag_recon_batch<1>, grid_for(n), dim3(256), 0, g_batch_stream, dst_frame, src_frame, ystride,
                       dst_offs, src1_offs, src2_offs, residue, n);
  else
    hipLaunchKernelGGL(k_frag_recon_batch<2>, grid_for(n), dim3(256), 0, g_batch_stream, dst_frame, src_frame, ystride,
                       dst_offs, src1_offs, src2_offs, residue, n);
  HIP_TRY(hipGetLastError());
  if (g_batch_sync) HIP_TRY(hipStreamSynchronize(g_batch_stream));
  return THIP_OK;
}

int thip_frag_copy_list_batch(uint8_t *dst_frame, const uint8_t *src_frame, int ystride,
                              const int32_t *fragis, int64_t nfragis, const int32_t *frag_buf_offs) {
  if (!dst_frame || !src_frame || !fragis || !frag_buf_offs) return THIP_EFAULT;
  if (nfragis < 0) return THIP_EINVAL;
  if (nfragis == 0) return THIP_OK;
  hipLaunchKernelGGL(k_frag_copy_list, grid_for(nfragis), dim3(256), 0, g_batch_stream, dst_frame, src_frame, ystride,
                     fragis, nfragis, frag_buf_offs);
  HIP_TRY(hipGetLastError());
  if (g_batch_sync) HIP_TRY(hipStreamSynchronize(g_batch_stream));
  return THIP_OK;
}

int thip_enc_mb_count(int frame_width, int frame_height) {
  if (frame_width <= 0 || frame_height <= 0 || (frame_width & 15) || (frame_height & 15)) return THIP_EINVAL;
  return 4 * (((frame_width >> 3) + 3) >> 2) * (((frame_height >> 3) + 3) >> 2);
}

int thip_enc_mb_cost_maps(const uint8_t *const planes[3], const int32_t strides[3], int frame_width, int frame_height, int pixel_fmt,
                          uint32_t *intra_satd, uint32_t *luma, uint32_t *activity, uint32_t *activity_fast) {
  if (!planes || !strides || !planes[0] || !planes[1] || !planes[2]) return THIP_EFAULT;
  if (frame_width <= 0 || frame_height <= 0 || (frame_width & 15) || (frame_height & 15) || frame_width >= 0x8000 || frame_height >= 0x8000 ||
      pixel_fmt < 0 || pixel_fmt > 3 || pixel_fmt == 1)
    return THIP_EINVAL;
  CostMapK K;
  memset(&K, 0, sizeof(K));
  K.fmt = pixel_fmt;
  K.hdec = !(pixel_fmt & 1);
  K.vdec = !(pixel_fmt & 2);
  for (int p = 0; p < 3; p++) {
    K.plane[p] = planes[p];
    K.stride[p] = strides[p];
    K.nh[p] = (frame_width >> 3) >> (p ? K.hdec : 0);
    K.nv[p] = (frame_height >> 3) >> (p ? K.vdec : 0);
    if (strides[p] < K.nh[p] * 8) return THIP_EINVAL;
  }
  K.nsbw = (K.nh[0] + 3) >> 2;
  K.n_luma = K.nh[0] * K.nv[0];
  K.n_all = K.n_luma + (intra_satd ? 2 * K.nh[1] * K.nv[1] : 0);
  K.intra_satd = intra_satd;
  K.luma = luma;
  K.activity = activity;
  K.activity_fast = activity_fast;
  // macro-block slots outside the frame are zeroed by lanes of their own (one launch, nothing to memset: thip_costmaps.h)
  K.n_pad = ((K.nh[0] & 3) || (K.nv[0] & 3)) ? thip_enc_mb_count(frame_width, frame_height) : 0;
  hipLaunchKernelGGL(k_enc_cost_maps, grid_for(K.n_all + K.n_pad), dim3(256), 0, g_batch_stream, K);
  HIP_TRY(hipGetLastError());
  if (g_batch_sync) HIP_TRY(hipStreamSynchronize(g_batch_stream));
  return THIP_OK;
}

int thip_enc_frag_metric_batch(int op, uint32_t *out, int32_t *dc_out, const uint8_t *src_plane,
                               const uint8_t *ref_plane, int ystride, const int32_t *src_offs,
                               const int32_t *ref_offs, const int32_t *ref2_offs, uint32_t thresh,
                               int64_t n) {
  if (!out || !src_plane || !src_offs) return THIP_EFAULT;
  if (op < 0 || op > THIP_ENC_SSD || n < 0) return THIP_EINVAL;
  const bool has_ref = op != THIP_ENC_INTRA_SAD && op != THIP_ENC_INTRA_SATD;
  const bool two = op == THIP_ENC_SAD2_THRESH || op == THIP_ENC_SATD2;
  if ((has_ref && (!ref_plane || !ref_offs)) || (two && !ref2_offs)) return THIP_EFAULT;
  if (n == 0) return THIP_OK;
#define LAUNCH_METRIC(OPC)                                                                              \
  case OPC:                                                                                             \
    hipLaunchKernelGGL(k_enc_metric<OPC>, grid_for(n), dim3(256), 0, g_batch_stream, out, dc_out, src_plane, ref_plane, \
                       ystride, src_offs, ref_offs, ref2_offs, thresh, n);                              \
    break;
  switch (op) {
    LAUNCH_METRIC(THIP_ENC_SAD)
    LAUNCH_METRIC(THIP_ENC_SAD_THRESH)
    LAUNCH_METRIC(THIP_ENC_SAD2_THRESH)
    LAUNCH_METRIC(THIP_ENC_INTRA_SAD)
    LAUNCH_METRIC(THIP_ENC_SATD)
    LAUNCH_METRIC(THIP_ENC_SATD2)
    LAUNCH_METRIC(THIP_ENC_INTRA_SATD)
    LAUNCH_METRIC(THIP_ENC_SSD)
  }
#undef LAUNCH_METRIC
  HIP_TRY(hipGetLastError());
  if (g_batch_sync) HIP_TRY(hipStreamSynchronize(g_batch_stream));
  return THIP_OK;
}

int thip_enc_frag_metric_sites_batch(int op, uint32_t *out, int32_t *dc_out, const uint8_t *src_plane, const uint8_t *ref_plane,
                                     int ystride, const int32_t *src_offs, const int32_t *ref_offs, const int8_t *site_dx,
                                     const int8_t *site_dy, int nsites, int64_t nblocks) {
  if (!out || !src_plane || !ref_plane || !src_offs || !ref_offs || !site_dx || !site_dy) return THIP_EFAULT;
  if ((op != THIP_ENC_SAD && op != THIP_ENC_SATD) || nsites < 1 || nsites > 9 || nblocks < 0) return THIP_EINVAL;
  SitesK K;
  for (int k = 0; k < 9; k++) K.site_of[k] = -1;
  K.nsites = nsites;
  for (int c = 0; c < nsites; c++) {
    if (site_dx[c] < -1 || site_dx[c] > 1 || site_dy[c] < -1 || site_dy[c] > 1) return THIP_EINVAL;
    int8_t &slot = K.site_of[(site_dy[c] + 1) * 3 + (site_dx[c] + 1)];
    if (slot >= 0) return THIP_EINVAL;   // a position asked for twice
    slot = (int8_t)c;
  }
  if (nblocks == 0) return THIP_OK;
  if (op == THIP_ENC_SAD)
    hipLaunchKernelGGL(k_enc_sites<THIP_ENC_SAD>, grid_for(3 * nblocks), dim3(256), 0, g_batch_stream, out, dc_out, src_plane, ref_plane,
                       ystride, src_offs, ref_offs, K, nblocks);
  else if (thip_option("enc_sites_lds") == 0)
    hipLaunchKernelGGL(k_enc_sites<THIP_ENC_SATD>, grid_for(3 * nblocks), dim3(256), 0, g_batch_stream, out, dc_out, src_plane, ref_plane,
                       ystride, src_offs, ref_offs, K, nblocks);
  else   // the source block shared by a block's three lanes through LDS (thip_enc.h)
    hipLaunchKernelGGL(k_enc_sites_satd, dim3((unsigned)((nblocks + 4 * kSiteBlocks - 1) / (4 * kSiteBlocks))), dim3(256), 0, g_batch_stream, out, dc_out,
                       src_plane, ref_plane, ystride, src_offs, ref_offs, K, nblocks);
  HIP_TRY(hipGetLastError());
  if (g_batch_sync) HIP_TRY(hipStreamSynchronize(g_batch_stream));
  return THIP_OK;
}

int thip_enc_frag_metric_halfpel_batch(int op, uint32_t *out, int32_t *dc_out, const uint8_t *src_plane, const uint8_t *ref_plane,
                                       int ystride, const int32_t *src_offs, const int32_t *ref_offs, const int16_t *vecs,
                                       const int8_t *site_dx, const int8_t *site_dy, int nsites, int64_t nblocks) {
  if (!out || !src_plane || !ref_plane || !src_offs || !ref_offs || !vecs || !site_dx || !site_dy) return THIP_EFAULT;
  if ((op != THIP_ENC_SAD2_THRESH && op != THIP_ENC_SATD2) || nsites < 1 || nsites > 8 || nblocks < 0) return THIP_EINVAL;
  SitesK K;
  for (int k = 0; k < 9; k++) K.site_of[k] = -1;
  K.nsites = nsites;
  for (int c = 0; c < nsites; c++) {
    if (site_dx[c] < -1 || site_dx[c] > 1 || site_dy[c] < -1 || site_dy[c] > 1 || (site_dx[c] == 0 && site_dy[c] == 0)) return THIP_EINVAL;
    int8_t &slot = K.site_of[(site_dy[c] + 1) * 3 + (site_dx[c] + 1)];
    if (slot >= 0) return THIP_EINVAL;   // a position asked for twice
    slot = (int8_t)c;
  }
  if (nblocks == 0) return THIP_OK;
  // (option enc_halfpel_lanes: 2 = a lane per side with four sites each -- every wave the same work, 7-12 % faster by rocprofv3 --,
  //  3 = a lane per dx)
  const int lanes = thip_option("enc_halfpel_lanes") == 3 ? 3 : 2;
  const dim3 grid((unsigned)((nblocks + 255) / 256), (unsigned)lanes);
#define THIP_HP_LAUNCH(OPK, LN)                                                                                                       \
  hipLaunchKernelGGL((k_enc_halfpel<OPK, LN>), grid, dim3(256), 0, g_batch_stream, out, dc_out, src_plane, ref_plane, ystride, src_offs, \
                     ref_offs, vecs, K, nblocks)
  if (op == THIP_ENC_SAD2_THRESH) {
    if (lanes == 2) THIP_HP_LAUNCH(THIP_ENC_SAD2_THRESH, 2);
    else THIP_HP_LAUNCH(THIP_ENC_SAD2_THRESH, 3);
  } else {
    if (lanes == 2) THIP_HP_LAUNCH(THIP_ENC_SATD2, 2);
    else THIP_HP_LAUNCH(THIP_ENC_SATD2, 3);
  }
#undef THIP_HP_LAUNCH
  HIP_TRY(hipGetLastError());
  if (g_batch_sync) HIP_TRY(hipStreamSynchronize(g_batch_stream));
  return THIP_OK;
}

int thip_enc_frag_border_ssd_batch(uint32_t *out, const uint8_t *src_plane, const uint8_t *ref_plane,
                                   int ystride, const int32_t *src_offs, const int32_t *ref_offs,
                                   const int64_t *masks, int64_t n) {
  if (!out || !src_plane || !ref_plane || !src_offs || !ref_offs || !masks) return THIP_EFAULT;
  if (n < 0) return THIP_EINVAL;
  if (n == 0) return THIP_OK;
  hipLaunchKernelGGL(k_enc_border_ssd, grid_for(n), dim3(256), 0, g_batch_stream, out, src_plane, ref_plane, ystride,
                     src_offs, ref_offs, masks, n);
  HIP_TRY(hipGetLastError());
  if (g_batch_sync) HIP_TRY(hipStreamSynchronize(g_batch_stream));
  return THIP_OK;
}

int thip_enc_frag_sub_batch(int16_t *diff, const uint8_t *src_plane, const uint8_t *ref_plane, int ystride,
                            const int32_t *src_offs, const int32_t *ref_offs, int64_t n) {
  if (!diff || !src_plane || !src_offs || (ref_offs && !ref_plane)) return THIP_EFAULT;
  if (n < 0) return THIP_EINVAL;
  if (n == 0) return THIP_OK;
  hipLaunchKernelGGL(k_enc_sub, grid_for(n), dim3(256), 0, g_batch_stream, diff, src_plane, ref_plane, ystride, src_offs,
                     ref_offs, n);
  HIP_TRY(hipGetLastError());
  if (g_batch_sync) HIP_TRY(hipStreamSynchronize(g_batch_stream));
  return THIP_OK;
}

int thip_enc_frag_copy2_batch(uint8_t *dst_plane, const uint8_t *src_plane, int ystride,
                              const int32_t *dst_offs, const int32_t *src1_offs, const int32_t *src2_offs,
                              int64_t n) {
  if (!dst_plane || !src_plane || !dst_offs || !src1_offs || !src2_offs) return THIP_EFAULT;
  if (n < 0) return THIP_EINVAL;
  if (n == 0) return THIP_OK;
  hipLaunchKernelGGL(k_enc_copy2, grid_for(n), dim3(256), 0, g_batch_stream, dst_plane, src_plane, ystride, dst_offs,
                     src1_offs, src2_offs, n);
  HIP_TRY(hipGetLastError());
  if (g_batch_sync) HIP_TRY(hipStreamSynchronize(g_batch_stream));
  return THIP_OK;
}

int thip_enc_fdct8x8_batch(int16_t *y, const int16_t *x, int64_t n) {
  if (!y || !x) return THIP_EFAULT;
  if (n < 0) return THIP_EINVAL;
  if (n == 0) return THIP_OK;
  // four lanes per block (option "enc_fdct_lanes" = 1: one block per lane, the kernel of rounds 1-5)
  if (thip_option("enc_fdct_lanes") == 1)
    hipLaunchKernelGGL(k_enc_fdct, grid_for(n), dim3(256), 0, g_batch_stream, y, x, n);
  else
    hipLaunchKernelGGL(k_enc_fdct4, grid_for(4 * n), dim3(256), 0, g_batch_stream, y, x, n);
  HIP_TRY(hipGetLastError());
  if (g_batch_sync) HIP_TRY(hipStreamSynchronize(g_batch_stream));
  return THIP_OK;
}

// ---------------------------------------------------------------------------------------
// The single-block slots of oc_enc_opt_vtable (encint.h:292-326) with the reference's own
// signatures -- HOST pointers, one 8x8 block, a value returned -- each bound to a one-element
// batch of the kernels above.  This is what oc_enc_accel_init_hip fills the vtable with
//  so that the encoder's existing call sites keep working and each slot
// can be compared with its C original one call at a time; an encoder that wants throughput calls
// the batch entry points.  A call stages its blocks through a small per-thread pinned/device
// scratch and waits for the result: tens of microseconds, by design.
// ---------------------------------------------------------------------------------------
namespace {
struct Enc1 {
  uint8_t *h = nullptr, *d = nullptr;   // 2 KB each: pinned host image and device copy of the layout below
  ~Enc1() {
    if (h) (void)hipHostFree(h);
    if (d) (void)hipFree(d);
  }
};
// layout (bytes): 0 src[64] | 64 ref1[64] | 128 ref2[64] | 192 dst[64] | 256 int16 in[64] | 384 int16 out[64] |
//                 512 u32 out | 516 i32 dc | 520 i32 nonzero | 528 offs {0, 64, +3 for copy2} | 552 mask i64 | 576 dequant u16[64] | 704 enquant[256]
constexpr int kE1Src = 0, kE1Ref1 = 64, kE1Ref2 = 128, kE1Dst = 192, kE1In = 256, kE1Out = 384, kE1Val = 512, kE1Dc = 516,
              kE1Nz = 520, kE1Offs = 528, kE1Mask = 552, kE1Deq = 576, kE1Enq = 704, kE1Bytes = 1024;
thread_local Enc1 t_e1;

bool e1_ready() {
  if (t_e1.h) return true;
  if (hipHostMalloc((void **)&t_e1.h, kE1Bytes, hipHostMallocDefault) != hipSuccess) return false;
  if (hipMalloc((void **)&t_e1.d, kE1Bytes) != hipSuccess) return false;
  memset(t_e1.h, 0, kE1Bytes);
  ((int32_t *)(t_e1.h + kE1Offs))[1] = 64;
  return true;
}
void e1_block(int at, const unsigned char *p, int ystride) {
  for (int r = 0; r < 8; r++) memcpy(t_e1.h + at + 8 * r, p + (ptrdiff_t)r * ystride, 8);
}
bool e1_up() { return hipMemcpy(t_e1.d, t_e1.h, kE1Bytes, hipMemcpyHostToDevice) == hipSuccess; }
bool e1_down() {
  if (hipStreamSynchronize(g_batch_stream) != hipSuccess) return false;
  return hipMemcpy(t_e1.h, t_e1.d, kE1Bytes, hipMemcpyDeviceToHost) == hipSuccess;
}
unsigned e1_metric(int op, int *dc, const unsigned char *src, const unsigned char *ref1, const unsigned char *ref2, int ystride,
                   unsigned thresh) {
  if (!e1_ready()) return 0;
  e1_block(kE1Src, src, ystride);
  if (ref1) e1_block(kE1Ref1, ref1, ystride);
  if (ref2) e1_block(kE1Ref2, ref2, ystride);
  if (!e1_up()) return 0;
  uint8_t *d = t_e1.d;
  const int32_t *offs = (const int32_t *)(d + kE1Offs);
  if (thip_enc_frag_metric_batch(op, (uint32_t *)(d + kE1Val), (int32_t *)(d + kE1Dc), d + kE1Src, d + kE1Ref1, 8, offs, offs,
                                 offs + 1, thresh, 1) < 0 || !e1_down())
    return 0;
  if (dc) *dc = *(int32_t *)(t_e1.h + kE1Dc);
  return *(uint32_t *)(t_e1.h + kE1Val);
}
}  // namespace

void thip_enc1_frag_sub(int16_t diff[64], const unsigned char *src, const unsigned char *ref, int ystride) {
  if (!e1_ready()) return;
  e1_block(kE1Src, src, ystride);
  e1_block(kE1Ref1, ref, ystride);
  const int32_t *offs = (const int32_t *)(t_e1.d + kE1Offs);
  if (!e1_up() || thip_enc_frag_sub_batch((int16_t *)(t_e1.d + kE1Out), t_e1.d + kE1Src, t_e1.d + kE1Ref1, 8, offs, offs, 1) < 0 ||
      !e1_down())
    return;
  memcpy(diff, t_e1.h + kE1Out, 128);
}
void thip_enc1_frag_sub_128(int16_t diff[64], const unsigned char *src, int ystride) {
  if (!e1_ready()) return;
  e1_block(kE1Src, src, ystride);
  const int32_t *offs = (const int32_t *)(t_e1.d + kE1Offs);
  if (!e1_up() || thip_enc_frag_sub_batch((int16_t *)(t_e1.d + kE1Out), t_e1.d + kE1Src, nullptr, 8, offs, nullptr, 1) < 0 ||
      !e1_down())
    return;
  memcpy(diff, t_e1.h + kE1Out, 128);
}
unsigned thip_enc1_frag_sad(const unsigned char *src, const unsigned char *ref, int ystride) {
  return e1_metric(THIP_ENC_SAD, nullptr, src, ref, nullptr, ystride, 0);
}
unsigned thip_enc1_frag_sad_thresh(const unsigned char *src, const unsigned char *ref, int ystride, unsigned thresh) {
  return e1_metric(THIP_ENC_SAD_THRESH, nullptr, src, ref, nullptr, ystride, thresh);
}
unsigned thip_enc1_frag_sad2_thresh(const unsigned char *src, const unsigned char *ref1, const unsigned char *ref2, int ystride,
                                    unsigned thresh) {
  return e1_metric(THIP_ENC_SAD2_THRESH, nullptr, src, ref1, ref2, ystride, thresh);
}
unsigned thip_enc1_frag_intra_sad(const unsigned char *src, int ystride) {
  return e1_metric(THIP_ENC_INTRA_SAD, nullptr, src, nullptr, nullptr, ystride, 0);
}
unsigned thip_enc1_frag_satd(int *dc, const unsigned char *src, const unsigned char *ref, int ystride) {
  return e1_metric(THIP_ENC_SATD, dc, src, ref, nullptr, ystride, 0);
}
unsigned thip_enc1_frag_satd2(int *dc, const unsigned char *src, const unsigned char *ref1, const unsigned char *ref2, int ystride) {
  return e1_metric(THIP_ENC_SATD2, dc, src, ref1, ref2, ystride, 0);
}
unsigned thip_enc1_frag_intra_satd(int *dc, const unsigned char *src, int ystride) {
  return e1_metric(THIP_ENC_INTRA_SATD, dc, src, nullptr, nullptr, ystride, 0);
}
unsigned thip_enc1_frag_ssd(const unsigned char *src, const unsigned char *ref, int ystride) {
  return e1_metric(THIP_ENC_SSD, nullptr, src, ref, nullptr, ystride, 0);
}
unsigned thip_enc1_frag_border_ssd(const unsigned char *src, const unsigned char *ref, int ystride, int64_t mask) {
  if (!e1_ready()) return 0;
  e1_block(kE1Src, src, ystride);
  e1_block(kE1Ref1, ref, ystride);
  *(int64_t *)(t_e1.h + kE1Mask) = mask;
  const int32_t *offs = (const int32_t *)(t_e1.d + kE1Offs);
  if (!e1_up() || thip_enc_frag_border_ssd_batch((uint32_t *)(t_e1.d + kE1Val), t_e1.d + kE1Src, t_e1.d + kE1Ref1, 8, offs, offs,
                                                 (const int64_t *)(t_e1.d + kE1Mask), 1) < 0 || !e1_down())
    return 0;
  return *(uint32_t *)(t_e1.h + kE1Val);
}
void thip_enc1_frag_copy2(unsigned char *dst, const unsigned char *src1, const unsigned char *src2, int ystride) {
  if (!e1_ready()) return;
  e1_block(kE1Ref1, src1, ystride);
  e1_block(kE1Ref2, src2, ystride);
  const int32_t *offs = (const int32_t *)(t_e1.d + kE1Offs);
  // dst at kE1Dst = kE1Ref1 + 128: one plane pointer (ref1), offsets 128 / 0 / 64
  static const int32_t h_offs3[3] = {kE1Dst - kE1Ref1, 0, kE1Ref2 - kE1Ref1};
  memcpy(t_e1.h + kE1Offs + 8, h_offs3, sizeof(h_offs3));
  if (!e1_up() || thip_enc_frag_copy2_batch(t_e1.d + kE1Ref1, t_e1.d + kE1Ref1, 8, offs + 2, offs + 3, offs + 4, 1) < 0 || !e1_down())
    return;
  for (int r = 0; r < 8; r++) memcpy(dst + (ptrdiff_t)r * ystride, t_e1.h + kE1Dst + 8 * r, 8);
}
int thip_enc1_quantize(int16_t qdct[64], const int16_t dct[64], const uint16_t dequant[64], const void *enquant) {
  if (!e1_ready()) return 0;
  memcpy(t_e1.h + kE1In, dct, 128);
  memcpy(t_e1.h + kE1Deq, dequant, 128);
  memcpy(t_e1.h + kE1Enq, enquant, THIP_ENQUANT_TABLE_SIZE);
  if (!e1_up() || thip_enc_quantize_tab_batch((int16_t *)(t_e1.d + kE1Out), (int32_t *)(t_e1.d + kE1Nz), (const int16_t *)(t_e1.d + kE1In),
                                              (const uint16_t *)(t_e1.d + kE1Deq), t_e1.d + kE1Enq, 1) < 0 || !e1_down())
    return 0;
  memcpy(qdct, t_e1.h + kE1Out, 128);
  return *(int32_t *)(t_e1.h + kE1Nz);
}
void thip_enc1_frag_recon_intra(unsigned char *dst, int ystride, const int16_t residue[64]) {
  if (!e1_ready()) return;
  memcpy(t_e1.h + kE1In, residue, 128);
  const int32_t *offs = (const int32_t *)(t_e1.d + kE1Offs);
  if (!e1_up() || thip_frag_recon_batch(t_e1.d + kE1Dst, nullptr, 8, 0, offs, nullptr, nullptr, (const int16_t *)(t_e1.d + kE1In), 1) < 0 ||
      !e1_down())
    return;
  for (int r = 0; r < 8; r++) memcpy(dst + (ptrdiff_t)r * ystride, t_e1.h + kE1Dst + 8 * r, 8);
}
void thip_enc1_frag_recon_inter(unsigned char *dst, const unsigned char *src, int ystride, const int16_t residue[64]) {
  if (!e1_ready()) return;
  e1_block(kE1Ref1, src, ystride);
  memcpy(t_e1.h + kE1In, residue, 128);
  const int32_t *offs = (const int32_t *)(t_e1.d + kE1Offs);
  if (!e1_up() || thip_frag_recon_batch(t_e1.d + kE1Dst, t_e1.d + kE1Ref1, 8, 1, offs, offs, nullptr,
                                        (const int16_t *)(t_e1.d + kE1In), 1) < 0 || !e1_down())
    return;
  for (int r = 0; r < 8; r++) memcpy(dst + (ptrdiff_t)r * ystride, t_e1.h + kE1Dst + 8 * r, 8);
}
void thip_enc1_fdct8x8(int16_t y[64], const int16_t x[64]) {
  if (!e1_ready()) return;
  memcpy(t_e1.h + kE1In, x, 128);
  if (!e1_up() || thip_enc_fdct8x8_batch((int16_t *)(t_e1.d + kE1Out), (const int16_t *)(t_e1.d + kE1In), 1) < 0 || !e1_down()) return;
  memcpy(y, t_e1.h + kE1Out, 128);
}

}  // extern "C"
