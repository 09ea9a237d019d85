// thip_decode.hip -- frame-scope reconstruction path for gfx950 (MI355X).
//
// Two kernels per batch of frames (one 8x8 block per lane, wave64):
//
//   k_recon       K1+K2.  One wave (= one workgroup) per TILE (4 super blocks = 16x4 fragments
//                 = 128x32 pixels), lanes in coded order (super block by super block, Hilbert
//                 inside).  Round trip 1: the 64 command words + the tile's first slot number.
//                 Round trip 2: coefficients (slot found by ballot / prefix count over the
//                 tile's mask, global -> LDS directly) and predictor windows, all issued
//                 before anything waits.  Then dequantised coefficients -> iDCT -> predictor
//                 -> pixels (state.c:959, idct.c:301, fragment.c:49-80); an uncoded fragment
//                 (fragment.c:37) is the zero-vector predictor plus a zero residual; DC-only
//                 fragments carry their value in the command word and own no slot.
//   k_loopfilter  K3.  Whole-frame in-loop deblocking (state.c:1055-1105), one 8x8 "corner
//                 cell" per lane; cells are independent, so the pass is fully parallel and a
//                 cell is one memory round trip.
//
// What makes these kernels fast or slow on this chip is not arithmetic but (a) how many
// DEPENDENT memory round trips a wave makes (kernel arguments indexed by a value the compiler
// cannot prove uniform become vector loads; a register copy behind a load is a hidden wait),
// and (b) straggler waves.  tools/wave_trace.py shows both; DESIGN.md section 5 has the story.
//
// K4 (UMV border fill, state.c:770-835) does not exist: device frames are unpadded and
// motion-compensated reads clamp their coordinates, which is bit-identical.
//
// Host side (C ABI, include/theora_hip.h): stream state = three device frames + ring,
// batched launch over up to THIP_MAX_BATCH independent streams per kernel, pinned staging
// for the one-fragment-at-a-time vtable slots.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <mutex>
#include <vector>

#include "../../include/theora_hip.h"
#include "thip_device.h"

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

// ---------------------------------------------------------------------------------------
// geometry shared by host and device
// ---------------------------------------------------------------------------------------
// (row, col) of the h-th block on the 4x4 Hilbert curve of a super block, the order of
// coded_fragis inside a super block (state.c:134-139), two bits per entry.
constexpr uint32_t kHilbRow = 0u | 0u << 2 | 1u << 4 | 1u << 6 | 2u << 8 | 3u << 10 | 3u << 12 | 2u << 14 |
                              2u << 16 | 3u << 18 | 3u << 20 | 2u << 22 | 1u << 24 | 1u << 26 | 0u << 28 | 0u << 30;
constexpr uint32_t kHilbCol = 0u | 1u << 2 | 1u << 4 | 0u << 6 | 0u << 8 | 0u << 10 | 1u << 12 | 1u << 14 |
                              2u << 16 | 2u << 18 | 3u << 20 | 3u << 22 | 3u << 24 | 2u << 26 | 2u << 28 | 3u << 30;
// inverse: Hilbert index of (row, col), four bits per entry, entry = row*4+col
constexpr uint64_t kHilbInv = 0ull | 1ull << 4 | 14ull << 8 | 15ull << 12 |      // row 0
                              3ull << 16 | 2ull << 20 | 13ull << 24 | 12ull << 28 |   // row 1
                              4ull << 32 | 7ull << 36 | 8ull << 40 | 11ull << 44 |    // row 2
                              5ull << 48 | 6ull << 52 | 9ull << 56 | 10ull << 60;     // row 3

__host__ __device__ inline int hilb_row(int h) { return (int)((kHilbRow >> (2 * h)) & 3u); }
__host__ __device__ inline int hilb_col(int h) { return (int)((kHilbCol >> (2 * h)) & 3u); }
__host__ __device__ inline int hilb_inv(int r, int c) { return (int)((kHilbInv >> (4 * (r * 4 + c))) & 15ull); }

struct PlaneK {
  int nh, nv;        // fragments across / down
  int stride;        // device pitch
  int off;           // byte offset of the plane in a frame
  int tiles_x;       // tiles across
  int tile_off;      // index of the plane's first tile
  int fro;           // raster index of the plane's first fragment
  float rcp_cx;      // 1/(nh+1)
  int nseg, seglen;  // a tile row is cut into nseg segments of seglen tiles (k_recon_lf)
  int seam_rows;     // cell rows filtered by k_lf_seam: m = 0,4,8,... and m = nv
};

struct StreamK {
  const uint2 *info;
  const int4 *coeffs;
  const uint32_t *tile_slot0;
  uint8_t *self;
  const uint8_t *prev;
  const uint8_t *gold;
  uint8_t *coded_map;     // 1 byte per fragment, raster order: written by k_recon, read by k_loopfilter
  int flimit2;            // 2*flimit
  int qpx, qpy;           // chroma axis decimated (quarter-pel chroma vectors)
  int tile_end[3];        // cumulative tile counts per plane (k_recon: one wave per tile)
  int cell_end[3];        // cumulative filter-cell counts per plane, each plane padded to 64 (k_loopfilter)
  int lf_y0[3], lf_y1[3]; // fragment-row range whose filter operations are applied
  int debug;              // ablation switches for profiling (THIP_DEBUG env), 0 in production
  // fused reconstruction + loop filter (k_recon_lf / k_lf_seam)
  int seg_end[3];         // cumulative workgroup counts per plane: one workgroup per (tile row, segment)
  int seam_end[3];        // cumulative seam-cell counts per plane, each plane padded to 64
  PlaneK pl[3];
};

struct BatchK {
  StreamK s[THIP_MAX_BATCH];
};

// ---------------------------------------------------------------------------------------
// in-loop filter on a register image of one 8x8 "cell"
// ---------------------------------------------------------------------------------------
// The reference filters, for every coded fragment in raster order, its left edge, its
// edge towards the previous fragment row, then its right / next-row edge when that
// neighbour is uncoded (state.c:1083-1104).  Order only matters where a vertical-edge
// filter and a horizontal-edge filter touch the same pixels: the 4x4 patch centred on a
// fragment corner.  The 8x8 "cell" centred on corner (k,m) -- pixels
// [8k-4,8k+4)x[8m-4,8m+4) -- contains, completely, four half-edge operations:
//   Vlo  vertical edge x=8k, fragment row m-1, its rows 4..7   (cell rows 0..3)
//   Vhi  vertical edge x=8k, fragment row m,   its rows 0..3   (cell rows 4..7)
//   Hl   horizontal edge y=8m, fragment column k-1, columns 4..7 (cell cols 0..3)
//   Hr   horizontal edge y=8m, fragment column k,   columns 0..3 (cell cols 4..7)
// and nothing else reads or writes those pixels, so cells are independent and tile the
// plane.  Inside a cell the operations run in the reference's order, which depends on which
// of the four fragments around the corner are coded (a=(k-1,m-1) b=(k,m-1) c=(k-1,m)
// d=(k,m)): raster time of an op = (row, column, slot) of the fragment that triggers it,
// slots left=0, previous-row=1, right=2, next-row=3.  Sorted, the eight candidates are
//   T1 Vlo by a (!b)   T2 Hl by a (!c)   T3 Vlo by b   T4 Hr by b (!d)
//   T5 Hl by c         T6 Vhi by c (!d)  T7 Vhi by d   T8 Hr by d
__device__ __forceinline__ void lf_vert(int P[64], int r0, int L2) {
#pragma unroll
  for (int r = r0; r < r0 + 4; r++) {
    int f = P[r * 8 + 2] - P[r * 8 + 5] + 3 * (P[r * 8 + 4] - P[r * 8 + 3]);  // state.c:1007
    f = lflim((f + 4) >> 3, L2);
    P[r * 8 + 3] = clamp255(P[r * 8 + 3] + f);
    P[r * 8 + 4] = clamp255(P[r * 8 + 4] - f);
  }
}
__device__ __forceinline__ void lf_horz(int P[64], int c0, int L2) {
#pragma unroll
  for (int c = c0; c < c0 + 4; c++) {
    int f = P[2 * 8 + c] - P[5 * 8 + c] + 3 * (P[4 * 8 + c] - P[3 * 8 + c]);  // state.c:1023
    f = lflim((f + 4) >> 3, L2);
    P[3 * 8 + c] = clamp255(P[3 * 8 + c] + f);
    P[4 * 8 + c] = clamp255(P[4 * 8 + c] - f);
  }
}

// Which of T1..T8 apply to cell (k,m) of a plane with nh x nv fragments, given the coded
// flags around the corner and the fragment-row range [fy0,fy1) being filtered.  Bit i-1 of
// the result = Ti.
__device__ __forceinline__ uint32_t lf_cell_ops(int k, int m, int nh, int nv, bool a, bool b, bool c, bool d,
                                                int fy0, int fy1) {
  const bool kin = k >= 1 && k <= nh - 1;   // a vertical edge exists at x=8k
  const bool min_ = m >= 1 && m <= nv - 1;  // a horizontal edge exists at y=8m
  a = a && k >= 1 && m >= 1;
  b = b && k <= nh - 1 && m >= 1;
  c = c && k >= 1 && m <= nv - 1;
  d = d && k <= nh - 1 && m <= nv - 1;
  const bool rlo = (m - 1) >= fy0 && (m - 1) < fy1;  // ops triggered from fragment row m-1
  const bool rhi = m >= fy0 && m < fy1;              // ops triggered from fragment row m
  uint32_t t = 0;
  t |= (kin && a && !b && rlo) ? 1u : 0u;
  t |= (min_ && k >= 1 && a && !c && rlo) ? 2u : 0u;
  t |= (kin && b && rlo) ? 4u : 0u;
  t |= (min_ && k <= nh - 1 && b && !d && rlo) ? 8u : 0u;
  t |= (min_ && k >= 1 && c && rhi) ? 16u : 0u;
  t |= (kin && c && !d && rhi) ? 32u : 0u;
  t |= (kin && d && rhi) ? 64u : 0u;
  t |= (min_ && k <= nh - 1 && d && rhi) ? 128u : 0u;
  return t;
}

__device__ __forceinline__ void lf_cell_apply(int P[64], uint32_t t, int L2) {
  if (__any(t & 1u)) { if (t & 1u) lf_vert(P, 0, L2); }
  if (__any(t & 2u)) { if (t & 2u) lf_horz(P, 0, L2); }
  if (__any(t & 4u)) { if (t & 4u) lf_vert(P, 0, L2); }
  if (__any(t & 8u)) { if (t & 8u) lf_horz(P, 4, L2); }
  if (__any(t & 16u)) { if (t & 16u) lf_horz(P, 0, L2); }
  if (__any(t & 32u)) { if (t & 32u) lf_vert(P, 4, L2); }
  if (__any(t & 64u)) { if (t & 64u) lf_vert(P, 4, L2); }
  if (__any(t & 128u)) { if (t & 128u) lf_horz(P, 4, L2); }
}

__device__ __forceinline__ void unpack_row(int *P, uint32_t lo, uint32_t hi) {
#pragma unroll
  for (int q = 0; q < 4; q++) {
    P[q] = byte_of(lo, q);
    P[4 + q] = byte_of(hi, q);
  }
}

// A cell directly on a plane in memory (k_loopfilter, thip_loop_filter_plane).  Neither the
// pixel loads nor the flag loads depend on anything loaded before, and none of them is
// predicated: coordinates are clamped into the plane instead (border cells read a valid
// neighbour whose value is never used), so a cell costs ONE memory round trip of eight
// 8-byte loads + four flag bytes; cells without work skip the stores.
struct CellPix {
  uint32_t lo[8], hi[8];
};
struct __attribute__((aligned(4))) Pix8 {
  uint32_t x, y;
};

// ---- the filter on the packed image: two pixels per register -------------------------------------
// f = P2 - P5 + 3*(P4 - P3), R = (f+4)>>3, lflim(R), P3 += ., P4 -= . (state.c:1002-1031) in
// 16-bit lanes: |f| <= 1020, every intermediate fits; the final clamp is v_sat_pk_u8_i16.
__device__ __forceinline__ pk16 pk_lf_delta(pk16 p2, pk16 p3, pk16 p4, pk16 p5, int L2) {
  const pk16 d = p4 - p3;
  const pk16 f = p2 - p5 + d + d + d;
  const pk16 R = (f + (short)4) >> 3;
  const pk16 a = __builtin_elementwise_max(R, -R);
  const pk16 l2 = {(short)L2, (short)L2};
  const pk16 z = {0, 0};
  const pk16 m = __builtin_elementwise_min(a, __builtin_elementwise_max(l2 - a, z));
  const pk16 s = R >> 15;                      // 0 or -1 per half
  return as_pk(as_u32(m) ^ as_u32(s)) - s;     // R < 0 ? -m : m
}
// horizontal edge y = 4 of the cell, columns 0..3 (half 0: the lo dwords) or 4..7 (half 1)
__device__ __forceinline__ void lf_horz_pk(CellPix &C, int half, int L2) {
  uint32_t *w = half ? C.hi : C.lo;
  const pk16 d01 = pk_lf_delta(pk_bytes01(w[2]), pk_bytes01(w[3]), pk_bytes01(w[4]), pk_bytes01(w[5]), L2);
  const pk16 d23 = pk_lf_delta(pk_bytes23(w[2]), pk_bytes23(w[3]), pk_bytes23(w[4]), pk_bytes23(w[5]), L2);
  const uint32_t n3 = sat_pk_u8(pk_bytes01(w[3]) + d01) | sat_pk_u8(pk_bytes23(w[3]) + d23) << 16;
  const uint32_t n4 = sat_pk_u8(pk_bytes01(w[4]) - d01) | sat_pk_u8(pk_bytes23(w[4]) - d23) << 16;
  w[3] = n3;
  w[4] = n4;
}
// vertical edge x = 4 of the cell, rows r0..r0+3: columns 2,3 are bytes 2,3 of lo, columns 4,5
// bytes 0,1 of hi; two rows per register
__device__ __forceinline__ void lf_vert_pk(CellPix &C, int r0, int L2) {
#pragma unroll
  for (int r = r0; r < r0 + 4; r += 2) {
    const pk16 p2 = as_pk(__builtin_amdgcn_perm(C.lo[r + 1], C.lo[r], 0x0c060c02u));
    const pk16 p3 = as_pk(__builtin_amdgcn_perm(C.lo[r + 1], C.lo[r], 0x0c070c03u));
    const pk16 p4 = as_pk(__builtin_amdgcn_perm(C.hi[r + 1], C.hi[r], 0x0c040c00u));
    const pk16 p5 = as_pk(__builtin_amdgcn_perm(C.hi[r + 1], C.hi[r], 0x0c050c01u));
    const pk16 d = pk_lf_delta(p2, p3, p4, p5, L2);
    const uint32_t n3 = sat_pk_u8(p3 + d), n4 = sat_pk_u8(p4 - d);   // byte 0: row r, byte 1: row r+1
    C.lo[r] = __builtin_amdgcn_perm(n3, C.lo[r], 0x04020100u);       // byte 3 <- n3.byte0
    C.lo[r + 1] = __builtin_amdgcn_perm(n3, C.lo[r + 1], 0x05020100u);
    C.hi[r] = __builtin_amdgcn_perm(n4, C.hi[r], 0x03020104u);       // byte 0 <- n4.byte0
    C.hi[r + 1] = __builtin_amdgcn_perm(n4, C.hi[r + 1], 0x03020105u);
  }
}
__device__ __forceinline__ void lf_cell_apply_pk(CellPix &C, uint32_t t, int L2) {
  if (__any(t & 1u)) { if (t & 1u) lf_vert_pk(C, 0, L2); }
  if (__any(t & 2u)) { if (t & 2u) lf_horz_pk(C, 0, L2); }
  if (__any(t & 4u)) { if (t & 4u) lf_vert_pk(C, 0, L2); }
  if (__any(t & 8u)) { if (t & 8u) lf_horz_pk(C, 1, L2); }
  if (__any(t & 16u)) { if (t & 16u) lf_horz_pk(C, 0, L2); }
  if (__any(t & 32u)) { if (t & 32u) lf_vert_pk(C, 4, L2); }
  if (__any(t & 64u)) { if (t & 64u) lf_vert_pk(C, 4, L2); }
  if (__any(t & 128u)) { if (t & 128u) lf_horz_pk(C, 1, L2); }
}
__device__ __forceinline__ void lf_cell_load(CellPix &C, const uint8_t *plane, int stride, int nh, int nv, int k,
                                             int m) {
  const int W = nh * 8, H = nv * 8;
  const int xb = min(max(8 * k - 4, 0), W - 8);
#pragma unroll
  for (int r = 0; r < 8; r++) {
    const int y = min(max(8 * m - 4 + r, 0), H - 1);
    const Pix8 v = *reinterpret_cast<const Pix8 *>(plane + (ptrdiff_t)y * stride + xb);
    C.lo[r] = k == nh ? v.y : v.x;   // right border: the half that exists is the upper dword of the clamped load
    C.hi[r] = k == 0 ? v.x : v.y;    // left border: ... the lower dword
  }
}
// Keeps the pixel loads where they are written: without it the compiler sinks them below the
// "no work in this cell" test, i.e. behind the flag loads' round trip.
__device__ __forceinline__ void lf_cell_pin(const CellPix &C) {
  asm volatile("" ::"v"(C.lo[0]), "v"(C.lo[1]), "v"(C.lo[2]), "v"(C.lo[3]), "v"(C.lo[4]), "v"(C.lo[5]), "v"(C.lo[6]),
               "v"(C.lo[7]), "v"(C.hi[0]), "v"(C.hi[1]), "v"(C.hi[2]), "v"(C.hi[3]), "v"(C.hi[4]), "v"(C.hi[5]),
               "v"(C.hi[6]), "v"(C.hi[7]));
}
// coded flags of the four fragments around corner (k,m), clamped the same way
__device__ __forceinline__ void lf_cell_flags(const uint8_t *coded, int nh, int nv, int k, int m, bool &a, bool &b,
                                              bool &c, bool &d) {
  const int ka = max(k - 1, 0), kb = min(k, nh - 1), ma = max(m - 1, 0), mb = min(m, nv - 1);
  const uint8_t fa = coded[ma * nh + ka], fb = coded[ma * nh + kb], fc = coded[mb * nh + ka], fd = coded[mb * nh + kb];
  a = (k >= 1) & (m >= 1) & (fa != 0);
  b = (k <= nh - 1) & (m >= 1) & (fb != 0);
  c = (k >= 1) & (m <= nv - 1) & (fc != 0);
  d = (k <= nh - 1) & (m <= nv - 1) & (fd != 0);
}
__device__ __forceinline__ void lf_cell_finish(const CellPix &Cin, uint8_t *plane, int stride, int nh, int nv, int k,
                                               int m, uint32_t t, int L2) {
  if (!t) return;
  const bool lo_ok = k >= 1, hi_ok = k <= nh - 1;
  uint8_t *base = plane + (ptrdiff_t)(8 * m - 4) * stride + (8 * k - 4);
  const int H = nv * 8;
  CellPix C = Cin;
  lf_cell_apply_pk(C, t, L2);
#pragma unroll
  for (int r = 0; r < 8; r++) {
    const int y = 8 * m - 4 + r;
    if (y >= 0 && y < H) {
      uint8_t *p = base + (ptrdiff_t)r * stride;
      if (lo_ok & hi_ok) {
        Pix8 o;
        o.x = C.lo[r];
        o.y = C.hi[r];
        *reinterpret_cast<Pix8 *>(p) = o;
      } else if (lo_ok) {
        *reinterpret_cast<uint32_t *>(p) = C.lo[r];
      } else {
        *reinterpret_cast<uint32_t *>(p + 4) = C.hi[r];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// k_recon (K1 + K2): one wave per tile, one fragment per lane, in coded order
// ---------------------------------------------------------------------------------------
// residual of this lane's block as eight rows of packed int16 pairs
__device__ __forceinline__ void load_slot(const int4 *coeffs, uint32_t slot, uint32_t P[32]) {
  const int4 *tp = coeffs + ((size_t)(slot >> 6) * 512 + (slot & 63));
#pragma unroll
  for (int q = 0; q < 8; q++) {
    const int4 w = tp[q * 64];
    P[q * 4 + 0] = (uint32_t)w.x;
    P[q * 4 + 1] = (uint32_t)w.y;
    P[q * 4 + 2] = (uint32_t)w.z;
    P[q * 4 + 3] = (uint32_t)w.w;
  }
}

// Predictor of an inter block (fragment.c:59-80 with the offsets of state.c:846-957): one
// reference block or the truncating average of two.  Split in two so that the loads are in
// flight while the inverse DCT runs: pred_issue() only computes addresses and issues the
// row loads, pred_finish() turns the raw windows into the eight predictor rows.
struct PredWin {
  Row12 w[9];             // rows clamp(ys+r, 0, H-1), 12 bytes from column xw (see pred_xw)
  int sx, sy, mx2, my2;   // first sample's position, second sample's offset (0 or +-1 per axis)
  bool border;            // the footprint leaves the frame: replicated-border addressing
};

// First column of the 12-byte window: the footprint's first column aligned down to 4, pulled
// inside the row.  Every column a sample can need -- after clamping to [0,W-1], which is what
// the reference's replicated UMV border amounts to (state.c:770-835) -- lies inside it.
__device__ __forceinline__ int pred_xw(int xs, int W) { return max(min(xs, W - 12), 0) & ~3; }

__device__ __forceinline__ void pred_issue(PredWin &Q, const uint8_t *ref, int stride, int W, int H, int x0, int y0,
                                           uint32_t flags, bool qpx, bool qpy) {
  const int dx = (int)(int8_t)(flags >> THIP_INFO_MVX_SHIFT);
  const int dy = (int)(int8_t)(flags >> THIP_INFO_MVY_SHIFT);
  int mx, my;
  mv_axis(dx, qpx, mx, Q.mx2);
  mv_axis(dy, qpy, my, Q.my2);
  Q.sx = x0 + mx;
  Q.sy = y0 + my;
  const int xs = Q.sx + min(Q.mx2, 0), ys = Q.sy + min(Q.my2, 0);
  // (bitwise | on purpose: one compare chain, no nest of divergent branches)
  Q.border = ((int)(xs < 0) | (int)(Q.sx + max(Q.mx2, 0) + 8 > W) | (int)(ys < 0) | (int)(Q.sy + max(Q.my2, 0) + 8 > H)) != 0;
  // Both samples of a row lie in the 9 bytes starting at xs, i.e. inside one 12-byte window
  // aligned down to 4: one dword-aligned dwordx3 load per source row.  Vertical half-pel needs
  // 9 source rows, not 16; without it the ninth load re-reads row 7.  Row and column clamps are
  // no-ops for a footprint inside the frame, so there is one code path and no straggler waves.
  const uint8_t *p1 = ref + pred_xw(xs, W);
#pragma unroll
  for (int r = 0; r < 9; r++) {
    const int y = min(max(ys + (r < 8 ? r : (Q.my2 != 0 ? 8 : 7)), 0), H - 1);
    Q.w[r] = load_row12(p1 + (ptrdiff_t)y * stride);
  }
}

// byte selectors for v_perm_b32: the four window columns col0+i (clamped to the row), as
// offsets from the window dword pair {4k..4k+7}
__device__ __forceinline__ void pred_sel(int col0, int xw, int W, uint32_t &sel, bool &k) {
  int c[4];
#pragma unroll
  for (int i = 0; i < 4; i++) c[i] = min(max(col0 + i, 0), W - 1) - xw;   // 0..11, non-decreasing
  k = c[0] >= 4;
  const int o = k ? 4 : 0;
  sel = (uint32_t)(c[0] - o) | (uint32_t)(c[1] - o) << 8 | (uint32_t)(c[2] - o) << 16 | (uint32_t)(c[3] - o) << 24;
}
__device__ __forceinline__ uint32_t pred_pick(const Row12 &w, uint32_t sel, bool k) {
  return __builtin_amdgcn_perm(k ? w.c : w.b, k ? w.b : w.a, sel);
}

__device__ __forceinline__ void pred_finish(const PredWin &Q, int W, uint2 pred[8]) {
  const int xw = pred_xw(Q.sx + min(Q.mx2, 0), W);
  const bool ra = Q.my2 < 0, rb = Q.my2 > 0;   // that sample starts one source row down
  const bool two = (Q.mx2 | Q.my2) != 0;
  if (!__any(Q.border)) {
    const int offA = Q.sx - xw, offB = Q.sx + Q.mx2 - xw;   // 0..4
#pragma unroll
    for (int r = 0; r < 8; r++) {
      const Row12 wa = ra ? Q.w[r + 1] : Q.w[r];
      pred[r] = extract8(wa, offA);
      if (two) {
        const Row12 wb = rb ? Q.w[r + 1] : Q.w[r];
        const uint2 b = extract8(wb, offB);
        pred[r].x = avg4_trunc(pred[r].x, b.x);
        pred[r].y = avg4_trunc(pred[r].y, b.y);
      }
    }
  } else {
    // some lane of the wave needs replicated columns: general byte gather for the whole wave
    uint32_t sa0, sa1, sb0, sb1;
    bool ka0, ka1, kb0, kb1;
    pred_sel(Q.sx, xw, W, sa0, ka0);
    pred_sel(Q.sx + 4, xw, W, sa1, ka1);
    pred_sel(Q.sx + Q.mx2, xw, W, sb0, kb0);
    pred_sel(Q.sx + Q.mx2 + 4, xw, W, sb1, kb1);
#pragma unroll
    for (int r = 0; r < 8; r++) {
      const Row12 wa = ra ? Q.w[r + 1] : Q.w[r];
      pred[r] = make_uint2(pred_pick(wa, sa0, ka0), pred_pick(wa, sa1, ka1));
      if (two) {
        const Row12 wb = rb ? Q.w[r + 1] : Q.w[r];
        pred[r].x = avg4_trunc(pred[r].x, pred_pick(wb, sb0, kb0));
        pred[r].y = avg4_trunc(pred[r].y, pred_pick(wb, sb1, kb1));
      }
    }
  }
}

#ifndef THIP_RECON_WAVES
#define THIP_RECON_WAVES 4
#endif
// Waves per workgroup of k_recon.  Waves never cooperate, and a workgroup's wave slots and
// LDS only become reusable together, so siblings of different length idle slots: 1 is best.
#ifndef THIP_RECON_WG_WAVES
#define THIP_RECON_WG_WAVES 1
#endif

// Optional wave-timeline instrumentation (tools/wave_trace.py builds a private copy of the
// library with -DTHIP_TRACE; never defined in the product build).
#ifdef THIP_TRACE
__device__ unsigned long long *g_trace_buf;   // [blockIdx.y][tile][8]
__device__ __forceinline__ unsigned long long trace_now() {
  unsigned long long t;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
#define THIP_TR(rec, i) do { if (rec) (rec)[i] = trace_now(); } while (0)
#else
#define THIP_TR(rec, i) do { } while (0)
#endif

// What a lane knows after the first round trip.
struct ReconLane {
  uint32_t flags, dcp;            // command word 0 (0 past the ragged edge), DC-only value
  bool coded, dc_only, has_coeff;
  int x0, y0;                     // pixel position of the block in its plane
};
struct ReconPlane {               // wave-uniform
  uint8_t *self;
  const uint8_t *prev, *gold;
  uint8_t *coded_map;             // already offset to the plane's first fragment
  int nh, nv, stride;
  bool qpx, qpy;
  int debug;
  unsigned long long *tr;         // THIP_TRACE: this wave's record (lane 0 only), else null
};

// Steps 3-5 of k_recon.  IDCT: the wave's coefficient loads into P are in flight.  Two
// instantiations instead of one body with a merged P: a merge makes the register allocator
// copy loaded registers right behind the loads, i.e. wait for them before the predictor
// loads are even issued.
// LDSOUT: the rows go to the tile image in LDS (128-byte pitch, lds_img = this lane's block) for
// the fused loop filter instead of to the frame.
template <bool IDCT, bool LDSOUT = false>
__device__ __forceinline__ void recon_tail(const ReconPlane &R, const ReconLane &L, const uint4 *lds_coef,
                                           uint8_t *lds_img = nullptr) {
  uint8_t *dst = R.self + (ptrdiff_t)L.y0 * R.stride + L.x0;

  // ---- 3. predictor loads.  An uncoded fragment (fragment.c:20-47) is the zero-vector
  //         predictor from the previous frame plus a zero residual: the same code path. ------------
  const int refi = L.coded ? (int)((L.flags >> THIP_INFO_REFI_SHIFT) & 3u) : THIP_FRAME_PREV;
  const bool inter = refi != THIP_FRAME_SELF && !(R.debug & 2);
  const uint8_t *const ref = refi == THIP_FRAME_PREV ? R.prev : R.gold;
  PredWin Q;
  Q.border = false;
  if (inter) pred_issue(Q, ref, R.stride, R.nh * 8, R.nv * 8, L.x0, L.y0, L.coded ? L.flags : 0u, R.qpx, R.qpy);
  R.coded_map[(L.y0 >> 3) * R.nh + (L.x0 >> 3)] = L.coded ? 1 : 0;
  THIP_TR(R.tr, 2);   // every load of the second round trip is issued

  // ---- 4. residual: DC-only shortcut (state.c:967-975) or inverse DCT (idct.c:301) ------------
  uint32_t Y[32];
  if (IDCT) {
    uint32_t P[32];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the caller's LDS-DMA loads have landed (see k_recon)
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const uint4 w = lds_coef[q * 64];
      P[q * 4 + 0] = w.x;
      P[q * 4 + 1] = w.y;
      P[q * 4 + 2] = w.z;
      P[q * 4 + 3] = w.w;
    }
#ifdef THIP_TRACE
    asm volatile("" : "+v"(P[31]));
    THIP_TR(R.tr, 3);   // coefficients (and, with them, the predictor windows) have arrived
#endif
    const int last_zzi = (int)((L.flags >> THIP_INFO_LAST_ZZI_SHIFT) & 0x7Fu);
    pk_mask_by_last_zzi(P, last_zzi);
    const bool all_zz10 = !__any(L.has_coeff && last_zzi > 10);
    pk_idct8x8(P, Y, all_zz10);
  }
  if (!IDCT || !L.has_coeff) {   // DC-only: the pre-rounded value; uncoded: zero residual
    const uint32_t fill = L.dc_only ? L.dcp : 0u;
#pragma unroll
    for (int i = 0; i < 32; i++) Y[i] = fill;
  }

  // ---- 5. predictor rows (fragment.c:49-80: 128, one block, or the average of two),
  //         reconstruct and store (8 aligned bytes per lane per row) ------------------------------
  uint2 pred[8];
#pragma unroll
  for (int r = 0; r < 8; r++) pred[r] = make_uint2(0x80808080u, 0x80808080u);
  if (inter) pred_finish(Q, R.nh * 8, pred);
  if (LDSOUT) {
#pragma unroll
    for (int r = 0; r < 8; r++)
      *reinterpret_cast<uint2 *>(lds_img + r * 128) =
          pk_recon_row(as_pk(Y[r * 4 + 0]), as_pk(Y[r * 4 + 1]), as_pk(Y[r * 4 + 2]), as_pk(Y[r * 4 + 3]), pred[r]);
  } else if (!(R.debug & 4)) {
#pragma unroll
    for (int r = 0; r < 8; r++)
      store_row8(dst + (ptrdiff_t)r * R.stride,
                 pk_recon_row(as_pk(Y[r * 4 + 0]), as_pk(Y[r * 4 + 1]), as_pk(Y[r * 4 + 2]), as_pk(Y[r * 4 + 3]),
                              pred[r]));
  }
  THIP_TR(R.tr, 4);   // stores issued
}

// ---- k_recon in three parts, so that the residual can be computed by ALL lanes of the wave ------
// (recon_tail above is the same thing in one piece; the fused variant still uses it.)
__device__ __forceinline__ void recon_issue(const ReconPlane &R, const ReconLane &L, PredWin &Q, bool &inter,
                                            const uint8_t *&ref) {
  const int refi = L.coded ? (int)((L.flags >> THIP_INFO_REFI_SHIFT) & 3u) : THIP_FRAME_PREV;
  inter = refi != THIP_FRAME_SELF && !(R.debug & 2);
  ref = refi == THIP_FRAME_PREV ? R.prev : R.gold;
  Q.border = false;
  if (inter) pred_issue(Q, ref, R.stride, R.nh * 8, R.nv * 8, L.x0, L.y0, L.coded ? L.flags : 0u, R.qpx, R.qpy);
  R.coded_map[(L.y0 >> 3) * R.nh + (L.x0 >> 3)] = L.coded ? 1 : 0;
}

__device__ __forceinline__ void recon_finish(const ReconPlane &R, const ReconLane &L, const PredWin &Q, bool inter,
                                             const uint32_t Y[32]) {
  uint8_t *dst = R.self + (ptrdiff_t)L.y0 * R.stride + L.x0;
  uint2 pred[8];
#pragma unroll
  for (int r = 0; r < 8; r++) pred[r] = make_uint2(0x80808080u, 0x80808080u);
  if (inter) pred_finish(Q, R.nh * 8, pred);
  if (!(R.debug & 4)) {
#pragma unroll
    for (int r = 0; r < 8; r++)
      store_row8(dst + (ptrdiff_t)r * R.stride,
                 pk_recon_row(as_pk(Y[r * 4 + 0]), as_pk(Y[r * 4 + 1]), as_pk(Y[r * 4 + 2]), as_pk(Y[r * 4 + 3]),
                              pred[r]));
  }
}

// Residual of the lanes that own coefficients, one block per lane (the whole wave executes the
// 16 one-dimensional transforms whether 1 lane or 64 need them).
__device__ __forceinline__ void residual_per_lane(const uint4 *lds_coef, const ReconLane &L, uint32_t Y[32]) {
  uint32_t P[32];
#pragma unroll
  for (int q = 0; q < 8; q++) {
    const uint4 w = lds_coef[q * 64];
    P[q * 4 + 0] = w.x;
    P[q * 4 + 1] = w.y;
    P[q * 4 + 2] = w.z;
    P[q * 4 + 3] = w.w;
  }
  const int last_zzi = (int)((L.flags >> THIP_INFO_LAST_ZZI_SHIFT) & 0x7Fu);
  pk_mask_by_last_zzi(P, last_zzi);
  const bool all_zz10 = !__any(L.has_coeff && last_zzi > 10);
  pk_idct8x8(P, Y, all_zz10);
}

// The same residuals when at most 64/LPB lanes of the wave own coefficients (the usual case
// outside synthetic worst cases: SURVEY section 6 has 80 % of the coded blocks DC-only): LPB lanes
// (4 or 2) share a block -- lane LPB*g+j takes row pairs j*NP..j*NP+NP-1 (NP = 4/LPB) of the
// g-th owner for the row pass and the same column pairs for the column pass, the transpose
// between goes through the wave's LDS area (free once the coefficients are in registers) -- so
// the wave executes 2*NP packed 1-D transforms instead of 8.  Bit-exact with residual_per_lane:
// the same operations on the same values.  Must be called by all 64 lanes.
// lds = the wave's 8 KB area as dwords; meta = 64/LPB dwords of LDS.
// The g-th owner's coefficients sit in slot slot0+g (slots are numbered in lane order inside a
// tile), so the sharing lanes fetch their own row pairs straight from the slot -- 32*NP bytes per
// lane instead of the whole wave staging 8 KB of which a fraction is used.
template <int LPB>
__device__ __forceinline__ void residual_shared_load(const int4 *coeffs, uint32_t slot0, int nown, int lane,
                                                     int4 W[4 / LPB][2]) {
  constexpr int NP = 4 / LPB;
  const int g = min(lane / LPB, nown - 1), j = lane % LPB;   // surplus groups re-read the last owner's slot
  const uint32_t slot = slot0 + (uint32_t)g;
  const int4 *tp = coeffs + ((size_t)(slot >> 6) * 512 + (slot & 63));
#pragma unroll
  for (int n = 0; n < NP; n++) {
    const int rp = j * NP + n;
    W[n][0] = tp[(2 * rp) * 64];
    W[n][1] = tp[(2 * rp + 1) * 64];
  }
}

template <int LPB>
__device__ __forceinline__ void residual_shared(const int4 W[4 / LPB][2], uint32_t *lds, uint32_t *meta, int lane,
                                                const ReconLane &L, uint32_t prefix, uint32_t Y[32]) {
  constexpr int NP = 4 / LPB;                        // row pairs (and column pairs) per lane
  const int last_zzi = (int)((L.flags >> THIP_INFO_LAST_ZZI_SHIFT) & 0x7Fu);
  if (L.has_coeff) meta[prefix] = (uint32_t)last_zzi;   // rank -> last_zzi of that owner
  const int g = lane / LPB, j = lane % LPB;
  const int lz = (int)(meta[g] & 0x7Fu);            // (garbage for g >= number of owners: results unused)
  const bool c3 = lz <= 3, c10 = lz <= 10;
  pk16 Rr[NP][8];
#pragma unroll
  for (int n = 0; n < NP; n++) {
    const int rp = j * NP + n;                       // row pair: rows 2rp, 2rp+1
    const int4 w0 = W[n][0], w1 = W[n][1];
    const uint32_t P8[8] = {(uint32_t)w0.x, (uint32_t)w0.y, (uint32_t)w0.z, (uint32_t)w0.w,
                            (uint32_t)w1.x, (uint32_t)w1.y, (uint32_t)w1.z, (uint32_t)w1.w};   // {x[2rp][c], x[2rp+1][c]}, c = 0..7
    // what the variant selected by last_zzi does not read is zero (pk_mask_by_last_zzi)
#pragma unroll
    for (int c = 0; c < 8; c++) {
      const uint32_t m10 = ((2 * rp + c <= 3) ? 0x0000FFFFu : 0u) | ((2 * rp + 1 + c <= 3) ? 0xFFFF0000u : 0u);
      const uint32_t m3 = ((rp == 0 && c <= 1) ? 0x0000FFFFu : 0u) | ((rp == 0 && c == 0) ? 0xFFFF0000u : 0u);
      Rr[n][c] = as_pk(P8[c] & (c3 ? m3 : (c10 ? m10 : 0xFFFFFFFFu)));
    }
    pk_idct8(Rr[n][0], Rr[n][1], Rr[n][2], Rr[n][3], Rr[n][4], Rr[n][5], Rr[n][6], Rr[n][7]);
  }
  // exchange inside the group: every lane publishes its row pairs, collects its column pairs
  uint32_t *xch = lds;                       // NP*8 dwords per lane
  uint32_t *res = lds + 64 * NP * 8;         // 32 dwords per owner
#pragma unroll
  for (int n = 0; n < NP; n++) {
    uint4 *x4 = reinterpret_cast<uint4 *>(xch + (lane * NP + n) * 8);
    x4[0] = make_uint4(as_u32(Rr[n][0]), as_u32(Rr[n][1]), as_u32(Rr[n][2]), as_u32(Rr[n][3]));
    x4[1] = make_uint4(as_u32(Rr[n][4]), as_u32(Rr[n][5]), as_u32(Rr[n][6]), as_u32(Rr[n][7]));
  }
#pragma unroll
  for (int n = 0; n < NP; n++) {
    const int cp = j * NP + n;                       // column pair: columns 2cp, 2cp+1
    pk16 Q[8];
#pragma unroll
    for (int rp = 0; rp < 4; rp++) {                 // row pair rp lives at slot (g*LPB*NP + rp) = g*4 + rp
      const uint2 ab = *reinterpret_cast<const uint2 *>(xch + (g * 4 + rp) * 8 + 2 * cp);
      Q[2 * rp] = as_pk(__builtin_amdgcn_perm(ab.y, ab.x, 0x05040100u));
      Q[2 * rp + 1] = as_pk(__builtin_amdgcn_perm(ab.y, ab.x, 0x07060302u));
    }
    pk_idct8(Q[0], Q[1], Q[2], Q[3], Q[4], Q[5], Q[6], Q[7]);
#pragma unroll
    for (int r = 0; r < 8; r++) res[g * 32 + r * 4 + cp] = as_u32(pk_descale(Q[r]));   // Y[r*4+k] layout of the owner
  }
  if (L.has_coeff) {
    const uint4 *y4 = reinterpret_cast<const uint4 *>(res + prefix * 32);
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const uint4 w = y4[q];
      Y[q * 4 + 0] = w.x;
      Y[q * 4 + 1] = w.y;
      Y[q * 4 + 2] = w.z;
      Y[q * 4 + 3] = w.w;
    }
  }
}

// A wave's life is exactly two memory round trips: (1) its 64 command words and the tile's
// first slot number, (2) coefficients and predictor windows, all issued before anything
// waits.  Everything read from the kernel arguments is wave-uniform and is forced into
// scalar registers (readfirstlane on the tile number), so the per-plane table lookups are
// scalar loads, not dependent vector loads.
__global__ __launch_bounds__(64 * THIP_RECON_WG_WAVES, THIP_RECON_WAVES) void k_recon(const BatchK B) {
  const StreamK &S = B.s[blockIdx.y];
  const int lane = (int)threadIdx.x & 63;
  // Workgroups are handed to the 8 XCDs round-robin (id mod 8) and each XCD has its own L2:
  // give XCD x the x-th contiguous band of tiles, so that the predictor windows of
  // neighbouring tiles -- which overlap by up to 16 pixels -- meet in one L2 instead of being
  // fetched from HBM by two.  gridDim.x is a multiple of 8.
  const int wg = ((int)blockIdx.x & 7) * ((int)gridDim.x >> 3) + ((int)blockIdx.x >> 3);
  const int unit = __builtin_amdgcn_readfirstlane(wg * THIP_RECON_WG_WAVES + ((int)threadIdx.x >> 6));  // tile
#ifdef THIP_TRACE
  unsigned long long *tr = nullptr;
  if (g_trace_buf && lane == 0) {
    tr = g_trace_buf + ((size_t)blockIdx.y * (gridDim.x * THIP_RECON_WG_WAVES) + unit) * 8;
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hwid), "=s"(xcc));
    tr[5] = hwid;
    tr[6] = xcc;
  }
  THIP_TR(tr, 0);
#endif
  // scalar batch 1: the stream's pointers and the plane boundaries (pinned by the empty asm:
  // left alone, the compiler sinks each scalar load to its first use, which turns one wait
  // into a chain of dependent ones)
  const uint2 *info_p = S.info;
  const int4 *coeffs_p = S.coeffs;
  const uint32_t *slot0_p = S.tile_slot0;
  uint8_t *self = S.self;
  const uint8_t *prev = S.prev, *gold = S.gold;
  uint8_t *coded_map = S.coded_map;
  const int te0 = S.tile_end[0], te1 = S.tile_end[1], te2 = S.tile_end[2];
  const int debug = S.debug, sqpx = S.qpx, sqpy = S.qpy;
  asm volatile("" ::"s"(info_p), "s"(coeffs_p), "s"(slot0_p), "s"(self), "s"(prev), "s"(gold), "s"(coded_map),
               "s"(te0), "s"(te1), "s"(te2), "s"(debug), "s"(sqpx), "s"(sqpy));
  if (unit >= te2) return;
  const int pli = (unit >= te0 ? 1 : 0) + (unit >= te1 ? 1 : 0);
  // scalar batch 2: the plane's geometry
  const PlaneK G = S.pl[pli];
  asm volatile("" ::"s"(G.nh), "s"(G.nv), "s"(G.stride), "s"(G.off), "s"(G.tiles_x), "s"(G.fro));

  // ---- 1. command word + first slot of the tile (one round trip) -------------------------------
  const uint32_t slot0 = slot0_p[unit];
  const uint2 info = info_p[(size_t)unit * THIP_TILE_FRAGS + lane];
  // both loads are consumed here as far as the compiler can tell, so the scalar load is issued
  // next to the vector load instead of being sunk behind the wait for it
  asm volatile("" ::"s"(slot0), "v"(info.x));
#ifdef THIP_TRACE
  THIP_TR(tr, 1);   // first round trip done
#endif

  const int rel = unit - (pli == 0 ? 0 : (pli == 1 ? te0 : te1));
  const int sby = rel / G.tiles_x;
  const int tx = rel - sby * G.tiles_x;
  const int h = lane & 15;
  const int bx = tx * 16 + (lane >> 4) * 4 + hilb_col(h);
  const int by = sby * 4 + hilb_row(h);
  const bool valid = bx < G.nh && by < G.nv;

  ReconLane L;
  L.flags = valid ? info.x : 0u;
  L.dcp = (info.y & 0xFFFFu) * 0x00010001u;   // {p, p}
  L.coded = (L.flags & THIP_INFO_CODED) != 0;
  L.dc_only = L.coded && (L.flags & THIP_INFO_DC_ONLY) != 0;
  L.has_coeff = L.coded && !L.dc_only;
  L.x0 = bx * 8;
  L.y0 = by * 8;
  ReconPlane R;
  R.self = self + G.off;
  R.prev = prev + G.off;
  R.gold = gold + G.off;
  R.coded_map = coded_map + G.fro;
  R.nh = G.nh;
  R.nv = G.nv;
  R.stride = G.stride;
  R.qpx = pli != 0 && sqpx;
  R.qpy = pli != 0 && sqpy;
  R.debug = debug;
#ifdef THIP_TRACE
  R.tr = tr;
#else
  R.tr = nullptr;
#endif

  // ---- 2. coefficient loads (slot by prefix count over the mask), issued, not waited for.
  //         The branch is wave-uniform and every lane loads (lanes without coefficients re-read
  //         the tile's first slot -- same cache lines, no extra traffic -- and are overridden
  //         in step 4). -----------------------------------------------------------------------------
  const uint64_t mask = __ballot(L.has_coeff);
#ifndef THIP_RECON_LDS_PAD
#define THIP_RECON_LDS_PAD 0
#endif
  __shared__ uint4 s_coef[THIP_RECON_WG_WAVES * 8 * 64 + THIP_RECON_LDS_PAD / 16];   // [wave][piece][lane]: 8 KB per wave, wave-private
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  uint4 *const lds_wave = s_coef + wave * 512;
  __shared__ uint32_t s_meta[THIP_RECON_WG_WAVES * 32];
  PredWin Q;
  bool inter = false;
  const uint8_t *ref = nullptr;
  uint32_t Y[32];
  const int nown = __popcll(mask);
  const uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
  uint32_t *const lds_dw = reinterpret_cast<uint32_t *>(lds_wave);
  if (nown == 0 || (debug & 9)) {
    if (valid) recon_issue(R, L, Q, inter, ref);   // (!valid: past the ragged edge of the plane)
  } else if (nown <= 16 && !(debug & 32)) {
    // ---- few owners: four lanes per block, pieces straight from the slots ------------------------
    int4 W[1][2];
    residual_shared_load<4>(coeffs_p, slot0, nown, lane, W);
    if (valid) recon_issue(R, L, Q, inter, ref);
    THIP_TR(R.tr, 2);
    residual_shared<4>(W, lds_dw, s_meta + wave * 32, lane, L, prefix, Y);
    THIP_TR(R.tr, 3);
  } else if (nown <= 32 && !(debug & 32)) {
    int4 W[2][2];
    residual_shared_load<2>(coeffs_p, slot0, nown, lane, W);
    if (valid) recon_issue(R, L, Q, inter, ref);
    THIP_TR(R.tr, 2);
    residual_shared<2>(W, lds_dw, s_meta + wave * 32, lane, L, prefix, Y);
    THIP_TR(R.tr, 3);
  } else {
    // ---- many owners: one lane per block.  Coefficients go global -> LDS directly (LDS address
    //      = wave-uniform base + lane*16): no VGPRs are tied up and nothing can make the compiler
    //      touch the data before the predictor loads are out.  Every lane loads (lanes without
    //      coefficients re-read the tile's first slot: same cache lines). ---------------------------
    const uint32_t slot = slot0 + (L.has_coeff ? prefix : 0u);
    const int4 *tp = coeffs_p + ((size_t)(slot >> 6) * 512 + (slot & 63));
    if (!(debug & 64)) {   // (ablation: transforms on whatever the LDS holds)
#pragma unroll
      for (int q = 0; q < 8; q++)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(tp + q * 64),
                                         (__attribute__((address_space(3))) void *)(lds_wave + q * 64), 16, 0, 0);
    }
    if (valid) recon_issue(R, L, Q, inter, ref);
    THIP_TR(R.tr, 2);
    // The LDS-DMA loads above are counted by vmcnt; the compiler's own wait before the LDS reads
    // below is not something to rely on (it vanished when the loads moved into a conditional
    // block and the reads returned stale LDS), so it is stated.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    residual_per_lane(lds_wave + lane, L, Y);
    THIP_TR(R.tr, 3);
  }
  if (!valid) return;
  if (!L.has_coeff || (debug & 9)) {   // DC-only: the pre-rounded value; uncoded: zero residual
    const uint32_t fill = L.dc_only ? L.dcp : 0u;
#pragma unroll
    for (int i = 0; i < 32; i++) Y[i] = fill;
  }
  recon_finish(R, L, Q, inter, Y);
  THIP_TR(R.tr, 4);
}

// ---------------------------------------------------------------------------------------
// k_recon_lf (K1+K2+most of K3): reconstruction fused with the loop filter
// ---------------------------------------------------------------------------------------
// One workgroup per SEGMENT of a tile row (up to 16 horizontally adjacent tiles, one wave each,
// exactly k_recon's two round trips per wave).  The reconstructed tiles go to LDS (each wave's
// 128x32 image re-uses its coefficient staging area), the workgroup meets at a barrier, and
// every filter cell that lies completely inside the segment's 32-pixel-high strip -- cell rows
// m = 1..3 of the tile row, all columns except the segment's outer edges -- is filtered there
// (cells are independent), then the strip is written to the frame ONCE.
// What is left for k_lf_seam are the cell rows on tile-row boundaries (m = 0 mod 4, and m = nv)
// and the columns on segment boundaries: about a quarter of the cells, read and written in full
// rows.  The separate k_loopfilter pass re-read and re-wrote every pixel.
constexpr int kSegMax = 8;             // waves per workgroup (8 x 8 KB of LDS: two workgroups per CU)
constexpr int kChunkBytes = 8192;      // LDS per wave: coefficient staging, then image (4096) + flags (64)

__device__ __forceinline__ uint32_t lds_px(const uint8_t *base, int x, int y) {   // 4 pixels at (x, y) of the strip
  return *reinterpret_cast<const uint32_t *>(base + (x >> 7) * kChunkBytes + y * 128 + (x & 127));
}
__device__ __forceinline__ void lds_px_store(uint8_t *base, int x, int y, uint32_t v) {
  *reinterpret_cast<uint32_t *>(base + (x >> 7) * kChunkBytes + y * 128 + (x & 127)) = v;
}

__global__ __launch_bounds__(64 * kSegMax, 1) void k_recon_lf(const BatchK B) {
  extern __shared__ uint4 s_dyn[];
  const StreamK &S = B.s[blockIdx.y];
  const int lane = (int)threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int wg = ((int)blockIdx.x & 7) * ((int)gridDim.x >> 3) + ((int)blockIdx.x >> 3);   // XCD bands, see k_recon
  const uint2 *info_p = S.info;
  const int4 *coeffs_p = S.coeffs;
  const uint32_t *slot0_p = S.tile_slot0;
  uint8_t *self = S.self;
  const uint8_t *prev = S.prev, *gold = S.gold;
  uint8_t *coded_map = S.coded_map;
  const int se0 = S.seg_end[0], se1 = S.seg_end[1], se2 = S.seg_end[2];
  const int te0 = S.tile_end[0], te1 = S.tile_end[1];
  const int debug = S.debug, sqpx = S.qpx, sqpy = S.qpy, L2 = S.flimit2;
  asm volatile("" ::"s"(info_p), "s"(coeffs_p), "s"(slot0_p), "s"(self), "s"(prev), "s"(gold), "s"(coded_map),
               "s"(se0), "s"(se1), "s"(se2), "s"(te0), "s"(te1), "s"(debug), "s"(sqpx), "s"(sqpy), "s"(L2));
  if (wg >= se2) return;
  const int pli = (wg >= se0 ? 1 : 0) + (wg >= se1 ? 1 : 0);
  const PlaneK G = S.pl[pli];
  const int fy0 = S.lf_y0[pli], fy1 = S.lf_y1[pli];
  asm volatile("" ::"s"(G.nh), "s"(G.nv), "s"(G.stride), "s"(G.off), "s"(G.tiles_x), "s"(G.fro), "s"(G.nseg),
               "s"(G.seglen), "s"(fy0), "s"(fy1));
  const int rel = wg - (pli == 0 ? 0 : (pli == 1 ? se0 : se1));
  const int sby = rel / G.nseg;                 // tile row
  const int seg = rel - sby * G.nseg;
  const int tx0 = seg * G.seglen;               // first tile of the segment
  const int ntx = min(G.seglen, G.tiles_x - tx0);
  if (wave >= ntx) return;                      // (whole waves only: the barriers below count live waves)
  const int tx = tx0 + wave;
  const int unit = (pli == 0 ? 0 : (pli == 1 ? te0 : te1)) + sby * G.tiles_x + tx;   // tile number, as in k_recon

  const uint32_t slot0 = slot0_p[unit];
  const uint2 info = info_p[(size_t)unit * THIP_TILE_FRAGS + lane];
  asm volatile("" ::"s"(slot0), "v"(info.x));

  const int h = lane & 15;
  const int lx = (lane >> 4) * 4 + hilb_col(h), ly = hilb_row(h);   // fragment inside the tile
  const int bx = tx * 16 + lx, by = sby * 4 + ly;
  const bool valid = bx < G.nh && by < G.nv;

  ReconLane L;
  L.flags = valid ? info.x : 0u;
  L.dcp = (info.y & 0xFFFFu) * 0x00010001u;
  L.coded = (L.flags & THIP_INFO_CODED) != 0;
  L.dc_only = L.coded && (L.flags & THIP_INFO_DC_ONLY) != 0;
  L.has_coeff = L.coded && !L.dc_only;
  L.x0 = bx * 8;
  L.y0 = by * 8;
  ReconPlane R;
  R.self = self + G.off;
  R.prev = prev + G.off;
  R.gold = gold + G.off;
  R.coded_map = coded_map + G.fro;
  R.nh = G.nh;
  R.nv = G.nv;
  R.stride = G.stride;
  R.qpx = pli != 0 && sqpx;
  R.qpy = pli != 0 && sqpy;
  R.debug = debug;
  R.tr = nullptr;

  uint8_t *const strip = reinterpret_cast<uint8_t *>(s_dyn);          // chunk w = tile w of the segment
  uint8_t *const chunk = strip + wave * kChunkBytes;
  uint8_t *const img = chunk + (ly * 8) * 128 + lx * 8;               // this lane's block in the tile image
  const uint64_t mask = __ballot(L.has_coeff);
  if (mask != 0 && !(debug & 9)) {
    const uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                      __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
    const uint32_t slot = slot0 + (L.has_coeff ? prefix : 0u);
    const int4 *tp = coeffs_p + ((size_t)(slot >> 6) * 512 + (slot & 63));
    uint4 *lds_wave = reinterpret_cast<uint4 *>(chunk);
#pragma unroll
    for (int q = 0; q < 8; q++)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(tp + q * 64),
                                       (__attribute__((address_space(3))) void *)(lds_wave + q * 64), 16, 0, 0);
    if (valid) recon_tail<true, true>(R, L, lds_wave + lane, img);
  } else {
    if (valid) recon_tail<false, true>(R, L, nullptr, img);
  }
  chunk[4096 + ly * 16 + lx] = (valid && L.coded) ? 1 : 0;           // coded flags of the tile, for the cells
  __syncthreads();

  // ---- loop filter on the cells inside the strip ---------------------------------------------------
  // columns: every corner from the segment's first to its last, except a segment edge that is
  // not a plane edge; rows: m = 1..3 of this tile row, below the plane's last corner row
  if (L2 != 0) {
    const int k0 = tx0 * 16, k1 = min(k0 + ntx * 16, G.nh);
    const int ka = k0 + (k0 > 0 ? 1 : 0), kb = k1 - (k1 < G.nh ? 1 : 0);
    const int ncols = kb - ka + 1;
    const int t = (int)threadIdx.x;
    const int mr = 1 + (t >= ncols ? 1 : 0) + (t >= 2 * ncols ? 1 : 0);
    const int k = ka + t - (mr - 1) * ncols, m = sby * 4 + mr;
    if (t < 3 * ncols && m < G.nv) {
      const int kx = (k - k0) * 8, my = mr * 8;                        // corner in strip pixels
      // flags of a=(k-1,m-1) b=(k,m-1) c=(k-1,m) d=(k,m); fragments outside the plane read as uncoded
      auto flag = [&](int fk, int fm) -> bool {
        const int rk = fk - k0;
        return fk >= 0 && fk < G.nh && strip[(rk >> 4) * kChunkBytes + 4096 + (fm - sby * 4) * 16 + (rk & 15)] != 0;
      };
      const bool a = flag(k - 1, m - 1), b = flag(k, m - 1), c = flag(k - 1, m), d = flag(k, m);
      const uint32_t ops = lf_cell_ops(k, m, G.nh, G.nv, a, b, c, d, fy0, fy1);
      if (ops) {
        const bool lo_ok = k >= 1, hi_ok = k <= G.nh - 1;
        int P[64];
#pragma unroll
        for (int r = 0; r < 8; r++)
          unpack_row(P + r * 8, lo_ok ? lds_px(strip, kx - 4, my - 4 + r) : 0u, hi_ok ? lds_px(strip, kx, my - 4 + r) : 0u);
        lf_cell_apply(P, ops, L2);
#pragma unroll
        for (int r = 0; r < 8; r++) {
          if (lo_ok) lds_px_store(strip, kx - 4, my - 4 + r, pack4(P[r * 8 + 0], P[r * 8 + 1], P[r * 8 + 2], P[r * 8 + 3]));
          if (hi_ok) lds_px_store(strip, kx, my - 4 + r, pack4(P[r * 8 + 4], P[r * 8 + 5], P[r * 8 + 6], P[r * 8 + 7]));
        }
      }
    }
    __syncthreads();
  }

  // ---- the strip goes to the frame, once --------------------------------------------------------------
  if (valid && !(debug & 4)) {
    uint8_t *dst = R.self + (ptrdiff_t)L.y0 * R.stride + L.x0;
#pragma unroll
    for (int r = 0; r < 8; r++) store_row8(dst + (ptrdiff_t)r * R.stride, *reinterpret_cast<const uint2 *>(img + r * 128));
  }
}

// The cells k_recon_lf leaves: per plane first the seam ROWS (m = 0, 4, 8, ... and m = nv, every
// column), then the seam COLUMNS (k on a segment boundary, the remaining rows).
__global__ __launch_bounds__(256) void k_lf_seam(const BatchK B) {
  const StreamK &S = B.s[blockIdx.y];
  const int lane = (int)threadIdx.x & 63;
  const int wbase = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 256u + (threadIdx.x & ~63u)));
  uint8_t *self = S.self;
  const uint8_t *cmap = S.coded_map;
  const int ce0 = S.seam_end[0], ce1 = S.seam_end[1], ce2 = S.seam_end[2], L2 = S.flimit2;
  asm volatile("" ::"s"(self), "s"(cmap), "s"(ce0), "s"(ce1), "s"(ce2), "s"(L2));
  if (wbase >= ce2 || L2 == 0) return;
  const int pli = (wbase >= ce0 ? 1 : 0) + (wbase >= ce1 ? 1 : 0);
  const PlaneK G = S.pl[pli];
  const int fy0 = S.lf_y0[pli], fy1 = S.lf_y1[pli];
  asm volatile("" ::"s"(G.nh), "s"(G.nv), "s"(G.stride), "s"(G.off), "s"(G.fro), "s"(G.rcp_cx), "s"(G.nseg),
               "s"(G.seglen), "s"(G.seam_rows), "s"(fy0), "s"(fy1));
  const int rel = wbase - (pli == 0 ? 0 : (pli == 1 ? ce0 : ce1)) + lane;
  const int nh = G.nh, nv = G.nv;
  const int nrowcells = G.seam_rows * (nh + 1);
  int k, m;
  if (rel < nrowcells) {
    uint32_t mu, ku;
    divmod_u24((uint32_t)rel, (uint32_t)(nh + 1), G.rcp_cx, mu, ku);
    k = (int)ku;
    m = min((int)mu * 4, nv);            // the last seam row is m = nv whether or not nv is a multiple of 4
  } else {
    // seam columns: (nseg-1) columns x (nv + 1 - seam_rows) rows; row index -> m skips the seam rows
    const int r2 = rel - nrowcells;
    const int ncol = G.nseg - 1, nrow = nv + 1 - G.seam_rows;
    if (ncol <= 0 || r2 >= ncol * nrow) return;
    const int ci = r2 / nrow, ri = r2 - ci * nrow;   // (small numbers: plain division)
    k = (ci + 1) * G.seglen * 16;
    m = ri + ri / 3 + 1;                 // rows 1,2,3, 5,6,7, 9,...: skip every multiple of 4
    if (k > nh - 1 || m >= nv) return;   // (a column beyond the plane cannot happen; m < nv by construction)
  }
  CellPix C;
  lf_cell_load(C, self + G.off, G.stride, nh, nv, k, m);
  bool a, b, c, d;
  lf_cell_flags(cmap + G.fro, nh, nv, k, m, a, b, c, d);
  const uint32_t t = lf_cell_ops(k, m, nh, nv, a, b, c, d, fy0, fy1);
  lf_cell_pin(C);
  lf_cell_finish(C, self + G.off, G.stride, nh, nv, k, m, t, L2);
}

// ---------------------------------------------------------------------------------------
// k_loopfilter (K3): one filter cell per lane over the whole frame
// ---------------------------------------------------------------------------------------
// One wave never straddles two planes (the cumulative cell counts in StreamK::cell_end are
// padded to whole waves), so the plane lookup is scalar, like k_recon's.
__global__ __launch_bounds__(256) void k_loopfilter(const BatchK B) {
  const StreamK &S = B.s[blockIdx.y];
  const int lane = (int)threadIdx.x & 63;
  const int wbase = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 256u + (threadIdx.x & ~63u)));
  uint8_t *self = S.self;
  const uint8_t *cmap = S.coded_map;
  const int ce0 = S.cell_end[0], ce1 = S.cell_end[1], ce2 = S.cell_end[2], L2 = S.flimit2;
  asm volatile("" ::"s"(self), "s"(cmap), "s"(ce0), "s"(ce1), "s"(ce2), "s"(L2));
  if (wbase >= ce2 || L2 == 0) return;
  const int pli = (wbase >= ce0 ? 1 : 0) + (wbase >= ce1 ? 1 : 0);
  const PlaneK G = S.pl[pli];
  const int fy0 = S.lf_y0[pli], fy1 = S.lf_y1[pli];
  asm volatile("" ::"s"(G.nh), "s"(G.nv), "s"(G.stride), "s"(G.off), "s"(G.fro), "s"(G.rcp_cx), "s"(fy0), "s"(fy1));
  const int rel = wbase - (pli == 0 ? 0 : (pli == 1 ? ce0 : ce1)) + lane;
  const int nh = G.nh, nv = G.nv;
  if (rel >= (nh + 1) * (nv + 1)) return;
  uint32_t mu, ku;
  divmod_u24((uint32_t)rel, (uint32_t)(nh + 1), G.rcp_cx, mu, ku);
  const int k = (int)ku, m = (int)mu;
  CellPix C;
  lf_cell_load(C, self + G.off, G.stride, nh, nv, k, m);
  bool a, b, c, d;
  lf_cell_flags(cmap + G.fro, nh, nv, k, m, a, b, c, d);
  const uint32_t t = lf_cell_ops(k, m, nh, nv, a, b, c, d, fy0, fy1);
  lf_cell_pin(C);
  lf_cell_finish(C, self + G.off, G.stride, nh, nv, k, m, t, L2);
}

// plane-level entry for the slot parity test (thip_loop_filter_plane)
__global__ __launch_bounds__(256) void k_loopfilter_plane(uint8_t *plane, int stride, int nh, int nv,
                                                         const uint8_t *coded, int L2, int fy0, int fy1,
                                                         float rcp_cx) {
  const int cell = (int)(blockIdx.x * 256u + threadIdx.x);
  if (cell >= (nh + 1) * (nv + 1)) return;
  uint32_t mu, ku;
  divmod_u24((uint32_t)cell, (uint32_t)(nh + 1), rcp_cx, mu, ku);
  const int k = (int)ku, m = (int)mu;
  CellPix C;
  lf_cell_load(C, plane, stride, nh, nv, k, m);
  bool a, b, c, d;
  lf_cell_flags(coded, nh, nv, k, m, a, b, c, d);
  lf_cell_finish(C, plane, stride, nh, nv, k, m, lf_cell_ops(k, m, nh, nv, a, b, c, d, fy0, fy1), L2);
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
struct thip_state {
  int frame_width, frame_height, pixel_fmt, hdec, vdec;
  thip_plane_geom geom[3];
  thip_tile_geom tiles;
  int64_t nfrags;
  size_t frame_bytes;
  uint8_t *frames[3];   // device
  uint8_t *coded_map;   // device, nfrags bytes
  int ref_idx[3];       // THIP_FRAME_* -> buffer index
  int last_decoded;     // buffer index of the most recently completed frame, -1 if none
  int lane;             // library-owned HIP stream this state is bound to, -1 until first use
  int32_t *frag_pos;    // host: raster fragment index -> tile*256+lane
  // host-enqueue staging (allocated on first use)
  int staging_ready;
  uint32_t *h_info, *d_info;
  int16_t *h_coeffs, *d_coeffs;
  uint32_t *h_slot0, *d_slot0;
  hipEvent_t ev_staging;     // recorded behind the kernels that read the staging buffers (enqueue path)
  hipStream_t last_stream;   // stream of the most recent launch for this state (ycbcr_out copies on it)
  uint8_t *h_out;       // pinned image of one frame for thip_state_ycbcr_out (allocated on first use)
  int32_t *enq_last_lane;   // per tile: last lane that received a slot (arrival-order check)
  int enq_ncoded, enq_nuncoded, enq_nslots, enq_frame_type, enq_flimit, enq_active, enq_last_tile;
  int enq_lf_y0[3], enq_lf_y1[3], enq_lf_any;
  int lf_y0[3], lf_y1[3], lf_rows_custom;
};

namespace {
std::mutex g_mu;
// Library-owned HIP streams ("lanes").  Every thip_state is bound to one lane for life, so
// the frames of a stream stay ordered; different lanes let one group's loop filter overlap
// another group's reconstruction (dependent kernels of one group cannot overlap).
constexpr int kMaxLanes = 4;
hipStream_t g_lanes[kMaxLanes] = {nullptr, nullptr, nullptr, nullptr};
int g_nlanes = 0;
int g_next_lane = 0;
int g_profile = 0;
const int g_debug = getenv("THIP_DEBUG") ? atoi(getenv("THIP_DEBUG")) : 0;
struct EvPair { hipEvent_t a, b; int kernel; };
std::vector<EvPair> g_events;
std::vector<hipEvent_t> g_pool;

std::mutex g_lanes_mu;
int ensure_lanes() {   // callable with or without g_mu held, from any host thread
  std::lock_guard<std::mutex> lk(g_lanes_mu);
  if (g_nlanes) return 0;
  int n = getenv("THIP_LANES") ? atoi(getenv("THIP_LANES")) : 2;
  if (n < 1) n = 1;
  if (n > kMaxLanes) n = kMaxLanes;
  for (int i = 0; i < n; i++) HIP_TRY(hipStreamCreateWithFlags(&g_lanes[i], hipStreamNonBlocking));
  g_nlanes = n;
  return 0;
}

hipEvent_t take_event() {
  if (!g_pool.empty()) {
    hipEvent_t e = g_pool.back();
    g_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

struct ScopedTimer {
  hipStream_t s;
  EvPair p;
  bool on;
  ScopedTimer(hipStream_t s_, int kernel) : s(s_), on(g_profile != 0) {
    if (on) {
      p.a = take_event();
      p.b = take_event();
      p.kernel = kernel;
      on = p.a && p.b;
      if (on) (void)hipEventRecord(p.a, s);
    }
  }
  ~ScopedTimer() {
    if (on) {
      (void)hipEventRecord(p.b, s);
      g_events.push_back(p);
    }
  }
};

// Fills the per-plane kernel geometry and the cumulative tile / cell counts.
void fill_stream_geom(StreamK &K, const thip_state *st) {
  int tiles = 0, cells = 0, segs = 0, seams = 0;
  for (int pli = 0; pli < 3; pli++) {
    const thip_plane_geom &g = st->geom[pli];
    PlaneK &k = K.pl[pli];
    k.nh = g.nhfrags;
    k.nv = g.nvfrags;
    k.stride = g.stride;
    k.off = g.plane_off;
    k.tiles_x = st->tiles.tiles_x[pli];
    k.tile_off = st->tiles.tile_off[pli];
    k.fro = g.froffset;
    k.rcp_cx = 1.0f / (float)(g.nhfrags + 1);
    tiles += st->tiles.tiles_x[pli] * st->tiles.tiles_y[pli];
    K.tile_end[pli] = tiles;
    cells += ((g.nhfrags + 1) * (g.nvfrags + 1) + 63) & ~63;   // whole waves per plane (k_loopfilter)
    K.cell_end[pli] = cells;
    // fused path: segments of at most kSegMax tiles, as equal as possible
    k.nseg = (k.tiles_x + kSegMax - 1) / kSegMax;
    k.seglen = (k.tiles_x + k.nseg - 1) / k.nseg;
    segs += st->tiles.tiles_y[pli] * k.nseg;
    K.seg_end[pli] = segs;
    k.seam_rows = g.nvfrags / 4 + 1 + (g.nvfrags % 4 ? 1 : 0);
    seams += (k.seam_rows * (g.nhfrags + 1) + (k.nseg - 1) * (g.nvfrags + 1 - k.seam_rows) + 63) & ~63;
    K.seam_end[pli] = seams;
  }
}
}  // namespace

extern "C" {

const char *thip_version_string(void) { return "theora_hip 0.2 (gfx950; libtheora 1.2.0 fragment path)"; }

int thip_state_create(thip_state **out, int frame_width, int frame_height, int pixel_fmt) {
  if (!out) return THIP_EFAULT;
  *out = nullptr;
  // state.c:712-727: coded size must be a positive multiple of 16, format not reserved
  if (frame_width <= 0 || frame_height <= 0 || (frame_width & 15) || (frame_height & 15) ||
      frame_width >= 0x100000 || frame_height >= 0x100000 || pixel_fmt < 0 || pixel_fmt > 3 ||
      pixel_fmt == 1)
    return THIP_EINVAL;
  thip_state *st = (thip_state *)calloc(1, sizeof(*st));
  if (!st) return THIP_EFAULT;
  st->frame_width = frame_width;
  st->frame_height = frame_height;
  st->pixel_fmt = pixel_fmt;
  st->hdec = !(pixel_fmt & 1);
  st->vdec = !(pixel_fmt & 2);
  int64_t fro = 0, ntiles = 0;
  size_t off = 0;
  for (int pli = 0; pli < 3; pli++) {
    thip_plane_geom &g = st->geom[pli];
    const int yh = frame_width >> 3, yv = frame_height >> 3;
    g.nhfrags = pli ? (yh + st->hdec) >> st->hdec : yh;   // state.c:443-449
    g.nvfrags = pli ? (yv + st->vdec) >> st->vdec : yv;
    const int64_t nf = (int64_t)g.nhfrags * g.nvfrags;
    g.froffset = (int32_t)fro;
    g.nfrags = (int32_t)nf;
    fro += nf;
    g.width = g.nhfrags * 8;
    g.height = g.nvfrags * 8;
    g.stride = g.width;
    g.plane_off = (int32_t)off;
    off += ((size_t)g.stride * g.height + 255) & ~(size_t)255;
    st->tiles.tiles_x[pli] = (g.nhfrags + 15) / 16;
    st->tiles.tiles_y[pli] = (g.nvfrags + 3) / 4;
    st->tiles.tile_off[pli] = (int32_t)ntiles;
    ntiles += (int64_t)st->tiles.tiles_x[pli] * st->tiles.tiles_y[pli];
    // 32-bit positions and byte offsets on the device (the reference has the same kind of
    // overflow guard, state.c:476-487, 583-590)
    if (fro >= (1 << 28) || ntiles * THIP_TILE_FRAGS >= (1ll << 30) || off >= 0x7FFFFFFFull) {
      free(st);
      return THIP_EIMPL;
    }
  }
  st->tiles.ntiles = (int32_t)ntiles;
  st->nfrags = fro;
  st->frame_bytes = off;
  st->frag_pos = (int32_t *)malloc(sizeof(int32_t) * (size_t)st->nfrags);
  if (!st->frag_pos) {
    free(st);
    return THIP_EFAULT;
  }
  for (int pli = 0; pli < 3; pli++) {
    const thip_plane_geom &g = st->geom[pli];
    for (int by = 0; by < g.nvfrags; by++)
      for (int bx = 0; bx < g.nhfrags; bx++)
        st->frag_pos[g.froffset + by * g.nhfrags + bx] =
            (st->tiles.tile_off[pli] + (by >> 2) * st->tiles.tiles_x[pli] + (bx >> 4)) * THIP_TILE_FRAGS +
            ((bx >> 2) & 3) * 16 + hilb_inv(by & 3, bx & 3);
  }
  hipError_t err = hipSuccess;
  // +256: aligned 12-byte predictor windows may read 3 bytes past a row end
  for (int b = 0; b < 3 && err == hipSuccess; b++) err = hipMalloc((void **)&st->frames[b], st->frame_bytes + 256);
  if (err == hipSuccess) err = hipMalloc((void **)&st->coded_map, (size_t)st->nfrags);
  if (err == hipSuccess) err = hipMemset(st->coded_map, 0, (size_t)st->nfrags);
  if (err != hipSuccess) {
    fprintf(stderr, "theora_hip: thip_state_create: device allocation failed: %s\n", hipGetErrorString(err));
    thip_state_free(st);
    return THIP_EFAULT;
  }
  st->ref_idx[0] = st->ref_idx[1] = st->ref_idx[2] = -1;   // state.c:658-663
  st->last_decoded = -1;
  st->lane = -1;
  *out = st;
  return THIP_OK;
}

void thip_state_free(thip_state *st) {
  if (!st) return;
  (void)hipDeviceSynchronize();
  for (int b = 0; b < 3; b++)
    if (st->frames[b]) (void)hipFree(st->frames[b]);
  if (st->coded_map) (void)hipFree(st->coded_map);
  if (st->h_info) (void)hipHostFree(st->h_info);
  if (st->h_coeffs) (void)hipHostFree(st->h_coeffs);
  if (st->h_slot0) (void)hipHostFree(st->h_slot0);
  if (st->h_out) (void)hipHostFree(st->h_out);
  if (st->ev_staging) (void)hipEventDestroy(st->ev_staging);
  if (st->d_info) (void)hipFree(st->d_info);
  if (st->d_coeffs) (void)hipFree(st->d_coeffs);
  if (st->d_slot0) (void)hipFree(st->d_slot0);
  free(st->enq_last_lane);
  free(st->frag_pos);
  free(st);
}

int thip_state_get_geom(const thip_state *st, thip_plane_geom geom[3], int64_t *nfrags,
                        int64_t *frame_bytes) {
  if (!st) return THIP_EFAULT;
  if (geom) memcpy(geom, st->geom, sizeof(st->geom));
  if (nfrags) *nfrags = st->nfrags;
  if (frame_bytes) *frame_bytes = (int64_t)st->frame_bytes;
  return THIP_OK;
}

int thip_state_get_tiles(const thip_state *st, thip_tile_geom *out) {
  if (!st || !out) return THIP_EFAULT;
  *out = st->tiles;
  return THIP_OK;
}

int64_t thip_state_frag_pos(const thip_state *st, int64_t fragi) {
  if (!st) return THIP_EFAULT;
  if (fragi < 0 || fragi >= st->nfrags) return THIP_EINVAL;
  return st->frag_pos[fragi];
}

int thip_state_ref_idx(const thip_state *st, int which) {
  if (!st || which < 0 || which > 2) return THIP_EINVAL;
  return st->ref_idx[which];
}

int thip_state_set_ref_idx(thip_state *st, int gold, int prev, int self) {
  if (!st) return THIP_EFAULT;
  if (gold < -1 || gold > 2 || prev < -1 || prev > 2 || self < -1 || self > 2) return THIP_EINVAL;
  st->ref_idx[THIP_FRAME_GOLD] = gold;
  st->ref_idx[THIP_FRAME_PREV] = prev;
  st->ref_idx[THIP_FRAME_SELF] = self;
  if (self >= 0) st->last_decoded = self;
  return THIP_OK;
}

uint8_t *thip_state_frame_ptr(const thip_state *st, int bufi) {
  if (!st || bufi < 0 || bufi > 2) return nullptr;
  return st->frames[bufi];
}

int thip_state_read_plane(thip_state *st, int bufi, int pli, uint8_t *host_out) {
  if (!st || !host_out) return THIP_EFAULT;
  if (bufi < 0 || bufi > 2 || pli < 0 || pli > 2) return THIP_EINVAL;
  const thip_plane_geom &g = st->geom[pli];
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy2D(host_out, g.width, st->frames[bufi] + g.plane_off, g.stride, g.width, g.height,
                      hipMemcpyDeviceToHost));
  return THIP_OK;
}

int thip_state_write_plane(thip_state *st, int bufi, int pli, const uint8_t *host_in) {
  if (!st || !host_in) return THIP_EFAULT;
  if (bufi < 0 || bufi > 2 || pli < 0 || pli > 2) return THIP_EINVAL;
  const thip_plane_geom &g = st->geom[pli];
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy2D(st->frames[bufi] + g.plane_off, g.stride, host_in, g.width, g.width, g.height,
                      hipMemcpyHostToDevice));
  return THIP_OK;
}

int thip_state_ycbcr_out(thip_state *st, uint8_t *const dst[3], const int32_t dst_stride[3]) {
  if (!st || !dst || !dst_stride) return THIP_EFAULT;
  if (st->last_decoded < 0) return THIP_EINVAL;
  for (int pli = 0; pli < 3; pli++)
    if (!dst[pli] || dst_stride[pli] < st->geom[pli].width) return THIP_EINVAL;
  // one DMA of the whole frame (planes are contiguous, pitch == width) into pinned memory, on
  // the stream that produced it, then the top-down flip on the host
  if (!st->h_out) HIP_TRY(hipHostMalloc((void **)&st->h_out, st->frame_bytes, hipHostMallocDefault));
  hipStream_t s = st->last_stream;   // null (legacy stream) if the frame came from write_plane only
  HIP_TRY(hipMemcpyAsync(st->h_out, st->frames[st->last_decoded], st->frame_bytes, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  for (int pli = 0; pli < 3; pli++) {
    const thip_plane_geom &g = st->geom[pli];
    // the device keeps row 0 at the bottom of the picture; hand the frame back top-down
    // (decode.c:2988-2992 flips pointers instead)
    const uint8_t *src = st->h_out + g.plane_off;
    uint8_t *last = dst[pli] + (size_t)(g.height - 1) * dst_stride[pli];
    for (int y = 0; y < g.height; y++) memcpy(last - (size_t)y * dst_stride[pli], src + (size_t)y * g.stride, g.width);
  }
  return THIP_OK;
}

int thip_synchronize(void) {
  for (int i = 0; i < g_nlanes; i++) HIP_TRY(hipStreamSynchronize(g_lanes[i]));
  return THIP_OK;
}

#ifdef THIP_TRACE
int thip_debug_trace_buffer(void *dev_buf) {
  HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_trace_buf), &dev_buf, sizeof(dev_buf)));
  return THIP_OK;
}
#endif

int thip_profile_enable(int on) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_profile = on;
  return THIP_OK;
}

int thip_profile_reset(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  HIP_TRY(hipDeviceSynchronize());
  for (auto &p : g_events) {
    g_pool.push_back(p.a);
    g_pool.push_back(p.b);
  }
  g_events.clear();
  return THIP_OK;
}

int thip_profile_read(int64_t launches[THIP_NKERNELS], double ms[THIP_NKERNELS]) {
  std::lock_guard<std::mutex> lk(g_mu);
  HIP_TRY(hipDeviceSynchronize());
  for (int k = 0; k < THIP_NKERNELS; k++) {
    launches[k] = 0;
    ms[k] = 0.0;
  }
  for (auto &p : g_events) {
    float t = 0.f;
    HIP_TRY(hipEventElapsedTime(&t, p.a, p.b));
    launches[p.kernel]++;
    ms[p.kernel] += (double)t;
  }
  return THIP_OK;
}

// Launch one chunk of <= THIP_MAX_BATCH streams.
static int launch_chunk(thip_state *const *states, const thip_frame_desc *descs, int n, hipStream_t s,
                        int32_t *results) {
  BatchK B;
  memset(&B, 0, sizeof(B));
  int max_wg = 0, max_seam_wg = 0, any_lf = 0, nlive = 0;
  int max_fwg = 0, max_seglen = 1, max_fseam_wg = 0;   // fused path
  int live_state[THIP_MAX_BATCH];
  for (int i = 0; i < n; i++) {
    thip_state *st = states[i];
    const thip_frame_desc &d = descs[i];
    if (!st) return THIP_EFAULT;
    if (d.ncoded < 0 || d.nslots < 0 || d.nslots > d.ncoded || d.ncoded > st->nfrags) return THIP_EINVAL;
    if (d.ncoded && (!d.frag_info || !d.tile_slot0 || (d.nslots && !d.coeffs))) return THIP_EFAULT;
    if (d.flimit < 0 || d.flimit > 127) return THIP_EINVAL;
    if (d.frame_type != THIP_INTRA_FRAME && d.frame_type != THIP_INTER_FRAME) return THIP_EINVAL;
    if (d.frame_type == THIP_INTRA_FRAME && d.ncoded != st->nfrags) return THIP_EINVAL;
    // decode.c:2757-2762: an inter frame without references decodes against mid-grey
    if (d.frame_type != THIP_INTRA_FRAME &&
        (st->ref_idx[THIP_FRAME_GOLD] < 0 || st->ref_idx[THIP_FRAME_PREV] < 0)) {
      HIP_TRY(hipMemsetAsync(st->frames[0], 0x80, st->frame_bytes, s));
      st->ref_idx[0] = st->ref_idx[1] = st->ref_idx[2] = 0;
      st->last_decoded = 0;
    }
    if (d.ncoded == 0) {  // decode.c:2764-2772
      if (results) results[i] = THIP_DUPFRAME;
      continue;
    }
    if (results) results[i] = THIP_OK;
    int bufi = 0;  // decode.c:2790-2794
    while (bufi == st->ref_idx[THIP_FRAME_GOLD] || bufi == st->ref_idx[THIP_FRAME_PREV]) bufi++;
    st->ref_idx[THIP_FRAME_SELF] = bufi;
    StreamK &K = B.s[nlive];
    K.info = reinterpret_cast<const uint2 *>(d.frag_info);
    K.coeffs = reinterpret_cast<const int4 *>(d.coeffs);
    K.tile_slot0 = d.tile_slot0;
    K.self = st->frames[bufi];
    K.prev = st->ref_idx[THIP_FRAME_PREV] >= 0 ? st->frames[st->ref_idx[THIP_FRAME_PREV]] : st->frames[bufi];
    K.gold = st->ref_idx[THIP_FRAME_GOLD] >= 0 ? st->frames[st->ref_idx[THIP_FRAME_GOLD]] : st->frames[bufi];
    K.coded_map = st->coded_map;
    K.flimit2 = 2 * d.flimit;
    K.debug = g_debug;
    K.qpx = st->hdec;
    K.qpy = st->vdec;
    fill_stream_geom(K, st);
    for (int pli = 0; pli < 3; pli++) {
      K.lf_y0[pli] = st->lf_rows_custom ? st->lf_y0[pli] : 0;
      K.lf_y1[pli] = st->lf_rows_custom ? st->lf_y1[pli] : st->geom[pli].nvfrags;
    }
    const int wgs = ((K.tile_end[2] + THIP_RECON_WG_WAVES - 1) / THIP_RECON_WG_WAVES + 7) & ~7;   // 8 XCD bands
    if (wgs > max_wg) max_wg = wgs;
    if (d.flimit) {
      any_lf = 1;
      const int swg = (K.cell_end[2] + 255) / 256;
      if (swg > max_seam_wg) max_seam_wg = swg;
      const int fswg = (K.seam_end[2] + 255) / 256;
      if (fswg > max_fseam_wg) max_fseam_wg = fswg;
    }
    const int fwg = (K.seg_end[2] + 7) & ~7;   // 8 XCD bands
    if (fwg > max_fwg) max_fwg = fwg;
    for (int pli = 0; pli < 3; pli++)
      if (K.pl[pli].seglen > max_seglen) max_seglen = K.pl[pli].seglen;
    live_state[nlive++] = i;
  }
  if (!nlive) return THIP_OK;
  // THIP_FUSE=1: reconstruct and filter a strip in LDS and write the frame once (k_recon_lf +
  // k_lf_seam) instead of k_recon + k_loopfilter over the whole frame.  Bit-exact, 70 MB less
  // traffic per 4K step, but measured SLOWER (58+17 us vs 44+23 us per 4-frame launch: the
  // workgroup-wide barriers and the filter's arithmetic cost more than the bytes saved), so it
  // is off by default; DESIGN.md section 5.
  static const int fuse = getenv("THIP_FUSE") ? atoi(getenv("THIP_FUSE")) : 0;
  if (fuse && any_lf) {
    static bool attr_set = false;
    if (!attr_set) {
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_recon_lf), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  kSegMax * kChunkBytes));
      attr_set = true;
    }
    {
      ScopedTimer t(s, THIP_KERNEL_RECON);
      hipLaunchKernelGGL(k_recon_lf, dim3(max_fwg, nlive), dim3(64 * max_seglen), (size_t)max_seglen * kChunkBytes, s, B);
    }
    {
      ScopedTimer t(s, THIP_KERNEL_LOOPFILTER);
      hipLaunchKernelGGL(k_lf_seam, dim3(max_fseam_wg, nlive), dim3(256), 0, s, B);
    }
  } else {
    {
      ScopedTimer t(s, THIP_KERNEL_RECON);
      hipLaunchKernelGGL(k_recon, dim3(max_wg, nlive), dim3(64 * THIP_RECON_WG_WAVES), 0, s, B);
    }
    if (any_lf) {
      ScopedTimer t(s, THIP_KERNEL_LOOPFILTER);
      hipLaunchKernelGGL(k_loopfilter, dim3(max_seam_wg, nlive), dim3(256), 0, s, B);
    }
  }
  HIP_TRY(hipGetLastError());
  // decode.c:2947-2962
  for (int j = 0; j < nlive; j++) {
    thip_state *st = states[live_state[j]];
    const int self = st->ref_idx[THIP_FRAME_SELF];
    if (descs[live_state[j]].frame_type == THIP_INTRA_FRAME) st->ref_idx[THIP_FRAME_GOLD] = self;
    st->ref_idx[THIP_FRAME_PREV] = self;
    st->last_decoded = self;
    st->last_stream = s;
  }
  return THIP_OK;
}

int thip_decode_frames(thip_state *const *states, const thip_frame_desc *descs, int nstreams,
                       void *stream, int32_t *results) {
  if (!states || !descs) return THIP_EFAULT;
  if (nstreams < 0) return THIP_EINVAL;
  std::lock_guard<std::mutex> lk(g_mu);
  if (stream) {   // caller-owned stream: everything in submission order on it
    hipStream_t s = (hipStream_t)stream;
    for (int i = 0; i < nstreams; i += THIP_MAX_BATCH) {
      const int n = nstreams - i < THIP_MAX_BATCH ? nstreams - i : THIP_MAX_BATCH;
      int rc = launch_chunk(states + i, descs + i, n, s, results ? results + i : nullptr);
      if (rc < 0) return rc;
    }
    return THIP_OK;
  }
  int rc = ensure_lanes();
  if (rc) return rc;
  for (int i = 0; i < nstreams; i++) {
    if (!states[i]) return THIP_EFAULT;
    if (states[i]->lane < 0) states[i]->lane = g_next_lane++ % g_nlanes;
  }
  // group by lane (order inside a lane preserved), launch chunk by chunk
  for (int lane = 0; lane < g_nlanes; lane++) {
    thip_state *ls[THIP_MAX_BATCH];
    thip_frame_desc ld[THIP_MAX_BATCH];
    int32_t lr[THIP_MAX_BATCH];
    int li[THIP_MAX_BATCH];
    int n = 0;
    for (int i = 0; i <= nstreams; i++) {
      if (i < nstreams && states[i]->lane == lane) {
        ls[n] = states[i];
        ld[n] = descs[i];
        li[n] = i;
        n++;
      }
      if (n == THIP_MAX_BATCH || (i == nstreams && n > 0)) {
        rc = launch_chunk(ls, ld, n, g_lanes[lane], lr);
        if (rc < 0) return rc;
        if (results)
          for (int k = 0; k < n; k++) results[li[k]] = lr[k];
        n = 0;
      }
    }
  }
  return THIP_OK;
}

int thip_loop_filter_plane(uint8_t *plane, int ystride, int nhfrags, int nvfrags, const uint8_t *coded,
                           int flimit, int fragy0, int fragy_end) {
  if (!plane || !coded) return THIP_EFAULT;
  if (nhfrags <= 0 || nvfrags <= 0 || flimit < 0 || flimit > 127 || ystride < nhfrags * 8 ||
      (int64_t)(nhfrags + 1) * (nvfrags + 1) >= (1 << 24))
    return THIP_EINVAL;
  if (flimit == 0) return THIP_OK;
  const int64_t cells = (int64_t)(nhfrags + 1) * (nvfrags + 1);
  hipLaunchKernelGGL(k_loopfilter_plane, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, 0, plane, ystride,
                     nhfrags, nvfrags, coded, 2 * flimit, fragy0, fragy_end, 1.0f / (float)(nhfrags + 1));
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  return THIP_OK;
}

// ---------------------------------------------------------------------------------------
// host-enqueue form (the vtable slots)
// ---------------------------------------------------------------------------------------
static int ensure_staging(thip_state *st) {
  if (st->staging_ready) return THIP_OK;
  const size_t npos = (size_t)st->tiles.ntiles * THIP_TILE_FRAGS;
  const size_t ngroups = ((size_t)st->nfrags + THIP_SLOT_GROUP - 1) / THIP_SLOT_GROUP;
  HIP_TRY(hipHostMalloc((void **)&st->h_info, npos * 8, hipHostMallocDefault));
  HIP_TRY(hipHostMalloc((void **)&st->h_coeffs, ngroups * THIP_SLOT_GROUP_BYTES, hipHostMallocDefault));
  HIP_TRY(hipHostMalloc((void **)&st->h_slot0, (size_t)st->tiles.ntiles * 4, hipHostMallocDefault));
  HIP_TRY(hipMalloc((void **)&st->d_info, npos * 8));
  HIP_TRY(hipMalloc((void **)&st->d_coeffs, ngroups * THIP_SLOT_GROUP_BYTES));
  HIP_TRY(hipMalloc((void **)&st->d_slot0, (size_t)st->tiles.ntiles * 4));
  st->enq_last_lane = (int32_t *)malloc(sizeof(int32_t) * (size_t)st->tiles.ntiles);
  if (!st->enq_last_lane) return THIP_EFAULT;
  st->staging_ready = 1;
  return THIP_OK;
}

int thip_frame_begin(thip_state *st, int frame_type) {
  if (!st) return THIP_EFAULT;
  if (frame_type != THIP_INTRA_FRAME && frame_type != THIP_INTER_FRAME) return THIP_EINVAL;
  int rc = ensure_staging(st);
  if (rc) return rc;
  // the previous frame's kernels must have read the staging buffers before they are reused;
  // only this stream's own work is waited for, so contexts on other host threads keep going
  if (st->ev_staging) HIP_TRY(hipEventSynchronize(st->ev_staging));
  memset(st->h_info, 0, (size_t)st->tiles.ntiles * THIP_TILE_FRAGS * 8);   // everything uncoded
  memset(st->h_slot0, 0, (size_t)st->tiles.ntiles * 4);
  for (int t = 0; t < st->tiles.ntiles; t++) st->enq_last_lane[t] = -1;
  st->enq_ncoded = st->enq_nuncoded = st->enq_nslots = 0;
  st->enq_last_tile = -1;
  st->enq_frame_type = frame_type;
  st->enq_flimit = 0;
  st->enq_lf_any = 0;
  for (int p = 0; p < 3; p++) {
    st->enq_lf_y0[p] = 0x7FFFFFFF;
    st->enq_lf_y1[p] = -1;
  }
  st->enq_active = 1;
  return THIP_OK;
}

int thip_state_frag_recon(thip_state *st, ptrdiff_t fragi, int pli, int16_t dct_coeffs[128],
                          int last_zzi, uint16_t dc_quant, int refi, int16_t mv) {
  if (!st || !dct_coeffs) return THIP_EFAULT;
  if (!st->enq_active || fragi < 0 || fragi >= st->nfrags || pli < 0 || pli > 2 || refi < 0 || refi > 2 ||
      last_zzi < 0 || last_zzi > 64 || (int64_t)st->enq_ncoded + st->enq_nuncoded >= st->nfrags)
    return THIP_EINVAL;
  const int32_t pos = st->frag_pos[fragi];
  if (st->h_info[2 * (size_t)pos] & THIP_INFO_CODED) return THIP_EINVAL;   // fragment enqueued twice
  uint32_t flags = THIP_INFO_CODED | ((uint32_t)refi << THIP_INFO_REFI_SHIFT) |
                   ((uint32_t)last_zzi << THIP_INFO_LAST_ZZI_SHIFT) |
                   ((uint32_t)(uint8_t)(mv & 0xFF) << THIP_INFO_MVX_SHIFT) |
                   ((uint32_t)(uint8_t)((mv >> 8) & 0xFF) << THIP_INFO_MVY_SHIFT);
  uint32_t word1 = 0;
  if (last_zzi < 2) {
    // state.c:967-975: the only rounded dequantisation of the path; no coefficient slot
    flags |= THIP_INFO_DC_ONLY;
    word1 = (uint32_t)(uint16_t)(int16_t)((dct_coeffs[0] * (int32_t)dc_quant + 15) >> 5);
  } else {
    // Slots are handed out in arrival order, which for the reference's caller is coded
    // order == tile/lane order (decode.c:1530-1586); the kernel re-derives a lane's slot
    // from the tile's first slot and a prefix count, so arrival must not jump backwards
    // inside a tile.
    const int tile = pos / THIP_TILE_FRAGS, lane = pos % THIP_TILE_FRAGS;
    if (lane <= st->enq_last_lane[tile]) return THIP_EINVAL;
    if (st->enq_last_lane[tile] < 0) st->h_slot0[tile] = (uint32_t)st->enq_nslots;
    else if (st->enq_last_tile != tile) return THIP_EINVAL;   // a tile's slots must be contiguous
    st->enq_last_lane[tile] = lane;
    st->enq_last_tile = tile;
    const int slot = st->enq_nslots++;
    dct_coeffs[0] = (int16_t)(dct_coeffs[0] * (int)dc_quant);   // state.c:978
    // piece q = 2*j+h of the block: columns c = 4h..4h+3 as pairs { x[2j][c], x[2j+1][c] }
    int16_t *blk = st->h_coeffs + (size_t)(slot >> 6) * (THIP_SLOT_GROUP_BYTES / 2) + (size_t)(slot & 63) * 8;
    for (int j = 0; j < 4; j++)
      for (int hh = 0; hh < 2; hh++) {
        int16_t *g = blk + (size_t)(2 * j + hh) * 512;
        for (int cc = 0; cc < 4; cc++) {
          g[2 * cc] = dct_coeffs[(2 * j) * 8 + 4 * hh + cc];
          g[2 * cc + 1] = dct_coeffs[(2 * j + 1) * 8 + 4 * hh + cc];
        }
      }
  }
  memset(dct_coeffs, 0, 64 * sizeof(int16_t));   // idct.c:245,276,295
  st->h_info[2 * (size_t)pos] = flags;
  st->h_info[2 * (size_t)pos + 1] = word1;
  st->enq_ncoded++;
  return THIP_OK;
}

int thip_frag_copy_list(thip_state *st, const ptrdiff_t *fragis, ptrdiff_t nfragis) {
  if (!st || (!fragis && nfragis)) return THIP_EFAULT;
  if (!st->enq_active || nfragis < 0 || (int64_t)st->enq_ncoded + st->enq_nuncoded + nfragis > st->nfrags)
    return THIP_EINVAL;
  for (ptrdiff_t k = 0; k < nfragis; k++) {
    if (fragis[k] < 0 || fragis[k] >= st->nfrags) return THIP_EINVAL;
    // an uncoded fragment is the default state of its command word; nothing to stage
    if (st->h_info[2 * (size_t)st->frag_pos[fragis[k]]] & THIP_INFO_CODED) return THIP_EINVAL;
  }
  st->enq_nuncoded += (int)nfragis;
  return THIP_OK;
}

void thip_loop_filter_init(signed char bv[256], int flimit) {
  // state.c:1036-1045 tabulates lflim(R,flimit) for R in [-127,128]
  for (int i = 0; i < 256; i++) {
    const int R = i - 127;
    const int a = R < 0 ? -R : R;
    int m = 2 * flimit - a;
    if (m < 0) m = 0;
    if (m > a) m = a;
    bv[i] = (signed char)(R < 0 ? -m : m);
  }
}

int thip_state_loop_filter_frag_rows(thip_state *st, int flimit, int refi, int pli, int fragy0,
                                     int fragy_end) {
  if (!st) return THIP_EFAULT;
  if (!st->enq_active || refi != THIP_FRAME_SELF || pli < 0 || pli > 2 || flimit < 0 || flimit > 127)
    return THIP_EINVAL;
  if (fragy0 < 0) fragy0 = 0;
  if (fragy_end > st->geom[pli].nvfrags) fragy_end = st->geom[pli].nvfrags;
  if (fragy_end <= fragy0) return THIP_OK;
  // successive calls extend the row range exactly as the MCU loop does (decode.c:2879-2884)
  if (st->enq_lf_y1[pli] >= 0 && fragy0 != st->enq_lf_y1[pli]) return THIP_EINVAL;
  if (fragy0 < st->enq_lf_y0[pli]) st->enq_lf_y0[pli] = fragy0;
  st->enq_lf_y1[pli] = fragy_end;
  st->enq_flimit = flimit;
  st->enq_lf_any = 1;
  return THIP_OK;
}

int thip_frame_flush(thip_state *st) {
  if (!st) return THIP_EFAULT;
  if (!st->enq_active) return THIP_EINVAL;
  st->enq_active = 0;
  // every fragment must have been reconstructed or copied exactly once
  if (st->enq_ncoded && (int64_t)st->enq_ncoded + st->enq_nuncoded != st->nfrags) return THIP_EINVAL;
  int rc = ensure_lanes();
  if (rc) return rc;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    if (st->lane < 0) st->lane = g_next_lane++ % g_nlanes;
  }
  hipStream_t s = g_lanes[st->lane];
  const size_t ngroups = ((size_t)st->enq_nslots + THIP_SLOT_GROUP - 1) / THIP_SLOT_GROUP;
  static const int zerocopy = getenv("THIP_ZEROCOPY") ? atoi(getenv("THIP_ZEROCOPY")) : 1;
  if (st->enq_ncoded && !zerocopy) {
    HIP_TRY(hipMemcpyAsync(st->d_info, st->h_info, (size_t)st->tiles.ntiles * THIP_TILE_FRAGS * 8,
                           hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(st->d_slot0, st->h_slot0, (size_t)st->tiles.ntiles * 4, hipMemcpyHostToDevice, s));
    if (ngroups)
      HIP_TRY(hipMemcpyAsync(st->d_coeffs, st->h_coeffs, ngroups * THIP_SLOT_GROUP_BYTES, hipMemcpyHostToDevice, s));
  }
  thip_frame_desc d;
  memset(&d, 0, sizeof(d));
  // zero-copy: the kernels read the pinned staging buffers across PCIe themselves (every
  // byte exactly once, coalesced); hipMemcpyAsync of ~1 MB costs more host time than that
  d.frag_info = zerocopy ? st->h_info : st->d_info;
  d.coeffs = zerocopy ? st->h_coeffs : st->d_coeffs;
  d.tile_slot0 = zerocopy ? st->h_slot0 : st->d_slot0;
  d.nslots = st->enq_nslots;
  d.ncoded = st->enq_ncoded;
  d.frame_type = st->enq_frame_type;
  d.flimit = st->enq_lf_any ? st->enq_flimit : 0;
  st->lf_rows_custom = st->enq_lf_any;
  for (int p = 0; p < 3; p++) {
    st->lf_y0[p] = st->enq_lf_y1[p] < 0 ? 0 : st->enq_lf_y0[p];
    st->lf_y1[p] = st->enq_lf_y1[p] < 0 ? 0 : st->enq_lf_y1[p];
  }
  int32_t res = 0;
  thip_state *sp = st;
  rc = thip_decode_frames(&sp, &d, 1, nullptr, &res);
  st->lf_rows_custom = 0;
  if (rc < 0) return rc;
  if (!st->ev_staging) HIP_TRY(hipEventCreateWithFlags(&st->ev_staging, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(st->ev_staging, s));
  return res;
}

}  // extern "C"
