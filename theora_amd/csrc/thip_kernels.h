// thip_kernels.h -- device side of the frame-scope path: the geometry constants shared with the
// host, the per-stream kernel-argument tables, and the kernels k_recon, k_loopfilter, k_loopfilter_plane,
// k_dc_unpredict, k_expand_tokens (k_recon_lf: thip_fused.h, k_dc_wave: thip_dc.h).  Included by thip_decode.hip only (which holds the host
// side and the C ABI); see the header comment there for the overall picture.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/theora_hip.h"
#include "thip_device.h"

using namespace thip;

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
  int tiles_y;       // tile rows
};

struct StreamK {
  const uint2 *info;
  const int4 *coeffs;
  const uint32_t *tile_slot0;
  uint8_t *self;
  const uint8_t *prev;
  const uint8_t *gold;
  uint8_t *coded_map;     // 1 byte per fragment, raster order: written by k_recon, read by k_loopfilter
  const uint8_t *coded_prev;   // the same map of the previous frame of this stream (two maps alternate)
  const int16_t *dc;      // null, or the un-predicted DC of every fragment (fragment-index order, k_dc_unpredict's
                          // output): used instead of the DC fields of the command words / coefficient slots
  const uint4 *dequant;   // null: the slots hold dequantised int16 coefficients (THIP_COEFFS_DEQUANT16).  Else THIP_COEFFS_LEVELS: the
                          // slots hold quantised LEVELS (64-byte int8 units; two units of int16 per block in the tiles whose
                          // tile_slot0 carries THIP_SLOT_WIDE) and this is the frame's 18 AC dequantisation tables in slot order
                          // (thip_pack_dequant_table): the multiplication of decode.c:1573 happens in the kernel
  int skip_ok;            // the buffer this frame goes to holds the frame before the previous one: an uncoded
                          // block that was not touched in the previous frame either is already in place
  int flimit2;            // 2*flimit
  int qpx, qpy;           // chroma axis decimated (quarter-pel chroma vectors)
  int tile_end[3];        // cumulative tile counts per plane (k_recon: one wave per tile)
  int cell_end[3];        // cumulative filter-cell counts per plane, each plane padded to 64 (k_loopfilter)
  int lf_y0[3], lf_y1[3]; // fragment-row range whose filter operations are applied
  int debug;              // ablation switches for profiling (THIP_DEBUG env), 0 in production
  int lf_sparse;          // k_loopfilter reads the coded flags first and skips waves without a coded block
  // reconstruction + loop filter in one pass (k_recon_lf, thip_fused.h)
  uint8_t *edge;          // kTfRec bytes per tile: the tile's edges for its neighbours
  uint32_t epoch;         // serial number that marks the edge records of this launch
  int band_u0[9];         // first tile of each of the 8 XCD bands (whole tile rows), [8] = number of tiles
  uint32_t *fault;        // pinned host word of the stream's state: set by a kernel whose bounded wait ran out (k_recon_lf's hand-over)
  // k_recon_lf, levels form: the frame has one coefficient unit for every block (nslots == the frame's fragments: the dense class, an
  // intra frame at a high bit rate), so a tile's first unit is known without its first-slot word wherever the tile rows before it
  // are whole -- spec_base[plane] + 4 nhfrags * (tile row) + 64 * (tile in its row), in the plane's whole tile rows: only the LAST tile of a
  // row can be ragged, so every tile's first unit is known and every tile of 64 blocks is a hit (round 6; round 5's patch asked for
  // planes a whole number of tiles wide, which left 1080p -- chroma 120 blocks across -- out) -- and the wave asks
  // for its coefficients when it starts, beside the command words, instead of a round trip later
  int spec_on;
  uint32_t spec_base[3];   // the plane's first unit
  int spec_rows[3];        // its whole tile rows (nvfrags / 4)
  int spec_rowunits[3];    // units a whole tile row holds (4 nhfrags)
  uint32_t spec_last;      // the frame's last unit (a ragged tile's guess reaches past its own units: never past the array)
  int spec_tx[3];          // tiles across (PlaneK::tiles_x, at hand before the plane's record is)
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
// lflim without the detour over |R| and the sign: see pk_lf_delta (six packed operations); 3 * (P4 - P3) + (P2 - P5) is one v_pk_mad_i16.
__device__ __forceinline__ pk16 pk_lf_delta(pk16 p2, pk16 p3, pk16 p4, pk16 p5, int L2) {
  const pk16 three = {(short)3, (short)3};
  const pk16 f = (p4 - p3) * three + (p2 - p5);
  const pk16 R = (f + (short)4) >> 3;
#ifdef THIP_LFLIM7   // (A/B: the seven-operation form of rounds 2 and 3)
  const pk16 l2 = {(short)L2, (short)L2};
  const pk16 z = {0, 0};
  const pk16 pos = __builtin_elementwise_max(__builtin_elementwise_min(R, l2 - R), z);
  const pk16 neg = __builtin_elementwise_min(__builtin_elementwise_max(R, z - l2 - R), z);
  return pos + neg;
#else
  // six: with a = clamp(R, -L, L), lflim(R) = a - clamp(R - a, -L, L) (R inside the limit: a = R, nothing taken off; between L and 2L:
  // a = +-L and the excess comes off; beyond: the excess is clamped to L as well and nothing is left) -- checked for every R and L
  const short L = (short)(L2 >> 1);
  const pk16 l = {L, L}, nl = {(short)-L, (short)-L};
  const pk16 a = __builtin_elementwise_max(__builtin_elementwise_min(R, l), nl);
  const pk16 b = __builtin_elementwise_max(__builtin_elementwise_min(R - a, l), nl);
  return a - b;
#endif
}
// horizontal edge y = 4 of the cell, columns 0..3 (half 0: the lo dwords) or 4..7 (half 1)
__device__ __forceinline__ void lf_horz_pk(CellPix &C, int half, int L2) {
  uint32_t *w = half ? C.hi : C.lo;
  const pk16 d01 = pk_lf_delta(pk_bytes01(w[2]), pk_bytes01(w[3]), pk_bytes01(w[4]), pk_bytes01(w[5]), L2);
  const pk16 d23 = pk_lf_delta(pk_bytes23(w[2]), pk_bytes23(w[3]), pk_bytes23(w[4]), pk_bytes23(w[5]), L2);
  const uint32_t n3 = sat_pk_u8x4(pk_bytes01(w[3]) + d01, pk_bytes23(w[3]) + d23);
  const uint32_t n4 = sat_pk_u8x4(pk_bytes01(w[4]) - d01, pk_bytes23(w[4]) - d23);
  w[3] = n3;
  w[4] = n4;
}
// vertical edge x = 4 of the cell, rows r0..r0+3: columns 2,3 are bytes 2,3 of lo, columns 4,5
// bytes 0,1 of hi; two rows per register
__device__ __forceinline__ void lf_vert_pair(uint32_t &lo0, uint32_t &lo1, uint32_t &hi0, uint32_t &hi1, int L2) {
  const pk16 p2 = as_pk(__builtin_amdgcn_perm(lo1, lo0, 0x0c060c02u));
  const pk16 p3 = as_pk(__builtin_amdgcn_perm(lo1, lo0, 0x0c070c03u));
  const pk16 p4 = as_pk(__builtin_amdgcn_perm(hi1, hi0, 0x0c040c00u));
  const pk16 p5 = as_pk(__builtin_amdgcn_perm(hi1, hi0, 0x0c050c01u));
  const pk16 d = pk_lf_delta(p2, p3, p4, p5, L2);
  const uint32_t n3 = sat_pk_u8(p3 + d), n4 = sat_pk_u8(p4 - d);   // byte 0: first row, byte 1: second row
  lo0 = __builtin_amdgcn_perm(n3, lo0, 0x04020100u);               // byte 3 <- n3.byte0
  lo1 = __builtin_amdgcn_perm(n3, lo1, 0x05020100u);
  hi0 = __builtin_amdgcn_perm(n4, hi0, 0x03020104u);               // byte 0 <- n4.byte0
  hi1 = __builtin_amdgcn_perm(n4, hi1, 0x03020105u);
}
__device__ __forceinline__ void lf_vert_pk(CellPix &C, int r0, int L2) {
#pragma unroll
  for (int r = r0; r < r0 + 4; r += 2) lf_vert_pair(C.lo[r], C.lo[r + 1], C.hi[r], C.hi[r + 1], L2);
}
// The operations of one cell in an order equivalent to T1..T8 with six slots instead of eight.  Only a vertical-edge and a
// horizontal-edge operation share pixels (Vlo / Vhi with Hl / Hr); Vlo and Vhi, Hl and Hr never do.  A cell has at most one
// of T1 / T3 (both are Vlo), of T2 / T5 (Hl), of T4 / T8 (Hr), of T6 / T7 (Vhi), so all that has to be kept is, for each
// of the four V-H pairs, which of the two comes first:  Hl precedes Vlo only as T2 before T3;  Vlo (T1, T3) always
// precedes Hr (T4, T8);  Hl (T2, T5) always precedes Vhi (T6, T7);  Hr precedes Vhi only as T4 (then Vhi is T6).
__device__ __forceinline__ void lf_cell_apply_pk(CellPix &C, uint32_t t, int L2) {
#ifdef THIP_LF_8SLOTS   // (A/B: the eight slots T1..T8 as they come)
  if (__any(t & 1u)) { if (t & 1u) lf_vert_pk(C, 0, L2); }
  if (__any(t & 2u)) { if (t & 2u) lf_horz_pk(C, 0, L2); }
  if (__any(t & 4u)) { if (t & 4u) lf_vert_pk(C, 0, L2); }
  if (__any(t & 8u)) { if (t & 8u) lf_horz_pk(C, 1, L2); }
  if (__any(t & 16u)) { if (t & 16u) lf_horz_pk(C, 0, L2); }
  if (__any(t & 32u)) { if (t & 32u) lf_vert_pk(C, 4, L2); }
  if (__any(t & 64u)) { if (t & 64u) lf_vert_pk(C, 4, L2); }
  if (__any(t & 128u)) { if (t & 128u) lf_horz_pk(C, 1, L2); }
  return;
#endif
  const bool hl_first = (t & 6u) == 6u;                 // T2 and T3: the horizontal edge first
  const bool vlo = (t & 5u) != 0, hl_late = (t & 18u) != 0 && !hl_first;
  const bool hr_first = (t & 8u) != 0, vhi = (t & 96u) != 0, hr_late = (t & 128u) != 0;
  if (__any(hl_first)) { if (hl_first) lf_horz_pk(C, 0, L2); }
  if (__any(vlo)) { if (vlo) lf_vert_pk(C, 0, L2); }
  if (__any(hl_late)) { if (hl_late) lf_horz_pk(C, 0, L2); }
  if (__any(hr_first)) { if (hr_first) lf_horz_pk(C, 1, L2); }
  if (__any(vhi)) { if (vhi) lf_vert_pk(C, 4, L2); }
  if (__any(hr_late)) { if (hr_late) lf_horz_pk(C, 1, L2); }
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
    // A row the filter left as it was is not written back: lflim() is zero for |R| >= 2L, i.e. across every
    // real edge in the picture (with the usual limits of 2..4 most pairs on textured content), and a store
    // that changes nothing still costs its bytes on the way to memory.
#ifndef THIP_LF_ALWAYS_STORE
    if (C.lo[r] == Cin.lo[r] && C.hi[r] == Cin.hi[r]) continue;
#endif
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

// (What pred_issue worked out of the command word is worked out again here -- a dozen instructions -- instead of
//  being carried in three registers through the inverse transform, which has none to spare.)
__device__ __forceinline__ void pred_finish(const PredWin &Qin, int W, uint2 pred[8], uint32_t flags, int x0, bool qpx, bool qpy) {
#ifndef THIP_NO_PRED_REMAT
  PredWin Q;
  {
    asm volatile("" : "+v"(flags), "+v"(x0));
    const int dx = (int)(int8_t)(flags >> THIP_INFO_MVX_SHIFT), dy = (int)(int8_t)(flags >> THIP_INFO_MVY_SHIFT);
    int mx, my;
    mv_axis(dx, qpx, mx, Q.mx2);
    mv_axis(dy, qpy, my, Q.my2);
    Q.sx = x0 + mx;
    Q.border = Qin.border;
#pragma unroll
    for (int r = 0; r < 9; r++) Q.w[r] = Qin.w[r];
  }
#else
  const PredWin &Q = Qin;
#endif
  const int xw = pred_xw(Q.sx + min(Q.mx2, 0), W);
  const bool ra = Q.my2 < 0, rb = Q.my2 > 0;   // that sample starts one source row down
  const bool two = (Q.mx2 | Q.my2) != 0;
  if (!__any(Q.border)) {
    // The two samples of a row are averaged, so which is called the first does not matter: T is the one whose rows are the window's
    // rows 0..7, U the other -- in the same rows (my2 = 0) or one further down.  One byte selector per sample for all rows, and the
    // choice of row is made on U's two extracted dwords (not on the three of the window, for both samples, as it used to be).
    const int offA = Q.sx - xw, offB = Q.sx + Q.mx2 - xw;   // 0..4
    const uint32_t selT = extract_sel(ra ? offB : offA), selU = extract_sel(ra ? offA : offB);
    const bool down = Q.my2 != 0;
#pragma unroll
    for (int r = 0; r < 8; r++) pred[r] = extract8s(Q.w[r], selT);
    if (two) {
      uint2 u = extract8s(Q.w[0], selU);
#pragma unroll
      for (int r = 0; r < 8; r++) {
        const uint2 n = extract8s(Q.w[r + 1], selU);
        pred[r].x = avg4_trunc(pred[r].x, down ? n.x : u.x);
        pred[r].y = avg4_trunc(pred[r].y, down ? n.y : u.y);
        u = n;
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
#ifndef THIP_COEF_CPOL
#define THIP_COEF_CPOL 0   // cache policy of the coefficient loads (experiment: 2 = non-temporal)
#endif
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
// state.c:978: x[0][0] = (int16)(dc * dc_quant) -- the low half of the block's first dword; the low
// 16 bits of a product depend on the low 16 bits of the factors only.
__device__ __forceinline__ uint32_t dequant_dc_lo(uint32_t w, uint32_t dcq) {
  return (w & 0xFFFF0000u) | (((w & 0xFFFFu) * dcq) & 0xFFFFu);
}
// ... with the raw DC coming from the DC array instead of the slot when dcraw says so
__device__ __forceinline__ uint32_t dequant_dc_lo(uint32_t w, uint32_t dcq, uint32_t dcraw) {
  return dequant_dc_lo((dcraw & 0x10000u) ? ((w & 0xFFFF0000u) | (dcraw & 0xFFFFu)) : w, dcq);
}

struct ReconLane {
  uint32_t flags, dcp, dcq;       // command word 0 (0 past the ragged edge), DC-only value {p,p}, dc_quant
  uint32_t dcraw;                 // bit 16 set: bits 0-15 are the block's raw DC (StreamK::dc, or command word 1 in the levels form), to be used instead of x[0][0] of its slot
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

// ---- the two coefficient forms (include/theora_hip.h: THIP_COEFFS_*) ---------------------------------------------------------
// DEQUANT16: one 128-byte slot per block, int16, AC already dequantised by the caller (what oc_state_frag_recon receives).
// LEVELS: the quantised values as the tokens carry them plus the frame's dequantisation tables; the kernel does
// `(ogg_int16_t)(coeff*ac_quant[zzi])` (decode.c:1573-1574, the table picked as decode.c:1537-1538 does: plane, qii,
// qti = the block is not intra) -- half the bytes of the largest term of a dense frame.  The slot array is then counted in
// 64-byte UNITS (groups of 64 units = 4 KB, piece-major like the int16 groups): a block of a narrow tile owns one unit, four
// 16-byte pieces, piece j = row pair j: for d = 0..3 the bytes { x[2j][2d], x[2j][2d+1], x[2j+1][2d], x[2j+1][2d+1] } (two
// packed shifts turn a dword into the two int16 pairs the row pass wants); a block of a WIDE tile (some level of the tile
// does not fit eight bits: tile_slot0 bit 31) owns two consecutive units holding the eight int16 pieces of the other form.
constexpr uint32_t kSlotWide = THIP_SLOT_WIDE;
constexpr int kLdsTabOff = 6144;       // the plane's six tables (768 bytes) in a wave's LDS area, behind anything the loads stage
struct CoefForm {                      // wave-uniform
  bool levels, wide;
  uint32_t slot0;                      // first slot (DEQUANT16) / first unit (LEVELS) of the tile
};
__device__ __forceinline__ CoefForm coef_form(uint32_t slot0_word, bool levels) {
  CoefForm F;
  F.levels = levels;
  F.wide = levels && (slot0_word & kSlotWide) != 0;
  F.slot0 = levels ? slot0_word & ~kSlotWide : slot0_word;
  return F;
}
// piece q (0..3) of unit `unit`
__device__ __forceinline__ const int4 *unit_piece(const int4 *coeffs, uint32_t unit, int q) {
  return coeffs + ((size_t)(unit >> 6) * 256 + (size_t)q * 64 + (unit & 63));
}
// piece q (0..7) of the wide block that starts at unit `unit`
__device__ __forceinline__ const int4 *wide_piece(const int4 *coeffs, uint32_t unit, int q) { return unit_piece(coeffs, unit + (uint32_t)(q >> 2), q & 3); }
// four int8 levels -> the pairs { x[2j][c], x[2j+1][c] } of an even and an odd column
__device__ __forceinline__ void unpack_levels(uint32_t w, uint32_t &even, uint32_t &odd) {
  const pk16 v = as_pk(w);
  odd = as_u32(v >> 8);            // bytes 1 and 3, sign-extended (v_pk_ashrrev_i16)
  even = as_u32((v << 8) >> 8);    // bytes 0 and 2
}
// Unpacking and multiplying in one go: `(ogg_int16_t)(level * ac_quant)` for the four int8 levels of a dword, the byte picked and
// sign-extended by the multiplier's SDWA source select, each product written to its half of the result: four operations where
// unpack_levels + two packed multiplies take five.
#ifndef THIP_NO_SDWA_DEQ
__device__ __forceinline__ void dequant_levels(uint32_t w, uint32_t t_even, uint32_t t_odd, uint32_t &even, uint32_t &odd) {
  asm("v_mul_lo_u16_sdwa %0, sext(%1), %2 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:WORD_0" : "=v"(even) : "v"(w), "v"(t_even));
  asm("v_mul_lo_u16_sdwa %0, sext(%1), %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2 src1_sel:WORD_1" : "+v"(even) : "v"(w), "v"(t_even));
  asm("v_mul_lo_u16_sdwa %0, sext(%1), %2 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:WORD_0" : "=v"(odd) : "v"(w), "v"(t_odd));
  asm("v_mul_lo_u16_sdwa %0, sext(%1), %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:WORD_1" : "+v"(odd) : "v"(w), "v"(t_odd));
}
#else
__device__ __forceinline__ void dequant_levels(uint32_t w, uint32_t t_even, uint32_t t_odd, uint32_t &even, uint32_t &odd) {
  uint32_t e, o;
  unpack_levels(w, e, o);
  even = as_u32(as_pk(e) * as_pk(t_even));
  odd = as_u32(as_pk(o) * as_pk(t_odd));
}
#endif
__device__ __forceinline__ uint32_t pk_mul_lo(uint32_t a, uint32_t b) { return as_u32(as_pk(a) * as_pk(b)); }   // low 16 bits of each product: the (ogg_int16_t) cast
// the dequantisation table of a block inside its plane's six: qii * 2 + qti (decode.c:1537-1538)
__device__ __forceinline__ uint32_t table_of(uint32_t flags) {
  return ((flags >> THIP_INFO_QII_SHIFT) & 3u) * 2u + (((flags >> THIP_INFO_REFI_SHIFT) & 3u) != (uint32_t)THIP_FRAME_SELF ? 1u : 0u);
}
// The plane's six tables into the wave's LDS (LDS-DMA, lanes 0..47 one 16-byte piece each).  Issued BEFORE the command words are
// requested: loads come back in order, so whoever has its command word also has the tables.
__device__ __forceinline__ void tables_to_lds(const uint4 *dequant, int pli, int lane, uint4 *lds_wave, int tab16 = kLdsTabOff / 16) {
  if (lane < 48)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(dequant + pli * 48 + lane),
                                     (__attribute__((address_space(3))) void *)(lds_wave + tab16), 16, 0, 0);
}

// ---- k_recon in three parts, so that the residual can be computed by ALL lanes of the wave ------
__device__ __forceinline__ void recon_issue(const ReconPlane &R, const ReconLane &L, PredWin &Q, bool &inter,
                                            const uint8_t *&ref, bool write_map = true) {
  const int refi = L.coded ? (int)((L.flags >> THIP_INFO_REFI_SHIFT) & 3u) : THIP_FRAME_PREV;
  inter = refi != THIP_FRAME_SELF && !(R.debug & 2);
  ref = refi == THIP_FRAME_PREV ? R.prev : R.gold;
  Q.border = false;
  if (inter) pred_issue(Q, ref, R.stride, R.nh * 8, R.nv * 8, L.x0, L.y0, L.coded ? L.flags : 0u, R.qpx, R.qpy);
  if (write_map) R.coded_map[(L.y0 >> 3) * R.nh + (L.x0 >> 3)] = L.coded ? 1 : 0;
}

// The eight reconstructed rows of this lane's block: predictor (fragment.c:49-80: 128, one block, or the
// average of two) + residual, clamped.
__device__ __forceinline__ void recon_rows(const ReconPlane &R, const PredWin &Q, bool inter, const uint32_t Y[32],
                                           uint2 rows[8], uint32_t Lflags, int Lx0) {
  uint2 pred[8];
#pragma unroll
  for (int r = 0; r < 8; r++) pred[r] = make_uint2(0x80808080u, 0x80808080u);
  if (inter) pred_finish(Q, R.nh * 8, pred, Lflags, Lx0, R.qpx, R.qpy);
#pragma unroll
  for (int r = 0; r < 8; r++)
    rows[r] = pk_recon_row(as_pk(Y[r * 4 + 0]), as_pk(Y[r * 4 + 1]), as_pk(Y[r * 4 + 2]), as_pk(Y[r * 4 + 3]), pred[r]);
}

__device__ __forceinline__ void recon_finish(const ReconPlane &R, const ReconLane &L, const PredWin &Q, bool inter,
                                             const uint32_t Y[32]) {
  uint8_t *dst = R.self + (ptrdiff_t)L.y0 * R.stride + L.x0;
  uint2 rows[8];
  recon_rows(R, Q, inter, Y, rows, L.coded ? L.flags : 0u, L.x0);
  if (!(R.debug & 4)) {
#pragma unroll
    for (int r = 0; r < 8; r++) store_row8(dst + (ptrdiff_t)r * R.stride, rows[r]);
  }
}

// ---- one block per lane (more than 32 lanes of the tile own coefficients) ----------------------------------------------------
// The loads: coefficients go global -> LDS directly (LDS address = wave-uniform base + lane * 16: no VGPRs are tied up and
// nothing can make the compiler touch -- i.e. wait for -- the data before the predictor loads are out); every lane loads (lanes
// without coefficients re-read the tile's first slot: same cache lines).  NLDS of the eight int16 pieces are staged (k_recon: 8;
// k_recon_lf: 7, its LDS area is 7 KB), six of a wide tile's (the tables sit behind them), the rest stays in registers; the
// four pieces of a narrow unit all go through LDS.
template <int NLDS>
__device__ __forceinline__ void dense_issue(const int4 *coeffs_p, const CoefForm &F, bool has_coeff, uint32_t prefix, uint4 *lds_wave, int4 &w7) {
  w7 = make_int4(0, 0, 0, 0);
  if (!F.levels) {
    const uint32_t slot = F.slot0 + (has_coeff ? prefix : 0u);
    const int4 *tp = coeffs_p + ((size_t)(slot >> 6) * 512 + (slot & 63));
#pragma unroll
    for (int q = 0; q < NLDS; q++)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(tp + q * 64),
                                       (__attribute__((address_space(3))) void *)(lds_wave + q * 64), 16, 0, THIP_COEF_CPOL);
    if (NLDS < 8) w7 = tp[7 * 64];
  } else if (!F.wide) {
    const uint32_t unit = F.slot0 + (has_coeff ? prefix : 0u);
#pragma unroll
    for (int q = 0; q < 4; q++)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)unit_piece(coeffs_p, unit, q),
                                       (__attribute__((address_space(3))) void *)(lds_wave + q * 64), 16, 0, THIP_COEF_CPOL);
  } else {
    // a wide tile (rare): six pieces now -- the tables sit behind them --, the last two in a second round (dense_finish)
    const uint32_t unit = F.slot0 + (has_coeff ? 2u * prefix : 0u);
#pragma unroll
    for (int q = 0; q < 6; q++)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)wide_piece(coeffs_p, unit, q),
                                       (__attribute__((address_space(3))) void *)(lds_wave + q * 64), 16, 0, THIP_COEF_CPOL);
  }
}

// Residual of the lanes that own coefficients, one block per lane (the whole wave executes the
// 16 one-dimensional transforms whether 1 lane or 64 need them).  Call it behind `s_waitcnt vmcnt(0)` (LDS-DMA data has landed).
template <int NLDS>
__device__ __forceinline__ void dense_finish(const int4 *coeffs_p, const CoefForm &F, bool has_coeff, uint32_t prefix, uint4 *lds_wave, int lane,
                                             const ReconLane &L, const int4 &w7, uint32_t Y[32]) {
  const uint4 *lds_coef = lds_wave + lane;
  const uint4 *tab = lds_wave + kLdsTabOff / 16 + table_of(L.flags) * 8;   // (levels form)
  uint32_t P[32];
  if (F.levels && !F.wide) {
    // row pair by row pair: its 16 levels, its two table pieces, `(ogg_int16_t)(coeff * ac_quant[zzi])` (decode.c:1573).  The
    // fences keep the scheduler from fetching all eight table pieces ahead of the first product -- 32 registers the wave does
    // not have next to its predictor windows (it spilled them, and waited for them first).
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const uint4 w = lds_coef[j * 64];
      const uint4 t0 = tab[2 * j], t1 = tab[2 * j + 1];
      dequant_levels(w.x, t0.x, t0.y, P[j * 8 + 0], P[j * 8 + 1]);
      dequant_levels(w.y, t0.z, t0.w, P[j * 8 + 2], P[j * 8 + 3]);
      dequant_levels(w.z, t1.x, t1.y, P[j * 8 + 4], P[j * 8 + 5]);
      dequant_levels(w.w, t1.z, t1.w, P[j * 8 + 6], P[j * 8 + 7]);
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
    const int nlds = F.levels ? 6 : NLDS;   // (compile time)
#pragma unroll
    for (int q = 0; q < 8; q++) {
      if (q == 6 && F.levels) {   // the wide tile's second round: pieces 6 and 7 over pieces 0 and 1
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const uint32_t unit = F.slot0 + (has_coeff ? 2u * prefix : 0u);
#pragma unroll
        for (int k = 0; k < 2; k++)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)wide_piece(coeffs_p, unit, 6 + k),
                                           (__attribute__((address_space(3))) void *)(lds_wave + k * 64), 16, 0, THIP_COEF_CPOL);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      uint4 w;
      if (q < nlds) w = lds_coef[q * 64];
      else if (F.levels) w = lds_coef[(q - 6) * 64];
      else w = make_uint4((uint32_t)w7.x, (uint32_t)w7.y, (uint32_t)w7.z, (uint32_t)w7.w);
      if (F.levels) {   // int16 levels of a wide tile
        const uint4 t = tab[q];
        w.x = pk_mul_lo(w.x, t.x);
        w.y = pk_mul_lo(w.y, t.y);
        w.z = pk_mul_lo(w.z, t.z);
        w.w = pk_mul_lo(w.w, t.w);
        __builtin_amdgcn_sched_barrier(0);
      }
      P[q * 4 + 0] = w.x;
      P[q * 4 + 1] = w.y;
      P[q * 4 + 2] = w.z;
      P[q * 4 + 3] = w.w;
    }
  }
  P[0] = dequant_dc_lo(P[0], L.dcq, L.dcraw);   // x[0][0] arrives raw
  const int last_zzi = (int)((L.flags >> THIP_INFO_LAST_ZZI_SHIFT) & 0x7Fu);
  // the masks are all ones for last_zzi > 10: ~100 instructions skipped when no owner needs them
  if (__any(L.has_coeff && last_zzi <= 10)) pk_mask_by_last_zzi(P, last_zzi);
  const bool all_zz10 = !__any(L.has_coeff && last_zzi > 10);
  // The int16 form keeps the 2 x 2 transposes between the passes: with the column pass straight off the row pass's registers
  // (pk_idct8_cols) k_recon_lf<false> spills a predictor window row -- 16 bytes of scratch a lane, VERDICT r05 -- because its 32
  // coefficient registers all arrive at once (the levels form's come row pair by row pair out of the dequantisation).
#ifndef THIP_INT16_FUSED_COLS
  if (F.levels || NLDS == 8) pk_idct8x8<true>(P, Y, all_zz10);    // (NLDS == 8: k_recon, which has the registers)
  else pk_idct8x8<false>(P, Y, all_zz10);
#else
  pk_idct8x8(P, Y, all_zz10);
#endif
}

// ---- at most 64/LPB lanes of the wave own coefficients (the usual case
// outside synthetic worst cases: SURVEY section 6 has 80 % of the coded blocks DC-only): LPB lanes
// (4 or 2) share a block -- lane LPB*g+j takes row pairs j*NP..j*NP+NP-1 (NP = 4/LPB) of the
// g-th owner for the row pass and the same column pairs for the column pass, the transpose
// between goes through the wave's LDS area (free once the coefficients are in registers) -- so
// the wave executes 2*NP packed 1-D transforms instead of 8.  Bit-exact with dense_finish:
// the same operations on the same values.  Must be called by all 64 lanes.
// lds = the wave's 8 KB area as dwords; meta = 64 dwords of LDS (two words per owner rank, up to 32 owners).
// The g-th owner's coefficients sit in slot slot0+g (slots are numbered in lane order inside a
// tile), so the sharing lanes fetch their own row pairs straight from the slot -- 32*NP bytes per
// lane (16*NP of a narrow unit) instead of the whole wave staging 8 KB of which a fraction is used.
template <int LPB>
__device__ __forceinline__ void residual_shared_load(const int4 *coeffs, const CoefForm &F, int nown, int lane,
                                                     int4 W[4 / LPB][2]) {
  constexpr int NP = 4 / LPB;
  const int g = min(lane / LPB, nown - 1), j = lane % LPB;   // surplus groups re-read the last owner's slot
#pragma unroll
  for (int n = 0; n < NP; n++) {
    const int rp = j * NP + n;
    if (!F.levels) {
      const uint32_t slot = F.slot0 + (uint32_t)g;
      const int4 *tp = coeffs + ((size_t)(slot >> 6) * 512 + (slot & 63));
      W[n][0] = tp[(2 * rp) * 64];
      W[n][1] = tp[(2 * rp + 1) * 64];
    } else if (!F.wide) {
      W[n][0] = *unit_piece(coeffs, F.slot0 + (uint32_t)g, rp);
      W[n][1] = make_int4(0, 0, 0, 0);
    } else {
      W[n][0] = *wide_piece(coeffs, F.slot0 + 2u * (uint32_t)g, 2 * rp);
      W[n][1] = *wide_piece(coeffs, F.slot0 + 2u * (uint32_t)g, 2 * rp + 1);
    }
  }
}

// COMPACT: the results go where the exchange was (4 KB of LDS instead of 8 for LPB = 2): every lane
// collects all its column pairs first, the wave's LDS traffic settles, then results are written.
// OUT = 32: the owner lane collects its whole block (Y[32]); OUT = 8 (k_recon_lf_sb, four lanes per block of the PICTURE too):
// lane 4b + p collects rows 2p, 2p+1 of block b, whose owner rank is `prefix` (Y[8]); OUT = 16 (k_recon_lf_h, two lanes per block):
// lane 2b + p collects rows 4p .. 4p + 3 (Y[16]).  tab16: where the tables are (levels form).
template <int LPB, bool COMPACT = false, int OUT = 32>
__device__ __forceinline__ void residual_shared(const int4 W[4 / LPB][2], const CoefForm &F, uint32_t *lds, uint32_t *meta, int lane,
                                                const ReconLane &L, uint32_t prefix, uint32_t *Y, int tab16 = kLdsTabOff / 16) {
  constexpr int NP = 4 / LPB;                        // row pairs (and column pairs) per lane
  const int last_zzi = (int)((L.flags >> THIP_INFO_LAST_ZZI_SHIFT) & 0x7Fu);
  if (L.has_coeff) {
    meta[prefix] = (uint32_t)last_zzi | table_of(L.flags) << 8 | (L.dcq << 16);   // rank -> last_zzi, table, dc_quant of that owner
    meta[32 + prefix] = L.dcraw;                          // ... and its raw DC when that does not come from the slot
  }
  const int g = lane / LPB, j = lane % LPB;
  const uint32_t mg = meta[g];                      // (garbage for g >= number of owners: results unused)
  const uint32_t dcraw_g = meta[32 + g];
  const int lz = (int)(mg & 0x7Fu);
  const uint32_t dcq_g = j == 0 ? mg >> 16 : 1u;    // the lane holding row pair 0 dequantises x[0][0]
  const bool c3 = lz <= 3, c10 = lz <= 10;
  const uint4 *tab = reinterpret_cast<const uint4 *>(lds) + tab16 + min((mg >> 8) & 7u, 5u) * 8;   // (levels form; surplus groups read garbage owner words)
  pk16 Rr[NP][8];
#pragma unroll
  for (int n = 0; n < NP; n++) {
    const int rp = j * NP + n;                       // row pair: rows 2rp, 2rp+1
    const int4 w0 = W[n][0], w1 = W[n][1];
    uint32_t X[8] = {(uint32_t)w0.x, (uint32_t)w0.y, (uint32_t)w0.z, (uint32_t)w0.w,
                     (uint32_t)w1.x, (uint32_t)w1.y, (uint32_t)w1.z, (uint32_t)w1.w};   // {x[2rp][c], x[2rp+1][c]}, c = 0..7
    if (F.levels && !F.wide) {
      const uint4 t0 = tab[2 * rp], t1 = tab[2 * rp + 1];   // decode.c:1573
      dequant_levels((uint32_t)w0.x, t0.x, t0.y, X[0], X[1]);
      dequant_levels((uint32_t)w0.y, t0.z, t0.w, X[2], X[3]);
      dequant_levels((uint32_t)w0.z, t1.x, t1.y, X[4], X[5]);
      dequant_levels((uint32_t)w0.w, t1.z, t1.w, X[6], X[7]);
    } else if (F.levels) {
      const uint4 t0 = tab[2 * rp], t1 = tab[2 * rp + 1];
      X[0] = pk_mul_lo(X[0], t0.x);
      X[1] = pk_mul_lo(X[1], t0.y);
      X[2] = pk_mul_lo(X[2], t0.z);
      X[3] = pk_mul_lo(X[3], t0.w);
      X[4] = pk_mul_lo(X[4], t1.x);
      X[5] = pk_mul_lo(X[5], t1.y);
      X[6] = pk_mul_lo(X[6], t1.z);
      X[7] = pk_mul_lo(X[7], t1.w);
    }
    if (n == 0) X[0] = dequant_dc_lo(X[0], dcq_g, j == 0 ? dcraw_g : 0u);
    // what the variant selected by last_zzi does not read is zero (pk_mask_by_last_zzi)
#pragma unroll
    for (int c = 0; c < 8; c++) {
      const uint32_t m10 = ((2 * rp + c <= 3) ? 0x0000FFFFu : 0u) | ((2 * rp + 1 + c <= 3) ? 0xFFFF0000u : 0u);
      const uint32_t m3 = ((rp == 0 && c <= 1) ? 0x0000FFFFu : 0u) | ((rp == 0 && c == 0) ? 0xFFFF0000u : 0u);
      Rr[n][c] = as_pk(X[c] & (c3 ? m3 : (c10 ? m10 : 0xFFFFFFFFu)));
    }
    pk_idct8(Rr[n][0], Rr[n][1], Rr[n][2], Rr[n][3], Rr[n][4], Rr[n][5], Rr[n][6], Rr[n][7]);
  }
  // exchange inside the group: every lane publishes its row pairs, collects its column pairs
  uint32_t *xch = lds;                                       // NP*8 dwords per lane
  uint32_t *res = COMPACT ? lds : lds + 64 * NP * 8;         // 32 dwords per owner
#pragma unroll
  for (int n = 0; n < NP; n++) {
    uint4 *x4 = reinterpret_cast<uint4 *>(xch + (lane * NP + n) * 8);
    x4[0] = make_uint4(as_u32(Rr[n][0]), as_u32(Rr[n][1]), as_u32(Rr[n][2]), as_u32(Rr[n][3]));
    x4[1] = make_uint4(as_u32(Rr[n][4]), as_u32(Rr[n][5]), as_u32(Rr[n][6]), as_u32(Rr[n][7]));
  }
  pk16 Qc[NP][8];
#if defined(THIP_HAVE_IDCT8_COLS) && !defined(THIP_NO_FUSED_TRANSPOSE)
  constexpr bool kFusedCols = true;    // the column pass reads the row pass's pairs as they are (pk_idct8_cols)
#else
  constexpr bool kFusedCols = false;
#endif
#pragma unroll
  for (int n = 0; n < NP; n++) {
    const int cp = j * NP + n;                       // column pair: columns 2cp, 2cp+1
#pragma unroll
    for (int rp = 0; rp < 4; rp++) {                 // row pair rp lives at slot (g*LPB*NP + rp) = g*4 + rp
      const uint2 ab = *reinterpret_cast<const uint2 *>(xch + (g * 4 + rp) * 8 + 2 * cp);
      if (kFusedCols) {
        Qc[n][rp] = as_pk(ab.x);                     // { r[2rp][2cp], r[2rp+1][2cp] }
        Qc[n][4 + rp] = as_pk(ab.y);                 // the same of column 2cp + 1
      } else {
        Qc[n][2 * rp] = as_pk(__builtin_amdgcn_perm(ab.y, ab.x, 0x05040100u));
        Qc[n][2 * rp + 1] = as_pk(__builtin_amdgcn_perm(ab.y, ab.x, 0x07060302u));
      }
    }
  }
  if (COMPACT) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // all of the exchange is read before results overwrite it
#pragma unroll
  for (int n = 0; n < NP; n++) {
    const int cp = j * NP + n;
#if defined(THIP_HAVE_IDCT8_COLS) && !defined(THIP_NO_FUSED_TRANSPOSE)
    {
      const pk16 A[4] = {Qc[n][0], Qc[n][1], Qc[n][2], Qc[n][3]}, B[4] = {Qc[n][4], Qc[n][5], Qc[n][6], Qc[n][7]};
      pk_idct8_cols(A, B, Qc[n][0], Qc[n][1], Qc[n][2], Qc[n][3], Qc[n][4], Qc[n][5], Qc[n][6], Qc[n][7]);
    }
#else
    pk_idct8(Qc[n][0], Qc[n][1], Qc[n][2], Qc[n][3], Qc[n][4], Qc[n][5], Qc[n][6], Qc[n][7]);
#endif
#pragma unroll
    for (int r = 0; r < 8; r++) res[g * 32 + r * 4 + cp] = as_u32(pk_descale(Qc[n][r]));   // Y[r*4+k] layout of the owner
  }
  if (L.has_coeff) {
    // (OUT < 32: the block belongs to 32 / OUT lanes of the picture too, each collects its rows)
    const uint4 *y4 = reinterpret_cast<const uint4 *>(res + prefix * 32) + (OUT == 8 ? 2 * (lane & 3) : (OUT == 16 ? 4 * (lane & 1) : 0));
#pragma unroll
    for (int q = 0; q < OUT / 4; q++) {
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
// One tile of one stream: k_recon's wave.  lds_wave: the wave's 8 KB of LDS, meta: its 64
// dwords for residual_shared.  (Lanes leave at different places; the caller gets the whole wave back.)
template <bool LEVELS>
__device__ __forceinline__ void recon_tile(const StreamK &S, const int unit, const int lane, uint4 *const lds_wave,
                                           uint32_t *const meta, unsigned long long *tr) {
  // scalar batch 1: the stream's pointers and the plane boundaries (pinned by the empty asm:
  // left alone, the compiler sinks each scalar load to its first use, which turns one wait
  // into a chain of dependent ones)
  const uint2 *info_p = S.info;
  const int4 *coeffs_p = S.coeffs;
  const uint32_t *slot0_p = S.tile_slot0;
  uint8_t *self = S.self;
  const uint8_t *prev = S.prev, *gold = S.gold;
  uint8_t *coded_map = S.coded_map;
  const uint8_t *coded_prev = S.coded_prev;
  const int16_t *dc_p = S.dc;
  const uint4 *dq_p = S.dequant;
  const int te0 = S.tile_end[0], te1 = S.tile_end[1], te2 = S.tile_end[2];
  const int debug = S.debug, sqpx = S.qpx, sqpy = S.qpy, skip_ok = S.skip_ok;
  asm volatile("" ::"s"(info_p), "s"(coeffs_p), "s"(slot0_p), "s"(self), "s"(prev), "s"(gold), "s"(coded_map),
               "s"(coded_prev), "s"(dc_p), "s"(dq_p), "s"(te0), "s"(te1), "s"(te2), "s"(debug), "s"(sqpx), "s"(sqpy), "s"(skip_ok));
  if (unit >= te2) return;
  const int pli = (unit >= te0 ? 1 : 0) + (unit >= te1 ? 1 : 0);
  constexpr bool levels = LEVELS;   // (one kernel per coefficient form: each is straight-line code for its own)
  if (levels) tables_to_lds(dq_p, pli, lane, lds_wave);   // (first: whoever has its command word has the tables)
  // scalar batch 2: the plane's geometry
  const PlaneK G = S.pl[pli];
  asm volatile("" ::"s"(G.nh), "s"(G.nv), "s"(G.stride), "s"(G.off), "s"(G.tiles_x), "s"(G.fro));

  const int rel = unit - (pli == 0 ? 0 : (pli == 1 ? te0 : te1));
  const int sby = rel / G.tiles_x;
  const int tx = rel - sby * G.tiles_x;
  const int h = lane & 15;
  const int bx = tx * 16 + (lane >> 4) * 4 + hilb_col(h);
  const int by = sby * 4 + hilb_row(h);
  const bool valid = bx < G.nh && by < G.nv;

  // ---- 1. command word + first slot of the tile (one round trip) -------------------------------
  const uint32_t slot0w = slot0_p[unit];
  const uint2 info = info_p[(size_t)unit * THIP_TILE_FRAGS + lane];
  // ... and, when the frame may leave blocks where they are (skip_ok, wave-uniform), what the
  // previous frame did to this block and to the four blocks it shares an edge with: if none of
  // them was coded, neither the reconstruction nor the loop filter of the previous frame changed
  // a pixel of it (an edge is filtered only if a block on one of its sides is coded).
  uint32_t touched = 1;
  if (skip_ok) {
    const int cx = min(bx, G.nh - 1), cy = min(by, G.nv - 1);
    const uint8_t *op = coded_prev + G.fro + cy * G.nh + cx;
    const int dl = cx > 0 ? 1 : 0, dr = cx < G.nh - 1 ? 1 : 0, du = cy > 0 ? G.nh : 0, dd = cy < G.nv - 1 ? G.nh : 0;
    // (a neighbour outside the plane reads the block's own flag again)
    touched = (uint32_t)op[0] | op[-dl] | op[dr] | op[-du] | op[dd];
  }
  // the block's un-predicted DC when it does not travel in the command stream (k_dc_unpredict's output)
  uint32_t dcv = 0;
  if (dc_p) dcv = 0x10000u | (uint16_t)dc_p[G.fro + min(by, G.nv - 1) * G.nh + min(bx, G.nh - 1)];
  // all loads are consumed here as far as the compiler can tell, so the scalar load is issued
  // next to the vector loads instead of being sunk behind the wait for them
  asm volatile("" ::"s"(slot0w), "v"(info.x), "v"(touched), "v"(dcv));
#ifdef THIP_TRACE
  THIP_TR(tr, 1);   // first round trip done
#endif
  if (levels && !dc_p) dcv = 0x10000u | (info.y & 0xFFFFu);   // levels form: every coded block's raw DC rides in command word 1
  const CoefForm F = coef_form(slot0w, levels);

  ReconLane L;
  L.flags = valid ? info.x : 0u;
  // DC dequantisation (state.c:967-979): command word 1 = dc_quant << 16 | raw DC of a DC-only block
  L.dcq = info.y >> 16;
  L.dcraw = dcv;
  L.dcp = ((uint32_t)(((int)(int16_t)((dcv ? dcv : info.y) & 0xFFFFu) * (int)L.dcq + 15) >> 5) & 0xFFFFu) * 0x00010001u;   // {p, p}
  L.coded = (L.flags & THIP_INFO_CODED) != 0;
  L.dc_only = L.coded && (L.flags & THIP_INFO_DC_ONLY) != 0;
  L.has_coeff = L.coded && !L.dc_only;
  L.x0 = bx * 8;
  L.y0 = by * 8;
  // Uncoded now, untouched by the previous frame, and the destination buffer holds the frame before
  // that: the copy PREV -> SELF of fragment.c:37 would write what is there already.
  const bool work = valid && (L.coded || touched != 0);
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

  if (valid && !work) R.coded_map[by * G.nh + bx] = 0;   // (recon_issue writes the flag of the others)
  if (!__any(work)) return;                               // a tile of static background

  // ---- 2. coefficient loads (slot by prefix count over the mask), issued, not waited for.
  //         The branch is wave-uniform and every lane loads (lanes without coefficients re-read
  //         the tile's first slot -- same cache lines, no extra traffic -- and are overridden
  //         in step 4). -----------------------------------------------------------------------------
  const uint64_t mask = __ballot(L.has_coeff);
  PredWin Q;
  bool inter = false;
  const uint8_t *ref = nullptr;
  uint32_t Y[32];
  const int nown = __popcll(mask);
  const uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
  uint32_t *const lds_dw = reinterpret_cast<uint32_t *>(lds_wave);
  if (nown == 0 || (debug & 9)) {
    if (work) recon_issue(R, L, Q, inter, ref);   // (!valid: past the ragged edge of the plane)
  } else if (nown <= 16 && !(debug & 32)) {
    // ---- few owners: four lanes per block, pieces straight from the slots ------------------------
    int4 W[1][2];
    residual_shared_load<4>(coeffs_p, F, nown, lane, W);
    if (work) recon_issue(R, L, Q, inter, ref);
    THIP_TR(R.tr, 2);
    residual_shared<4>(W, F, lds_dw, meta, lane, L, prefix, Y);
    THIP_TR(R.tr, 3);
  } else if (nown <= 32 && !(debug & 32)) {
    int4 W[2][2];
    residual_shared_load<2>(coeffs_p, F, nown, lane, W);
    if (work) recon_issue(R, L, Q, inter, ref);
    THIP_TR(R.tr, 2);
    residual_shared<2, true>(W, F, lds_dw, meta, lane, L, prefix, Y);   // (results over the exchange: 4 KB, the meta words behind)
    THIP_TR(R.tr, 3);
  } else {
    // ---- many owners: one lane per block, coefficients global -> LDS directly (dense_issue) -------
    int4 w7;
    dense_issue<8>(coeffs_p, F, L.has_coeff, prefix, lds_wave, w7);
    if (work) recon_issue(R, L, Q, inter, ref);
    THIP_TR(R.tr, 2);
    // The LDS-DMA loads above are counted by vmcnt; the compiler's own wait before the LDS reads
    // below is not something to rely on (it vanished when the loads moved into a conditional
    // block and the reads returned stale LDS), so it is stated.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    dense_finish<8>(coeffs_p, F, L.has_coeff, prefix, lds_wave, lane, L, w7, Y);
    THIP_TR(R.tr, 3);
  }
  if (!work) return;
  if (!L.has_coeff || (debug & 9)) {   // DC-only: the rounded value (state.c:972); uncoded: zero residual
    const uint32_t fill = L.dc_only ? L.dcp : 0u;
#pragma unroll
    for (int i = 0; i < 32; i++) Y[i] = fill;
  }
  recon_finish(R, L, Q, inter, Y);
  THIP_TR(R.tr, 4);
}

template <bool LEVELS>
__global__ __launch_bounds__(64 * THIP_RECON_WG_WAVES, THIP_RECON_WAVES) void k_recon(const BatchK B) {
  const StreamK &S = B.s[blockIdx.y];
  const int lane = (int)threadIdx.x & 63;
  // Workgroups are handed to the 8 XCDs round-robin (id mod 8) and each XCD has its own L2:
  // give XCD x the x-th contiguous band of tiles, so that the predictor windows of
  // neighbouring tiles -- which overlap by up to 16 pixels -- meet in one L2 instead of being
  // fetched from HBM by two.  gridDim.x is a multiple of 8.
  const int wg = ((int)blockIdx.x & 7) * ((int)gridDim.x >> 3) + ((int)blockIdx.x >> 3);
  const int unit = __builtin_amdgcn_readfirstlane(wg * THIP_RECON_WG_WAVES + ((int)threadIdx.x >> 6));  // tile
  unsigned long long *tr = nullptr;
#ifdef THIP_TRACE
  if (g_trace_buf && lane == 0) {
    tr = g_trace_buf + ((size_t)blockIdx.y * (gridDim.x * THIP_RECON_WG_WAVES) + unit) * 8;
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hwid), "=s"(xcc));
    tr[5] = hwid;
    tr[6] = xcc;
  }
  THIP_TR(tr, 0);
#endif
#ifndef THIP_RECON_LDS_PAD
#define THIP_RECON_LDS_PAD 0
#endif
  // 8 KB per wave and not a byte more -- 20 waves are 160 KB, a CU's LDS: the paths that need the meta words
  // (residual_shared) use the first 4 KB only and find them at 4 KB
  __shared__ uint4 s_coef[THIP_RECON_WG_WAVES * 8 * 64 + THIP_RECON_LDS_PAD / 16];   // [wave][piece][lane]: 8 KB per wave, wave-private
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  recon_tile<LEVELS>(S, unit, lane, s_coef + wave * 512, reinterpret_cast<uint32_t *>(s_coef + wave * 512) + 1024, tr);
}

__device__ __forceinline__ void lds_settle() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// ---------------------------------------------------------------------------------------
// k_dc_unpredict: oc_dec_dc_unpredict_mcu_plane (decode.c:1392-1500) on the device
// ---------------------------------------------------------------------------------------
// The predictor of a fragment reads the FINAL DC of its left, up-left, up and up-right neighbours
// (those that are coded and predicted from the same reference frame), so fragment (x, y) can be done
// once (x-1, y) and (x+1, y-1) are: all fragments with equal x + 2y are independent -- an anti-diagonal
// wavefront of slope 2.  One work group per plane, one thread per fragment ROW: a thread walks its row
// left to right and may take fragment x when the row above has finished fragment x+1; the work group
// steps in lock-step (one barrier per step, rows publish their progress in LDS), so a plane of
// nh x nv fragments takes about nh + 2*nv steps instead of nh*nv.
// What breaks the wavefront is pred_last (decode.c:1448: "default: pred=pred_last[refi]"): a fragment
// none of whose four neighbours qualifies takes the DC of the LAST fragment with its reference frame in
// raster order, which may sit at the far right end of the row above.  A thread therefore keeps its own
// row's last value per reference frame, and when its row has none yet it needs pred_last as it stood at
// the END of the row above (`carry`, handed down row by row): it waits until that row is complete.  Rows
// below wait behind it through the same progress test, so the result is exact whatever the picture;
// key frames never take that branch after row 0, inter frames rarely.
struct DcPlaneK {
  const int16_t *in;        // DC values as decoded from the tokens, raster order of the plane
  int16_t *out;             // un-predicted DC values, raster (may equal `in`: a fragment is read before it is written)
  const uint8_t *flags;     // raster flags (bit 0 coded, bits 1-2 refi) -- or null:
  const uint32_t *info;     // ... the stream's tile-ordered command words (word 0 of each pair)
  int nh, nv, tiles_x, tile_base;
  uint4 *ent;               // k_dc_prepare -> k_dc_wave: 16 bytes per fragment (thip_dc.h)
  uint8_t *rowhas;          // ... and per fragment row which references it has (bit r)
};
struct DcBatchK {
  DcPlaneK p[THIP_MAX_BATCH][3];
};

__device__ __forceinline__ uint32_t dc_flag(const DcPlaneK &P, int x, int y) {
  if (P.flags) return P.flags[(size_t)y * P.nh + x] & 7u;
  const int pos = (P.tile_base + (y >> 2) * P.tiles_x + (x >> 4)) * THIP_TILE_FRAGS + ((x >> 2) & 3) * 16 + hilb_inv(y & 3, x & 3);
  const uint32_t w = P.info[2 * (size_t)pos];
  return (w & THIP_INFO_CODED) ? (w & 7u) : 0u;   // coded | refi << 1; an uncoded fragment never matches (refi NONE, decode.c:658)
}

constexpr int kDcMaxRows = 1024;   // fragment rows per plane the kernel handles (one thread each): 8192 pixels

__global__ __launch_bounds__(kDcMaxRows) void k_dc_unpredict(const DcBatchK B) {
  const DcPlaneK &P = B.p[blockIdx.y][blockIdx.x];
  const int nh = P.nh, nv = P.nv;
  if (nh <= 0 || nv <= 0) return;
  __shared__ int s_prog[2][kDcMaxRows];        // fragments finished per row, double-buffered by step parity
  __shared__ short s_carry[kDcMaxRows][4];     // pred_last[refi] as it stands at the end of each row
  const int y = (int)threadIdx.x;
  const bool active = y < nv;
  s_prog[0][y] = active ? 0 : nh;
  s_prog[1][y] = active ? 0 : nh;
  __syncthreads();
  int x = 0;
  uint32_t f_l = 0, f_ul = 0, f_u = 0, f_ur = 0;    // flags of the neighbours of fragment x (0 = does not count)
  int d_l = 0, d_ul = 0, d_u = 0, d_ur = 0;         // their final DC values
  int pl0 = 0, pl1 = 0, pl2 = 0;                    // this row's last DC per reference frame ...
  uint32_t plv = 0;                                 // ... and which of them exist
  const int16_t *up_dc = P.out + (size_t)(y > 0 ? y - 1 : 0) * nh;
  const long long max_steps = (long long)nh * nv + 2 * nv + 8;   // even a fully serial picture ends
  for (long long step = 0; step < max_steps; step++) {
    const int cur = (int)(step & 1);
    if (s_prog[cur][nv - 1] >= nh) break;           // the last row is the last to finish
    int xn = x;
    if (active && x < nh) {
      const int above = y > 0 ? s_prog[cur][y - 1] : nh;
      if (above >= min(x + 2, nh)) {
        // the window over the row above slides with x; its new right end is final by the test above
        if (y > 0) {
          if (x == 0) {
            f_u = dc_flag(P, 0, y - 1);
            d_u = up_dc[0];
          }
          if (x + 1 < nh) {
            f_ur = dc_flag(P, x + 1, y - 1);
            d_ur = up_dc[x + 1];
          } else {
            f_ur = 0;
          }
        }
        const uint32_t f = dc_flag(P, x, y);
        bool done = true;
        int dc = 0;
        if (f & 1u) {
          const int r = (int)(f >> 1);
          const int mask = (f_l == f ? 1 : 0) | (f_ul == f ? 2 : 0) | (f_u == f ? 4 : 0) | (f_ur == f ? 8 : 0);
          int pred = 0;
          switch (mask) {                                       // decode.c:1450-1485
            case 0:
              if (plv >> r & 1u) pred = r == 0 ? pl0 : (r == 1 ? pl1 : pl2);
              else if (y == 0) pred = 0;                        // decode.c:1367: pred_last starts at 0
              else if (above >= nh) pred = s_carry[y - 1][r];
              else done = false;                                // the row above must finish first
              break;
            case 1: case 3: pred = d_l; break;
            case 2: pred = d_ul; break;
            case 4: case 6: case 12: pred = d_u; break;
            case 5: pred = (d_l + d_u) / 2; break;
            case 8: pred = d_ur; break;
            case 9: case 11: case 13: pred = (75 * d_l + 53 * d_ur) / 128; break;
            case 10: pred = (d_ul + d_ur) / 2; break;
            case 14: pred = (3 * (d_ul + d_ur) + 10 * d_u) / 16; break;
            default:   // 7, 15
              pred = (29 * (d_l + d_u) - 26 * d_ul) / 32;
              if (abs(pred - d_u) > 128) pred = d_u;
              else if (abs(pred - d_l) > 128) pred = d_l;
              else if (abs(pred - d_ul) > 128) pred = d_ul;
              break;
          }
          if (done) {
            dc = (int)(short)(P.in[(size_t)y * nh + x] + pred);   // a signed 16-bit bit-field in the reference (state.h:321)
            P.out[(size_t)y * nh + x] = (int16_t)dc;
            if (r == 0) pl0 = dc;
            else if (r == 1) pl1 = dc;
            else pl2 = dc;
            plv |= 1u << r;
          }
        }
        if (done) {
          f_l = f;
          d_l = dc;
          f_ul = f_u;
          d_ul = d_u;
          f_u = f_ur;
          d_u = d_ur;
          xn = x + 1;
          if (xn == nh) {
#pragma unroll
            for (int r = 0; r < 3; r++) {
              const int own = r == 0 ? pl0 : (r == 1 ? pl1 : pl2);
              s_carry[y][r] = (short)((plv >> r & 1u) ? own : (y > 0 ? (int)s_carry[y - 1][r] : 0));
            }
          }
        }
      }
    }
    x = xn;
    if (active) s_prog[cur ^ 1][y] = x;
    __syncthreads();   // stores of this step are complete and the progress is published before anyone reads either
  }
}

// ---------------------------------------------------------------------------------------
// k_expand_tokens: token -> coefficient expansion and AC dequantisation on the device
// ---------------------------------------------------------------------------------------
// What oc_dec_frags_recon_mcu_plane does per coded fragment before it calls oc_state_frag_recon
// (decode.c:1540-1581): the block's DCT tokens become 64 coefficients -- zeros, and at the zig-zag
// position of each token its value times the AC quantiser of that position, `(ogg_int16_t)(coeff *
// ac_quant[zzi])` (decode.c:1573) -- in natural order (dct_fzig_zag, decode.c:1574).  Here the host has
// only delimited the tokens (which tokens belong to which fragment is a serial walk through the per-index
// lists with their shared end-of-block runs, decode.c:1544-1570, that stays with the entropy decoder);
// what crosses PCIe is 4 bytes per non-zero coefficient instead of a 128-byte block, and the zero fill,
// the scatter through the zig-zag table, the dequantisation and the tile layout happen here.
// One wave per group of 64 coefficient slots: lane = slot; the group's 8 KB image is built in LDS in the
// piece-major layout k_recon wants and written out with eight coalesced 1 KB stores per wave.
struct TokK {
  const uint32_t *tok;       // zig-zag position << 16 | quantised value (16 bits, two's complement); position 0: the raw DC
  const uint2 *slot_tok;     // per slot: first token, count | dequant table number << 8
  const uint16_t *dq;        // up to 18 tables of 64, zig-zag order (oc_dequant tables, pli x qii x qti)
  int4 *coeffs;              // out: slot groups of 8192 bytes
  int nslots;
};
// natural position of zig-zag index i (internal.c:27, OC_FZIG_ZAG), four per dword
__host__ __device__ __forceinline__ int fzig_zag(int i) {
  constexpr uint8_t T[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
  return T[i];
}
__global__ __launch_bounds__(256) void k_expand_tokens(const TokK K) {
  __shared__ int4 s_img[4 * 512];                 // [wave][piece q][lane]: 8 KB per wave
  __shared__ uint16_t s_dq[18 * 64];
  for (int i = (int)threadIdx.x; i < 18 * 64; i += 256) s_dq[i] = K.dq[i];
  const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
  const int group = (int)blockIdx.x * 4 + wave;
  int4 *img = s_img + wave * 512;
#pragma unroll
  for (int q = 0; q < 8; q++) img[q * 64 + lane] = make_int4(0, 0, 0, 0);
  __syncthreads();
  const int slot = group * 64 + lane;
  if (slot < K.nslots) {
    const uint2 st = K.slot_tok[slot];
    const uint32_t *t = K.tok + st.x;
    const int n = (int)(st.y & 0xFFu), sel = (int)(st.y >> 8) & 31;
    const uint16_t *dq = s_dq + sel * 64;
    int16_t *mine = reinterpret_cast<int16_t *>(img);
    for (int k = 0; k < n; k++) {
      const uint32_t w = t[k];
      const int zz = (int)(w >> 16) & 63, v = (int)(int16_t)(w & 0xFFFFu);
      const int c = zz == 0 ? v : (int)(int16_t)(v * (int)dq[zz]);     // decode.c:1573; the DC stays raw (state.c:967-979)
      const int nat = fzig_zag(zz), r = nat >> 3, col = nat & 7;
      // piece q = 2*(r>>1) + (col>>2) at q*1024 + lane*16; inside it the pairs {x[2j][c], x[2j+1][c]}
      mine[((2 * (r >> 1) + (col >> 2)) * 1024 + lane * 16 + ((col & 3) * 2 + (r & 1)) * 2) >> 1] = (int16_t)c;
    }
  }
  __syncthreads();
  if ((size_t)group * 64 < (size_t)K.nslots) {
    int4 *out = K.coeffs + (size_t)group * 512;
#pragma unroll
    for (int q = 0; q < 8; q++) out[q * 64 + lane] = img[q * 64 + lane];
  }
}

// ---------------------------------------------------------------------------------------
// k_loopfilter (K3): one filter cell per lane over the whole frame
// ---------------------------------------------------------------------------------------
// One wave never straddles two planes (the cumulative cell counts in StreamK::cell_end are
// padded to whole waves), so the plane lookup is scalar, like k_recon's.
#ifndef THIP_LF_WG
#define THIP_LF_WG 256   // threads per workgroup of k_loopfilter
#endif
// The cell of plane-relative index rel (row-major over the (nh+1) x (nv+1) corners) of plane pli: k_loopfilter's lane.
__device__ __forceinline__ void lf_wave(const StreamK &S, const int pli, const int rel, const int rel_end) {
  uint8_t *self = S.self;
  const uint8_t *cmap = S.coded_map;
  const int L2 = S.flimit2;
  const PlaneK G = S.pl[pli];
  const int fy0 = S.lf_y0[pli], fy1 = S.lf_y1[pli];
  asm volatile("" ::"s"(self), "s"(cmap), "s"(L2), "s"(G.nh), "s"(G.nv), "s"(G.stride), "s"(G.off), "s"(G.fro), "s"(G.rcp_cx), "s"(fy0),
               "s"(fy1));
  const int nh = G.nh, nv = G.nv;
  if (rel >= rel_end || L2 == 0) return;
  uint32_t mu, ku;
  divmod_u24((uint32_t)rel, (uint32_t)(nh + 1), G.rcp_cx, mu, ku);
  const int k = (int)ku, m = (int)mu;
  CellPix C;
  bool a, b, c, d;
  if (S.lf_sparse) {
    // A frame with uncoded regions (wave-uniform choice, made on the host from the coded count):
    // the coded flags come first, and a wave none of whose 64 cells has an edge to filter -- a
    // static background -- ends here without touching a pixel.  The price is a second, dependent
    // round trip for the waves that stay, which is why fully coded frames take the other branch.
    lf_cell_flags(cmap + G.fro, nh, nv, k, m, a, b, c, d);
    const uint32_t t = lf_cell_ops(k, m, nh, nv, a, b, c, d, fy0, fy1);
    if (!__any(t != 0)) return;
    lf_cell_load(C, self + G.off, G.stride, nh, nv, k, m);
    lf_cell_pin(C);
    lf_cell_finish(C, self + G.off, G.stride, nh, nv, k, m, t, L2);
    return;
  }
  lf_cell_load(C, self + G.off, G.stride, nh, nv, k, m);
  lf_cell_flags(cmap + G.fro, nh, nv, k, m, a, b, c, d);
  const uint32_t t = lf_cell_ops(k, m, nh, nv, a, b, c, d, fy0, fy1);
  lf_cell_pin(C);
  lf_cell_finish(C, self + G.off, G.stride, nh, nv, k, m, t, L2);
}

__global__ __launch_bounds__(THIP_LF_WG) void k_loopfilter(const BatchK B) {
  const StreamK &S = B.s[blockIdx.y];
  const int lane = (int)threadIdx.x & 63;
  // XCD bands over the work groups, like k_recon's (gridDim.x is a multiple of 8): the cells of a band are
  // the pixels the same XCD's k_recon waves have just written, so what is still in that L2 is not
  // fetched again
  const int wgb = ((int)blockIdx.x & 7) * ((int)gridDim.x >> 3) + ((int)blockIdx.x >> 3);
  const int wbase = __builtin_amdgcn_readfirstlane(wgb * THIP_LF_WG + (int)(threadIdx.x & ~63u));
  const int ce0 = S.cell_end[0], ce1 = S.cell_end[1], ce2 = S.cell_end[2], L2 = S.flimit2;
  asm volatile("" ::"s"(ce0), "s"(ce1), "s"(ce2), "s"(L2));
  if (wbase >= ce2 || L2 == 0) return;
  const int pli = (wbase >= ce0 ? 1 : 0) + (wbase >= ce1 ? 1 : 0);
  const int rel = wbase - (pli == 0 ? 0 : (pli == 1 ? ce0 : ce1)) + lane;
  lf_wave(S, pli, rel, (S.pl[pli].nh + 1) * (S.pl[pli].nv + 1));
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

