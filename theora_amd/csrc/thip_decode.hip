// thip_decode.hip -- frame-scope reconstruction path for gfx950 (MI355X).
//
// Kernels (one 8x8 block per lane, 64-wide wavefronts, 256-thread workgroups):
//   k_recon       K1+K2: coded fragments  -> dequantised coeffs -> iDCT -> predictor -> pixels
//                        uncoded fragments -> 8x8 copy PREV->SELF          (state.c:959, fragment.c:37)
//   k_loopfilter  K3:    whole-frame in-loop deblocking, one 8x8 "corner cell" per lane
//                                                                         (state.c:1055-1105)
// K4 (UMV border fill, state.c:770-835) does not exist here: device frames are unpadded
// and motion-compensated reads clamp their coordinates, which is bit-identical.
//
// Host side (C ABI in include/theora_hip.h): stream state = three device frames + ring,
// batched launch over up to THIP_MAX_BATCH independent streams, pinned staging for the
// one-fragment-at-a-time vtable slots.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

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
// kernel arguments
// ---------------------------------------------------------------------------------------
struct PlaneK {
  int nh, nv;       // fragments across / down
  int fro;          // index of the plane's first fragment
  int stride;       // device pitch
  int off;          // byte offset of the plane in a frame
  int ncells_x;     // nh+1 (loop-filter cells across)
  float rcp_nh;     // 1/nh
  float rcp_cx;     // 1/(nh+1)
};

struct StreamK {
  const uint2 *cmds;
  const int4 *coeffs;
  const uint32_t *uncoded;
  uint8_t *self;
  const uint8_t *prev;
  const uint8_t *gold;
  uint8_t *coded_map;
  int ncoded, nuncoded;
  int nwg_coded;    // workgroups of k_recon that do reconstruction; the rest copy
  int flimit2;      // 2*flimit
  int cell_end0, cell_end1, cell_end2;  // cumulative loop-filter cell counts per plane
  int qpx, qpy;     // chroma axis decimated (quarter-pel chroma vectors)
  int lf_y0[3], lf_y1[3];  // fragment-row range to filter per plane
  int debug;        // ablation switches for profiling (THIP_DEBUG env), 0 in production
  PlaneK pl[3];
};

struct BatchK {
  StreamK s[THIP_MAX_BATCH];
};

// ---------------------------------------------------------------------------------------
// K1 + K2
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void locate_fragment(const StreamK &S, uint32_t fragi, int &pli, int &fx,
                                                int &fy, int &stride, int &off, int &W, int &H) {
  const bool c1 = fragi >= (uint32_t)S.pl[1].fro;
  const bool c2 = fragi >= (uint32_t)S.pl[2].fro;
  pli = (int)c1 + (int)c2;
  const int nh = c1 ? S.pl[1].nh : S.pl[0].nh;
  const int nv = c1 ? S.pl[1].nv : S.pl[0].nv;
  const float rcp = c1 ? S.pl[1].rcp_nh : S.pl[0].rcp_nh;
  const int fro = c2 ? S.pl[2].fro : (c1 ? S.pl[1].fro : 0);
  stride = c1 ? S.pl[1].stride : S.pl[0].stride;
  off = c2 ? S.pl[2].off : (c1 ? S.pl[1].off : S.pl[0].off);
  uint32_t q, r;
  divmod_u24(fragi - (uint32_t)fro, (uint32_t)nh, rcp, q, r);
  fx = (int)r;
  fy = (int)q;
  W = nh * 8;
  H = nv * 8;
}

__device__ __forceinline__ void recon_lane(const StreamK &S, uint32_t slot) {
  const bool active = slot < (uint32_t)S.ncoded;
  // 1. this lane's coefficients: four row pairs x two 16-byte halves; every load of the
  //    wave is 1 KiB contiguous.  P[j*8+c] = { x[2j][c], x[2j+1][c] }.
  const int4 *tp = S.coeffs + ((size_t)(slot >> 6) * 512 + (slot & 63));
  uint32_t P[32];
#pragma unroll
  for (int q = 0; q < 8; q++) {
    const int4 w = (S.debug & 8) ? make_int4((int)slot, q, 3, 4) : tp[q * 64];
    P[q * 4 + 0] = (uint32_t)w.x;
    P[q * 4 + 1] = (uint32_t)w.y;
    P[q * 4 + 2] = (uint32_t)w.z;
    P[q * 4 + 3] = (uint32_t)w.w;
  }
  uint2 cmd = make_uint2(0u, 0u);
  if (active) cmd = S.cmds[slot];
  const uint32_t fragi = cmd.x, flags = cmd.y;
  const int refi = (int)(flags & THIP_CMD_REFI_MASK);
  const bool dc_only = (flags & THIP_CMD_DC_ONLY) != 0;
  const int last_zzi = (int)((flags >> THIP_CMD_LAST_ZZI_SHIFT) & 0x7Fu);
  int pli, fx, fy, stride, off, W, H;
  locate_fragment(S, fragi, pli, fx, fy, stride, off, W, H);
  const int x0 = fx * 8, y0 = fy * 8;

  // 2. predictor (fragment.c:49-80): 128, one reference block, or the truncating
  //    average of two (state.c:986-998)
  uint2 pred[8];
#pragma unroll
  for (int r = 0; r < 8; r++) pred[r] = make_uint2(0x80808080u, 0x80808080u);
  if (active && refi != THIP_FRAME_SELF && !(S.debug & 2)) {
    const uint8_t *ref = (refi == THIP_FRAME_PREV ? S.prev : S.gold) + off;
    const int dx = (int)(int8_t)(flags >> THIP_CMD_MVX_SHIFT);
    const int dy = (int)(int8_t)(flags >> THIP_CMD_MVY_SHIFT);
    int mx, my, mx2, my2;
    mv_axis(dx, pli != 0 && S.qpx, mx, mx2);
    mv_axis(dy, pli != 0 && S.qpy, my, my2);
    const int sx = x0 + mx, sy = y0 + my;
    const bool two = (mx2 | my2) != 0;
    const bool inside = sx + min(mx2, 0) >= 0 && sx + max(mx2, 0) + 8 <= W &&
                        sy + min(my2, 0) >= 0 && sy + max(my2, 0) + 8 <= H;
    if (inside) {
      // Both samples of a row lie in the 9 bytes starting at xs, i.e. inside one 12-byte
      // window aligned down to 4: one DWORD-ALIGNED dwordx3 load per source row and byte
      // funnel shifts, instead of byte-unaligned loads (which the memory pipeline
      // serialises).  Vertical half-pel needs 9 source rows, not 16.
      const int xs = sx + min(mx2, 0);
      const int xw = xs & ~3;
      const int offA = sx - xw, offB = sx + mx2 - xw;   // 0..4
      const int ys = sy + min(my2, 0);
      const int ra = sy - ys, rb = sy + my2 - ys;        // first source row of each sample: 0 or 1
      const uint8_t *p1 = ref + (ptrdiff_t)ys * stride + xw;
      Row12 w[9];
#pragma unroll
      for (int r = 0; r < 8; r++) w[r] = load_row12(p1 + (ptrdiff_t)r * stride);
      w[8] = w[7];
      if (my2 != 0) w[8] = load_row12(p1 + (ptrdiff_t)8 * stride);
#pragma unroll
      for (int r = 0; r < 8; r++) {
        const Row12 wa = ra ? w[r + 1] : w[r];
        pred[r] = extract8(wa, offA);
        if (two) {
          const Row12 wb = rb ? w[r + 1] : w[r];
          const uint2 b = extract8(wb, offB);
          pred[r].x = avg4_trunc(pred[r].x, b.x);
          pred[r].y = avg4_trunc(pred[r].y, b.y);
        }
      }
    } else {
      // the block reaches into the reference's UMV border: clamp every coordinate
      // (== replicated padding, state.c:770-835)
#pragma unroll 1
      for (int r = 0; r < 8; r++) {
        const int ya = min(max(sy + r, 0), H - 1);
        const int yb = min(max(sy + my2 + r, 0), H - 1);
        uint32_t w[2] = {0u, 0u};
#pragma unroll
        for (int c = 0; c < 8; c++) {
          const int xa = min(max(sx + c, 0), W - 1);
          int v = ref[(ptrdiff_t)ya * stride + xa];
          if (two) {
            const int xb = min(max(sx + mx2 + c, 0), W - 1);
            v = (v + ref[(ptrdiff_t)yb * stride + xb]) >> 1;
          }
          w[c >> 2] |= (uint32_t)v << (8 * (c & 3));
        }
        // runtime row index: select into the register array without dynamic indexing
#pragma unroll
        for (int rr = 0; rr < 8; rr++)
          if (rr == r) pred[rr] = make_uint2(w[0], w[1]);
      }
    }
  }

  // 3. residual: DC-only shortcut (state.c:967-975) or inverse DCT (idct.c:301), packed
  //    two int16 per register
  uint32_t Y[32];
  const uint32_t dcp = (P[0] & 0xFFFFu) * 0x00010001u;   // {p, p}
  // wave-uniform choice of the cheapest transform that covers every lane
  const bool need_any = __any(active && !dc_only);
  if (S.debug & 1) {
#pragma unroll
    for (int i = 0; i < 32; i++) Y[i] = P[i];
  } else if (need_any) {
    pk_mask_by_last_zzi(P, last_zzi);
    const bool all_zz10 = !__any(active && !dc_only && last_zzi > 10);
    pk_idct8x8(P, Y, all_zz10);
  }
  if (dc_only || !need_any) {
#pragma unroll
    for (int i = 0; i < 32; i++) Y[i] = dcp;
  }

  // 4. reconstruct and store (8 B per lane per row)
  if (active && (S.debug & 4)) {
    // ablation: keep the values live with one store
    uint32_t acc = 0;
#pragma unroll
    for (int r = 0; r < 8; r++) {
      const uint2 o = pk_recon_row(as_pk(Y[r * 4 + 0]), as_pk(Y[r * 4 + 1]), as_pk(Y[r * 4 + 2]),
                                   as_pk(Y[r * 4 + 3]), pred[r]);
      acc ^= o.x ^ o.y;
    }
    if (acc == 0x12345678u) S.coded_map[fragi] = 2;
  } else if (active) {
    uint8_t *dst = S.self + off + (ptrdiff_t)y0 * stride + x0;
#pragma unroll
    for (int r = 0; r < 8; r++)
      store_row8(dst + (ptrdiff_t)r * stride,
                 pk_recon_row(as_pk(Y[r * 4 + 0]), as_pk(Y[r * 4 + 1]), as_pk(Y[r * 4 + 2]),
                              as_pk(Y[r * 4 + 3]), pred[r]));
    S.coded_map[fragi] = 1;
  }
}

__device__ __forceinline__ void copy_lane(const StreamK &S, uint32_t idx) {
  if (idx >= (uint32_t)S.nuncoded) return;
  const uint32_t fragi = S.uncoded[idx];
  int pli, fx, fy, stride, off, W, H;
  locate_fragment(S, fragi, pli, fx, fy, stride, off, W, H);
  const ptrdiff_t o = off + (ptrdiff_t)fy * 8 * stride + fx * 8;
  uint2 t[8];
#pragma unroll
  for (int r = 0; r < 8; r++) t[r] = *reinterpret_cast<const uint2 *>(S.prev + o + (ptrdiff_t)r * stride);
#pragma unroll
  for (int r = 0; r < 8; r++) store_row8(S.self + o + (ptrdiff_t)r * stride, t[r]);
  S.coded_map[fragi] = 0;
}

__global__ __launch_bounds__(256) void k_recon(const BatchK B) {
  const StreamK &S = B.s[blockIdx.y];
  const int b = (int)blockIdx.x;
  if (b < S.nwg_coded) {
    const uint32_t slot = (uint32_t)b * 256u + threadIdx.x;
    // whole waves past the end have no coefficient tile to read
    if ((slot & ~63u) >= (uint32_t)S.ncoded) return;
    recon_lane(S, slot);
  } else {
    copy_lane(S, (uint32_t)(b - S.nwg_coded) * 256u + threadIdx.x);
  }
}

// ---------------------------------------------------------------------------------------
// K3: in-loop deblocking
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
// and nothing else reads or writes those pixels, so cells are independent.  Inside a
// cell the operations run in the reference's order, which depends on which of the four
// fragments around the corner are coded (a=(k-1,m-1) b=(k,m-1) c=(k-1,m) d=(k,m)):
//   raster time of an op = (row, column, slot) of the fragment that triggers it, slots
//   left=0, previous-row=1, right=2, next-row=3.  Sorted, the eight candidates are
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

__device__ __forceinline__ void loopfilter_cell(uint8_t *plane, int stride, int nh, int nv,
                                                const uint8_t *cm, int k, int m, int L2, int fy0,
                                                int fy1) {
  const bool kin = k >= 1 && k <= nh - 1;  // a vertical edge exists at x=8k
  const bool min_ = m >= 1 && m <= nv - 1; // a horizontal edge exists at y=8m
  const bool a = k >= 1 && m >= 1 && cm[(m - 1) * nh + k - 1];
  const bool b = k <= nh - 1 && m >= 1 && cm[(m - 1) * nh + k];
  const bool c = k >= 1 && m <= nv - 1 && cm[m * nh + k - 1];
  const bool d = k <= nh - 1 && m <= nv - 1 && cm[m * nh + k];
  const bool rlo = (m - 1) >= fy0 && (m - 1) < fy1;  // ops triggered from fragment row m-1
  const bool rhi = m >= fy0 && m < fy1;              // ops triggered from fragment row m
  const bool t1 = kin && a && !b && rlo;
  const bool t2 = min_ && k >= 1 && a && !c && rlo;
  const bool t3 = kin && b && rlo;
  const bool t4 = min_ && k <= nh - 1 && b && !d && rlo;
  const bool t5 = min_ && k >= 1 && c && rhi;
  const bool t6 = kin && c && !d && rhi;
  const bool t7 = kin && d && rhi;
  const bool t8 = min_ && k <= nh - 1 && d && rhi;
  if (!(t1 | t2 | t3 | t4 | t5 | t6 | t7 | t8)) return;
  const bool lo_ok = k >= 1, hi_ok = k <= nh - 1;
  uint8_t *base = plane + (ptrdiff_t)(8 * m - 4) * stride + (8 * k - 4);
  const int H = nv * 8;
  int P[64];
#pragma unroll
  for (int r = 0; r < 8; r++) {
    const int y = 8 * m - 4 + r;
    uint32_t lo = 0, hi = 0;
    if (y >= 0 && y < H) {
      const uint32_t *p = reinterpret_cast<const uint32_t *>(base + (ptrdiff_t)r * stride);
      if (lo_ok) lo = p[0];
      if (hi_ok) hi = p[1];
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      P[r * 8 + q] = byte_of(lo, q);
      P[r * 8 + 4 + q] = byte_of(hi, q);
    }
  }
  if (__any(t1)) { if (t1) lf_vert(P, 0, L2); }
  if (__any(t2)) { if (t2) lf_horz(P, 0, L2); }
  if (__any(t3)) { if (t3) lf_vert(P, 0, L2); }
  if (__any(t4)) { if (t4) lf_horz(P, 4, L2); }
  if (__any(t5)) { if (t5) lf_horz(P, 0, L2); }
  if (__any(t6)) { if (t6) lf_vert(P, 4, L2); }
  if (__any(t7)) { if (t7) lf_vert(P, 4, L2); }
  if (__any(t8)) { if (t8) lf_horz(P, 4, L2); }
#pragma unroll
  for (int r = 0; r < 8; r++) {
    const int y = 8 * m - 4 + r;
    if (y >= 0 && y < H) {
      uint32_t *p = reinterpret_cast<uint32_t *>(base + (ptrdiff_t)r * stride);
      if (lo_ok) p[0] = pack4(P[r * 8 + 0], P[r * 8 + 1], P[r * 8 + 2], P[r * 8 + 3]);
      if (hi_ok) p[1] = pack4(P[r * 8 + 4], P[r * 8 + 5], P[r * 8 + 6], P[r * 8 + 7]);
    }
  }
}

__global__ __launch_bounds__(256) void k_loopfilter(const BatchK B) {
  const StreamK &S = B.s[blockIdx.y];
  const int cell = (int)(blockIdx.x * 256u + threadIdx.x);
  if (cell >= S.cell_end2 || S.flimit2 == 0) return;
  const bool c1 = cell >= S.cell_end0, c2 = cell >= S.cell_end1;
  const int pli = (int)c1 + (int)c2;
  const int rel = cell - (c2 ? S.cell_end1 : (c1 ? S.cell_end0 : 0));
  const int nh = c1 ? S.pl[1].nh : S.pl[0].nh;
  const int nv = c1 ? S.pl[1].nv : S.pl[0].nv;
  const int stride = c1 ? S.pl[1].stride : S.pl[0].stride;
  const int off = c2 ? S.pl[2].off : (c1 ? S.pl[1].off : S.pl[0].off);
  const int fro = c2 ? S.pl[2].fro : (c1 ? S.pl[1].fro : 0);
  const float rcp = c1 ? S.pl[1].rcp_cx : S.pl[0].rcp_cx;
  uint32_t m, k;
  divmod_u24((uint32_t)rel, (uint32_t)(nh + 1), rcp, m, k);
  const int y0 = pli == 0 ? S.lf_y0[0] : (pli == 1 ? S.lf_y0[1] : S.lf_y0[2]);
  const int y1 = pli == 0 ? S.lf_y1[0] : (pli == 1 ? S.lf_y1[1] : S.lf_y1[2]);
  loopfilter_cell(S.self + off, stride, nh, nv, S.coded_map + fro, (int)k, (int)m, S.flimit2, y0, y1);
}

// plane-level entry for the slot parity test (thip_loop_filter_plane)
__global__ __launch_bounds__(256) void k_loopfilter_plane(uint8_t *plane, int stride, int nh, int nv,
                                                         const uint8_t *coded, int L2, int fy0, int fy1,
                                                         float rcp_cx) {
  const int cell = (int)(blockIdx.x * 256u + threadIdx.x);
  if (cell >= (nh + 1) * (nv + 1)) return;
  uint32_t m, k;
  divmod_u24((uint32_t)cell, (uint32_t)(nh + 1), rcp_cx, m, k);
  loopfilter_cell(plane, stride, nh, nv, coded, (int)k, (int)m, L2, fy0, fy1);
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
struct thip_state {
  int frame_width, frame_height, pixel_fmt, hdec, vdec;
  thip_plane_geom geom[3];
  int64_t nfrags;
  size_t frame_bytes;
  uint8_t *frames[3];   // device
  uint8_t *coded_map;   // device, nfrags bytes
  int ref_idx[3];       // THIP_FRAME_* -> buffer index
  int last_decoded;     // buffer index of the most recently completed frame, -1 if none
  // host-enqueue staging (allocated on first use)
  int staging_ready;
  uint32_t *h_cmds, *d_cmds;
  int16_t *h_coeffs, *d_coeffs;
  uint32_t *h_uncoded, *d_uncoded;
  int32_t *frag_xy;     // host table: not needed by the device, kept for validation
  int enq_ncoded, enq_nuncoded, enq_frame_type, enq_flimit, enq_active;
  int enq_lf_y0[3], enq_lf_y1[3], enq_lf_any;
  int lf_y0[3], lf_y1[3], lf_rows_custom;
  int lane;             // library-owned HIP stream this state is bound to, -1 until first use
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

int ensure_lanes() {
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

void fill_plane_k(PlaneK &k, const thip_plane_geom &g) {
  k.nh = g.nhfrags;
  k.nv = g.nvfrags;
  k.fro = g.froffset;
  k.stride = g.stride;
  k.off = g.plane_off;
  k.ncells_x = g.nhfrags + 1;
  k.rcp_nh = 1.0f / (float)g.nhfrags;
  k.rcp_cx = 1.0f / (float)(g.nhfrags + 1);
}
}  // namespace

extern "C" {

const char *thip_version_string(void) { return "theora_hip 0.1 (gfx950; libtheora 1.2.0 fragment path)"; }

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
  int64_t fro = 0;
  size_t off = 0;
  for (int pli = 0; pli < 3; pli++) {
    thip_plane_geom &g = st->geom[pli];
    const int yh = frame_width >> 3, yv = frame_height >> 3;
    g.nhfrags = pli ? (yh + st->hdec) >> st->hdec : yh;   // state.c:443-449
    g.nvfrags = pli ? (yv + st->vdec) >> st->vdec : yv;
    const int64_t nf = (int64_t)g.nhfrags * g.nvfrags;
    if (nf >= (1 << 24) || fro + nf >= 0x7FFFFFFFll) {  // divmod_u24 range; 32-bit indices
      free(st);
      return THIP_EIMPL;
    }
    g.froffset = (int32_t)fro;
    g.nfrags = (int32_t)nf;
    fro += nf;
    g.width = g.nhfrags * 8;
    g.height = g.nvfrags * 8;
    g.stride = g.width;
    g.plane_off = (int32_t)off;
    off += ((size_t)g.stride * g.height + 255) & ~(size_t)255;
    if (off >= 0x7FFFFFFFull) {
      free(st);
      return THIP_EIMPL;
    }
  }
  st->nfrags = fro;
  st->frame_bytes = off;
  hipError_t err = hipSuccess;
  for (int b = 0; b < 3 && err == hipSuccess; b++) err = hipMalloc((void **)&st->frames[b], st->frame_bytes + 256);  /* +256: aligned 12-byte windows may read 3 bytes past a row end */
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
  if (st->h_cmds) (void)hipHostFree(st->h_cmds);
  if (st->h_coeffs) (void)hipHostFree(st->h_coeffs);
  if (st->h_uncoded) (void)hipHostFree(st->h_uncoded);
  if (st->d_cmds) (void)hipFree(st->d_cmds);
  if (st->d_coeffs) (void)hipFree(st->d_coeffs);
  if (st->d_uncoded) (void)hipFree(st->d_uncoded);
  free(st->frag_xy);
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
  HIP_TRY(hipDeviceSynchronize());
  for (int pli = 0; pli < 3; pli++) {
    const thip_plane_geom &g = st->geom[pli];
    if (!dst[pli] || dst_stride[pli] < g.width) return THIP_EINVAL;
    // the device keeps row 0 at the bottom of the picture; hand the frame back top-down
    // (decode.c:2988-2992 flips pointers instead)
    std::vector<uint8_t> tmp((size_t)g.width * g.height);
    HIP_TRY(hipMemcpy2D(tmp.data(), g.width, st->frames[st->last_decoded] + g.plane_off, g.stride, g.width,
                        g.height, hipMemcpyDeviceToHost));
    uint8_t *last = dst[pli] + (size_t)(g.height - 1) * dst_stride[pli];
    for (int y = 0; y < g.height; y++)
      memcpy(last - (size_t)y * dst_stride[pli], tmp.data() + (size_t)y * g.width, g.width);
  }
  return THIP_OK;
}

int thip_synchronize(void) {
  for (int i = 0; i < g_nlanes; i++) HIP_TRY(hipStreamSynchronize(g_lanes[i]));
  return THIP_OK;
}

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
  int max_wg = 0, max_cells_wg = 0, any_lf = 0, nlive = 0;
  int live_state[THIP_MAX_BATCH];
  for (int i = 0; i < n; i++) {
    thip_state *st = states[i];
    const thip_frame_desc &d = descs[i];
    if (!st) return THIP_EFAULT;
    if (d.ncoded < 0 || d.nuncoded < 0 || (int64_t)d.ncoded + d.nuncoded != st->nfrags) return THIP_EINVAL;
    if ((d.ncoded && (!d.cmds || !d.coeffs)) || (d.nuncoded && !d.uncoded)) return THIP_EFAULT;
    if (d.flimit < 0 || d.flimit > 127) return THIP_EINVAL;
    if (d.frame_type == THIP_INTRA_FRAME && d.nuncoded) return THIP_EINVAL;
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
    K.cmds = reinterpret_cast<const uint2 *>(d.cmds);
    K.coeffs = reinterpret_cast<const int4 *>(d.coeffs);
    K.uncoded = d.uncoded;
    K.self = st->frames[bufi];
    K.prev = st->ref_idx[THIP_FRAME_PREV] >= 0 ? st->frames[st->ref_idx[THIP_FRAME_PREV]] : st->frames[bufi];
    K.gold = st->ref_idx[THIP_FRAME_GOLD] >= 0 ? st->frames[st->ref_idx[THIP_FRAME_GOLD]] : st->frames[bufi];
    K.coded_map = st->coded_map;
    K.ncoded = d.ncoded;
    K.nuncoded = d.nuncoded;
    K.nwg_coded = (d.ncoded + 255) / 256;
    K.flimit2 = 2 * d.flimit;
    K.debug = g_debug;
    K.qpx = st->hdec;
    K.qpy = st->vdec;
    int cells = 0;
    for (int pli = 0; pli < 3; pli++) {
      fill_plane_k(K.pl[pli], st->geom[pli]);
      cells += (st->geom[pli].nhfrags + 1) * (st->geom[pli].nvfrags + 1);
      (pli == 0 ? K.cell_end0 : pli == 1 ? K.cell_end1 : K.cell_end2) = cells;
      K.lf_y0[pli] = st->lf_rows_custom ? st->lf_y0[pli] : 0;
      K.lf_y1[pli] = st->lf_rows_custom ? st->lf_y1[pli] : st->geom[pli].nvfrags;
    }
    const int wg = K.nwg_coded + (d.nuncoded + 255) / 256;
    if (wg > max_wg) max_wg = wg;
    if (d.flimit) {
      any_lf = 1;
      const int cwg = (cells + 255) / 256;
      if (cwg > max_cells_wg) max_cells_wg = cwg;
    }
    live_state[nlive++] = i;
  }
  if (!nlive) return THIP_OK;
  {
    ScopedTimer t(s, THIP_KERNEL_RECON);
    hipLaunchKernelGGL(k_recon, dim3(max_wg, nlive), dim3(256), 0, s, B);
  }
  if (any_lf) {
    ScopedTimer t(s, THIP_KERNEL_LOOPFILTER);
    hipLaunchKernelGGL(k_loopfilter, dim3(max_cells_wg, nlive), dim3(256), 0, s, B);
  }
  HIP_TRY(hipGetLastError());
  // decode.c:2947-2962
  for (int j = 0; j < nlive; j++) {
    thip_state *st = states[live_state[j]];
    const int self = st->ref_idx[THIP_FRAME_SELF];
    if (descs[live_state[j]].frame_type == THIP_INTRA_FRAME) st->ref_idx[THIP_FRAME_GOLD] = self;
    st->ref_idx[THIP_FRAME_PREV] = self;
    st->last_decoded = self;
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
  hipLaunchKernelGGL(k_loopfilter_plane, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, 0, plane, ystride, nhfrags, nvfrags,
                     coded, 2 * flimit, fragy0, fragy_end, 1.0f / (float)(nhfrags + 1));
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  return THIP_OK;
}

// ---------------------------------------------------------------------------------------
// host-enqueue form (the vtable slots)
// ---------------------------------------------------------------------------------------
static int ensure_staging(thip_state *st) {
  if (st->staging_ready) return THIP_OK;
  const size_t n = (size_t)st->nfrags;
  const size_t ntiles = (n + THIP_TILE_BLOCKS - 1) / THIP_TILE_BLOCKS;
  HIP_TRY(hipHostMalloc((void **)&st->h_cmds, n * 8, hipHostMallocDefault));
  HIP_TRY(hipHostMalloc((void **)&st->h_coeffs, ntiles * THIP_TILE_BYTES, hipHostMallocDefault));
  HIP_TRY(hipHostMalloc((void **)&st->h_uncoded, n * 4, hipHostMallocDefault));
  HIP_TRY(hipMalloc((void **)&st->d_cmds, n * 8));
  HIP_TRY(hipMalloc((void **)&st->d_coeffs, ntiles * THIP_TILE_BYTES));
  HIP_TRY(hipMalloc((void **)&st->d_uncoded, n * 4));
  st->staging_ready = 1;
  return THIP_OK;
}

int thip_frame_begin(thip_state *st, int frame_type) {
  if (!st) return THIP_EFAULT;
  if (frame_type != THIP_INTRA_FRAME && frame_type != THIP_INTER_FRAME) return THIP_EINVAL;
  int rc = ensure_staging(st);
  if (rc) return rc;
  // the previous frame's upload must have drained before the staging buffers are reused
  rc = thip_synchronize();
  if (rc) return rc;
  st->enq_ncoded = st->enq_nuncoded = 0;
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
  const int slot = st->enq_ncoded++;
  uint32_t flags = (uint32_t)refi | ((uint32_t)last_zzi << THIP_CMD_LAST_ZZI_SHIFT) |
                   ((uint32_t)(uint8_t)(mv & 0xFF) << THIP_CMD_MVX_SHIFT) |
                   ((uint32_t)(uint8_t)((mv >> 8) & 0xFF) << THIP_CMD_MVY_SHIFT);
  // tile layout (theora_hip.h): 16-byte group q = 2*j+h of the block holds, for columns
  // c = 4h..4h+3, the int16 pairs { x[2j][c], x[2j+1][c] }
  int16_t *tile = st->h_coeffs + (size_t)(slot >> 6) * (THIP_TILE_BYTES / 2) + (size_t)(slot & 63) * 8;
  if (last_zzi < 2) {
    // state.c:967-975: the only rounded dequantisation of the path
    flags |= THIP_CMD_DC_ONLY;
    const int16_t p = (int16_t)((dct_coeffs[0] * (int32_t)dc_quant + 15) >> 5);
    for (int q = 0; q < 8; q++) memset(tile + (size_t)q * 512, 0, 16);
    tile[0] = p;
  } else {
    dct_coeffs[0] = (int16_t)(dct_coeffs[0] * (int)dc_quant);   // state.c:978
    for (int j = 0; j < 4; j++)
      for (int h = 0; h < 2; h++) {
        int16_t *g = tile + (size_t)(2 * j + h) * 512;
        for (int cc = 0; cc < 4; cc++) {
          g[2 * cc] = dct_coeffs[(2 * j) * 8 + 4 * h + cc];
          g[2 * cc + 1] = dct_coeffs[(2 * j + 1) * 8 + 4 * h + cc];
        }
      }
  }
  memset(dct_coeffs, 0, 64 * sizeof(int16_t));   // idct.c:245,276,295
  st->h_cmds[2 * (size_t)slot] = (uint32_t)fragi;
  st->h_cmds[2 * (size_t)slot + 1] = flags;
  return THIP_OK;
}

int thip_frag_copy_list(thip_state *st, const ptrdiff_t *fragis, ptrdiff_t nfragis) {
  if (!st || (!fragis && nfragis)) return THIP_EFAULT;
  if (!st->enq_active || nfragis < 0 || (int64_t)st->enq_ncoded + st->enq_nuncoded + nfragis > st->nfrags)
    return THIP_EINVAL;
  for (ptrdiff_t k = 0; k < nfragis; k++) {
    if (fragis[k] < 0 || fragis[k] >= st->nfrags) return THIP_EINVAL;
    st->h_uncoded[st->enq_nuncoded++] = (uint32_t)fragis[k];
  }
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
  int rc = ensure_lanes();
  if (rc) return rc;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    if (st->lane < 0) st->lane = g_next_lane++ % g_nlanes;
  }
  hipStream_t g_stream = g_lanes[st->lane];
  const size_t ntiles = ((size_t)st->enq_ncoded + THIP_TILE_BLOCKS - 1) / THIP_TILE_BLOCKS;
  if (st->enq_ncoded) {
    HIP_TRY(hipMemcpyAsync(st->d_cmds, st->h_cmds, (size_t)st->enq_ncoded * 8, hipMemcpyHostToDevice, g_stream));
    HIP_TRY(hipMemcpyAsync(st->d_coeffs, st->h_coeffs, ntiles * THIP_TILE_BYTES, hipMemcpyHostToDevice, g_stream));
  }
  if (st->enq_nuncoded)
    HIP_TRY(hipMemcpyAsync(st->d_uncoded, st->h_uncoded, (size_t)st->enq_nuncoded * 4, hipMemcpyHostToDevice, g_stream));
  thip_frame_desc d;
  memset(&d, 0, sizeof(d));
  d.cmds = st->d_cmds;
  d.coeffs = st->d_coeffs;
  d.uncoded = st->d_uncoded;
  d.ncoded = st->enq_ncoded;
  d.nuncoded = st->enq_nuncoded;
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
  return res;
}

}  // extern "C"
