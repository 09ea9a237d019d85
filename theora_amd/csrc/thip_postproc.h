// thip_postproc.h -- out-of-loop post-processing on the device (decode.c:1608-1957): de-blocking and
// de-ringing of the finished frame into a separate picture, th_decode_ctl's TH_DECCTL_SET_PPLEVEL.
// Included by thip_decode.hip only.
//
// The reference drives both filters MCU by MCU with one-row delays (decode.c:2895-2911); run once over all
// fragment rows of a plane they give the same picture and the same variances (tests/test_oracle.py::
// test_postprocessing_is_independent_of_the_mcu_chunking), which is the form used here:
//   k_pp_hedge   every horizontal block edge (decode.c:1610-1661): reads the decoded frame, writes the
//                post-processed one -- no dependency at all, one thread per four pixel columns of an edge;
//                also copies the first and last four rows (decode.c:1734-1738, :1768-1772)
//   k_pp_vedge   every vertical block edge (decode.c:1663-1694), in place on the result of k_pp_hedge: the
//                filter of edge x reads a pixel the filter of edge x-8 has just written, so a pixel row is a
//                chain -- one thread per pixel row walks it left to right, the previous edge's last output
//                carried in a register and the next eight bytes requested a step ahead; rows are independent
//   k_pp_dering  oc_dering_block (decode.c:1788-1890) works IN PLACE, block after block in raster order and,
//                inside a block, pixel after pixel: a pixel sees the new value of its left and upper
//                neighbours and the old value of its right and lower ones.  Blocks on an anti-diagonal are
//                independent, and so are the pixels on an anti-diagonal of a block: one wave per group of 4x4
//                blocks (its 32x32 pixels plus a one-pixel halo in LDS; the halo clamped to the plane is what
//                the reference's border cases amount to), one launch per anti-diagonal of groups, 16 blocks in
//                raster order inside, 15 pixel steps per pass.
// Variances (decode.c:1636-1637, :1679-1680) are integer sums: any order of addition gives the reference's.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct PpPlaneK {
  const uint8_t *src;   // decoded frame, this plane
  uint8_t *dst;         // post-processed frame, this plane
  int stride, width, height, nh, nv;
  int *variances;       // per fragment of the plane
  const uint8_t *dc_qis, *frag_qi;
};
struct PpK {
  PpPlaneK p[3];
  int active[3];        // plane is post-processed at this level
  int dering[3], strong[3];
  int dc_scale[64], sharp_mod[64];
};

__device__ __forceinline__ int pp_abs(int v) { return v < 0 ? -v : v; }
struct __attribute__((aligned(4))) PpPix8 {
  uint32_t x, y;
};

// The two outputs' worth of one edge: r[0..9] across the edge -> filtered r[1..8] (decode.c:1639-1653).
__device__ __forceinline__ bool pp_edge(const int r[10], int qstep, int flimit, int o[8], int &v0, int &v1) {
  int sum0 = 0, sum1 = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    sum0 += pp_abs(r[k + 1] - r[k]);
    sum1 += pp_abs(r[k + 5] - r[k + 6]);
  }
  v0 = min(255, sum0);
  v1 = min(255, sum1);
  const bool f = sum0 < flimit && sum1 < flimit && r[5] - r[4] < qstep && r[4] - r[5] < qstep;
  o[0] = (r[0] * 3 + r[1] * 2 + r[2] + r[3] + r[4] + 4) >> 3;
  o[1] = (r[0] * 2 + r[1] + r[2] * 2 + r[3] + r[4] + r[5] + 4) >> 3;
#pragma unroll
  for (int k = 0; k < 4; k++) o[2 + k] = (r[k] + r[k + 1] + r[k + 2] + r[k + 3] * 2 + r[k + 4] + r[k + 5] + r[k + 6] + 4) >> 3;
  o[6] = (r[4] + r[5] + r[6] + r[7] * 2 + r[8] + r[9] * 2 + 4) >> 3;
  o[7] = (r[5] + r[6] + r[7] + r[8] * 2 + r[9] * 3 + 4) >> 3;
  return f;
}

// grid: (ceil(width / 256), nv + 1, 3); thread = four pixel columns; blockIdx.y = 0: rows 0..3 copied,
// 1..nv-1: the edge between fragment rows y-1 and y, nv: the last four rows copied
__global__ __launch_bounds__(64) void k_pp_hedge(const PpK K) {
  const PpPlaneK &P = K.p[blockIdx.z];
  if (!K.active[blockIdx.z]) return;
  const int x = ((int)blockIdx.x * 64 + (int)threadIdx.x) * 4, k = (int)blockIdx.y;
  if (x >= P.width || k > P.nv) return;
  if (k == 0 || k == P.nv) {
    const int y0 = k == 0 ? 0 : P.height - 4;
#pragma unroll
    for (int r = 0; r < 4; r++)
      *reinterpret_cast<uint32_t *>(P.dst + (size_t)(y0 + r) * P.stride + x) =
          *reinterpret_cast<const uint32_t *>(P.src + (size_t)(y0 + r) * P.stride + x);
    return;
  }
  const int fx = x >> 3;
  const int qstep = K.dc_scale[P.dc_qis[(k - 1) * P.nh + fx]], flimit = (qstep * 3) >> 2;
  uint32_t w[10];
#pragma unroll
  for (int r = 0; r < 10; r++) w[r] = *reinterpret_cast<const uint32_t *>(P.src + (size_t)(8 * k - 5 + r) * P.stride + x);
  uint32_t out[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int s0 = 0, s1 = 0;
#pragma unroll
  for (int c = 0; c < 4; c++) {
    int r[10], o[8], v0, v1;
#pragma unroll
    for (int i = 0; i < 10; i++) r[i] = (int)((w[i] >> (8 * c)) & 0xFFu);
    const bool f = pp_edge(r, qstep, flimit, o, v0, v1);
    s0 += v0;
    s1 += v1;
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] |= (uint32_t)((f ? o[i] : r[1 + i]) & 0xFF) << (8 * c);
  }
#pragma unroll
  for (int i = 0; i < 8; i++) *reinterpret_cast<uint32_t *>(P.dst + (size_t)(8 * k - 4 + i) * P.stride + x) = out[i];
  atomicAdd(P.variances + (k - 1) * P.nh + fx, s0);
  atomicAdd(P.variances + k * P.nh + fx, s1);
}

// grid: (ceil(height / 64), 1, 3); thread = one pixel row, lanes 8j..8j+7 = the rows of one fragment row
__global__ __launch_bounds__(64) void k_pp_vedge(const PpK K) {
  const PpPlaneK &P = K.p[blockIdx.z];
  if (!K.active[blockIdx.z]) return;
  const int y = (int)blockIdx.x * 64 + (int)threadIdx.x;
  const bool live = y < P.height;
  const int yy = live ? y : P.height - 1, fy = yy >> 3;
  uint8_t *row = P.dst + (size_t)yy * P.stride;
  if (P.width < 16) return;
  // bytes x-4..x+3 around edge x; the first edge is x = 8
  PpPix8 cur = *reinterpret_cast<const PpPix8 *>(row + 4);
  int carry = (int)row[3];
  for (int x = 8; x < P.width; x += 8) {
    PpPix8 nxt;   // bytes x+4 .. x+11: the first of them is this edge's last tap, all of them the next edge's window
    nxt.x = *reinterpret_cast<const uint32_t *>(row + x + 4);
    nxt.y = x + 8 < P.width ? *reinterpret_cast<const uint32_t *>(row + x + 8) : 0u;
    const int qstep = K.dc_scale[P.dc_qis[fy * P.nh + (x >> 3)]], flimit = (qstep * 3) >> 2;
    int r[10], o[8], v0, v1;
    r[0] = carry;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      r[1 + i] = (int)((cur.x >> (8 * i)) & 0xFFu);
      r[5 + i] = (int)((cur.y >> (8 * i)) & 0xFFu);
    }
    r[9] = (int)(nxt.x & 0xFFu);
    const bool f = pp_edge(r, qstep, flimit, o, v0, v1);
    if (f && live) {
      PpPix8 wv;
      wv.x = (uint32_t)o[0] | (uint32_t)o[1] << 8 | (uint32_t)o[2] << 16 | (uint32_t)o[3] << 24;
      wv.y = (uint32_t)o[4] | (uint32_t)o[5] << 8 | (uint32_t)o[6] << 16 | (uint32_t)o[7] << 24;
      *reinterpret_cast<PpPix8 *>(row + x - 4) = wv;
    }
    carry = f ? o[7] : r[8];
    // the eight rows of a fragment row add up, one lane writes (this kernel is the only writer now)
    if (!live) v0 = v1 = 0;
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) {
      v0 += __shfl_xor(v0, m);
      v1 += __shfl_xor(v1, m);
    }
    if (live && (threadIdx.x & 7) == 0) {
      P.variances[fy * P.nh + (x >> 3) - 1] += v0;
      P.variances[fy * P.nh + (x >> 3)] += v1;
    }
    cur = nxt;
  }
}

constexpr int kPpGroup = 4;                       // blocks per group side
constexpr int kPpRegion = kPpGroup * 8 + 2;       // pixels per side with the halo
#define THIP_DERING_THRESH1 384
#define THIP_DERING_THRESH2 (4 * THIP_DERING_THRESH1)
#define THIP_DERING_THRESH3 (5 * THIP_DERING_THRESH1)
#define THIP_DERING_THRESH4 (10 * THIP_DERING_THRESH1)

// one pass of oc_dering_block on block (lbx, lby) of the region in LDS; lane = pixel (px, py)
__device__ __forceinline__ void pp_dering_pass(uint8_t *reg, int lbx, int lby, int dc_scale, int sharp_mod, int strong, int lane) {
  const int px = lane & 7, py = lane >> 3;
  uint8_t *c = reg + (1 + lby * 8 + py) * kPpRegion + (1 + lbx * 8 + px);
  const int mod_hi = min(3 * dc_scale, strong ? 32 : 24), sh = strong ? 0 : 1;
  const int me = c[0], up = c[-kPpRegion], dn = c[kPpRegion], lf = c[-1], rt = c[1];
  auto modf = [&](int d) {
    const int mod = 32 + dc_scale - (pp_abs(d) << sh);
    return mod < -64 ? sharp_mod : min(max(mod, 0), mod_hi);
  };
  const int w_up = modf(me - up), w_dn = modf(dn - me), w_lf = modf(me - lf), w_rt = modf(rt - me);   // from the block as it is before the pass
  const int a = 128 - w_up - w_dn - w_lf - w_rt;
  // pixel after pixel in raster order: new left and upper neighbour, old right and lower one -> anti-diagonals
  for (int t = 0; t < 15; t++) {
    if (px + py == t) {
      const int nl = c[-1], nu = c[-kPpRegion];          // already new when they lie inside the block
      const int b = 64 + w_lf * nl + w_up * nu + w_dn * dn + w_rt * rt;
      c[0] = (uint8_t)min(max((a * me + b) >> 7, 0), 255);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

// grid: (groups on anti-diagonal d of the plane's group grid); one wave per group
__global__ __launch_bounds__(64) void k_pp_dering(const PpK K, int pli, int d) {
  const PpPlaneK &P = K.p[pli];
  __shared__ uint8_t s_reg[kPpRegion * kPpRegion + 4];
  const int gnx = (P.nh + kPpGroup - 1) / kPpGroup, gny = (P.nv + kPpGroup - 1) / kPpGroup;
  const int gx0 = max(0, d - (gny - 1));
  const int gx = gx0 + (int)blockIdx.x, gy = d - gx;
  if (gx >= gnx || gy < 0 || gy >= gny) return;
  const int lane = (int)threadIdx.x;
  const int X0 = gx * kPpGroup * 8, Y0 = gy * kPpGroup * 8;
  // region with halo, coordinates clamped into the plane (= the reference's border cases, decode.c:1803-1826)
  for (int i = lane; i < kPpRegion * kPpRegion; i += 64) {
    const int ry = i / kPpRegion, rx = i - ry * kPpRegion;
    const int x = min(max(X0 - 1 + rx, 0), P.width - 1), y = min(max(Y0 - 1 + ry, 0), P.height - 1);
    s_reg[i] = P.dst[(size_t)y * P.stride + x];
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  const int strong_plane = K.strong[pli];
  const int sthresh = pli ? THIP_DERING_THRESH4 : THIP_DERING_THRESH3;
  bool touched = false;
  for (int lby = 0; lby < kPpGroup; lby++)
    for (int lbx = 0; lbx < kPpGroup; lbx++) {
      const int fx = gx * kPpGroup + lbx, fy = gy * kPpGroup + lby;
      if (fx >= P.nh || fy >= P.nv) continue;                                   // (wave-uniform)
      const int fi = fy * P.nh + fx;
      const int var = P.variances[fi], qi = P.frag_qi[fi];
      const int dcs = K.dc_scale[qi], shm = K.sharp_mod[qi];
      // a block at the plane's edge whose halo is a copy of its own pixels: refresh the halo from the block after each
      // pass?  Not needed: the clamped halo is only read through differences with / weights on the same pixel
      // -- but its VALUE is used (b += w * neighbour), and the neighbour is the block's own, possibly new, pixel.
      const bool eL = fx == 0, eR = fx == P.nh - 1, eT = fy == 0, eB = fy == P.nv - 1;
      int passes = 0, strong = 1;
      if (strong_plane && var > sthresh) {
        passes = 1;
        if (pli || (!eL && P.variances[fi - 1] > THIP_DERING_THRESH4) || (!eR && P.variances[fi + 1] > THIP_DERING_THRESH4) ||
            (!eT && P.variances[fi - P.nh] > THIP_DERING_THRESH4) || (!eB && P.variances[fi + P.nh] > THIP_DERING_THRESH4))
          passes = 3;
      } else if (var > THIP_DERING_THRESH2) {
        passes = 1;
      } else if (var > THIP_DERING_THRESH1) {
        passes = 1;
        strong = 0;
      }
      for (int pss = 0; pss < passes; pss++) {
        // the clamped halo of a block on the plane's edge mirrors the block's own border pixels AS THEY ARE NOW
        // (the reference reads the pixel itself there): bring it up to date before every pass
        if (eL | eR | eT | eB) {
          if (lane < 8) {
            uint8_t *b0 = s_reg + (1 + lby * 8) * kPpRegion + (1 + lbx * 8);
            if (eL) b0[lane * kPpRegion - 1] = b0[lane * kPpRegion];
            if (eR) b0[lane * kPpRegion + 8] = b0[lane * kPpRegion + 7];
            if (eT) b0[-kPpRegion + lane] = b0[lane];
            if (eB) b0[8 * kPpRegion + lane] = b0[7 * kPpRegion + lane];
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_wave_barrier();
        }
        pp_dering_pass(s_reg, lbx, lby, dcs, shm, strong, lane);
        touched = true;
      }
    }
  if (!touched) return;
  for (int i = lane; i < kPpGroup * 8 * kPpGroup * 8; i += 64) {
    const int ry = i / (kPpGroup * 8), rx = i - ry * (kPpGroup * 8);
    const int x = X0 + rx, y = Y0 + ry;
    if (x < P.width && y < P.height) P.dst[(size_t)y * P.stride + x] = s_reg[(1 + ry) * kPpRegion + 1 + rx];
  }
}
