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
//                independent, and so are the pixels on an anti-diagonal of a block: one work group of eight waves
//                per group of 8x8 blocks (its 64x64 pixels plus a one-pixel halo in LDS; the halo clamped to the
//                plane is what the reference's border cases amount to), ONE launch for the three planes: a group
//                waits for its left and upper neighbour groups (see k_pp_dering), 15 block diagonals inside,
//                15 pixel steps per pass.
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
  // de-ringing as ONE launch: a group of blocks waits for its left and upper neighbour groups (k_pp_dering)
  uint32_t *done[3];    // per plane, per group: the serial number of the launch that finished it
  uint32_t serial;      // of this launch (never 0)
  uint32_t *fault;      // the state's pinned host word: set (bit 1) when a bounded wait runs out
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

constexpr int kPpGroup = 8;                       // blocks per group side
constexpr int kPpPitch = kPpGroup * 8 + 8;        // LDS row: 3 spare bytes, the left halo, 64 pixels (dword aligned), the right halo, 3 spare
constexpr int kPpRows = kPpGroup * 8 + 2;         // with the halo rows
constexpr int kPpX0 = 4;                          // column of the group's first pixel in an LDS row
#define THIP_DERING_THRESH1 384
#define THIP_DERING_THRESH2 (4 * THIP_DERING_THRESH1)
#define THIP_DERING_THRESH3 (5 * THIP_DERING_THRESH1)
#define THIP_DERING_THRESH4 (10 * THIP_DERING_THRESH1)

__device__ __forceinline__ void pp_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// oc_dering_block (decode.c:1795-1890) filters a block in place, pixel after pixel in raster order -- a pixel sees
// the NEW left and upper neighbour and the OLD right and lower one, its weights come from the block as it was before
// the call -- and the blocks of a plane one after the other in raster order, each up to three times in a row.  A
// block reads its four neighbours' border pixels only, so the blocks on an anti-diagonal of the block grid are
// independent, and so are the pixels on an anti-diagonal of a block.  One work group of eight waves takes a group of
// 8 x 8 blocks (its pixels and a one-pixel halo in LDS; the halo clamped into the plane is what the reference's border
// cases amount to) and walks the group's 15 block diagonals, wave w filtering the diagonal's block in block row w:
// all 64 lanes work out, for their pixel, what does not depend on the order (the weights and the terms of the old
// neighbours), then lanes 0..7 run the recurrence over the block's 15 pixel diagonals in registers, handing the new
// values from lane to lane (DPP).  Groups on an anti-diagonal of the group grid are independent, and a group needs its left
// and upper neighbour groups finished (their new border pixels; and they have read this group's old ones by then) and
// nothing else.  ONE launch: grid (groups on the longest diagonal, 3 planes, diagonals), so that the dispatcher hands the
// diagonals out in order -- blockIdx.z is the slowest index -- and a group's two predecessors were dispatched before it: it
// waits for their entries in K.done to carry this launch's serial number (bounded; they depend on nothing that comes later,
// so the wait cannot deadlock).  Producer and consumer may sit on different XCDs: a group writes its pixels through to
// memory (sc0 sc1) before it sets its entry, and reads the two halo lines its predecessors wrote past its own L2.
__device__ __forceinline__ uint32_t pp_load_through(const uint32_t *p) {
  uint32_t v;
  asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ uint32_t pp_load_byte_through(const uint8_t *p) {
  uint32_t v;
  asm volatile("global_load_ubyte %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void pp_store_through(uint32_t *p, uint32_t v) {
  asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
__global__ __launch_bounds__(64 * kPpGroup) void k_pp_dering(const PpK K) {
  const int d = (int)blockIdx.z;
  const int pli = (int)blockIdx.y;
  if (!K.dering[pli]) return;
  const PpPlaneK &P = K.p[pli];
  __shared__ __attribute__((aligned(16))) uint8_t s_reg[kPpRows * kPpPitch];
  __shared__ int2 s_pre[kPpGroup][64];             // per wave, per pixel: k0, w_lf | w_up << 16
  const int gnx = (P.nh + kPpGroup - 1) / kPpGroup, gny = (P.nv + kPpGroup - 1) / kPpGroup;
  if (d >= gnx + gny - 1) return;
  const int gx0 = max(0, d - (gny - 1));
  const int gx = gx0 + (int)blockIdx.x, gy = d - gx;
  if (gx >= gnx || gy < 0 || gy >= gny) return;
  const int tid = (int)threadIdx.x, lane = tid & 63, T = 64 * kPpGroup;
  const int b = __builtin_amdgcn_readfirstlane(tid >> 6);   // this wave's block row in the group
  const int X0 = gx * kPpGroup * 8, Y0 = gy * kPpGroup * 8;
  uint32_t *const done = K.done[pli];
  // the predecessors: lanes 0 and 1 of the first wave look until both entries carry this launch's number
  if (tid < 64) {
    const uint32_t *src = nullptr;
    if (tid == 0 && gx > 0) src = done + gy * gnx + gx - 1;
    if (tid == 1 && gy > 0) src = done + (gy - 1) * gnx + gx;
    bool ok = src == nullptr;
    for (int spins = 0; spins < (1 << 22); spins++) {
      if (!ok) ok = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == K.serial;
      if (__all(ok)) break;
      __builtin_amdgcn_s_sleep(2);
    }
    if (!ok && K.fault) __hip_atomic_store(K.fault + 9, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (a word of its own: thip_decode.hip, check_fault)
  }
  __syncthreads();
  // the region: rows Y0-1 .. Y0+64 clamped into the plane; the 64 pixels as dwords (plane widths are multiples of 8),
  // the two halo columns as bytes, clamped.  The upper halo row and the left halo column are this launch's work (when
  // those groups exist): read past L2.
  for (int i = tid; i < kPpRows * 16; i += T) {
    const int ry = i >> 4, c4 = (i & 15) * 4;
    const int y = min(max(Y0 - 1 + ry, 0), P.height - 1), x = X0 + c4;
    if (x < P.width) {
      const uint32_t *src = reinterpret_cast<const uint32_t *>(P.dst + (size_t)y * P.stride + x);
      *reinterpret_cast<uint32_t *>(s_reg + ry * kPpPitch + kPpX0 + c4) = (ry == 0 && gy > 0) ? pp_load_through(src) : *src;
    }
  }
  for (int i = tid; i < kPpRows * 2; i += T) {
    const int ry = i >> 1, right = i & 1;
    const int y = min(max(Y0 - 1 + ry, 0), P.height - 1);
    const int x = right ? min(X0 + kPpGroup * 8, P.width - 1) : max(X0 - 1, 0);
    const uint8_t *src = P.dst + (size_t)y * P.stride + x;
    // (left column: the left group's; its row -1 the upper-left group's, finished before the left one could start)
    s_reg[ry * kPpPitch + (right ? kPpX0 + kPpGroup * 8 : kPpX0 - 1)] = (!right && gx > 0) ? (uint8_t)pp_load_byte_through(src) : *src;
  }
  __syncthreads();
  const int strong_plane = K.strong[pli];
  const int sthresh = pli ? THIP_DERING_THRESH4 : THIP_DERING_THRESH3;
  const int i8 = lane & 7;
  const int qx = lane & 7, qy = lane >> 3;         // the pixel this lane prepares
  int any_touched = 0;
  for (int D = 0; D < 2 * kPpGroup - 1; D++) {
    const int lby = b, lbx = D - b;
    const int fx = gx * kPpGroup + lbx, fy = gy * kPpGroup + lby;
    const bool blk = lbx >= 0 && lbx < kPpGroup && fx < P.nh && fy < P.nv;   // (wave-uniform)
    int passes = 0, strong = 1, dcs = 0, shm = 0;
    bool eL = false, eR = false, eT = false, eB = false;
    if (blk) {
      const int fi = fy * P.nh + fx;
      const int var = P.variances[fi], qi = P.frag_qi[fi];
      dcs = K.dc_scale[qi];
      shm = K.sharp_mod[qi];
      eL = fx == 0;
      eR = fx == P.nh - 1;
      eT = fy == 0;
      eB = fy == P.nv - 1;
      if (strong_plane && var > sthresh) {         // decode.c:1926-1950
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
    }
    uint8_t *const b0 = s_reg + (1 + lby * 8) * kPpPitch + kPpX0 + lbx * 8;   // the block's first pixel
    const int mod_hi = min(3 * dcs, strong ? 32 : 24), sh = strong ? 0 : 1;
    if (passes) any_touched = 1;
    for (int pss = 0; pss < passes; pss++) {       // (the waves of a diagonal do not read each other's blocks: no group barrier inside)
      // the clamped halo of a block on the plane's edge mirrors the block's own border pixels AS THEY ARE NOW
      // (the reference reads the pixel itself there): bring it up to date before every pass
      if (lane < 8) {
        if (eL) b0[lane * kPpPitch - 1] = b0[lane * kPpPitch];
        if (eR) b0[lane * kPpPitch + 8] = b0[lane * kPpPitch + 7];
        if (eT) b0[-kPpPitch + lane] = b0[lane];
        if (eB) b0[8 * kPpPitch + lane] = b0[7 * kPpPitch + lane];
      }
      pp_wave_sync();
      // What does not depend on the order, every lane for its pixel, from the block as it is before the pass:
      // new = (k0 + w_lf * new_left + w_up * new_up) >> 7 with k0 = a * me + 64 + w_dn * dn + w_rt * rt (the lower and
      // right neighbours are still the old ones when a pixel's turn comes).  A left / upper neighbour outside the block
      // does not change during the pass either: its term goes into k0 and its weight to zero.
      {
        const uint8_t *c = b0 + qy * kPpPitch + qx;
        const int me = c[0], up = c[-kPpPitch], dn = c[kPpPitch], lf = c[-1], rt = c[1];
        auto modf = [&](int dd) {
          const int mod = 32 + dcs - (pp_abs(dd) << sh);
          return mod < -64 ? shm : min(max(mod, 0), mod_hi);
        };
        const int w_up = modf(me - up), w_dn = modf(dn - me), w_lf = modf(me - lf), w_rt = modf(rt - me);
        const int k0 = (128 - w_up - w_dn - w_lf - w_rt) * me + 64 + w_dn * dn + w_rt * rt + (qx == 0 ? w_lf * lf : 0) + (qy == 0 ? w_up * up : 0);
        s_pre[b][lane] = make_int2(k0, ((qx == 0 ? 0 : w_lf) & 0xFFFF) | (qy == 0 ? 0 : w_up) << 16);
      }
      pp_wave_sync();   // every lane has read the block before anybody writes
      // The pixel diagonals, lanes 0..7, in registers: on the way down (t <= 7) lane i holds column i -- the new upper
      // neighbour is its own previous result, the new left one its left lane's; on the way out (t >= 8) lane i holds row
      // 7 - i -- the new left neighbour is its own previous result, the new upper one the next lane's.
      int2 pre[15];
#pragma unroll
      for (int t = 0; t < 15; t++) {
        const int px = (t > 7 ? t - 7 : 0) + i8, py = t - px;
        pre[t] = s_pre[b][(min(max(py, 0), 7)) * 8 + min(px, 7)];
      }
      int prev = 0;
#pragma unroll
      for (int t = 0; t < 15; t++) {
        const int px = (t > 7 ? t - 7 : 0) + i8, py = t - px;
        const int other = t <= 7 ? __builtin_amdgcn_update_dpp(0, prev, 0x111, 0xF, 0xF, true)    // row_shr:1: lane i reads lane i-1
                                 : __builtin_amdgcn_update_dpp(0, prev, 0x101, 0xF, 0xF, true);   // row_shl:1: lane i reads lane i+1
        const int nl = t <= 7 ? other : prev, nu = t <= 7 ? prev : other;
        if (lane < 8 && px <= 7 && py >= 0) {
          prev = min(max((pre[t].x + (int)(int16_t)(pre[t].y & 0xFFFF) * nl + (pre[t].y >> 16) * nu) >> 7, 0), 255);
          b0[py * kPpPitch + px] = (uint8_t)prev;
        }
      }
      pp_wave_sync();
    }
    __syncthreads();   // the diagonal is done before the next one reads its borders
  }
  if (__syncthreads_or(any_touched)) {
    for (int i = tid; i < kPpGroup * 8 * 16; i += T) {
      const int ry = i >> 4, c4 = (i & 15) * 4;
      const int x = X0 + c4, y = Y0 + ry;
      if (x < P.width && y < P.height)
        pp_store_through(reinterpret_cast<uint32_t *>(P.dst + (size_t)y * P.stride + x),
                         *reinterpret_cast<const uint32_t *>(s_reg + (1 + ry) * kPpPitch + kPpX0 + c4));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pixels have arrived
  }
  __syncthreads();                                      // ... and every wave's
  if (tid == 0) __hip_atomic_store(done + gy * gnx + gx, K.serial, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
