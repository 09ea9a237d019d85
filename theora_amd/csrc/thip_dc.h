// thip_dc.h -- oc_dec_dc_unpredict_mcu_plane (decode.c:1392-1500) for whole planes on the device: k_dc_prepare + k_dc_wave.
// Included by thip_decode.hip behind thip_kernels.h (DcPlaneK, dc_flag).
//
// The predictor of a coded fragment reads the FINAL DC of its left, upper-left, upper and upper-right neighbours --
// those that are coded and predicted from the same reference frame -- and falls back to pred_last[refi], the DC of the last
// fragment with that reference frame in raster order (decode.c:1448).  Fragment (x, y) can therefore be done once
// (x-1, y) and (x+1, y-1) are: a wavefront of slope 2 over the rows, and a chain wherever pred_last reaches back.
// Everything about a fragment's predictor except the neighbours' VALUES is known from the coded flags and reference
// indices alone, so it is worked out beforehand, for all fragments at once:
//   k_dc_prepare  one thread per fragment: which neighbours count (decode.c:1450-1485's mask), hence the four weights
//                 and the shift of the truncating division -- 16 bytes per fragment together with its token value;
//                 whether the fragment falls back to pred_last and then to a value of its own row or of an earlier
//                 one; whether it is the last fragment of its reference in its row (the one later rows may ask for).
//   k_dc_wave     ONE wave per plane does the serial part.  Lane l walks rows l, l + 64, l + 128, ...; the 64 rows in flight
//                 keep their whole DC rows in LDS (row y in slot y mod 64), so nothing a row ever needs from the row above can
//                 be gone when it gets there, however far it fell behind.  No barrier, no trip to memory inside a step:
//                 the lanes of a wave run in lock step and a wave's LDS operations are performed in order, so a lane that
//                 reads another row's progress word and then its DC sees at least what the progress word promised.  A step
//                 is ONE batch of LDS reads (progress of the row above, the fragment's entry, three DCs of the row above),
//                 two v_dot2_i32_i16 for the weighted sum, the reference's division towards zero and its outlier clamp,
//                 three LDS writes.  Entries come in and DCs go out through per-lane windows in LDS that ALL lanes refill
//                 and drain together every eighth step: the wave has one instruction stream, so a load in flight for one
//                 lane is a wait for all 64 at the next s_waitcnt; batching puts that wait eight steps behind the
//                 accesses it covers, where it costs nothing.
// pred_last across rows is resolved to the fragment it names: a row publishes the DC of its last fragment of each
// reference when it passes it (lastval / lastrdy), srcrow[y][r] = the nearest row at or above y that has one, and a fragment
// without a usable neighbour and without a predecessor in its own row waits for exactly that -- not for the row above to
// end.  A picture whose rows chain through pred_last is as serial here as anywhere.
#pragma once

constexpr int kDcwRows = 64;              // rows in flight = lanes
constexpr int kDcwWin = 16;               // per-lane entry window
constexpr int kDcwBatch = 8;              // steps between two refills / drains
constexpr int kDcwLdsMax = 160 * 1024 - 512;
// entry word 0: bits 0-15 token value, 16-17 code (0 = not coded, 1 + refi), 18 needs an earlier row's pred_last,
// 19 takes its own row's last value, 20 last fragment of its reference in its row, 21 outlier clamp, 24-27 shift
constexpr uint32_t kDceSrc = 1u << 18, kDceOwn = 1u << 19, kDceLast = 1u << 20, kDceClamp = 1u << 21;

struct DcwLayout {
  int nhp;                                // row pitch of the DC rows, int16 units
  int o_srcrow, o_lastval, o_lastrdy, o_prog, o_win, o_outq, bytes;
};
__host__ __device__ inline DcwLayout dcw_layout(int nh, int nv) {
  DcwLayout L;
  L.nhp = (nh + 2) | 1;                   // (odd: the rows' first elements fall into different banks)
  int o = kDcwRows * L.nhp * 2;
  o = (o + 15) & ~15;
  L.o_srcrow = o;
  o += nv * 4 * 2;                        // int16 [nv][4]
  L.o_lastval = o;
  o += nv * 4 * 2;
  L.o_lastrdy = o;
  o += nv * 4;                            // uint8 [nv][4]
  o = (o + 15) & ~15;
  L.o_prog = o;
  o += kDcwRows * 4;
  L.o_win = o;
  o += kDcwRows * kDcwWin * 16;
  L.o_outq = o;
  o += kDcwRows * kDcwBatch * 8;          // {fragment index, value} per entry
  L.bytes = o;
  return L;
}
__host__ __device__ inline bool dcw_fits(int nh, int nv) {
  return nh >= 1 && nv >= 1 && nh <= 1024 && nv < 32768 && dcw_layout(nh, nv).bytes <= kDcwLdsMax;
}

// weights {left, upper-left, upper, upper-right} and the shift of the truncating division, by neighbour mask (decode.c:1450-1485)
__device__ __forceinline__ void dcw_weights(int mask, int &wl, int &wul, int &wu, int &wur, int &sh) {
  wl = wul = wu = wur = sh = 0;
  switch (mask) {
    case 1: case 3: wl = 1; break;
    case 2: wul = 1; break;
    case 4: case 6: case 12: wu = 1; break;
    case 5: wl = 1; wu = 1; sh = 1; break;
    case 8: wur = 1; break;
    case 9: case 11: case 13: wl = 75; wur = 53; sh = 7; break;
    case 10: wul = 1; wur = 1; sh = 1; break;
    case 14: wul = 3; wu = 10; wur = 3; sh = 4; break;
    case 7: case 15: wl = 29; wul = -26; wu = 29; sh = 5; break;
    default: break;
  }
}

// One work group per fragment row (blockIdx.x = row, .y = plane, .z = stream), one thread per fragment.
__global__ __launch_bounds__(1024) void k_dc_prepare(const DcBatchK B) {
  const DcPlaneK &P = B.p[blockIdx.z][blockIdx.y];
  const int nh = P.nh, nv = P.nv, y = (int)blockIdx.x, x = (int)threadIdx.x;
  if (y >= nv) return;
  __shared__ uint32_t s_has[16];           // per wave: bit r = the wave has a fragment of reference r
  const int lane = x & 63, wave = x >> 6, nw = ((int)blockDim.x + 63) >> 6;
  const bool in = x < nh;
  uint32_t f = 0, fl = 0, ful = 0, fu = 0, fur = 0;   // codes: 0 = does not count, 1 + refi
  if (in) {
    const uint32_t w = dc_flag(P, x, y);
    f = (w & 1u) ? 1u + (w >> 1) : 0u;
    if (f) {
      uint32_t v;
      if (x > 0) { v = dc_flag(P, x - 1, y); fl = (v & 1u) ? 1u + (v >> 1) : 0u; }
      if (y > 0) {
        v = dc_flag(P, x, y - 1); fu = (v & 1u) ? 1u + (v >> 1) : 0u;
        if (x > 0) { v = dc_flag(P, x - 1, y - 1); ful = (v & 1u) ? 1u + (v >> 1) : 0u; }
        if (x + 1 < nh) { v = dc_flag(P, x + 1, y - 1); fur = (v & 1u) ? 1u + (v >> 1) : 0u; }
      }
    }
  }
  // which references does the row have before / after this fragment
  const uint64_t m1 = __ballot(f == 1u), m2 = __ballot(f == 2u), m3 = __ballot(f == 3u);
  if (lane == 0) s_has[wave] = (m1 ? 1u : 0u) | (m2 ? 2u : 0u) | (m3 ? 4u : 0u);
  __syncthreads();
  uint32_t before = 0, after = 0, all = 0;
  for (int w = 0; w < nw; w++) {
    const uint32_t h = s_has[w];
    before |= w < wave ? h : 0u;
    after |= w > wave ? h : 0u;
    all |= h;
  }
  if (x == 0) P.rowhas[y] = (uint8_t)all;
  if (!in) return;
  const uint64_t mine = f == 1u ? m1 : (f == 2u ? m2 : m3);
  const uint64_t below = ((uint64_t)1 << lane) - 1u, above = lane == 63 ? 0ull : ~(uint64_t)0 << (lane + 1);
  const uint32_t bit = f ? 1u << (f - 1u) : 0u;
  const bool has_before = f && ((mine & below) != 0 || (before & bit) != 0);
  const bool is_last = f && (mine & above) == 0 && (after & bit) == 0;
  uint32_t e0 = (uint32_t)(uint16_t)P.in[(size_t)y * nh + x] | f << 16, e1 = 0, e2 = 0;
  if (f) {
    const int mask = (fl == f ? 1 : 0) | (ful == f ? 2 : 0) | (fu == f ? 4 : 0) | (fur == f ? 8 : 0);
    int wl, wul, wu, wur, sh;
    dcw_weights(mask, wl, wul, wu, wur, sh);
    e1 = (uint32_t)(uint16_t)wl | (uint32_t)(uint16_t)wul << 16;
    e2 = (uint32_t)(uint16_t)wu | (uint32_t)(uint16_t)wur << 16;
    e0 |= (uint32_t)sh << 24;
    if (mask == 0) e0 |= has_before ? kDceOwn : kDceSrc;
    if ((mask & 7) == 7) e0 |= kDceClamp;
    if (is_last) e0 |= kDceLast;
  }
  P.ent[(size_t)y * nh + x] = make_uint4(e0, e1, e2, 0u);
}

typedef short dcw_s2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int dcw_dot2(uint32_t a, uint32_t b, int c) {
  return __builtin_amdgcn_sdot2(__builtin_bit_cast(dcw_s2, a), __builtin_bit_cast(dcw_s2, b), c, false);
}

__global__ __launch_bounds__(64) void k_dc_wave(const DcBatchK B) {
  extern __shared__ __attribute__((aligned(16))) uint8_t s_dcw[];
  const DcPlaneK &P = B.p[blockIdx.y][blockIdx.x];
  const int nh = P.nh, nv = P.nv;
  if (nh <= 0 || nv <= 0) return;
  const int lane = (int)threadIdx.x;
  const DcwLayout L = dcw_layout(nh, nv);
  // (plain LDS pointers and compiler barriers, not `volatile`: a volatile access through a pointer the compiler no longer
  //  knows to be LDS becomes a FLAT instruction with a full wait behind it)
  int16_t *rows = reinterpret_cast<int16_t *>(s_dcw);
  int16_t *srcrow = reinterpret_cast<int16_t *>(s_dcw + L.o_srcrow);
  int16_t *lastval = reinterpret_cast<int16_t *>(s_dcw + L.o_lastval);
  uint8_t *lastrdy = s_dcw + L.o_lastrdy;
  uint32_t *prog = reinterpret_cast<uint32_t *>(s_dcw + L.o_prog);
  uint4 *win = reinterpret_cast<uint4 *>(s_dcw + L.o_win) + lane * kDcwWin;
  uint2 *outq = reinterpret_cast<uint2 *>(s_dcw + L.o_outq) + lane * kDcwBatch;

  // ---- 0. tables: nothing is ready, nobody has started; per row and reference the nearest row at or above that has one
  prog[lane] = 0xFFFF0000u;                 // "row -1, nothing done": below every row a reader can ask for
  for (int i = lane; i < nv * 4; i += 64) {
    lastrdy[i] = 0;
    srcrow[i] = (int16_t)P.rowhas[i >> 2];  // (the row's reference bits for now)
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  if (lane < 3) {
    int src = -1;
    for (int y = 0; y < nv; y++) {
      if ((int)srcrow[y * 4 + lane] >> lane & 1) src = y;
      srcrow[y * 4 + lane] = (int16_t)src;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  // ---- 1. the walk ------------------------------------------------------------------------------------------------
  // entry stream of the lane: its rows one after the other.  cons = entries used, land = entries in the window,
  // (pf_y, pf_x) = the next entry to ask for.
  int y = lane, x = 0;
  int cons = 0, land = 0, pf_y = lane, pf_x = 0;
  uint4 pend[kDcwBatch];                   // entries asked for at the last refill, on their way
#pragma unroll
  for (int k = 0; k < kDcwBatch; k++) pend[k] = make_uint4(0u, 0u, 0u, 0u);
  int npend = 0;
  int nq = 0;                              // finished DCs waiting in outq
  int d_l = 0;                             // the left neighbour's DC
  int pl0 = 0, pl1 = 0, pl2 = 0;           // this row's last DC per reference frame
  const uint4 *ent = P.ent;
  int16_t *out = P.out;
  const int slot_up = (lane + 63) & 63;
  const int nmax = nh * nv;
  const int max_it = 2 * nmax + 16 * nv + 1024;      // even a fully serial picture ends
  for (int it = 0; it < max_it; it++) {
    if ((it & (kDcwBatch - 1)) == 0) {
      // ---- every eighth step, all lanes: land the entries asked for last time, drain the finished DCs, ask for more
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int k = 0; k < kDcwBatch; k++)
        if (k < npend) win[(land + k) & (kDcwWin - 1)] = pend[k];
      land += npend;
#pragma unroll
      for (int k = 0; k < kDcwBatch; k++)
        if (k < nq) {
          const uint2 e = outq[k];
          out[e.x] = (int16_t)e.y;
        }
      nq = 0;
      // (every lane loads, a lane with nothing to ask for re-reads entry 0: a load under a condition is merged with the
      //  register's old value behind it, and the merge waits for the data on the spot)
      const bool want = pf_y < nv && land - cons <= kDcwWin - kDcwBatch;
      npend = 0;
#pragma unroll
      for (int k = 0; k < kDcwBatch; k++) {
        const bool go = want && pf_y < nv;
        pend[k] = ent[go ? (size_t)pf_y * nh + pf_x : (size_t)0];
        if (go) {
          npend = k + 1;
          if (++pf_x == nh) {
            pf_x = 0;
            pf_y += kDcwRows;
          }
        }
      }
      if (!__any(y < nv)) break;
    }
    // ---- one batch of LDS reads: progress of the row above, this fragment's entry, three DCs of the row above
    asm volatile("" ::: "memory");      // (what other lanes wrote in earlier steps is read again)
    const uint32_t p = prog[slot_up];
    asm volatile("" ::: "memory");      // (the progress word first, then what it vouches for)
    const uint4 e = win[cons & (kDcwWin - 1)];
    const int16_t *up = rows + slot_up * L.nhp;
    const int d_ul = up[max(x - 1, 0)], d_u = up[x], d_ur = up[min(x + 1, nh - 1)];
    const int prow = (int)(int16_t)(p >> 16), px = (int)(p & 0xFFFFu);
    const int above = y == 0 ? nh : (prow > y - 1 ? nh : (prow == y - 1 ? px : 0));
    const bool can = y < nv && cons < land && above >= min(x + 2, nh);
    const uint32_t code = (e.x >> 16) & 3u;
    bool done = can;
    int pred = 0;
    if (__any(can && (e.x & kDceSrc) != 0)) {
      if (can && (e.x & kDceSrc)) {
        const int ys = y > 0 ? (int)srcrow[(y - 1) * 4 + (int)code - 1] : -1;      // decode.c:1367: pred_last starts at 0
        if (ys >= 0) {
          done = lastrdy[ys * 4 + (int)code - 1] != 0;                            // that row has passed its last fragment of the reference
          asm volatile("" ::: "memory");
          pred = lastval[ys * 4 + (int)code - 1];
        }
      }
    }
    if (done) {
      int dc = 0;
      if (code) {
        if (e.x & kDceOwn) {
          pred = code == 1u ? pl0 : (code == 2u ? pl1 : pl2);
        } else if (!(e.x & kDceSrc)) {
          const int sh = (int)(e.x >> 24) & 15;
          const int t = dcw_dot2((uint32_t)(uint16_t)d_l | (uint32_t)d_ul << 16, e.y, dcw_dot2((uint32_t)(uint16_t)d_u | (uint32_t)d_ur << 16, e.z, 0));
          pred = (t + ((t >> 31) & ((1 << sh) - 1))) >> sh;                        // C division by 2^sh: towards zero
          if (e.x & kDceClamp) {                                                  // decode.c:1481-1483
            if (abs(pred - d_u) > 128) pred = d_u;
            else if (abs(pred - d_l) > 128) pred = d_l;
            else if (abs(pred - d_ul) > 128) pred = d_ul;
          }
        }
        dc = (int)(int16_t)((int)(int16_t)(e.x & 0xFFFFu) + pred);                // a signed 16-bit bit-field in the reference (state.h:321)
        rows[lane * L.nhp + x] = (int16_t)dc;
        outq[nq++] = make_uint2((uint32_t)(y * nh + x), (uint32_t)dc);
        pl0 = code == 1u ? dc : pl0;
        pl1 = code == 2u ? dc : pl1;
        pl2 = code == 3u ? dc : pl2;
        if (e.x & kDceLast) {
          lastval[y * 4 + (int)code - 1] = (int16_t)dc;
          asm volatile("" ::: "memory");
          lastrdy[y * 4 + (int)code - 1] = 1;
        }
      }
      d_l = dc;
      cons++;
      x++;
      asm volatile("" ::: "memory");
      prog[lane] = (uint32_t)y << 16 | (uint32_t)x;                               // (behind the row and lastval writes: LDS keeps a wave's order)
      if (x == nh) {
        y += kDcwRows;
        x = 0;
        d_l = 0;
      }
    }
  }
  // (the loop leaves at a refill point, after the drain: nothing is pending)
}
