// thip_fused.h -- k_recon_lf: reconstruction AND the whole in-loop filter in one pass, one wave per tile.
// Included by thip_decode.hip behind thip_kernels.h (whose recon / filter building blocks it uses).
//
// k_loopfilter costs a second read and write of every pixel (100 MB of the two passes' 311 MB per
// 4 x 4K step).  A filter cell (lf_cell_ops in thip_kernels.h) needs the UNFILTERED reconstruction of the
// four blocks around its corner and nothing else, so a wave that has just reconstructed a tile
// (16 x 4 blocks, 128 x 32 pixels) can close every cell whose four blocks it holds -- and the cells on its
// left and upper boundary too, if the neighbour tiles hand over their last block column / block row.
// They do, through L2:
//   * every tile PUBLISHES, straight from its registers, its last four pixel rows, its last four pixel
//     columns (and, where the tile above belongs to another XCD's band, its first four rows) into a
//     1.25-KB record of its own, waits for those stores to be acknowledged and sets the record's flag
//     word to the launch's serial number (+ the coded flags of the blocks on those edges);
//   * the tile to the right / below CONSUMES: it polls the flag words of its left, upper and upper-left
//     neighbours, copies their edges into the margins of its LDS image and becomes 16 x 4 filter cells
//     shifted by half a block: lane (kx, m) takes the cell on corner (16t + kx, 4 sby + m).  The wave
//     therefore stores the region [128t-4, 128t+124) x [32sby-4, 32sby+28): every byte of the frame is
//     written exactly once, final, and there is no second kernel.
// Who waits for whom: tiles are numbered plane by plane, tile row by tile row, left to right; the
// launch gives XCD x (work-group id mod 8) the x-th contiguous BAND of whole tile rows (StreamK::band_u0)
// and hands the tiles of a band out in order.  Left, upper and upper-left neighbours of a tile inside
// its band therefore have lower work-group ids on the same XCD: they were dispatched earlier, are
// resident or finished, and depend on nothing themselves (a tile publishes before it waits), so the
// poll cannot deadlock; it is bounded all the same (a wrong picture is a failed test, a hang is a
// dead GPU).  Producer and consumer share one L2: plain stores stop there, device-scope loads go past
// the CU's L1 and find them.
// Band boundaries: the first tile row of band x (D tiles) runs at the START of the launch, the last row of
// band x-1 (U tiles) at its END, so there the hand-over runs upwards: a D tile publishes its first four rows
// with device-scope stores (through to memory: the reader sits on another XCD), leaves the cells on
// its upper boundary alone, and the U tile above -- last of the two by construction -- closes them
// as a 17th cell row.  The same extra pass closes the plane's own border cells (k = nh, m = nv) where
// the plane ends exactly on a tile boundary.
// Order of operations inside every cell: the reference's (state.c:1055-1105), via lf_cell_ops; the
// fragment-row range of the enqueue slot (state.c:1066) is honoured the same way as in k_loopfilter.
#pragma once

#ifndef THIP_TF_PITCH
#define THIP_TF_PITCH 144
#endif
constexpr int kTfPitch = THIP_TF_PITCH;                  // LDS image row: 8-byte left margin (4 used), 128 pixels, 8 spare
constexpr int kTfX0 = 8;                       // byte offset of pixel column 0 in an image row
constexpr int kTfImgRows = 40;                 // pixel rows -4 .. 35 (upper neighbour's last 4, the tile, lower neighbour's first 4)
constexpr int kTfFlagOff = kTfImgRows * kTfPitch;   // coded flags: 6 rows (block rows -1..4) of kTfFlagPitch bytes
constexpr int kTfFlagPitch = 20;               // [0] block column -1, [1..16] the tile, [17] column 16
// The wave's LDS: 7 KB, not 8.  This chip hands LDS out in 1280-byte granules, so 8 KB costs 8960 bytes and a CU holds 18 such
// waves; 7 KB costs 7680 and it holds the 20 the registers allow.  Seven of a tile's eight 1-KB coefficient pieces are staged
// here (LDS-DMA), the eighth stays in registers.
constexpr int kTfLds = 7168;
static_assert(kTfFlagOff + 6 * kTfFlagPitch <= kTfLds, "the image lives in the wave's staging area");
// a tile's record in StreamK::edge
constexpr int kTfBot = 0;                      // pixel rows 28..31: 4 x 128 bytes
constexpr int kTfRight = 512;                  // pixel columns 124..127: 32 rows x 4 bytes
constexpr int kTfTop = 640;                    // pixel rows 0..3: 4 x 128 bytes (tiles that open a band only)
constexpr int kTfFlag = 1152;                  // 8 bytes: {serial << 20 | right4 << 16 | bottom16, top16}
constexpr int kTfRec = 1280;

// scope of the record stores: 0 = this XCD's L2 is enough, 1 = through to memory (a reader on another XCD)
template <int AGENT>
__device__ __forceinline__ void tf_store64(uint8_t *p, uint32_t lo, uint32_t hi) {
  const unsigned long long v = (unsigned long long)hi << 32 | lo;
  if (AGENT)
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// (device scope: past the CU's L1 -- served by the L2 when the producer is on this XCD, by memory otherwise)
__device__ __forceinline__ uint2 tf_load64(const uint8_t *p) {
  const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
}

template <int AGENT>
__device__ __forceinline__ void tf_publish(uint8_t *rec, const uint2 rows[8], int lx, int ly, bool valid, bool pub_top) {
  // rows 28..31 / 0..3: the lanes of block row 3 / 0, four 8-byte pieces each
  const bool bot = ly == 3, top = ly == 0 && pub_top;
  if (bot || top) {
    uint8_t *p = rec + (bot ? kTfBot : kTfTop) + lx * 8;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const uint2 v = bot ? rows[4 + r] : rows[r];
      tf_store64<AGENT>(p + r * 128, v.x, v.y);
    }
  }
  // columns 124..127: the lanes of block column 15, their rows' upper dwords two by two
  if (lx == 15) {
    uint8_t *p = rec + kTfRight + ly * 32;
#pragma unroll
    for (int r = 0; r < 4; r++) tf_store64<AGENT>(p + r * 8, rows[2 * r].y, rows[2 * r + 1].y);
  }
  (void)valid;
}

// One filter cell out of the LDS image: corner column kx (0..16) of the tile, cell row m (0..4).
__device__ __forceinline__ void tf_cell(const uint8_t *lds, uint8_t *plane, int stride, int nh, int nv, int t, int sby, int kx, int m,
                                        bool active, int L2, int fy0, int fy1) {
  const int k = 16 * t + kx, mm = 4 * sby + m;
  active = active && k <= nh && mm <= nv;
  CellPix C;
  const uint8_t *img = lds + (8 * m) * kTfPitch + kTfX0 + 8 * kx - 4;   // image row index = pixel row + 4
#pragma unroll
  for (int r = 0; r < 8; r++) {
    const uint32_t *q = reinterpret_cast<const uint32_t *>(img + r * kTfPitch);
    C.lo[r] = q[0];
    C.hi[r] = q[1];
  }
  const uint8_t *fl = lds + kTfFlagOff + m * kTfFlagPitch + kx;   // flag of block (kx-1, m-1)
  const bool a = fl[0] != 0, b = fl[1] != 0, c = fl[kTfFlagPitch] != 0, d = fl[kTfFlagPitch + 1] != 0;
  uint32_t ops = lf_cell_ops(k, mm, nh, nv, a, b, c, d, fy0, fy1);
  if (L2 == 0 || !active) ops = 0;
  lf_cell_apply_pk(C, ops, L2);
  const bool lo_ok = active && k >= 1, hi_ok = active && k <= nh - 1;
  const bool up_ok = mm >= 1, dn_ok = mm <= nv - 1;
  uint8_t *base = plane + (ptrdiff_t)(8 * mm - 4) * stride + (8 * k - 4);
#pragma unroll
  for (int r = 0; r < 8; r++) {
    uint8_t *p = base + (ptrdiff_t)r * stride;
    if (r < 4 ? up_ok : dn_ok) {
      if (lo_ok & hi_ok) {
        Pix8 o;
        o.x = C.lo[r];
        o.y = C.hi[r];
        *reinterpret_cast<Pix8 *>(p) = o;
      } else if (lo_ok) {
        *reinterpret_cast<uint32_t *>(p) = C.lo[r];
      } else if (hi_ok) {
        *reinterpret_cast<uint32_t *>(p + 4) = C.hi[r];
      }
    }
  }
}

// The neighbours' flag words: lane i < 3 looks at recs[i] when need[i].  tf_poll asks once (the answer travels while the
// wave does something else), tf_wait takes that answer and keeps asking until all carry this launch's serial number;
// returns the words in w[0..2], the second word of recs[0] in w[3] (wave-uniform).
struct TfPoll {
  const uint8_t *rec;
  bool need;
  uint2 f;
};
// (every lane loads -- the lanes with nothing to ask read the wave's own record: a load under a condition would be
//  merged with a default value behind it, and the merge waits for the data on the spot)
__device__ __forceinline__ void tf_poll(TfPoll &P, const uint8_t *rec0, const uint8_t *rec1, const uint8_t *rec2, bool need0, bool need1,
                                        bool need2, int lane, const uint8_t *myrec) {
  const uint8_t *rec = lane == 0 ? rec0 : (lane == 1 ? rec1 : rec2);
  P.need = lane == 0 ? need0 : (lane == 1 ? need1 : (lane == 2 ? need2 : false));
  P.rec = P.need ? rec : myrec;
  P.f = tf_load64(P.rec + kTfFlag);
}
// (fault: the device's pinned host word.  A wait that runs out -- it cannot, as long as work groups are dispatched in order --
//  leaves a wrong picture behind; the word turns that into an error the next synchronising call of the ABI returns.)
__device__ __forceinline__ void tf_wait(const TfPoll &P, uint32_t ep, uint32_t w[4], uint32_t *fault) {
  uint2 f = P.f;
  bool ok = !P.need || (f.x >> 20) == ep;
  for (int spins = 0; spins < (1 << 20) && !__all(ok); spins++) {
    __builtin_amdgcn_s_sleep(2);
    if (!ok) {
      f = tf_load64(P.rec + kTfFlag);
      ok = (f.x >> 20) == ep;
    }
  }
  if (!__all(ok) && fault && !ok) __hip_atomic_store(fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  w[0] = (uint32_t)__builtin_amdgcn_readlane((int)f.x, 0);
  w[1] = (uint32_t)__builtin_amdgcn_readlane((int)f.x, 1);
  w[2] = (uint32_t)__builtin_amdgcn_readlane((int)f.x, 2);
  w[3] = (uint32_t)__builtin_amdgcn_readlane((int)f.y, 0);   // second flag word of rec0
}

#ifndef THIP_TF_WAVES_PER_EU
#define THIP_TF_WAVES_PER_EU 5    // 96 VGPRs; with 8 KB of LDS per wave that is 20 waves per CU
#endif
__global__ __launch_bounds__(64, THIP_TF_WAVES_PER_EU) void k_recon_lf(const BatchK B) {
  __shared__ uint4 s_tf[kTfLds / 16];   // wave-private (one wave per work group): coefficient staging, then the tile image
  const StreamK &S = B.s[blockIdx.y];
  const int lane = (int)threadIdx.x & 63;
  const int band = (int)blockIdx.x & 7, jb = (int)blockIdx.x >> 3;
  // scalar batch 1 (see recon_tile)
  const uint2 *info_p = S.info;
  const int4 *coeffs_p = S.coeffs;
  const uint32_t *slot0_p = S.tile_slot0;
  uint8_t *self = S.self;
  const uint8_t *prev = S.prev, *gold = S.gold;
  uint8_t *coded_map = S.coded_map;
  const int16_t *dc_p = S.dc;
  uint8_t *edge_p = S.edge;
  const int te0 = S.tile_end[0], te1 = S.tile_end[1];
  const int sqpx = S.qpx, sqpy = S.qpy, L2 = S.flimit2;
  const uint32_t ep = S.epoch;
  const int bu0 = S.band_u0[band], bu1 = S.band_u0[band + 1];
  asm volatile("" ::"s"(info_p), "s"(coeffs_p), "s"(slot0_p), "s"(self), "s"(prev), "s"(gold), "s"(coded_map), "s"(dc_p), "s"(edge_p),
               "s"(te0), "s"(te1), "s"(sqpx), "s"(sqpy), "s"(L2), "s"(ep), "s"(bu0), "s"(bu1));
  const int u = bu0 + jb;
  if (u >= bu1) return;
  [[maybe_unused]] unsigned long long *tr = nullptr;   // tools/lf_trace.py: lane 0 stamps the phases of the wave's life (THIP_TRACE builds only)
#ifdef THIP_TRACE
  if (g_trace_buf && lane == 0) {
    tr = g_trace_buf + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 12;
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hwid), "=s"(xcc));
    tr[9] = hwid;
    tr[10] = (unsigned long long)xcc | (unsigned long long)u << 8;
  }
#endif
  THIP_TR(tr, 0);
  const int pli = (u >= te0 ? 1 : 0) + (u >= te1 ? 1 : 0);
  const PlaneK G = S.pl[pli];
  const int fy0 = S.lf_y0[pli], fy1 = S.lf_y1[pli];
  asm volatile("" ::"s"(G.nh), "s"(G.nv), "s"(G.stride), "s"(G.off), "s"(G.tiles_x), "s"(G.tiles_y), "s"(G.fro), "s"(fy0), "s"(fy1));
  const int nh = G.nh, nv = G.nv, tiles_x = G.tiles_x;
  const int rel = u - (pli == 0 ? 0 : (pli == 1 ? te0 : te1));
  const int sby = rel / tiles_x, t = rel - sby * tiles_x;
  // neighbours: inside the plane, and inside this band (bands are whole tile rows, so left / right always are)
  const bool has_left = t > 0, row_end = t == tiles_x - 1;
  const bool has_up = sby > 0, has_dn = sby < G.tiles_y - 1;
  const bool up_in = has_up && u - tiles_x >= bu0, dn_in = has_dn && u + tiles_x < bu1;
  const bool xb_up = has_up && !up_in;     // opens a band below another band's rows: the tile above closes the shared cells
  const bool xb_dn = has_dn && !dn_in;     // ends a band: closes the cells shared with the tile below, which ran long ago

  const int hh = lane & 15;
  const int lx = (lane >> 4) * 4 + hilb_col(hh), ly = hilb_row(hh);
  const int bx = t * 16 + lx, by = sby * 4 + ly;
  const bool valid = bx < nh && by < nv;
  // ---- 1. command word + first slot of the tile ------------------------------------------------------
  const uint32_t slot0 = slot0_p[u];
  const uint2 info = info_p[(size_t)u * THIP_TILE_FRAGS + lane];
  uint32_t dcv = 0;   // the block's un-predicted DC when it does not travel in the command stream
  if (dc_p) dcv = 0x10000u | (uint16_t)dc_p[G.fro + min(by, nv - 1) * nh + min(bx, nh - 1)];
  asm volatile("" ::"s"(slot0), "v"(info.x), "v"(dcv));
#ifdef THIP_TRACE
  THIP_TR(tr, 1);   // command words are here
#endif
  ReconLane L;
  L.flags = valid ? info.x : 0u;
  L.dcq = info.y >> 16;
  L.dcraw = dcv;
  L.dcp = ((uint32_t)(((int)(int16_t)((dcv ? dcv : info.y) & 0xFFFFu) * (int)L.dcq + 15) >> 5) & 0xFFFFu) * 0x00010001u;
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
  R.nh = nh;
  R.nv = nv;
  R.stride = G.stride;
  R.qpx = pli != 0 && sqpx;
  R.qpy = pli != 0 && sqpy;
  R.debug = 0;
  R.tr = nullptr;
  uint8_t *const lds = reinterpret_cast<uint8_t *>(s_tf);
  uint32_t *const lds_dw = reinterpret_cast<uint32_t *>(s_tf);
  uint32_t *const meta = lds_dw + 1024;

  // ---- 2. coefficients + predictor: k_recon's second round trip ---------------------------------------
  const uint64_t mask = __ballot(L.has_coeff);
  const int nown = __popcll(mask);
  const uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
  PredWin Q;
  Q.border = false;
  bool inter = false;
  const uint8_t *ref = nullptr;
  const uint32_t fill = L.dc_only ? L.dcp : 0u;   // DC-only: the rounded value (state.c:972); uncoded: zero residual
  uint32_t Y[32];
  if (nown == 0) {
    if (valid) recon_issue(R, L, Q, inter, ref);
  } else if (nown <= 16) {
    int4 Wc[1][2];
    residual_shared_load<4>(coeffs_p, slot0, nown, lane, Wc);
    if (valid) recon_issue(R, L, Q, inter, ref);
    residual_shared<4>(Wc, lds_dw, meta, lane, L, prefix, Y);
  } else if (nown <= 32) {
    int4 Wc[2][2];
    residual_shared_load<2>(coeffs_p, slot0, nown, lane, Wc);
    if (valid) recon_issue(R, L, Q, inter, ref);
    residual_shared<2, true>(Wc, lds_dw, meta, lane, L, prefix, Y);
  } else {
    const uint32_t slot = slot0 + (L.has_coeff ? prefix : 0u);
    const int4 *tp = coeffs_p + ((size_t)(slot >> 6) * 512 + (slot & 63));
#pragma unroll
    for (int q = 0; q < 7; q++)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(tp + q * 64),
                                       (__attribute__((address_space(3))) void *)(s_tf + q * 64), 16, 0, 0);
    const int4 w7i = tp[7 * 64];
    if (valid) recon_issue(R, L, Q, inter, ref);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // LDS-DMA data has landed (see k_recon)
    const uint4 w7 = make_uint4((uint32_t)w7i.x, (uint32_t)w7i.y, (uint32_t)w7i.z, (uint32_t)w7i.w);
    residual_per_lane(s_tf + lane, L, Y, &w7);
  }
  if (!L.has_coeff) {
#pragma unroll
    for (int i = 0; i < 32; i++) Y[i] = fill;
  }
  uint2 rows[8];
  recon_rows(R, Q, inter, Y, rows);
#ifdef THIP_TRACE
  asm volatile("" : "+v"(rows[0].x), "+v"(rows[7].y));
  THIP_TR(tr, 2);   // pixels done (coefficients and predictor windows had arrived)
#endif

  // ---- 3. publish the edges (from the registers), ask for the neighbours' flag words, image into LDS ----------
  uint8_t *const myrec = edge_p + (size_t)u * kTfRec;
  if (xb_up)
    tf_publish<1>(myrec, rows, lx, ly, valid, true);
  else
    tf_publish<0>(myrec, rows, lx, ly, valid, false);
  const uint8_t *const rec_up = myrec - (ptrdiff_t)tiles_x * kTfRec, *const rec_left = myrec - kTfRec;
  const uint8_t *const rec_ul = rec_up - kTfRec;
  const bool need_ul = up_in && has_left;
  TfPoll poll;                                    // first look at the neighbours' flag words: in flight with the record stores
  tf_poll(poll, rec_up, rec_left, rec_ul, up_in, has_left, need_ul, lane, myrec);
  lds_settle();                                   // every lane is done with the staging area
  if (valid) {
    uint8_t *img = lds + (ly * 8 + 4) * kTfPitch + kTfX0 + lx * 8;
#pragma unroll
    for (int r = 0; r < 8; r++) *reinterpret_cast<uint2 *>(img + r * kTfPitch) = rows[r];
  }
  lds[kTfFlagOff + (ly + 1) * kTfFlagPitch + lx + 1] = (valid && L.coded) ? 1 : 0;
  lds_settle();
  // the coded flags of the published edges, one ballot: bits 0..15 block row 3, 16..19 block column 15, 32..47 block row 0
  {
    const int fi = lane < 16 ? 4 * kTfFlagPitch + lane + 1
                             : (lane < 20 ? (lane - 16 + 1) * kTfFlagPitch + 16 : (lane >= 32 && lane < 48 ? kTfFlagPitch + (lane - 32) + 1 : 0));
    const bool fb = (lane < 20 || (lane >= 32 && lane < 48)) && lds[kTfFlagOff + fi] != 0;
    const uint64_t fm = __ballot(fb);
    THIP_TR(tr, 3);   // edges on their way, image in LDS
    // the flag word goes out when the record is in place: stores are acknowledged by the L2 (or by memory)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) {
      const uint32_t w0 = ep << 20 | (uint32_t)(fm & 0xFFFFFu), w1 = (uint32_t)(fm >> 32) & 0xFFFFu;
      if (xb_up)
        tf_store64<1>(myrec + kTfFlag, w0, w1);
      else
        tf_store64<0>(myrec + kTfFlag, w0, w1);
    }
    THIP_TR(tr, 4);   // record acknowledged, flag word out
  }

  // ---- 4. the neighbours' edges into the image margins --------------------------------------------------
  {
    uint32_t w[4];
    tf_wait(poll, ep, w, B.fault);
    THIP_TR(tr, 5);   // the neighbours' records are there
    // lanes 0..31: the upper tile's rows 28..31, 16 bytes each; 32..39: the left tile's columns 124..127, four rows each;
    // 40: the upper-left tile's corner (rows 28..31 of its column record)
    const uint8_t *src = nullptr;
    if (lane < 32) {
      if (up_in) src = rec_up + kTfBot + lane * 16;
    } else if (lane < 40) {
      if (has_left) src = rec_left + kTfRight + (lane - 32) * 16;
    } else if (lane == 40) {
      if (need_ul) src = rec_ul + kTfRight + 112;
    }
    uint2 d0 = make_uint2(0u, 0u), d1 = d0;
    if (src) {
      d0 = tf_load64(src);
      d1 = tf_load64(src + 8);
    }
    if (lane < 32) {
      uint8_t *p = lds + (lane >> 3) * kTfPitch + kTfX0 + (lane & 7) * 16;
      *reinterpret_cast<uint2 *>(p) = d0;
      *reinterpret_cast<uint2 *>(p + 8) = d1;
    } else if (lane <= 40) {
      uint8_t *p = lds + (lane == 40 ? 0 : 4 + 4 * (lane - 32)) * kTfPitch + kTfX0 - 4;
      *reinterpret_cast<uint32_t *>(p) = d0.x;
      *reinterpret_cast<uint32_t *>(p + kTfPitch) = d0.y;
      *reinterpret_cast<uint32_t *>(p + 2 * kTfPitch) = d1.x;
      *reinterpret_cast<uint32_t *>(p + 3 * kTfPitch) = d1.y;
    }
    // their coded flags: block row -1 (columns 0..15), block column -1 (rows 0..3), block (-1, -1)
    if (lane < 16)
      lds[kTfFlagOff + lane + 1] = (uint8_t)((w[0] >> lane) & 1u);
    else if (lane < 20)
      lds[kTfFlagOff + (lane - 16 + 1) * kTfFlagPitch] = (uint8_t)((w[1] >> lane) & 1u);
    else if (lane == 20)
      lds[kTfFlagOff] = (uint8_t)((w[2] >> 19) & 1u);
    lds_settle();
    THIP_TR(tr, 6);   // ... and in the margins
  }

  // ---- 5. the cells: lane (kx, m) on corner (16t + kx, 4 sby + m) ---------------------------------------------
  tf_cell(lds, R.self, R.stride, nh, nv, t, sby, lane & 15, lane >> 4, !(xb_up && lane < 16), L2, fy0, fy1);
  THIP_TR(tr, 7);   // cells filtered, stores issued

  // ---- 6. a 17th cell column where the plane ends on this tile's right boundary (k = nh), a 5th cell row where the plane
  //         ends on its lower boundary (m = nv) or where the tile below belongs to another band ----------------------------
  const bool extra_col = row_end && (nh & 15) == 0;
  const bool extra_row = xb_dn || (!has_dn && (nv & 3) == 0);
  if (extra_col || extra_row) {
    if (xb_dn) {
      const uint8_t *const rec_dn = myrec + (ptrdiff_t)tiles_x * kTfRec, *const rec_dl = rec_dn - kTfRec;
      uint32_t w[4];
      TfPoll pd;
      tf_poll(pd, rec_dn, rec_dl, rec_dl, true, has_left, false, lane, myrec);
      tf_wait(pd, ep, w, B.fault);
      const uint8_t *src = nullptr;
      if (lane < 32)
        src = rec_dn + kTfTop + lane * 16;          // the lower tile's rows 0..3
      else if (lane == 32 && has_left)
        src = rec_dl + kTfRight;                    // the lower-left tile's corner (rows 0..3 of its column record)
      uint2 d0 = make_uint2(0u, 0u), d1 = d0;
      if (src) {
        d0 = tf_load64(src);
        d1 = tf_load64(src + 8);
      }
      if (lane < 32) {
        uint8_t *p = lds + (36 + (lane >> 3)) * kTfPitch + kTfX0 + (lane & 7) * 16;
        *reinterpret_cast<uint2 *>(p) = d0;
        *reinterpret_cast<uint2 *>(p + 8) = d1;
      } else if (lane == 32) {
        uint8_t *p = lds + 36 * kTfPitch + kTfX0 - 4;
        *reinterpret_cast<uint32_t *>(p) = d0.x;
        *reinterpret_cast<uint32_t *>(p + kTfPitch) = d0.y;
        *reinterpret_cast<uint32_t *>(p + 2 * kTfPitch) = d1.x;
        *reinterpret_cast<uint32_t *>(p + 3 * kTfPitch) = d1.y;
      }
      // block row 4: the lower tile's first block row (word 1 of its flags), the lower-left tile's block (15, 0)
      if (lane < 16)
        lds[kTfFlagOff + 5 * kTfFlagPitch + lane + 1] = (uint8_t)((w[3] >> lane) & 1u);
      else if (lane == 16)
        lds[kTfFlagOff + 5 * kTfFlagPitch] = (uint8_t)((w[1] >> 16) & 1u);
      lds_settle();
    }
    // lanes 0..16: cells (lane, 4); lanes 32..35: cells (16, lane - 32)
    const bool rowl = lane <= 16, coll = lane >= 32 && lane < 36;
    const int kx = rowl ? lane : 16, m = rowl ? 4 : (lane - 32) & 3;
    const bool act = rowl ? (extra_row && (lane < 16 || extra_col)) : (coll && extra_col && !(xb_up && m == 0));
    tf_cell(lds, R.self, R.stride, nh, nv, t, sby, kx, m, act, L2, fy0, fy1);
  }
  THIP_TR(tr, 8);
}
