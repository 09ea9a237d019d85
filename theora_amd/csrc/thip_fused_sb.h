// thip_fused_sb.h -- k_recon_lf_sb: k_recon_lf (thip_fused.h) for launches that leave the chip empty.
// Included by thip_decode.hip behind thip_fused.h, whose hand-over (units, bands, who waits for whom) and cell arithmetic it shares.
//
// One 1080p frame is 782 tiles: k_recon_lf gives every tile a wave of its own, so a single stream puts fewer waves on the chip than
// it has SIMDs (1024), every wave runs its ~1 700 vector instructions one after the other with nothing to interleave -- a wave
// alone on a SIMD issues one every 5.2 clocks -- and each of its three memory round trips is idle time: 9.3 us of wave life,
// 11.9 us per frame (profiles/r04_lf_trace_1080p_single.txt), and the frames of one stream cannot overlap (decode.c:2858-2945
// is one frame after the other: every frame predicts from the one before).  Here the unit of work is ONE SUPER BLOCK (4 x 4
// blocks, 32 x 32 pixels) per wave and a block belongs to FOUR lanes: lane 4 b + p owns rows 2p, 2p+1 of block b (b = the
// block's place on the super block's Hilbert curve, so the command words, the coefficient slots and the tile numbering are
// k_recon_lf's: a wave reads its tile's 64 command words, finds the slots of its super block behind those of the super blocks
// before it with the same ballot, and fetches the words of its own blocks with a cross-lane read).  Four times the waves, each
// with a quarter of the reconstruction: the inverse transforms are residual_shared<4> as they stand (four lanes per block was
// their shape already), a lane loads and averages three predictor rows instead of nine and clamps two rows instead of eight;
// the filter cells of the super block -- 4 x 4, plus the fifth row / column where the plane or the band ends -- are one pass of 25
// lanes.  The cells keep the idle lanes idle (a cell's four operations share its centre pixels: it does not split), so the
// chip executes about 1.7 times the instructions of k_recon_lf -- which is why the launch takes this kernel only while its
// tiles are fewer than two per SIMD (option sb_tiles) -- but three or four waves per SIMD now overlap each other's round trips.
// LDS: 5 KB per wave (exchange 2 KB | results 2 KB | owner words | the plane's tables), the image over the exchange afterwards.
//
// Round 6: the same kernel with TWO super blocks a wave and two lanes a block (k_recon_lf_h; NSB = 2 below: a half tile of 8 x 4
// blocks, lane 2 b + p owns rows 4p .. 4p + 3 of block b) for launches between the two -- one 1080p stream is 782 tiles on 1 024
// SIMDs: with a wave per tile every wave runs alone at the single-wave issue rate, with a wave per super block (3 128 waves) the
// cells' idle lanes and three waves a SIMD eat what the shorter chain gains (DESIGN_HISTORY.md section 5h).  1 564 waves with half
// the transforms, predictor rows and image rows each, and the cells of 9 x 5 corners on 45 lanes (sb: 25).  Same records, tags,
// bands and recovery with TfGeom<8>, a buffer and a serial number of its own.  LDS: exchange 4 KB with the results over it
// (residual_shared's COMPACT form) | owner words | tables: the same 5 KB.
#pragma once

typedef TfGeom<4> Tf4;
typedef TfGeom<8> Tf8;
constexpr int kSbLds = 5120;
constexpr int kSbTab16 = 4352 / 16;     // the tables: behind the exchange, the results and the 64 owner words
static_assert(Tf4::kFlagOff + 6 * Tf4::kFlagPitch <= 2048, "the image lives over the exchange area");
static_assert(Tf8::kFlagOff + 6 * Tf8::kFlagPitch <= 4096, "the image lives over the exchange area");

// Predictor of the NR rows a lane owns (pred_issue / pred_finish of thip_kernels.h for rows y0 .. y0 + NR - 1 of a block)
template <int NR>
struct PredWinN {
  Row12 w[NR + 1];
  int sx, sy, mx2, my2;
  bool border;
};
template <int NR>
__device__ __forceinline__ void predn_issue(PredWinN<NR> &Q, const uint8_t *ref, int stride, int W, int H, int x0, int y0, uint32_t flags, bool qpx,
                                            bool qpy) {
  const int dx = (int)(int8_t)(flags >> THIP_INFO_MVX_SHIFT);
  const int dy = (int)(int8_t)(flags >> THIP_INFO_MVY_SHIFT);
  int mx, my;
  mv_axis(dx, qpx, mx, Q.mx2);
  mv_axis(dy, qpy, my, Q.my2);
  Q.sx = x0 + mx;
  Q.sy = y0 + my;
  const int xs = Q.sx + min(Q.mx2, 0), ys = Q.sy + min(Q.my2, 0);
  Q.border = ((int)(xs < 0) | (int)(Q.sx + max(Q.mx2, 0) + 8 > W)) != 0;   // (rows are clamped where they are loaded)
  const uint8_t *p1 = ref + pred_xw(xs, W);
#pragma unroll
  for (int r = 0; r < NR + 1; r++) {
    const int y = min(max(ys + (r < NR ? r : (Q.my2 != 0 ? NR : NR - 1)), 0), H - 1);
    Q.w[r] = load_row12(p1 + (ptrdiff_t)y * stride);
  }
}
template <int NR>
__device__ __forceinline__ void predn_finish(const PredWinN<NR> &Q, int W, uint2 pred[NR]) {
  const int xw = pred_xw(Q.sx + min(Q.mx2, 0), W);
  const bool ra = Q.my2 < 0, rb = Q.my2 > 0;
  const bool two = (Q.mx2 | Q.my2) != 0;
  if (!__any(Q.border)) {
    const int offA = Q.sx - xw, offB = Q.sx + Q.mx2 - xw;
#pragma unroll
    for (int r = 0; r < NR; r++) {
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
    uint32_t sa0, sa1, sb0, sb1;
    bool ka0, ka1, kb0, kb1;
    pred_sel(Q.sx, xw, W, sa0, ka0);
    pred_sel(Q.sx + 4, xw, W, sa1, ka1);
    pred_sel(Q.sx + Q.mx2, xw, W, sb0, kb0);
    pred_sel(Q.sx + Q.mx2 + 4, xw, W, sb1, kb1);
#pragma unroll
    for (int r = 0; r < NR; r++) {
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

// A work group is the four super blocks of a tile, one wave each: the waves never synchronise with each other (the left
// neighbour's units travel through memory like everybody else's), they only arrive together -- the dispatcher starts some 700
// work groups a microsecond, and 3 128 single-wave groups for a 1080p frame would spend a third of the frame's time being started.
// NSB: super blocks a wave takes (1: k_recon_lf_sb, four lanes a block; 2: k_recon_lf_h, two lanes a block).  The work group is the
// tile: 4 / NSB waves.
template <bool LEVELS, int NSB>
__device__ __forceinline__ void recon_lf_small(const BatchK &B, uint4 *const s_sb, const int wave) {
  constexpr int LPB = 4 / NSB;             // lanes a block
  constexpr int NR = 8 / LPB;              // pixel rows a lane
  constexpr int BW = 4 * NSB;              // blocks across
  constexpr int RPT = 4 / NSB;             // waves (and records) a tile
  typedef TfGeom<BW> Tf;
  const StreamK &S = B.s[blockIdx.y];
  const int lane = (int)threadIdx.x & 63;
  const int band = (int)blockIdx.x & 7;
  const uint2 *info_p = S.info;
  const int4 *coeffs_p = S.coeffs;
  const uint32_t *slot0_p = S.tile_slot0;
  uint8_t *self = S.self;
  const uint8_t *prev = S.prev, *gold = S.gold;
  uint8_t *coded_map = S.coded_map;
  const int16_t *dc_p = S.dc;
  const uint4 *dq_p = S.dequant;
  uint8_t *edge_p = S.edge;
  const int te0 = S.tile_end[0], te1 = S.tile_end[1];
  const int sqpx = S.qpx, sqpy = S.qpy;
  const int L2 = (S.debug & 256) ? 0 : S.flimit2;
  const bool poison = (S.debug & 512) != 0;           // (see k_recon_lf)
  const int max_spins = poison ? 256 : (1 << 20);
  uint32_t *fault_p = S.fault;
  const uint32_t ep = S.epoch;
  const int bu0 = S.band_u0[band], bu1 = S.band_u0[band + 1];
  asm volatile("" ::"s"(info_p), "s"(coeffs_p), "s"(slot0_p), "s"(self), "s"(prev), "s"(gold), "s"(coded_map), "s"(dc_p), "s"(dq_p), "s"(edge_p),
               "s"(te0), "s"(te1), "s"(sqpx), "s"(sqpy), "s"(L2), "s"(ep), "s"(bu0), "s"(bu1), "s"(fault_p));
  const int u = bu0 + ((int)blockIdx.x >> 3), sub = wave;   // tile, the wave's place in it
  if (u >= bu1) return;
  const int pli = (u >= te0 ? 1 : 0) + (u >= te1 ? 1 : 0);
  constexpr bool levels = LEVELS;
  const PlaneK G = S.pl[pli];
  const int fy0 = S.lf_y0[pli], fy1 = S.lf_y1[pli];
  asm volatile("" ::"s"(G.nh), "s"(G.nv), "s"(G.stride), "s"(G.off), "s"(G.tiles_x), "s"(G.tiles_y), "s"(G.fro), "s"(fy0), "s"(fy1));
  const int nh = G.nh, nv = G.nv, tiles_x = G.tiles_x;
  const int rel = u - (pli == 0 ? 0 : (pli == 1 ? te0 : te1));
  const int sby = rel / tiles_x, t = rel - sby * tiles_x;
  const int sbx = 4 * t + NSB * sub;                  // the wave's first super block
  const int tw = RPT * t + sub;                       // ... its number across the plane in units of BW blocks
  const int nsbw = (nh + 3) >> 2;
  if (sbx >= nsbw) return;                            // super blocks past the plane's right edge: nobody waits for them
  [[maybe_unused]] unsigned long long *tr = nullptr;   // tools/lf_trace.py (THIP_TRACE builds only): the phases of k_recon_lf's record
#ifdef THIP_TRACE
  if (g_trace_buf && lane == 0) {
    tr = g_trace_buf + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * RPT + wave) * 12;
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hwid), "=s"(xcc));
    tr[9] = hwid;
    tr[10] = (unsigned long long)xcc | (unsigned long long)u << 8;
  }
#endif
  THIP_TR(tr, 0);
  if (levels) tables_to_lds(dq_p, pli, lane, s_sb, kSbTab16);   // (first: whoever has its command word has the tables)
  const bool has_left = sbx > 0, row_end = sbx + NSB - 1 >= nsbw - 1;
  const bool has_up = sby > 0, has_dn = sby < G.tiles_y - 1;
  const bool up_in = has_up && u - tiles_x >= bu0, dn_in = has_dn && u + tiles_x < bu1;
  const bool xb_up = has_up && !up_in;
  const bool xb_dn = has_dn && !dn_in;

  // ---- 1. the tile's 64 command words (one per lane) and its first slot ----------------------------------------------------------
  const uint32_t slot0w = slot0_p[u];
  const uint2 info = info_p[(size_t)u * THIP_TILE_FRAGS + lane];
  uint32_t flags_t, dcv_t = 0;
  {
    const int hh = lane & 15;
    const int bxt = t * 16 + (lane >> 4) * 4 + hilb_col(hh), byt = sby * 4 + hilb_row(hh);
    if (dc_p) dcv_t = 0x10000u | (uint16_t)dc_p[G.fro + min(byt, nv - 1) * nh + min(bxt, nh - 1)];
    asm volatile("" ::"s"(slot0w), "v"(info.x), "v"(dcv_t));
    flags_t = (bxt < nh && byt < nv) ? info.x : 0u;
  }
  THIP_TR(tr, 1);
  const uint64_t mask = __ballot((flags_t & THIP_INFO_CODED) != 0 && (flags_t & THIP_INFO_DC_ONLY) == 0);
  // ... and this lane's block: b on the Hilbert curves of the wave's super blocks, part p
  const int b = lane / LPB, part = lane % LPB;
  const int src = 16 * NSB * sub + b;
  const uint32_t flags = (uint32_t)__shfl((int)flags_t, src), w1 = (uint32_t)__shfl((int)info.y, src);
  uint32_t dcv = (uint32_t)__shfl((int)dcv_t, src);
  if (levels && !dc_p) dcv = 0x10000u | (w1 & 0xFFFFu);
  const int lx = 4 * (b >> 4) + hilb_col(b & 15), ly = hilb_row(b & 15);
  const int bx = 4 * sbx + lx, by = 4 * sby + ly;
  const bool valid = bx < nh && by < nv;
  ReconLane L;
  L.flags = flags;                                    // (0 past the ragged edge)
  L.dcq = w1 >> 16;
  L.dcraw = dcv;
  L.dcp = ((uint32_t)(((int)(int16_t)((dcv ? dcv : w1) & 0xFFFFu) * (int)L.dcq + 15) >> 5) & 0xFFFFu) * 0x00010001u;
  L.coded = (L.flags & THIP_INFO_CODED) != 0;
  L.dc_only = L.coded && (L.flags & THIP_INFO_DC_ONLY) != 0;
  L.has_coeff = L.coded && !L.dc_only;
  L.x0 = bx * 8;
  L.y0 = by * 8;
  const uint64_t mask_w = (mask >> (16 * NSB * sub)) & ((1ull << (16 * NSB)) - 1ull);   // the owners among the wave's blocks
  const int nown = __popcll(mask_w);
  const uint32_t prefix = (uint32_t)__popcll(mask_w & ((1ull << b) - 1ull));
  CoefForm F = coef_form(slot0w, levels);
  F.slot0 += (uint32_t)__popcll(mask & ((1ull << (16 * NSB * sub)) - 1ull)) * (F.wide ? 2u : 1u);
  uint8_t *const lds = reinterpret_cast<uint8_t *>(s_sb);
  uint32_t *const lds_dw = reinterpret_cast<uint32_t *>(s_sb);
  uint32_t *const meta = lds_dw + 1024;
  uint8_t *const plane_self = self + G.off;
  const int W = nh * 8, H = nv * 8;

  // ---- 2. coefficients + the lane's predictor rows -----------------------------------------------------------------------------------
  const int refi = L.coded ? (int)((L.flags >> THIP_INFO_REFI_SHIFT) & 3u) : THIP_FRAME_PREV;
  const bool inter = valid && refi != THIP_FRAME_SELF;
  PredWinN<NR> Q;
  Q.border = false;
  Q.sx = Q.sy = Q.mx2 = Q.my2 = 0;
  uint32_t Y[4 * NR];
  int4 Wc[4 / LPB][2];
  if (nown) residual_shared_load<LPB>(coeffs_p, F, nown, lane, Wc);
  if (inter)
    predn_issue<NR>(Q, (refi == THIP_FRAME_PREV ? prev : gold) + G.off, G.stride, W, H, L.x0, L.y0 + NR * part, L.coded ? L.flags : 0u, pli != 0 && sqpx,
                    pli != 0 && sqpy);
  if (valid && part == 0) coded_map[G.fro + by * nh + bx] = L.coded ? 1 : 0;
  // (two lanes a block: the results go where the exchange was -- 4 KB for 32 owners --, the owner words and the tables behind it)
  if (nown) residual_shared<LPB, NSB == 2, 4 * NR>(Wc, F, lds_dw, meta, lane, L, prefix, Y, kSbTab16);
  if (!L.has_coeff) {
    const uint32_t fill = L.dc_only ? L.dcp : 0u;
#pragma unroll
    for (int i = 0; i < 4 * NR; i++) Y[i] = fill;
  }
#ifdef THIP_TRACE
  asm volatile("" : "+v"(Y[0]), "+v"(Y[4 * NR - 1]));
  THIP_TR(tr, 11);
#endif
  uint2 pred[NR];
#pragma unroll
  for (int r = 0; r < NR; r++) pred[r] = make_uint2(0x80808080u, 0x80808080u);
  if (__any(inter)) {   // (predn_finish looks at the whole wave for the border case)
    uint2 pr[NR];
    predn_finish<NR>(Q, W, pr);
    if (inter) {
#pragma unroll
      for (int r = 0; r < NR; r++) pred[r] = pr[r];
    }
  }
  uint2 rows[NR];
#pragma unroll
  for (int r = 0; r < NR; r++) rows[r] = pk_recon_row(as_pk(Y[r * 4 + 0]), as_pk(Y[r * 4 + 1]), as_pk(Y[r * 4 + 2]), as_pk(Y[r * 4 + 3]), pred[r]);

#ifdef THIP_TRACE
  asm volatile("" : "+v"(rows[0].x), "+v"(rows[NR - 1].y));
  THIP_TR(tr, 2);
#endif
  // ---- 3. the super block's image into LDS, its edges out as units ------------------------------------------------------------------
  const size_t rec_i = (size_t)RPT * u + sub;
  uint8_t *const myrec = edge_p + rec_i * Tf::kRec;
  const uint8_t *const rec_up = myrec - (ptrdiff_t)RPT * tiles_x * Tf::kRec, *const rec_left = myrec - Tf::kRec;
  const uint8_t *const rec_ul = rec_up - Tf::kRec;
  const bool need_ul = up_in && has_left;
  lds_settle();
  if (valid) {
    uint8_t *img = lds + (ly * 8 + 4 + NR * part) * Tf::kPitch + Tf::kX0 + lx * 8;
#pragma unroll
    for (int r = 0; r < NR; r++) *reinterpret_cast<uint2 *>(img + r * Tf::kPitch) = rows[r];
  }
  if (part == 0) lds[Tf::kFlagOff + (ly + 1) * Tf::kFlagPitch + lx + 1] = (valid && L.coded) ? 1 : 0;
  lds_settle();
  {
    // the coded flags of the published edges: bits 0..BW-1 block row 3, 16..19 block column BW - 1, 32..32+BW-1 block row 0
    const int fi = lane < BW ? 4 * Tf::kFlagPitch + lane + 1
                             : (lane >= 16 && lane < 20 ? (lane - 16 + 1) * Tf::kFlagPitch + BW : (lane >= 32 && lane < 32 + BW ? Tf::kFlagPitch + (lane - 32) + 1 : 0));
    const bool fb = (lane < BW || (lane >= 16 && lane < 20) || (lane >= 32 && lane < 32 + BW)) && lds[Tf::kFlagOff + fi] != 0;
    const uint64_t fm = __ballot(fb);
    const uint32_t tag_ep = (poison && rec_i == 1) ? 0u : ep;
    constexpr uint32_t kRowBits = (1u << BW) - 1u;
    tf_publish_units<Tf>(lds, myrec + Tf::kBot, myrec + Tf::kRight, 34, true, tag_ep << 20 | (uint32_t)(fm & (0xF0000u | kRowBits)), lane, xb_up);
    if (xb_up) tf_publish_units<Tf>(lds, myrec + Tf::kTop, nullptr, 4, false, tag_ep << 20 | ((uint32_t)(fm >> 32) & kRowBits), lane, true);
    THIP_TR(tr, 3);
    THIP_TR(tr, 4);
  }

  // ---- 4. the neighbours' edges into the image margins: lanes 0..5 the upper super block's rows 30, 31, 6..16 the left one's
  //         columns 28..31, 17 the upper-left one's corner ----------------------------------------------------------------------
  {
    constexpr int kB = Tf::kBotUnits, kR = Tf::kRightUnits;
    constexpr uint32_t kFaultTag = NSB == 1 ? 0x1000u : 0x2000u;   // (which kernel's serial number a failed wait reports)
    const uint8_t *usrc = nullptr;
    if (lane < kB) {
      if (up_in) usrc = rec_up + Tf::kBot + lane * kTfUnit;
    } else if (lane < kB + kR) {
      if (has_left) usrc = rec_left + Tf::kRight + (lane - kB) * kTfUnit;
    } else if (lane == kB + kR) {
      if (need_ul) usrc = rec_ul + Tf::kRight + 10 * kTfUnit;
    }
    const uint4 un = tf_fetch_unit(usrc, ep, fault_p, max_spins, ep | kFaultTag);
    THIP_TR(tr, 5);
    const uint32_t d[3] = {un.x, un.y, un.z};
    if (lane < kB) {
#pragma unroll
      for (int j = 0; j < 3; j++) {
        const int idx = 3 * lane + j;
        if (idx < 2 * Tf::kRowDwords) *reinterpret_cast<uint32_t *>(lds + tf_rows_at<Tf>(idx, 2)) = d[j];
      }
    } else if (lane < kB + kR) {
#pragma unroll
      for (int j = 0; j < 3; j++) {
        const int idx = 3 * (lane - kB) + j;
        if (idx < 32) *reinterpret_cast<uint32_t *>(lds + tf_col_at<Tf>(idx, 4, -4)) = d[j];
      }
    } else if (lane == kB + kR) {
#pragma unroll
      for (int j = 0; j < 2; j++) *reinterpret_cast<uint32_t *>(lds + tf_col_at<Tf>(2 + j, 0, -4)) = d[j];   // dwords 30, 31
    }
    const uint32_t w_up = (uint32_t)__builtin_amdgcn_readlane((int)un.w, 0);
    const uint32_t w_left = (uint32_t)__builtin_amdgcn_readlane((int)un.w, kB);
    const uint32_t w_ul = (uint32_t)__builtin_amdgcn_readlane((int)un.w, kB + kR);
    if (lane < BW)
      lds[Tf::kFlagOff + lane + 1] = (uint8_t)((w_up >> lane) & 1u);
    else if (lane >= 16 && lane < 20)
      lds[Tf::kFlagOff + (lane - 16 + 1) * Tf::kFlagPitch] = (uint8_t)((w_left >> lane) & 1u);
    else if (lane == 20)
      lds[Tf::kFlagOff] = (uint8_t)((w_ul >> 19) & 1u);
    if (xb_dn) {
      // the super block below ran at the start of the launch: its first two rows (lanes 0..5), the lower-left one's corner (lane 6)
      const uint8_t *const rec_dn = myrec + (ptrdiff_t)RPT * tiles_x * Tf::kRec, *const rec_dl = rec_dn - Tf::kRec;
      const uint8_t *s2 = nullptr;
      if (lane < kB) s2 = rec_dn + Tf::kTop + lane * kTfUnit;
      else if (lane == kB && has_left) s2 = rec_dl + Tf::kRight;
      const uint4 u2 = tf_fetch_unit(s2, ep, fault_p, max_spins, ep | kFaultTag);
      const uint32_t e[3] = {u2.x, u2.y, u2.z};
      if (lane < kB) {
#pragma unroll
        for (int j = 0; j < 3; j++) {
          const int idx = 3 * lane + j;
          if (idx < 2 * Tf::kRowDwords) *reinterpret_cast<uint32_t *>(lds + tf_rows_at<Tf>(idx, 36)) = e[j];
        }
      } else if (lane == kB) {
#pragma unroll
        for (int j = 0; j < 2; j++) *reinterpret_cast<uint32_t *>(lds + tf_col_at<Tf>(j, 36, -4)) = e[j];
      }
      const uint32_t w_dn = (uint32_t)__builtin_amdgcn_readlane((int)u2.w, 0);
      const uint32_t w_dl = (uint32_t)__builtin_amdgcn_readlane((int)u2.w, kB);
      if (lane < BW)
        lds[Tf::kFlagOff + 5 * Tf::kFlagPitch + lane + 1] = (uint8_t)((w_dn >> lane) & 1u);
      else if (lane == 16)
        lds[Tf::kFlagOff + 5 * Tf::kFlagPitch] = (uint8_t)((w_dl >> 16) & 1u);
    }
    lds_settle();
    THIP_TR(tr, 6);
  }

  // ---- 5. the cells, one pass: lane c < 25 takes the cell on corner (4 sbx + c % 5, 4 sby + c / 5); the fifth column where the
  //         plane ends on this super block's right boundary, the fifth row where it ends on the lower one or where the super
  //         block below belongs to another band (row and operation ranges: see k_recon_lf, sections 5 and 6) ------------------
  {
    const bool extra_col = row_end && (nh % BW) == 0;
    const bool extra_row = xb_dn || (!has_dn && (nv & 3) == 0);
    const int top_lo = has_up ? (xb_up ? 6 : 2) : 0;
    const uint32_t top_mask = xb_up ? 96u : 0xFFu;
    constexpr int NC = BW + 1;                            // corners across
    const bool cl = lane < 5 * NC;
    const int kx = cl ? lane % NC : 0, m = cl ? lane / NC : 0;
    const bool act = cl && (kx < BW || extra_col) && (m < 4 || extra_row);
    tf_cell<Tf>(lds, plane_self, G.stride, nh, nv, tw, sby, kx, m, act, L2, fy0, fy1, m == 4 ? 2 : (m == 0 ? top_lo : 0),
                 m == 4 ? (xb_dn ? 6 : 8) : 8, m == 0 ? top_mask : 0xFFu, m == 3);
  }
  THIP_TR(tr, 7);
  THIP_TR(tr, 8);
}

template <bool LEVELS>
__global__ __launch_bounds__(256) void k_recon_lf_sb(const BatchK B) {
  __shared__ uint4 s_sb4[4 * (kSbLds / 16)];
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  recon_lf_small<LEVELS, 1>(B, s_sb4 + wave * (kSbLds / 16), wave);
}
template <bool LEVELS>
__global__ __launch_bounds__(128) void k_recon_lf_h(const BatchK B) {
  __shared__ uint4 s_sb2[2 * (kSbLds / 16)];
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  recon_lf_small<LEVELS, 2>(B, s_sb2 + wave * (kSbLds / 16), wave);
}
