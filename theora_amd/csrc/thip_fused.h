// thip_fused.h -- k_recon_lf: reconstruction AND the whole in-loop filter in one pass, one wave per tile.
// Included by thip_decode.hip behind thip_kernels.h (whose recon / filter building blocks it uses).
//
// k_loopfilter costs a second read and write of every pixel (100 MB of the two passes' 311 MB per
// 4 x 4K step).  A filter cell (lf_cell_ops in thip_kernels.h) needs the UNFILTERED reconstruction of the
// four blocks around its corner and nothing else, so a wave that has just reconstructed a tile
// (16 x 4 blocks, 128 x 32 pixels) can close every cell whose four blocks it holds -- and the cells on its
// left and upper boundary too, if the neighbour tiles hand over their last block column / block row.
// They do, through L2, in self-validating 16-byte UNITS -- 12 bytes of pixels and a tag word {launch serial << 20 | coded flags
// of the edge blocks} -- so that data and "it is there" arrive in the same access:
//   * every tile PUBLISHES its last TWO pixel rows (22 units), its last four pixel columns (11 units) and, where the tile
//     above belongs to another XCD's band, its first two rows: one 16-byte store per lane, assembled from the tile image in
//     LDS, no wait for an acknowledgement, no separate flag;
//   * the tile to the right / below CONSUMES: 34 lanes load one unit each of the left, upper and upper-left neighbours' records
//     (past the CU's L1) and look again until every tag carries this launch's serial number -- one round trip when the
//     neighbour was done first -- then scatter the pixels into the margins of the LDS image and the wave becomes 16 x 4 filter
//     cells shifted by half a block: lane (kx, m) takes the cell on corner (16t + kx, 4 sby + m).  The wave therefore stores
//     the region [128t-4, 128t+124) x [32sby-2, 32sby+30): every byte of the frame is written exactly once, final, and there
//     is no second kernel.
// Two rows, not the four a cell has above its corner: the horizontal edge at the tile boundary reads two rows on either side
// and changes one, so rows 28 and 29 of a tile are touched by vertical edges only -- row by row, no order to keep -- and the
// tile finishes them itself (the cells of its last cell row run their lower half's vertical edge over one more row pair,
// tf_cell's `ext`), while the cells on the tile's upper boundary own their rows 2..7 only.  Every byte not handed over is a
// byte not written to and read from memory: the records cost 13 MB of a 4 x 4K step's 227 MB this way, 22 MB with four rows
// (measured before it was built, with a build that simply dropped half the units: the step time follows the bytes).
// (A unit is one aligned 16-byte access of one lane: a single request to the L2 that owns the line, which is what makes the
//  tag vouch for the twelve bytes in front of it.  Round 3's first version kept data and a flag word apart: the producer
//  waited for its stores to be acknowledged before it set the flag, the consumer polled the flag and then fetched the data --
//  two more dependent round trips in a wave's life, which is what bounds the kernel on content that is not byte-bound.)
// Who waits for whom: tiles are numbered plane by plane, tile row by tile row, left to right; the
// launch gives XCD x (work-group id mod 8) the x-th contiguous BAND of whole tile rows (StreamK::band_u0)
// and hands the tiles of a band out in order.  Left, upper and upper-left neighbours of a tile inside
// its band therefore have lower work-group ids on the same XCD: they were dispatched earlier, are
// resident or finished, and depend on nothing themselves (a tile publishes before it waits), so the
// poll cannot deadlock; it is bounded all the same (a wrong picture is a failed test, a hang is a
// dead GPU).  Producer and consumer share one L2: plain stores stop there, device-scope loads go past
// the CU's L1 and find them.
// Band boundaries: the first tile row of band x (D tiles) runs at the START of the launch, the last row of
// band x-1 (U tiles) at its END, so there the hand-over runs upwards: a D tile publishes its first two rows
// with device-scope stores (through to memory: the reader sits on another XCD), runs only the vertical
// edge of its rows 2 and 3 in the cells on its upper boundary, and the U tile above -- last of the two by
// construction -- closes the boundary (its rows 30, 31 and the D tile's rows 0, 1) as a 17th cell row.  The same extra
// pass closes the plane's own border cells (k = nh, m = nv) where the plane ends exactly on a tile boundary.
// Order of operations inside every cell: the reference's (state.c:1055-1105), via lf_cell_ops; the
// fragment-row range of the enqueue slot (state.c:1066) is honoured the same way as in k_loopfilter.
#pragma once

#ifndef THIP_TF_PITCH
#define THIP_TF_PITCH 144
#endif
// The geometry of a tile's LDS image and of its edge record, for tiles of BW x 4 blocks: BW = 16 is k_recon_lf's tile (four
// super blocks, one block per lane), BW = 4 k_recon_lf_sb's (one super block, four lanes per block: thip_fused_sb.h).
template <int BW>
struct TfGeom {
  static constexpr int kBW = BW;
  static constexpr int kPitch = BW == 16 ? THIP_TF_PITCH : 8 * BW + 16;   // LDS image row: 8-byte left margin (4 used), 8 BW pixels, 8 spare
  static constexpr int kX0 = 8;                       // byte offset of pixel column 0 in an image row
  static constexpr int kImgRows = 40;                 // pixel rows -4 .. 35 (of the upper neighbour's rows only -2, -1 are filled, of the lower one's 32, 33)
  static constexpr int kFlagOff = kImgRows * kPitch;  // coded flags: 6 rows (block rows -1..4) of kFlagPitch bytes
  static constexpr int kFlagPitch = BW + 4;           // [0] block column -1, [1..BW] the tile, [BW + 1] column BW
  static constexpr int kRowDwords = 2 * BW;           // dwords of one pixel row of the tile
  // a tile's record in StreamK::edge: units of 16 bytes = 3 dwords of pixels + tag
  static constexpr int kBotUnits = (2 * kRowDwords + 2) / 3, kRightUnits = 11;   // two pixel rows; four pixel columns x 32 rows = 128 bytes
  static constexpr int kBot = 0;                                  // pixel rows 30, 31 (row-major), tag flags: right4 << 16 | bottom BW
  static constexpr int kRight = kBotUnits * 16;                   // pixel columns 8 BW - 4 .. 8 BW - 1 (32 rows x 4 bytes), same tag flags
  static constexpr int kTop = (kBotUnits + kRightUnits) * 16;     // pixel rows 0, 1 (tiles that open a band only), tag flags: top BW
  static constexpr int kRec = ((2 * kBotUnits + kRightUnits) * 16 + 127) & ~127;
};
typedef TfGeom<16> Tf16;
constexpr int kTfPitch = Tf16::kPitch, kTfX0 = Tf16::kX0, kTfImgRows = Tf16::kImgRows, kTfFlagOff = Tf16::kFlagOff, kTfFlagPitch = Tf16::kFlagPitch;
// The wave's LDS: 7 KB, not 8.  This chip hands LDS out in 1280-byte granules, so 8 KB costs 8960 bytes and a CU holds 18 such
// waves; 7 KB costs 7680 and it holds the 20 the registers allow.  Seven of a tile's eight 1-KB int16 coefficient pieces are staged
// here (LDS-DMA), the eighth stays in registers; in the levels form the four 1-KB pieces of int8 units (six of the eight of a
// wide tile) and, at kLdsTabOff, the plane's dequantisation tables.
constexpr int kTfLds = 7168;
static_assert(kTfFlagOff + 6 * kTfFlagPitch <= kTfLds, "the image lives in the wave's staging area");
static_assert(kLdsTabOff + 768 <= kTfLds && kLdsTabOff >= 6 * 1024, "the tables sit behind six staged pieces");
constexpr int kTfUnit = 16;
constexpr int kTfBotUnits = Tf16::kBotUnits, kTfRightUnits = Tf16::kRightUnits;
constexpr int kTfBot = Tf16::kBot, kTfRight = Tf16::kRight, kTfTop = Tf16::kTop, kTfRec = Tf16::kRec;
static_assert(kTfBotUnits == 22 && kTfRight == 352 && kTfTop == 528 && kTfRec == 896, "the record layout of rounds 3 and 4");

// A unit goes out with one 16-byte store (through to memory where the reader sits on another XCD) and comes in with one
// 16-byte load that bypasses the CU's L1.  Inline assembly: there is no 16-byte atomic to ask the compiler for, and the
// cache-policy bits are the point.
typedef uint32_t tf_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void tf_store_unit(uint8_t *p, uint4 v, bool through) {
  if (through) {
    const tf_u32x4 w = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(w) : "memory");
  } else {
    *reinterpret_cast<uint4 *>(p) = v;
  }
}
__device__ __forceinline__ uint4 tf_load_unit(const uint8_t *p) {
  tf_u32x4 w;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(w) : "v"(p) : "memory");
  return make_uint4(w.x, w.y, w.z, w.w);
}

// Where dword `idx` of a 4 x 128-byte row block / of a 32-row column lies in the LDS image (byte offsets), given the image
// row of its first row.
template <class G>
__device__ __forceinline__ int tf_rows_at(int idx, int ri0) { return (ri0 + idx / G::kRowDwords) * G::kPitch + G::kX0 + (idx % G::kRowDwords) * 4; }
template <class G>
__device__ __forceinline__ int tf_col_at(int idx, int ri0, int x) { return (ri0 + idx) * G::kPitch + G::kX0 + x; }

// Lanes 0..21 assemble the units of the two pixel rows that start at image row ri0, lanes 22..32 (when `right`) the units of
// the column at pixel x = 124 (tiles of 16 blocks across; 0..5, 6..16 and x = 28 for 4); everything out of the LDS image.
template <class G>
__device__ __forceinline__ void tf_publish_units(const uint8_t *lds, uint8_t *rec_rows, uint8_t *rec_right, int ri0, bool right, uint32_t tag,
                                                 int lane, bool through) {
  uint32_t d[3] = {0u, 0u, 0u};
  uint8_t *dst = nullptr;
  if (lane < G::kBotUnits) {
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const int idx = 3 * lane + j;
      if (idx < 2 * G::kRowDwords) d[j] = *reinterpret_cast<const uint32_t *>(lds + tf_rows_at<G>(idx, ri0));
    }
    dst = rec_rows + lane * kTfUnit;
  } else if (right && lane < G::kBotUnits + G::kRightUnits) {
    const int v = lane - G::kBotUnits;
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const int idx = 3 * v + j;
      if (idx < 32) d[j] = *reinterpret_cast<const uint32_t *>(lds + tf_col_at<G>(idx, 4, 8 * G::kBW - 4));
    }
    dst = rec_right + v * kTfUnit;
  }
  if (dst) tf_store_unit(dst, make_uint4(d[0], d[1], d[2], tag), through);
}

// The consumer's side: every lane with a source loads its unit and looks again until its tag carries the launch's serial
// number (bounded: see tf header; `fault` is the state's pinned host word: the host notices it at its next synchronising call
// and decodes the frame again with the two passes, thip_decode.hip: recover_fault).  Returns the unit.
__device__ __forceinline__ uint4 tf_fetch_unit(const uint8_t *src, uint32_t ep, uint32_t *fault, int max_spins, uint32_t fault_id) {
  uint4 v = make_uint4(0u, 0u, 0u, 0u);
  bool ok = src == nullptr;
  for (int spins = 0; spins < max_spins; spins++) {
    if (!ok) {
      v = tf_load_unit(src);
      ok = (v.w >> 20) == ep;
    }
    if (__all(ok)) break;
    __builtin_amdgcn_s_sleep(1);
  }
  // WHICH launch it was goes into one of eight words behind the flag (fault_id = the launch's serial number, 0x1000 added by
  // k_recon_lf_sb, whose buffer counts on its own): the host decodes a frame again only if every launch that reports a failed
  // wait is the frame it can still repeat -- a frame launched on top of a failed one is not made right by repeating it.
  // (Eight words, ids modulo eight: two failing launches eight serial numbers apart -- eight frames decoded without a synchronising
  //  call -- share a word.  Whoever finds another launch's id in its word says so in word 10, and the host then repeats nothing:
  //  ADVICE r05.)
  if (!ok && fault) {
    const uint32_t before = __hip_atomic_load(fault + 1 + (fault_id & 7u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (before != 0u && before != fault_id) __hip_atomic_store(fault + 10, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(fault + 1 + (fault_id & 7u), fault_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  return v;
}

// One filter cell out of the LDS image: corner column kx (0..16) of the tile, cell row m (0..4).  Of its eight rows the
// cell stores [r_lo, r_hi) -- a tile hands only its last TWO pixel rows down, so a cell on the tile's upper boundary owns its
// rows 2..7 (6, 7 where the tile above belongs to another band and closes the boundary itself) and applies `opmask` of its
// operations; `ext` cells (cell row 3) own the two rows below them as well, pixel rows 28 and 29 of the tile: nothing touches
// those but the vertical edge of the cell's lower half (Vhi, one more row pair of the same operation), so they need not wait
// for the tile below.
template <class G>
__device__ __forceinline__ void tf_cell(const uint8_t *lds, uint8_t *plane, int stride, int nh, int nv, int t, int sby, int kx, int m,
                                        bool active, int L2, int fy0, int fy1, int r_lo, int r_hi, uint32_t opmask, bool ext) {
  constexpr int kTfPitch = G::kPitch, kTfX0 = G::kX0, kTfFlagOff = G::kFlagOff, kTfFlagPitch = G::kFlagPitch;
  const int k = G::kBW * t + kx, mm = 4 * sby + m;
  active = active && k <= nh && mm <= nv;
  CellPix C;
  const uint8_t *img = lds + (8 * m) * kTfPitch + kTfX0 + 8 * kx - 4;   // image row index = pixel row + 4
#pragma unroll
  for (int r = 0; r < 8; r++) {
    const uint32_t *q = reinterpret_cast<const uint32_t *>(img + r * kTfPitch);
    C.lo[r] = q[0];
    C.hi[r] = q[1];
  }
  uint32_t xlo[2] = {0u, 0u}, xhi[2] = {0u, 0u};
  const bool any_ext = __any(ext);
  if (any_ext) {
#pragma unroll
    for (int r = 0; r < 2; r++) {
      const uint32_t *q = reinterpret_cast<const uint32_t *>(img + (8 + r) * kTfPitch);
      xlo[r] = q[0];
      xhi[r] = q[1];
    }
  }
  const uint8_t *fl = lds + kTfFlagOff + m * kTfFlagPitch + kx;   // flag of block (kx-1, m-1)
  const bool a = fl[0] != 0, b = fl[1] != 0, c = fl[kTfFlagPitch] != 0, d = fl[kTfFlagPitch + 1] != 0;
  uint32_t ops = lf_cell_ops(k, mm, nh, nv, a, b, c, d, fy0, fy1) & opmask;
  if (L2 == 0 || !active) ops = 0;
  lf_cell_apply_pk(C, ops, L2);
  if (any_ext) {
    const bool vx = ext && (ops & 96u) != 0;
    if (__any(vx)) { if (vx) lf_vert_pair(xlo[0], xlo[1], xhi[0], xhi[1], L2); }
  }
  const bool lo_ok = active && k >= 1, hi_ok = active && k <= nh - 1;
  const bool up_ok = mm >= 1, dn_ok = mm <= nv - 1;
  uint8_t *base = plane + (ptrdiff_t)(8 * mm - 4) * stride + (8 * k - 4);
  // the rows this lane stores, one bit each (0..7 the cell, 8 and 9 the two rows below it)
  uint32_t rows = (0xFFu >> (8 - r_hi)) & (0xFFu << r_lo) & ((up_ok ? 0x0Fu : 0u) | (dn_ok ? 0xF0u : 0u));
  if (ext && dn_ok) rows |= 0x300u;
  // Three cases, each entered by the wave once: both halves (every lane away from the plane's left and right border), the
  // left half only, the right half only.
  if (lo_ok && hi_ok) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
      if (r >= 8 && !any_ext) break;
      if (rows >> r & 1u) {
        Pix8 o;
        o.x = r < 8 ? C.lo[r] : xlo[r - 8];
        o.y = r < 8 ? C.hi[r] : xhi[r - 8];
        *reinterpret_cast<Pix8 *>(base + (ptrdiff_t)r * stride) = o;
      }
    }
  } else if (lo_ok) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
      if (r >= 8 && !any_ext) break;
      if (rows >> r & 1u) *reinterpret_cast<uint32_t *>(base + (ptrdiff_t)r * stride) = r < 8 ? C.lo[r] : xlo[r - 8];
    }
  } else if (hi_ok) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
      if (r >= 8 && !any_ext) break;
      if (rows >> r & 1u) *reinterpret_cast<uint32_t *>(base + (ptrdiff_t)r * stride + 4) = r < 8 ? C.hi[r] : xhi[r - 8];
    }
  }
}

// The same cell when the whole wave knows the answer to every question tf_cell asks (k_recon_lf decides, wave-uniformly): the tile
// touches no border of its plane and no band boundary above it, every block of the tile and of the neighbours' edges is coded, the
// filter is on and the whole tile lies inside the fragment-row range.  Then every cell applies T3, T5, T7, T8 of lf_cell_ops'
// list -- the vertical edge of its upper half, the horizontal edge's left half, the vertical edge of its lower half, the horizontal
// edge's right half, in that order (state.c:1083-1104 with all four blocks coded) -- and stores both halves of its rows: no flag
// bytes, no per-lane operation word, no exec masks but the two that follow from the cell row (m = 0: rows 2..7; m = 3: rows 8, 9 too).
template <class G>
__device__ __forceinline__ void tf_cell_all_coded(const uint8_t *lds, uint8_t *plane, int stride, int t, int sby, int kx, int m, int L2) {
  constexpr int kTfPitch = G::kPitch, kTfX0 = G::kX0;
  const int k = G::kBW * t + kx, mm = 4 * sby + m;
  CellPix C;
  const uint8_t *img = lds + (8 * m) * kTfPitch + kTfX0 + 8 * kx - 4;
#pragma unroll
  for (int r = 0; r < 8; r++) {
    const uint32_t *q = reinterpret_cast<const uint32_t *>(img + r * kTfPitch);
    C.lo[r] = q[0];
    C.hi[r] = q[1];
  }
  uint32_t xlo[2], xhi[2];
#pragma unroll
  for (int r = 0; r < 2; r++) {
    const uint32_t *q = reinterpret_cast<const uint32_t *>(img + (8 + r) * kTfPitch);
    xlo[r] = q[0];
    xhi[r] = q[1];
  }
  lf_vert_pk(C, 0, L2);
  lf_horz_pk(C, 0, L2);
  lf_vert_pk(C, 4, L2);
  lf_horz_pk(C, 1, L2);
  lf_vert_pair(xlo[0], xlo[1], xhi[0], xhi[1], L2);   // (rows 28, 29 of the tile: stored by cell row 3 only)
  uint8_t *base = plane + (ptrdiff_t)(8 * mm - 4) * stride + (8 * k - 4);
  if (m != 0) {
#pragma unroll
    for (int r = 0; r < 2; r++) {
      Pix8 o;
      o.x = C.lo[r];
      o.y = C.hi[r];
      *reinterpret_cast<Pix8 *>(base + (ptrdiff_t)r * stride) = o;
    }
  }
#pragma unroll
  for (int r = 2; r < 8; r++) {
    Pix8 o;
    o.x = C.lo[r];
    o.y = C.hi[r];
    *reinterpret_cast<Pix8 *>(base + (ptrdiff_t)r * stride) = o;
  }
  if (m == 3) {
#pragma unroll
    for (int r = 0; r < 2; r++) {
      Pix8 o;
      o.x = xlo[r];
      o.y = xhi[r];
      *reinterpret_cast<Pix8 *>(base + (ptrdiff_t)(8 + r) * stride) = o;
    }
  }
}

// Experiment hooks: the wave's issue priority by phase (0 start, 1 loads out, 2 pixels done, 3 cells).  THIP_PRIO_SEQ is four digits,
// e.g. 0x3003: priority 3 until the loads are out, 0 through the transforms and the hand-over, 3 for the cells.  Not defined: nothing.
#ifdef THIP_PRIO_SEQ
#define THIP_PRIO_AT(i) __builtin_amdgcn_s_setprio((THIP_PRIO_SEQ >> (12 - 4 * (i))) & 3)
#else
#define THIP_PRIO_AT(i) do { } while (0)
#endif
#ifndef THIP_TF_WAVES_PER_EU
#define THIP_TF_WAVES_PER_EU 5    // 96 VGPRs; with 8 KB of LDS per wave that is 20 waves per CU
#endif
template <bool LEVELS>
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
  const uint4 *dq_p = S.dequant;
  uint8_t *edge_p = S.edge;
  const int te0 = S.tile_end[0], te1 = S.tile_end[1];
  const int sqpx = S.qpx, sqpy = S.qpy;
  const int L2 = (S.debug & 256) ? 0 : S.flimit2;   // (ablation switch for profiling, option debug = 256: cells copy, no filtering)
  // (test switch, option debug = 512: tile 1 of every stream tags its units with the WRONG serial number, so that its neighbours'
  //  waits run out -- quickly -- and the host's recovery can be exercised: tests/test_gpu_frames.py::test_a_failed_hand_over_is_decoded_again)
  const bool poison = (S.debug & 512) != 0;
  const int max_spins = poison ? 256 : (1 << 20);
  uint32_t *fault_p = S.fault;
  const uint32_t ep = S.epoch;
  const int bu0 = S.band_u0[band], bu1 = S.band_u0[band + 1];
  asm volatile("" ::"s"(info_p), "s"(coeffs_p), "s"(slot0_p), "s"(self), "s"(prev), "s"(gold), "s"(coded_map), "s"(dc_p), "s"(dq_p), "s"(edge_p),
               "s"(te0), "s"(te1), "s"(sqpx), "s"(sqpy), "s"(L2), "s"(ep), "s"(bu0), "s"(bu1), "s"(fault_p));
  const int u = bu0 + jb;
  if (u >= bu1) return;
  // the tile's plane and its place in it (tiles across from StreamK::spec_tx: the plane's record is not at hand yet)
  const int pli = (u >= te0 ? 1 : 0) + (u >= te1 ? 1 : 0);
  const int rel = u - (pli == 0 ? 0 : (pli == 1 ? te0 : te1));
  const int tiles_x = S.spec_tx[pli];
  const int sby = rel / tiles_x, t = rel - sby * tiles_x;
  // ---- 0. (levels form, a frame with a unit for every block) the coefficients asked for NOW: the tile's first unit follows from
  //      its place in the plane wherever its tile row is whole, and what comes back is checked against the first-slot word when that
  //      arrives -- a wave's second round trip, 3 of its 15 us, is gone for every tile the guess is right for; a wrong guess (and
  //      the ragged tile at the end of a row, whose blocks do not all own a unit) costs its 4 KB and nothing else: the real loads
  //      land behind it, in order, in the same place
  bool spec = false;
  uint32_t spec_slot0 = 0;
  if (LEVELS && S.spec_on) {
    if (sby < S.spec_rows[pli]) {
      spec = true;
      spec_slot0 = S.spec_base[pli] + (uint32_t)(sby * S.spec_rowunits[pli]) + 64u * (uint32_t)t;
      const uint32_t unit = min(spec_slot0 + (uint32_t)lane, S.spec_last);
#pragma unroll
      for (int q = 0; q < 4; q++)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)unit_piece(coeffs_p, unit, q),
                                         (__attribute__((address_space(3))) void *)(s_tf + q * 64), 16, 0, THIP_COEF_CPOL);
    }
  }
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
  THIP_PRIO_AT(0);
  constexpr bool levels = LEVELS;   // (one kernel per coefficient form: each is straight-line code for its own)
  if (levels) tables_to_lds(dq_p, pli, lane, s_tf);   // (first: whoever has its command word has the tables)
  const PlaneK G = S.pl[pli];
  const int fy0 = S.lf_y0[pli], fy1 = S.lf_y1[pli];
  asm volatile("" ::"s"(G.nh), "s"(G.nv), "s"(G.stride), "s"(G.off), "s"(G.tiles_x), "s"(G.tiles_y), "s"(G.fro), "s"(fy0), "s"(fy1));
  const int nh = G.nh, nv = G.nv;
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
  const uint32_t slot0w = slot0_p[u];
  const uint2 info = info_p[(size_t)u * THIP_TILE_FRAGS + lane];
  uint32_t dcv = 0;   // the block's un-predicted DC when it does not travel in the command stream
  if (dc_p) dcv = 0x10000u | (uint16_t)dc_p[G.fro + min(by, nv - 1) * nh + min(bx, nh - 1)];
  asm volatile("" ::"s"(slot0w), "v"(info.x), "v"(dcv));
#ifdef THIP_TRACE
  THIP_TR(tr, 1);   // command words are here
#endif
  if (levels && !dc_p) dcv = 0x10000u | (info.y & 0xFFFFu);   // levels form: every coded block's raw DC rides in command word 1
  const CoefForm F = coef_form(slot0w, levels);
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
  const bool own_all_coded = __all(valid && L.coded);   // (cells' fast path, step 5)
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
    THIP_PRIO_AT(1);
  } else if (nown <= 16) {
    int4 Wc[1][2];
    residual_shared_load<4>(coeffs_p, F, nown, lane, Wc);
    if (valid) recon_issue(R, L, Q, inter, ref);
    THIP_PRIO_AT(1);
    residual_shared<4>(Wc, F, lds_dw, meta, lane, L, prefix, Y);
  } else if (nown <= 32) {
    int4 Wc[2][2];
    residual_shared_load<2>(coeffs_p, F, nown, lane, Wc);
    if (valid) recon_issue(R, L, Q, inter, ref);
    THIP_PRIO_AT(1);
    residual_shared<2, true>(Wc, F, lds_dw, meta, lane, L, prefix, Y);
  } else {
    int4 w7 = make_int4(0, 0, 0, 0);
    // (the units asked for in step 0 are this tile's: all 64 blocks own one, in lane order from the guessed first unit)
    const bool spec_hit = spec && nown == 64 && !F.wide && F.slot0 == spec_slot0;
    if (!spec_hit) dense_issue<7>(coeffs_p, F, L.has_coeff, prefix, s_tf, w7);
    if (valid) recon_issue(R, L, Q, inter, ref);
    THIP_PRIO_AT(1);
    // LDS-DMA data has landed (see k_recon) -- on a hit it has landed already: the guess was asked for before the command words, loads
    // come back in order, and the command words are here; the transform then runs UNDER the predictor windows' round trip instead
    // of behind it (what the guess is worth where the chip is nearly empty and that round trip is as long as the coefficients')
    if (!spec_hit) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    dense_finish<7>(coeffs_p, F, L.has_coeff, prefix, s_tf, lane, L, w7, Y);
  }
  if (!L.has_coeff) {
#pragma unroll
    for (int i = 0; i < 32; i++) Y[i] = fill;
  }
#ifdef THIP_TRACE
  asm volatile("" : "+v"(Y[0]), "+v"(Y[31]));
  THIP_TR(tr, 11);  // residual there (coefficients had arrived, transform done); the predictor windows may still be on their way
#endif
  uint2 rows[8];
  recon_rows(R, Q, inter, Y, rows, L.coded ? L.flags : 0u, L.x0);
#ifdef THIP_TRACE
  asm volatile("" : "+v"(rows[0].x), "+v"(rows[7].y));
  THIP_TR(tr, 2);   // pixels done (coefficients and predictor windows had arrived)
#endif

  THIP_PRIO_AT(2);
  // ---- 3. the tile image into LDS, its edges out as units ---------------------------------------------------------
  uint8_t *const myrec = edge_p + (size_t)u * kTfRec;
  const uint8_t *const rec_up = myrec - (ptrdiff_t)tiles_x * kTfRec, *const rec_left = myrec - kTfRec;
  const uint8_t *const rec_ul = rec_up - kTfRec;
  const bool need_ul = up_in && has_left;
  lds_settle();                                   // every lane is done with the staging area
  if (valid) {
    uint8_t *img = lds + (ly * 8 + 4) * kTfPitch + kTfX0 + lx * 8;
#pragma unroll
    for (int r = 0; r < 8; r++) *reinterpret_cast<uint2 *>(img + r * kTfPitch) = rows[r];
  }
  lds[kTfFlagOff + (ly + 1) * kTfFlagPitch + lx + 1] = (valid && L.coded) ? 1 : 0;
  lds_settle();
  {
    // the coded flags of the published edges, one ballot: bits 0..15 block row 3, 16..19 block column 15, 32..47 block row 0
    const int fi = lane < 16 ? 4 * kTfFlagPitch + lane + 1
                             : (lane < 20 ? (lane - 16 + 1) * kTfFlagPitch + 16 : (lane >= 32 && lane < 48 ? kTfFlagPitch + (lane - 32) + 1 : 0));
    const bool fb = (lane < 20 || (lane >= 32 && lane < 48)) && lds[kTfFlagOff + fi] != 0;
    const uint64_t fm = __ballot(fb);
    const uint32_t tag_ep = (poison && u == 1) ? 0u : ep;   // (serial numbers are never 0)
    tf_publish_units<Tf16>(lds, myrec + kTfBot, myrec + kTfRight, 34, true, tag_ep << 20 | (uint32_t)(fm & 0xFFFFFu), lane, xb_up);
    if (xb_up) tf_publish_units<Tf16>(lds, myrec + kTfTop, nullptr, 4, false, tag_ep << 20 | ((uint32_t)(fm >> 32) & 0xFFFFu), lane, true);
    THIP_TR(tr, 3);   // image in LDS, edges on their way
    THIP_TR(tr, 4);
  }

  bool nb_all_coded = false;   // every block on the upper, left and upper-left neighbours' edges is coded (their tags)
  // ---- 4. the neighbours' edges into the image margins: lanes 0..21 the upper tile's rows 30, 31, 22..32 the left tile's
  //         columns 124..127, 33 the upper-left tile's corner (dwords 30, 31 of its column: unit 10) ------------------------
  {
    const uint8_t *src = nullptr;
    if (lane < kTfBotUnits) {
      if (up_in) src = rec_up + kTfBot + lane * kTfUnit;
    } else if (lane < kTfBotUnits + kTfRightUnits) {
      if (has_left) src = rec_left + kTfRight + (lane - kTfBotUnits) * kTfUnit;
    } else if (lane == kTfBotUnits + kTfRightUnits) {
      if (need_ul) src = rec_ul + kTfRight + 10 * kTfUnit;
    }
    const uint4 un = tf_fetch_unit(src, ep, fault_p, max_spins, ep);
    THIP_TR(tr, 5);   // the neighbours' units are there
    const uint32_t d[3] = {un.x, un.y, un.z};
    if (lane < kTfBotUnits) {
#pragma unroll
      for (int j = 0; j < 3; j++) {
        const int idx = 3 * lane + j;
        if (idx < 64) *reinterpret_cast<uint32_t *>(lds + tf_rows_at<Tf16>(idx, 2)) = d[j];
      }
    } else if (lane < kTfBotUnits + kTfRightUnits) {
#pragma unroll
      for (int j = 0; j < 3; j++) {
        const int idx = 3 * (lane - kTfBotUnits) + j;
        if (idx < 32) *reinterpret_cast<uint32_t *>(lds + tf_col_at<Tf16>(idx, 4, -4)) = d[j];
      }
    } else if (lane == kTfBotUnits + kTfRightUnits) {
#pragma unroll
      for (int j = 0; j < 2; j++) *reinterpret_cast<uint32_t *>(lds + tf_col_at<Tf16>(2 + j, 0, -4)) = d[j];   // dwords 30, 31
    }
    // their coded flags (the tags of the first unit of each record): block row -1 (columns 0..15), block column -1 (rows 0..3),
    // block (-1, -1)
    const uint32_t w_up = (uint32_t)__builtin_amdgcn_readlane((int)un.w, 0);
    const uint32_t w_left = (uint32_t)__builtin_amdgcn_readlane((int)un.w, kTfBotUnits);
    const uint32_t w_ul = (uint32_t)__builtin_amdgcn_readlane((int)un.w, kTfBotUnits + kTfRightUnits);
    nb_all_coded = (w_up & 0xFFFFu) == 0xFFFFu && (w_left & 0xF0000u) == 0xF0000u && (w_ul & 0x80000u) != 0;
    if (lane < 16)
      lds[kTfFlagOff + lane + 1] = (uint8_t)((w_up >> lane) & 1u);
    else if (lane < 20)
      lds[kTfFlagOff + (lane - 16 + 1) * kTfFlagPitch] = (uint8_t)((w_left >> lane) & 1u);
    else if (lane == 20)
      lds[kTfFlagOff] = (uint8_t)((w_ul >> 19) & 1u);
    lds_settle();
    THIP_TR(tr, 6);   // ... and in the margins
  }

  THIP_PRIO_AT(3);
  // ---- 5. the cells: lane (kx, m) on corner (16t + kx, 4 sby + m) ---------------------------------------------
  // (cell row 0 below a tile of this band: rows 2..7, the two above them are the upper tile's; below another band's tile: only
  //  the lower half's vertical edge, rows 6 and 7 -- the upper tile closes the rest as its 17th..20th pixel rows, see 6.)
  const int top_lo = has_up ? (xb_up ? 6 : 2) : 0;
  const uint32_t top_mask = xb_up ? 96u : 0xFFu;
  {
    const int m = lane >> 4;
#ifndef THIP_NO_CELL_FAST
    // (wave-uniform, scalar: see tf_cell_all_coded)
    const bool plain = own_all_coded && nb_all_coded && L2 != 0 && has_left && !row_end && up_in && 4 * sby + 3 <= nv - 1 &&
                       fy0 <= 4 * sby - 1 && fy1 >= 4 * sby + 4;
    if (plain)
      tf_cell_all_coded<Tf16>(lds, R.self, R.stride, t, sby, lane & 15, m, L2);
    else
#endif
      tf_cell<Tf16>(lds, R.self, R.stride, nh, nv, t, sby, lane & 15, m, true, L2, fy0, fy1, m == 0 ? top_lo : 0, 8, m == 0 ? top_mask : 0xFFu, m == 3);
  }
  THIP_TR(tr, 7);   // cells filtered, stores issued

  // ---- 6. a 17th cell column where the plane ends on this tile's right boundary (k = nh), a 5th cell row where the plane
  //         ends on its lower boundary (m = nv) or where the tile below belongs to another band ----------------------------
  const bool extra_col = row_end && (nh & 15) == 0;
  const bool extra_row = xb_dn || (!has_dn && (nv & 3) == 0);
  if (extra_col || extra_row) {
    if (xb_dn) {
      // the tile below ran at the start of the launch: its first two rows (lanes 0..21) and the lower-left tile's corner
      // (dwords 0, 1 of its column: unit 0, lane 22), into image rows 36, 37
      const uint8_t *const rec_dn = myrec + (ptrdiff_t)tiles_x * kTfRec, *const rec_dl = rec_dn - kTfRec;
      const uint8_t *src = nullptr;
      if (lane < kTfBotUnits) src = rec_dn + kTfTop + lane * kTfUnit;
      else if (lane == kTfBotUnits && has_left) src = rec_dl + kTfRight;
      const uint4 un = tf_fetch_unit(src, ep, fault_p, max_spins, ep);
      const uint32_t d[3] = {un.x, un.y, un.z};
      if (lane < kTfBotUnits) {
#pragma unroll
        for (int j = 0; j < 3; j++) {
          const int idx = 3 * lane + j;
          if (idx < 64) *reinterpret_cast<uint32_t *>(lds + tf_rows_at<Tf16>(idx, 36)) = d[j];
        }
      } else if (lane == kTfBotUnits) {
#pragma unroll
        for (int j = 0; j < 2; j++) *reinterpret_cast<uint32_t *>(lds + tf_col_at<Tf16>(j, 36, -4)) = d[j];
      }
      // block row 4: the lower tile's first block row (its top units' tags), the lower-left tile's block (15, 0)
      const uint32_t w_dn = (uint32_t)__builtin_amdgcn_readlane((int)un.w, 0);
      const uint32_t w_dl = (uint32_t)__builtin_amdgcn_readlane((int)un.w, kTfBotUnits);
      if (lane < 16)
        lds[kTfFlagOff + 5 * kTfFlagPitch + lane + 1] = (uint8_t)((w_dn >> lane) & 1u);
      else if (lane == 16)
        lds[kTfFlagOff + 5 * kTfFlagPitch] = (uint8_t)((w_dl >> 16) & 1u);
      lds_settle();
    }
    // lanes 0..16: cells (lane, 4); lanes 32..35: cells (16, lane - 32)
    const bool rowl = lane <= 16, coll = lane >= 32 && lane < 36;
    const int kx = rowl ? lane : 16, m = rowl ? 4 : (lane - 32) & 3;
    const bool act = rowl ? (extra_row && (lane < 16 || extra_col)) : (coll && extra_col);
    // row cells: pixel rows 30, 31 and, of another band's tile below, its rows 0 and 1 (28 and 29 went out with cell row 3)
    tf_cell<Tf16>(lds, R.self, R.stride, nh, nv, t, sby, kx, m, act, L2, fy0, fy1, rowl ? 2 : (m == 0 ? top_lo : 0), rowl ? (xb_dn ? 6 : 8) : 8,
            (!rowl && m == 0) ? top_mask : 0xFFu, !rowl && m == 3);
  }
  THIP_TR(tr, 8);
}
