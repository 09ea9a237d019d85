// thip_costmaps.h -- the encoder's per-macro-block cost maps for a WHOLE frame in one launch:
// oc_mb_intra_satd (analyze.c:1360-1403), oc_mb_activity (:1152-1237) and oc_mb_activity_fast (:1239-1251), which the
// reference computes macro block by macro block inside its mode-decision loop (:1697-1718, :2372-2398) although they depend
// on the input picture alone.  Included by thip_slots.hip.
//
// One 8x8 block per lane.  A luma lane loads the ten rows around its block ONCE -- 16 aligned bytes each, coordinates clamped
// into the plane, which is what the replicated border of the encoder's input frame amounts to (encode.c:1735-1744) -- and takes
// everything from them: the intra SATD and its DC (= the pixel sum oc_mb_intra_satd adds up as `luma`), sum and sum of squares
// (v_sad_u8, v_dot4_u32_u8), and, for the blocks that are not flat, the four directional edge energies of oc_mb_activity on
// packed 16-bit column pairs.  With V = up + 2 mid + dn and D = dn - up (column vectors of a row triple) the four sums are
//   e1 = sum |V[c+1] - V[c-1]|                e2 = sum |D[c-1] + 2 D[c] + D[c+1]|
//   e3 = sum |2 (dn[c+1] - up[c-1]) + dn[c] - mid[c-1] + mid[c+1] - up[c]|
//   e4 = sum |2 (dn[c-1] - up[c+1]) + dn[c] - mid[c+1] + mid[c-1] - up[c]|
// over the block's 64 positions (analyze.c:1209-1221 with s = block - 1), every term a packed add/sub between a row's "aligned"
// pairs R = {c-1,c0},{c1,c2},... and its "shifted" pairs S = {c0,c1},{c2,c3},...; the logarithm / exponential of the edge
// classification are the reference's integer polynomials (mathops.c:294-314).  A chroma lane computes its block's intra SATD.
// Results go to the macro block's entry in the reference's numbering: mbi = super block << 2 | quadrant (state.c:300-330), luma
// blocks in sb_maps order (state.c:134-139), chroma blocks in OC_MB_MAP_IDXS order (internal.c:67-76).
#pragma once

namespace thip {

struct CostMapK {
  const uint8_t *plane[3];
  int stride[3];
  int nh[3], nv[3];        // blocks across / down
  int nsbw;                // luma super blocks across
  int hdec, vdec, fmt;
  int n_luma, n_all;       // lanes: [0, n_luma) luma blocks (macro block by macro block, four lanes each), [n_luma, n_all) Cb then Cr blocks
  int n_pad;               // ... [n_all, n_all + n_pad): one lane per macro-block SLOT, zeroing the slots outside the frame (0: there are none)
  uint32_t *intra_satd;    // [nmbs][12], may be null
  uint32_t *luma;          // [nmbs], may be null
  uint32_t *activity;      // [nmbs][4], may be null
  uint32_t *activity_fast; // [nmbs][4], may be null
};

// mathops.c:294-314
__device__ __forceinline__ uint32_t cm_bexp32_q10(int z) {
  const int ipart = z >> 10;
  uint32_t n = (uint32_t)(z & 1023) << 4;
  n = (n * ((n * ((n * ((n * 3548u >> 15) + 6817u) >> 15) + 15823u) >> 15) + 22708u) >> 15) + 16384u;
  return 14 - ipart > 0 ? (n + (1u << (13 - ipart))) >> (14 - ipart) : n << (ipart - 14);
}
__device__ __forceinline__ int cm_blog32_q10(uint32_t w) {
  if (w == 0) return -1;
  const int ipart = 32 - __builtin_clz(w);
  const int n = (int)(ipart - 16 > 0 ? w >> (ipart - 16) : w << (16 - ipart)) - 32768 - 16384;
  const int fpart = ((n * ((n * ((n * ((n * -1402 >> 15) + 2546) >> 15) - 5216) >> 15) + 15745)) >> 15) - 6793;
  return (ipart << 10) + (fpart >> 4);
}

// One source row of a luma block as twelve bytes N = columns -1 .. 10 of the block (clamped into the plane): the 16-byte
// aligned window that starts at xw, shifted so that column -1 sits in byte 0.  Three cases: the window starts 4 pixels left of
// the block (interior), AT the block (x0 = 0: column -1 is column 0 again), or 8 pixels left of it (x0 = W - 8: column 8 is
// column 7 again).
struct CmRow {
  uint32_t n0, n1, n2;
};
__device__ __forceinline__ CmRow cm_row(const uint8_t *row_xw, int edge /* 0 interior, 1 left, 2 right */) {
  uint4 w;   // (the caller's planes: no alignment beyond the byte is promised)
  __builtin_memcpy(&w, row_xw, 16);
  uint32_t a0, a1, a2, a3;
  if (edge == 1) {
    a0 = w.x << 24;
    a1 = w.x;
    a2 = w.y;
    a3 = w.z;
  } else if (edge == 2) {
    a0 = w.y;
    a1 = w.z;
    a2 = w.w;
    a3 = w.w >> 24;
  } else {
    a0 = w.x;
    a1 = w.y;
    a2 = w.z;
    a3 = w.w;
  }
  CmRow r;
  r.n0 = __builtin_amdgcn_alignbyte(a1, a0, 3u);
  r.n1 = __builtin_amdgcn_alignbyte(a2, a1, 3u);
  r.n2 = __builtin_amdgcn_alignbyte(a3, a2, 3u);
  return r;
}
// the pairs of a row as 16-bit lanes: R[k] = {c(2k-1), c(2k)}, k = 0..4; S[k] = {c(2k), c(2k+1)}, k = 0..3
__device__ __forceinline__ void cm_pairs(const CmRow &r, pk16 R[5], pk16 S[4]) {
  R[0] = as_pk(__builtin_amdgcn_perm(0u, r.n0, 0x0c010c00u));
  R[1] = as_pk(__builtin_amdgcn_perm(0u, r.n0, 0x0c030c02u));
  R[2] = as_pk(__builtin_amdgcn_perm(0u, r.n1, 0x0c010c00u));
  R[3] = as_pk(__builtin_amdgcn_perm(0u, r.n1, 0x0c030c02u));
  R[4] = as_pk(__builtin_amdgcn_perm(0u, r.n2, 0x0c010c00u));
  S[0] = as_pk(__builtin_amdgcn_perm(0u, r.n0, 0x0c020c01u));
  S[1] = as_pk(__builtin_amdgcn_perm(r.n1, r.n0, 0x0c040c03u));
  S[2] = as_pk(__builtin_amdgcn_perm(0u, r.n1, 0x0c020c01u));
  S[3] = as_pk(__builtin_amdgcn_perm(r.n2, r.n1, 0x0c040c03u));
}
__device__ __forceinline__ uint32_t cm_abs_sum(pk16 v, uint32_t acc) { return sum2_u16(pk_abs(v), acc); }

// where a block's results go: luma block (bx, by) -> (mbi, entry of sb_maps); SB_MAP of state.c:134-139 as two nibble tables
__device__ __forceinline__ void cm_luma_slot(int bx, int by, int nsbw, uint32_t &mbi, int &bi) {
  // position on the 4x4 Hilbert curve of block (row, column) of a super block, four bits per entry: quadrant = h >> 2, entry of
  // the quadrant's sb_map = h & 3 (SB_MAP of state.c:134-139 lists exactly these pairs)
  constexpr uint64_t kHilb = 0ull | 1ull << 4 | 14ull << 8 | 15ull << 12 | 3ull << 16 | 2ull << 20 | 13ull << 24 | 12ull << 28 |
                             4ull << 32 | 7ull << 36 | 8ull << 40 | 11ull << 44 | 5ull << 48 | 6ull << 52 | 9ull << 56 | 10ull << 60;
  const int hidx = (int)((kHilb >> (4 * ((by & 3) * 4 + (bx & 3)))) & 15ull);
  mbi = (uint32_t)((by >> 2) * nsbw + (bx >> 2)) << 2 | (uint32_t)(hidx >> 2);
  bi = hidx & 3;
}

// ONE launch writes every entry of the four maps (round 6; the call used to be four hipMemsetAsync + the kernel, 14.8 us for a 7.9 us
// kernel): the four luma blocks of a macro block sit in four neighbouring lanes, which add their DC terms among themselves (no
// atomics into a zeroed array) and whose first lane zeroes the chroma entries its pixel format leaves unused; the macro-block slots
// a frame does not fill (widths or heights that are not a multiple of 32: the maps are indexed by super block) are zeroed by lanes
// of their own.
__global__ __launch_bounds__(256) void k_enc_cost_maps(const CostMapK K) {
  const int u = (int)(blockIdx.x * 256 + threadIdx.x);
  if (u >= K.n_all + K.n_pad) return;
  if (u >= K.n_all) {
    // ---- a macro-block slot: zero it when its macro block lies outside the frame (OC_MB_MAP, internal.c:63: quadrant -> place) ----
    const int slot = u - K.n_all, sb = slot >> 2, quad = slot & 3;
    const int sby = sb / K.nsbw, sbx = sb - sby * K.nsbw;
    const int mx = quad >> 1, my = (quad == 1 || quad == 2) ? 1 : 0;      // 0: (0,0)  1: (0,1)  2: (1,1)  3: (1,0)   (x, y)
    if (sbx * 4 + mx * 2 < K.nh[0] && sby * 4 + my * 2 < K.nv[0]) return;
    if (K.intra_satd) {
#pragma unroll
      for (int k = 0; k < 12; k++) K.intra_satd[(size_t)slot * 12 + k] = 0u;
    }
    if (K.luma) K.luma[slot] = 0u;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (K.activity) K.activity[(size_t)slot * 4 + k] = 0u;
      if (K.activity_fast) K.activity_fast[(size_t)slot * 4 + k] = 0u;
    }
    return;
  }
  if (u >= K.n_luma) {
    // ---- a chroma block: its intra SATD into the macro block's entry (analyze.c:1393-1400) ---------------------------------
    int v = u - K.n_luma;
    const int nc = K.nh[1] * K.nv[1];
    const int pli = v >= nc ? 2 : 1;
    v -= (pli - 1) * nc;
    const int cy = v / K.nh[1], cx = v - cy * K.nh[1];
    uint2 s[8];
    load_rows8(s, K.plane[pli] + (ptrdiff_t)(cy * 8) * K.stride[pli] + cx * 8, K.stride[pli]);
    pk16 D[8][4];
#pragma unroll
    for (int r = 0; r < 8; r++) row_sd(D[r], s[r]);
    int dc;
    const uint32_t satd = satd_sd(D, dc);
    // the macro block of this block and the block's place in it (state.c:200-290: mb_map[pli][i << 1 | j])
    const int mbx = (cx << K.hdec) & ~1, mby = (cy << K.vdec) & ~1;
    const int ci = K.vdec ? 0 : cy & 1, cj = K.hdec ? 0 : cx & 1;
    const int quad = ((mby >> 1) & 1) ? (((mbx >> 1) & 1) ? 2 : 1) : (((mbx >> 1) & 1) ? 3 : 0);   // OC_MB_MAP, internal.c:63
    const uint32_t mbi = (uint32_t)((mby >> 2) * K.nsbw + (mbx >> 2)) << 2 | (uint32_t)quad;
    // OC_MB_MAP_IDXS (internal.c:67-72): 4:2:0 {.., 4, 8}; 4:2:2 {.., 4, 6, 8, 10}; 4:4:4 {.., 4..11}
    const int bi = ci << 1 | cj;
    const int per = K.fmt == 3 ? 4 : (K.fmt == 2 ? 2 : 1);
    const int idx = K.fmt == 3 ? bi : (K.fmt == 2 ? ci : 0);
    if (K.intra_satd) K.intra_satd[(size_t)mbi * 12 + 4 + (pli - 1) * per + idx] = satd;
    return;
  }
  // ---- a luma block ---------------------------------------------------------------------------------------------------------
  const int nh = K.nh[0], nv = K.nv[0];
  // lane 4 m + q: block q (raster inside the macro block) of macro block m (raster); nh and nv are even
  const int mbw = nh >> 1, mb_r = u >> 2, mb_y = mb_r / mbw, mb_x = mb_r - mb_y * mbw;
  const int by = 2 * mb_y + ((u >> 1) & 1), bx = 2 * mb_x + (u & 1);
  const int W = nh * 8, H = nv * 8, x0 = bx * 8, y0 = by * 8;
  const int edge = x0 == 0 ? 1 : (x0 + 8 == W ? 2 : 0);
  const int xw = edge == 1 ? 0 : (edge == 2 ? W - 16 : x0 - 4);
  const uint8_t *base = K.plane[0] + xw;
  CmRow rows[10];
#pragma unroll
  for (int r = 0; r < 10; r++) {
    const int y = min(max(y0 - 1 + r, 0), H - 1);
    rows[r] = cm_row(base + (ptrdiff_t)y * K.stride[0], edge);
  }
  // the block itself: columns 0..7 = bytes 1..8 of N
  uint2 s[8];
#pragma unroll
  for (int r = 0; r < 8; r++)
    s[r] = make_uint2(__builtin_amdgcn_alignbyte(rows[r + 1].n1, rows[r + 1].n0, 1u), __builtin_amdgcn_alignbyte(rows[r + 1].n2, rows[r + 1].n1, 1u));
  uint32_t satd;
  int dc;
  {
    pk16 D[8][4];
#pragma unroll
    for (int r = 0; r < 8; r++) row_sd(D[r], s[r]);
    satd = satd_sd(D, dc);
  }
  uint32_t x = 0, x2 = 0;
#pragma unroll
  for (int r = 0; r < 8; r++) {
    x = sad_row(s[r], make_uint2(0u, 0u), x);
    x2 = dot4(s[r].x, s[r].x, x2);
    x2 = dot4(s[r].y, s[r].y, x2);
  }
  uint32_t act = (x2 << 6) - x * x;
  if (act < (8u << 12)) {
    act = min(act, 5u << 12);   // the region is flat
  } else {
    uint32_t e1 = 0, e2 = 0, e3 = 0, e4 = 0;
    pk16 Ru[5], Su[4], Rm[5], Sm[4], Rd[5], Sd[4];
    cm_pairs(rows[0], Ru, Su);
    cm_pairs(rows[1], Rm, Sm);
#pragma unroll
    for (int r = 0; r < 8; r++) {
      cm_pairs(rows[r + 2], Rd, Sd);
      pk16 V[5], D[5], Ds[4];
#pragma unroll
      for (int k = 0; k < 5; k++) {
        V[k] = Ru[k] + Rm[k] + Rm[k] + Rd[k];
        D[k] = Rd[k] - Ru[k];
      }
#pragma unroll
      for (int k = 0; k < 4; k++) Ds[k] = Sd[k] - Su[k];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        e1 = cm_abs_sum(V[k + 1] - V[k], e1);
        e2 = cm_abs_sum(D[k] + Ds[k] + Ds[k] + D[k + 1], e2);
        const pk16 t3 = Rd[k + 1] - Ru[k], t4 = Rd[k] - Ru[k + 1];
        e3 = cm_abs_sum(t3 + t3 + Sd[k] - Rm[k] + Rm[k + 1] - Su[k], e3);
        e4 = cm_abs_sum(t4 + t4 + Sd[k] - Rm[k + 1] + Rm[k] - Su[k], e4);
      }
#pragma unroll
      for (int k = 0; k < 5; k++) {
        Ru[k] = Rm[k];
        Rm[k] = Rd[k];
      }
#pragma unroll
      for (int k = 0; k < 4; k++) {
        Su[k] = Sm[k];
        Sm[k] = Sd[k];
      }
    }
    // an edge block when the largest component is at least 40 % of the total (analyze.c:1224-1231)
    if (5u * max(max(e1, e2), max(e3, e4)) > 2u * (e1 + e2 + e3 + e4))
      act = cm_bexp32_q10(0x394A + (7 * (cm_blog32_q10(act) - 0x394A + 5) / 10));
  }
  uint32_t mbi;
  int bi;
  cm_luma_slot(bx, by, K.nsbw, mbi, bi);
  if (K.intra_satd) K.intra_satd[(size_t)mbi * 12 + bi] = satd;
  // the macro block's luma sum: its four lanes are neighbours (and all here: n_luma is a multiple of four)
  int dcs = dc + __shfl_xor(dc, 1);
  dcs += __shfl_xor(dcs, 2);
  if ((u & 3) == 0) {
    if (K.luma) K.luma[mbi] = (uint32_t)dcs;
    if (K.intra_satd) {   // OC_MB_MAP_IDXS (internal.c:67-72): 4:2:0 fills entries 4, 5; 4:2:2 4..7; 4:4:4 4..11
      const int used = K.fmt == 3 ? 12 : (K.fmt == 2 ? 8 : 6);
      for (int k = used; k < 12; k++) K.intra_satd[(size_t)mbi * 12 + k] = 0u;
    }
  }
  if (K.activity) K.activity[(size_t)mbi * 4 + bi] = act;
  if (K.activity_fast) {
    uint32_t fa = (11u * satd >> 8) * satd;   // analyze.c:1244
    if (fa < (8u << 12)) fa = min(fa, 5u << 12);
    K.activity_fast[(size_t)mbi * 4 + bi] = fa;
  }
}

}  // namespace thip
