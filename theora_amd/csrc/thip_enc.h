// thip_enc.h -- the encoder's block metrics (oc_enc_opt_vtable: SAD, SATD, SSD families, encfrag.c:42-378) for gfx950.
// Included by thip_slots.hip only.
//
// A block stays what memory gives: eight rows of eight bytes (uint2), never 64 integers.
//   * SAD family: v_sad_u8 takes four byte pairs per instruction (|a-b| summed into an accumulator), a row is two
//     instructions; the half-pel average of two references is v_lerp_u8 (encfrag.c:79).
//   * SSD: sum (a-b)^2 = sum a^2 + sum b^2 - 2 sum a b, three v_dot4_u32_u8 per four pixels, exact in 32-bit
//     modular arithmetic (the result is < 2^23).
//   * SATD family: the 8x8 Hadamard transform of the difference is a 64-point transform over the six index bits
//     (row bits r2 r1 r0, column bits c2 c1 c0), one butterfly level per bit IN ANY ORDER, and the sum of absolute
//     values does not care where an output lands.  The column bit c0 is taken first, on the bytes: a register
//     holds {x[2j] + x[2j+1], x[2j] - x[2j+1]} (two v_perm_b32 and one v_pk_mad_i16 per pixel pair, done once for
//     the source block and once per reference row, whatever the number of candidates that use them); the
//     difference of two such registers is the same level of the difference block (the transform is linear);
//     c1, c2, r2, r1 are 16 packed add/sub pairs each between whole registers -- no shuffles anywhere; and the
//     last level r0 is never computed: |p + q| + |p - q| = 2 max(|p|, |q|).  Every intermediate fits 16 bits
//     (|coefficient| <= 64 * 255), which is also why the reference's int16 buffer between its two passes
//     (encfrag.c:147-154) changes nothing for pixel input.  The DC coefficient (the sum of all differences)
//     is the one output the reference leaves out of the sum and returns separately (encfrag.c:302,312).
//   * motion search (k_enc_sites): the candidates of a block are positions (dx, dy) in {-1,0,1}^2 around one
//     reference position (the square pattern of mcenc.c:50-53).  A lane takes one block and one dx: the ten
//     reference rows it needs are loaded and prepared once and serve its three dy; results are stored candidate
//     by candidate (out[c * nblocks + block]).
#pragma once

namespace thip {

__device__ __forceinline__ void load_rows8(uint2 r[8], const uint8_t *p, int ystride) {
#pragma unroll
  for (int i = 0; i < 8; i++) r[i] = load_row8(p + (ptrdiff_t)i * ystride);
}
__device__ __forceinline__ uint2 avg_row(uint2 a, uint2 b) { return make_uint2(avg4_trunc(a.x, b.x), avg4_trunc(a.y, b.y)); }

// sum |a - b| over one row, added to acc
__device__ __forceinline__ uint32_t sad_row(uint2 a, uint2 b, uint32_t acc) {
  return __builtin_amdgcn_sad_u8(a.y, b.y, __builtin_amdgcn_sad_u8(a.x, b.x, acc));
}
// encfrag.c:42-86: whole-block SAD; with a threshold the rows after the one that crosses it are not added
template <bool THRESH>
__device__ __forceinline__ uint32_t sad_rows(const uint2 a[8], const uint2 b[8], uint32_t thresh) {
  uint32_t v = 0;
  bool live = true;
#pragma unroll
  for (int r = 0; r < 8; r++) {
    const uint32_t nv = sad_row(a[r], b[r], v);
    if (!THRESH) {
      v = nv;
    } else {
      v = live ? nv : v;
      live = live && v <= thresh;
    }
  }
  return v;
}
// encfrag.c:88-107: SAD against the block's rounded mean
__device__ __forceinline__ uint32_t intra_sad_rows(const uint2 a[8]) {
  uint32_t sum = 0;
  const uint2 z = make_uint2(0u, 0u);
#pragma unroll
  for (int r = 0; r < 8; r++) sum = sad_row(a[r], z, sum);
  const uint32_t m = ((sum + 32) >> 6) * 0x01010101u;
  const uint2 mm = make_uint2(m, m);
  uint32_t v = 0;
#pragma unroll
  for (int r = 0; r < 8; r++) v = sad_row(a[r], mm, v);
  return v;
}
// encfrag.c:338-350
// (the builtin, not inline assembly: the dot instructions have issue hazards against their neighbours that only the compiler
//  can pad when it knows what the instruction is)
__device__ __forceinline__ uint32_t dot4(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_udot4(a, b, c, false); }
__device__ __forceinline__ uint32_t ssd_rows(const uint2 a[8], const uint2 b[8]) {
  uint32_t sq = 0, ab = 0;
#pragma unroll
  for (int r = 0; r < 8; r++) {
    sq = dot4(a[r].x, a[r].x, sq);
    sq = dot4(a[r].y, a[r].y, sq);
    sq = dot4(b[r].x, b[r].x, sq);
    sq = dot4(b[r].y, b[r].y, sq);
    ab = dot4(a[r].x, b[r].x, ab);
    ab = dot4(a[r].y, b[r].y, ab);
  }
  return sq - 2u * ab;
}

// ---- SATD -------------------------------------------------------------------------------------------------------
// one row of pixels -> four registers {x[2j] + x[2j+1], x[2j] - x[2j+1]}, j = 0..3 (the c0 level)
// (one perm spreads the two bytes over the halves; the butterfly INSIDE the register is one packed multiply-add whose operand
//  selects read the upper half twice and the lower half twice -- {hi, hi} * {1, -1} + {lo, lo}: the compiler folds the two
//  broadcasts into op_sel / op_sel_hi, no second perm)
__device__ __forceinline__ pk16 pair_sd(uint32_t w, bool upper) {
  const pk16 x = as_pk(__builtin_amdgcn_perm(0u, w, upper ? 0x0c030c02u : 0x0c010c00u));   // {even, odd}
  const pk16 xh = {x.y, x.y}, xl = {x.x, x.x};
  const pk16 k = {(short)1, (short)-1};
  return xh * k + xl;
}
__device__ __forceinline__ void row_sd(pk16 out[4], uint2 row) {
  out[0] = pair_sd(row.x, false);
  out[1] = pair_sd(row.x, true);
  out[2] = pair_sd(row.y, false);
  out[3] = pair_sd(row.y, true);
}
__device__ __forceinline__ void bfly(pk16 &a, pk16 &b) {
  const pk16 s = a + b;
  b = a - b;
  a = s;
}
__device__ __forceinline__ pk16 pk_abs(pk16 x) {
  const pk16 z = {0, 0};
  return __builtin_elementwise_max(x, z - x);
}
typedef unsigned short upk16 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t sum2_u16(pk16 m, uint32_t acc) {   // acc + m.lo + m.hi, both halves unsigned
  const upk16 one = {(unsigned short)1, (unsigned short)1};
  return __builtin_amdgcn_udot2(__builtin_bit_cast(upk16, m), one, acc, false);
}
// The remaining horizontal levels (c1, c2) of one row whose c0 level is in R[0..3]
__device__ __forceinline__ void row_h12(pk16 R[4]) {
  bfly(R[0], R[1]);
  bfly(R[2], R[3]);
  bfly(R[0], R[2]);
  bfly(R[1], R[3]);
}
// D[r][j]: the block with all three horizontal levels done.  The vertical levels r2, r1, the last level folded into a maximum;
// returns sum |coefficient| without the DC term; dc = the DC coefficient = the sum of all differences (encfrag.c:264-315).
__device__ __forceinline__ uint32_t satd_vert(pk16 D[8][4], int &dc) {
#pragma unroll
  for (int j = 0; j < 4; j++) {   // r2, r1
#pragma unroll
    for (int r = 0; r < 4; r++) bfly(D[r][j], D[r + 4][j]);
    bfly(D[0][j], D[2][j]);
    bfly(D[1][j], D[3][j]);
    bfly(D[4][j], D[6][j]);
    bfly(D[5][j], D[7][j]);
  }
  // r0: |p + q| + |p - q| = 2 max(|p|, |q|); the all-plus output p + q of pair (D[0][0], D[1][0]), lower half, is the DC
  dc = (int)D[0][0].x + (int)D[1][0].x;
  uint32_t acc = 0;
#pragma unroll
  for (int r = 0; r < 8; r += 2)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      // max(|p|, |q|) = max(max(p, q), -min(p, q)): four packed operations instead of the five of two absolute values and a maximum
      const pk16 p = D[r][j], q = D[r + 1][j], z = {0, 0};
      acc = sum2_u16(__builtin_elementwise_max(__builtin_elementwise_max(p, q), z - __builtin_elementwise_min(p, q)), acc);
    }
  return 2u * acc - (uint32_t)(dc < 0 ? -dc : dc);
}
// D[r][j]: the c0 level of the difference block (row r, pixel pair j): both remaining horizontal levels, then satd_vert.
__device__ __forceinline__ uint32_t satd_sd(pk16 D[8][4], int &dc) {
#pragma unroll
  for (int r = 0; r < 8; r++) row_h12(D[r]);
  return satd_vert(D, dc);
}

// ---------------------------------------------------------------------------------------------------------------
// the vtable's metrics on lists of (source block, reference block[, second reference block]): one pair per lane
// ---------------------------------------------------------------------------------------------------------------
template <int OP>
__global__ __launch_bounds__(256) void k_enc_metric(uint32_t *out, int32_t *dc_out, const uint8_t *src_plane, const uint8_t *ref_plane,
                                                   int ystride, const int32_t *src_offs, const int32_t *ref_offs,
                                                   const int32_t *ref2_offs, uint32_t thresh, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  constexpr bool kHasRef = OP != THIP_ENC_INTRA_SAD && OP != THIP_ENC_INTRA_SATD;
  constexpr bool kTwo = OP == THIP_ENC_SAD2_THRESH || OP == THIP_ENC_SATD2;
  uint2 s[8], p[8];
  load_rows8(s, src_plane + src_offs[i], ystride);
  if (kHasRef) {
    load_rows8(p, ref_plane + ref_offs[i], ystride);
    if (kTwo) {
      uint2 q[8];
      load_rows8(q, ref_plane + ref2_offs[i], ystride);
#pragma unroll
      for (int r = 0; r < 8; r++) p[r] = avg_row(p[r], q[r]);   // encfrag.c:79,172
    }
  }
  uint32_t v;
  int dc = 0;
  if (OP == THIP_ENC_SAD) {
    v = sad_rows<false>(s, p, 0u);
  } else if (OP == THIP_ENC_SAD_THRESH || OP == THIP_ENC_SAD2_THRESH) {
    v = sad_rows<true>(s, p, thresh);
  } else if (OP == THIP_ENC_INTRA_SAD) {
    v = intra_sad_rows(s);
  } else if (OP == THIP_ENC_SSD) {
    v = ssd_rows(s, p);
  } else {   // SATD family, encfrag.c:317-336
    pk16 D[8][4];
#pragma unroll
    for (int r = 0; r < 8; r++) {
      row_sd(D[r], s[r]);
      if (kHasRef) {
        pk16 B[4];
        row_sd(B, p[r]);
#pragma unroll
        for (int j = 0; j < 4; j++) D[r][j] = D[r][j] - B[j];
      }
    }
    v = satd_sd(D, dc);
  }
  out[i] = v;
  if (dc_out) dc_out[i] = dc;
}

// ---------------------------------------------------------------------------------------------------------------
// motion search: every block against the candidates (dx, dy) in {-1,0,1}^2 around its reference position
// ---------------------------------------------------------------------------------------------------------------
struct SitesK {
  int8_t site_of[9];   // candidate number of (dy + 1) * 3 + (dx + 1), -1 if that position is not asked for
  int nsites;
};
// unit u = 3 * block + dxi: the lane's candidates are (dxi - 1, dy), dy = -1..1
template <int OP>
__global__ __launch_bounds__(256) void k_enc_sites(uint32_t *out, int32_t *dc_out, const uint8_t *src_plane, const uint8_t *ref_plane,
                                                  int ystride, const int32_t *src_offs, const int32_t *ref_offs, const SitesK K,
                                                  int64_t nblocks) {
  const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (u >= 3 * nblocks) return;
  const int64_t i = u / 3;
  const int dxi = (int)(u - 3 * i);
  int c[3];
#pragma unroll
  for (int dyi = 0; dyi < 3; dyi++) c[dyi] = K.site_of[dyi * 3 + dxi];
  if ((c[0] & c[1] & c[2]) < 0) return;   // none of this column's positions is a candidate
  uint2 s[8], e[10];
  load_rows8(s, src_plane + src_offs[i], ystride);
  const uint8_t *rp = ref_plane + ref_offs[i] - ystride + (dxi - 1);
#pragma unroll
  for (int r = 0; r < 10; r++) e[r] = load_row8(rp + (ptrdiff_t)r * ystride);
  if (OP == THIP_ENC_SAD) {
#pragma unroll
    for (int dyi = 0; dyi < 3; dyi++) {
      if (c[dyi] < 0) continue;
      uint32_t v = 0;
#pragma unroll
      for (int r = 0; r < 8; r++) v = sad_row(s[r], e[r + dyi], v);
      out[(int64_t)c[dyi] * nblocks + i] = v;
    }
  } else {
    // The c0 level (pixel pairs -> sum, difference) of the ten reference rows is taken ONCE and serves the three vertical
    // candidates (the transform is linear: the level of the difference is the difference of the levels): 40 registers instead
    // of the 20 of the rows as bytes -- 104 with the source block's 32 and the difference block's 32, four waves per SIMD
    // instead of five -- for 168 instructions a lane less (10 row preparations instead of 24).
    pk16 S[8][4], E[10][4];
    // (... and so are the two other HORIZONTAL levels: the transform is linear, the horizontal levels of the difference are the
    //  difference of the rows' horizontal levels, whichever rows meet -- only the vertical levels depend on dy)
#pragma unroll
    for (int r = 0; r < 8; r++) {
      row_sd(S[r], s[r]);
      row_h12(S[r]);
    }
#pragma unroll
    for (int r = 0; r < 10; r++) {
      row_sd(E[r], e[r]);
      row_h12(E[r]);
    }
#pragma unroll
    for (int dyi = 0; dyi < 3; dyi++) {
      if (c[dyi] < 0) continue;
      pk16 D[8][4];
#pragma unroll
      for (int r = 0; r < 8; r++)
#pragma unroll
        for (int j = 0; j < 4; j++) D[r][j] = S[r][j] - E[r + dyi][j];
      int dc;
      const uint32_t v = satd_vert(D, dc);
      out[(int64_t)c[dyi] * nblocks + i] = v;
      if (dc_out) dc_out[(int64_t)c[dyi] * nblocks + i] = dc;
    }
  }
}

// The SATD search with the source block SHARED by the three lanes of a block (round 6; option enc_sites_lds = 0: k_enc_sites<SATD>).
// k_enc_sites<SATD> keeps three things per lane -- the source block's 32 registers of horizontal levels, the ten reference rows'
// 40, the difference block's 32: 109 registers, four waves per SIMD, and a 1080p 4:4:4 frame is 4 590 waves on 4 096 slots: a second
// round of waves for an eighth of the work.  Here a wave takes 21 blocks (63 lanes: lane 3b + dxi), the three lanes of a block
// prepare its eight source rows BETWEEN them (rows dxi, dxi + 3, dxi + 6: three row preparations a lane instead of eight) into the
// wave's LDS (36 dwords a block: 32 + 4 of padding, so that sixteen blocks' reads fall on different banks), and the vertical levels
// run one register column at a time straight off LDS and the prepared reference rows (d[r] = S[r][j] - E[r + dy][j]: the vertical
// levels mix rows of ONE column only), so that neither the source block nor the difference block ever sits in registers.
constexpr int kSiteBlocks = 21, kSitePitch = 36;
#ifndef THIP_ENC_SITES_LDS_WAVES
#define THIP_ENC_SITES_LDS_WAVES 6
#endif
__global__ __launch_bounds__(256, THIP_ENC_SITES_LDS_WAVES) void k_enc_sites_satd(uint32_t *out, int32_t *dc_out, const uint8_t *src_plane,
                                                                               const uint8_t *ref_plane, int ystride, const int32_t *src_offs,
                                                                               const int32_t *ref_offs, const SitesK K, int64_t nblocks) {
  __shared__ __attribute__((aligned(16))) uint32_t s_S[4][kSiteBlocks * kSitePitch];
  const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
  uint32_t *const S = s_S[wave];
  const int lb = lane / 3, dxi = lane - 3 * lb;   // (lane 63: block 21, idle)
  const int64_t i = ((int64_t)blockIdx.x * 4 + wave) * kSiteBlocks + lb;
  const bool live = lb < kSiteBlocks && i < nblocks;
  int c[3];
#pragma unroll
  for (int dyi = 0; dyi < 3; dyi++) c[dyi] = K.site_of[dyi * 3 + dxi];
  const bool mine = live && (c[0] & c[1] & c[2]) >= 0;   // some position of this lane's column is a candidate
  uint2 srow[3], e[10];
  if (live) {
    const uint8_t *sp = src_plane + src_offs[i];
#pragma unroll
    for (int k = 0; k < 3; k++) srow[k] = load_row8(sp + (ptrdiff_t)min(dxi + 3 * k, 7) * ystride);
  }
  if (mine) {
    const uint8_t *rp = ref_plane + ref_offs[i] - ystride + (dxi - 1);
#pragma unroll
    for (int r = 0; r < 10; r++) e[r] = load_row8(rp + (ptrdiff_t)r * ystride);
  }
  if (live) {
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const int r = dxi + 3 * k;
      pk16 R[4];
      row_sd(R, srow[k]);
      row_h12(R);
      if (r < 8) {
#pragma unroll
        for (int j = 0; j < 4; j++) S[lb * kSitePitch + j * 8 + r] = as_u32(R[j]);
      }
    }
  }
  if (!mine) return;   // (its share of the source block is in LDS: the wave's loads and stores are issued in order, nobody waits for this lane)
  pk16 E[10][4];
#pragma unroll
  for (int r = 0; r < 10; r++) {
    row_sd(E[r], e[r]);
    row_h12(E[r]);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int dyi = 0; dyi < 3; dyi++) {
    if (c[dyi] < 0) continue;
    uint32_t acc = 0;
    int dc = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const uint4 s0 = *reinterpret_cast<const uint4 *>(S + lb * kSitePitch + j * 8), s1 = *reinterpret_cast<const uint4 *>(S + lb * kSitePitch + j * 8 + 4);
      pk16 d[8] = {as_pk(s0.x) - E[dyi + 0][j], as_pk(s0.y) - E[dyi + 1][j], as_pk(s0.z) - E[dyi + 2][j], as_pk(s0.w) - E[dyi + 3][j],
                   as_pk(s1.x) - E[dyi + 4][j], as_pk(s1.y) - E[dyi + 5][j], as_pk(s1.z) - E[dyi + 6][j], as_pk(s1.w) - E[dyi + 7][j]};
#pragma unroll
      for (int r = 0; r < 4; r++) bfly(d[r], d[r + 4]);   // r2, r1 (satd_vert)
      bfly(d[0], d[2]);
      bfly(d[1], d[3]);
      bfly(d[4], d[6]);
      bfly(d[5], d[7]);
      if (j == 0) dc = (int)d[0].x + (int)d[1].x;
#pragma unroll
      for (int r = 0; r < 8; r += 2) {
        const pk16 p = d[r], q = d[r + 1], z = {0, 0};
        acc = sum2_u16(__builtin_elementwise_max(__builtin_elementwise_max(p, q), z - __builtin_elementwise_min(p, q)), acc);
      }
    }
    out[(int64_t)c[dyi] * nblocks + i] = 2u * acc - (uint32_t)(dc < 0 ? -dc : dc);
    if (dc_out) dc_out[(int64_t)c[dyi] * nblocks + i] = dc;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// half-pel refinement: every block against the half-pel positions around ONE whole-pel vector (mcenc.c:596-657)
// ---------------------------------------------------------------------------------------------------------------
// The reference's refinement (oc_mcenc_ysatd_halfpel_mbrefine, mcenc.c:606-657; the SAD form :551-594) tries the eight half-pel
// vectors 2 * vec + (dx, dy), (dx, dy) in {-1, 0, 1}^2 without the centre, each as oc_enc_frag_satd2 / oc_enc_frag_sad2_thresh
// (encfrag.c:62-86, 323-328) of the source block against the truncating average of TWO whole-pel blocks: per axis the pair is
// {0, d} and WHICH of the two blocks gets the step follows from the signs (mcenc.c:644-647: the block towards zero comes first,
// as oc_state_get_mv_offsets has it); the average does not care which is first, so all that matters is whether the x step and the
// y step of a diagonal site land on the same block -- the pair {(0, 0), (dx, dy)} -- or on different ones -- {(dx, 0), (0, dy)}.
// One lane = one block and one side (the grid's y: dx = -1 or +1, so a wave's branches are scalar): the ten rows around the whole-pel
// position are loaded once (one 12-byte window a row holds columns 0 and dx) and serve the lane's four sites -- the three with its dx
// and the vertical site on its side, (0, dx).  Results site-major.
__device__ __forceinline__ bool halfpel_first_gets_step(int v, int d) { return (((2 * v + d) ^ d) < 0); }   // OC_SIGNMASK(((vec<<1)+d)^d), mcenc.c:644-645
// the three sites with one dx (DX = -1 / +1), and with VERT the vertical site of the same side, (0, DX)
template <int OP, int DX, bool VERT>
__device__ __forceinline__ void halfpel_side(uint32_t *out, int32_t *dc_out, const uint8_t *src_plane, const uint8_t *ref_plane,
                                             int ystride, const int32_t *src_offs, const int32_t *ref_offs, const int16_t *vecs,
                                             const SitesK &K, int64_t nblocks, int64_t i) {
  int c[4];
#pragma unroll
  for (int dyi = 0; dyi < 3; dyi++) c[dyi] = K.site_of[dyi * 3 + DX + 1];
  c[3] = VERT ? K.site_of[(DX + 1) * 3 + 1] : -1;
  if ((c[0] & c[1] & c[2] & c[3]) < 0) return;
  const int vv = vecs[i];
  const int vx = (int)(int8_t)(vv & 0xFF), vy = (int)(int8_t)(vv >> 8);   // OC_MV_X / OC_MV_Y, state.h:232-240
  uint2 s[8], e0[10], e1[10];
  load_rows8(s, src_plane + src_offs[i], ystride);
  // columns 0 and DX of a row are nine consecutive bytes: one 12-byte load a row, the second column shifted out of it
  const uint8_t *rq = ref_plane + ref_offs[i] - ystride + (DX < 0 ? -1 : 0);
#pragma unroll
  for (int r = 0; r < 10; r++) {
    Row12 w;
    __builtin_memcpy(&w, rq + (ptrdiff_t)r * ystride, 12);
    const uint2 lo = make_uint2(w.a, w.b);
    const uint2 hi = make_uint2(__builtin_amdgcn_alignbyte(w.b, w.a, 1), __builtin_amdgcn_alignbyte(w.c, w.b, 1));
    e0[r] = DX < 0 ? hi : lo;
    e1[r] = DX < 0 ? lo : hi;
  }
  constexpr bool kSad = OP == THIP_ENC_SAD2_THRESH;
  pk16 S[8][4];
  if (!kSad) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
      row_sd(S[r], s[r]);
      row_h12(S[r]);
    }
  }
  const bool xfirst = halfpel_first_gets_step(vx, DX);
#pragma unroll
  for (int k = 0; k < (VERT ? 4 : 3); k++) {
    if (c[k] < 0) continue;
    const int dy = k < 3 ? k - 1 : DX;
    // the two blocks: {(0, 0), (dx, dy)} when both steps go to the same one, {(dx, 0), (0, dy)} otherwise (dy = 0: {(0, 0), (dx, 0)};
    // the vertical site: {(0, 0), (0, dy)})
    const bool same = dy == 0 || xfirst == halfpel_first_gets_step(vy, dy);
    uint2 a[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
      if (k == 3) {
        a[r] = avg_row(e0[r + 1], e0[r + 1 + dy]);
      } else {
        const uint2 p0 = e0[r + 1], p1 = e1[r + 1], q0 = e0[r + 1 + dy], q1 = e1[r + 1 + dy];
        a[r] = avg_row(same ? p0 : p1, same ? q1 : q0);
      }
    }
    if (kSad) {
      uint32_t v = 0;
#pragma unroll
      for (int r = 0; r < 8; r++) v = sad_row(s[r], a[r], v);
      out[(int64_t)c[k] * nblocks + i] = v;
    } else {
      pk16 D[8][4];
#pragma unroll
      for (int r = 0; r < 8; r++) {
        pk16 B[4];
        row_sd(B, a[r]);
        row_h12(B);
#pragma unroll
        for (int j = 0; j < 4; j++) D[r][j] = S[r][j] - B[j];
      }
      int dc;
      const uint32_t v = satd_vert(D, dc);
      out[(int64_t)c[k] * nblocks + i] = v;
      if (dc_out) dc_out[(int64_t)c[k] * nblocks + i] = dc;
    }
  }
}
// the two vertical sites, (0, -1) and (0, +1): rows r, r + 1 averaged for r = 0..8 (pixel rows -1..7), prepared once -- the site above
// takes the first eight, the site below the last eight
template <int OP>
__device__ __forceinline__ void halfpel_vertical(uint32_t *out, int32_t *dc_out, const uint8_t *src_plane, const uint8_t *ref_plane,
                                                 int ystride, const int32_t *src_offs, const int32_t *ref_offs, const SitesK &K,
                                                 int64_t nblocks, int64_t i) {
  const int c[2] = {K.site_of[1], K.site_of[7]};
  if ((c[0] & c[1]) < 0) return;
  uint2 s[8], e0[10], a[9];
  load_rows8(s, src_plane + src_offs[i], ystride);
  const uint8_t *rp = ref_plane + ref_offs[i] - ystride;
#pragma unroll
  for (int r = 0; r < 10; r++) e0[r] = load_row8(rp + (ptrdiff_t)r * ystride);
#pragma unroll
  for (int r = 0; r < 9; r++) a[r] = avg_row(e0[r], e0[r + 1]);
  if (OP == THIP_ENC_SAD2_THRESH) {
#pragma unroll
    for (int k = 0; k < 2; k++) {
      if (c[k] < 0) continue;
      uint32_t v = 0;
#pragma unroll
      for (int r = 0; r < 8; r++) v = sad_row(s[r], a[r + k], v);
      out[(int64_t)c[k] * nblocks + i] = v;
    }
  } else {
    pk16 S[8][4], E[9][4];
#pragma unroll
    for (int r = 0; r < 8; r++) {
      row_sd(S[r], s[r]);
      row_h12(S[r]);
    }
#pragma unroll
    for (int r = 0; r < 9; r++) {
      row_sd(E[r], a[r]);
      row_h12(E[r]);
    }
#pragma unroll
    for (int k = 0; k < 2; k++) {
      if (c[k] < 0) continue;
      pk16 D[8][4];
#pragma unroll
      for (int r = 0; r < 8; r++)
#pragma unroll
        for (int j = 0; j < 4; j++) D[r][j] = S[r][j] - E[r + k][j];
      int dc;
      const uint32_t v = satd_vert(D, dc);
      out[(int64_t)c[k] * nblocks + i] = v;
      if (dc_out) dc_out[(int64_t)c[k] * nblocks + i] = dc;
    }
  }
}
// LANES = 3: the grid's y is dx + 1 (two or three sites a lane); LANES = 2: y is the side, four sites a lane
template <int OP, int LANES>
__global__ __launch_bounds__(256) void k_enc_halfpel(uint32_t *out, int32_t *dc_out, const uint8_t *src_plane, const uint8_t *ref_plane,
                                                    int ystride, const int32_t *src_offs, const int32_t *ref_offs, const int16_t *vecs,
                                                    const SitesK K, int64_t nblocks) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= nblocks) return;
  const int y = (int)blockIdx.y;
  if (LANES == 2) {
    if (y == 0) halfpel_side<OP, -1, true>(out, dc_out, src_plane, ref_plane, ystride, src_offs, ref_offs, vecs, K, nblocks, i);
    else halfpel_side<OP, 1, true>(out, dc_out, src_plane, ref_plane, ystride, src_offs, ref_offs, vecs, K, nblocks, i);
  } else {
    if (y == 0) halfpel_side<OP, -1, false>(out, dc_out, src_plane, ref_plane, ystride, src_offs, ref_offs, vecs, K, nblocks, i);
    else if (y == 1) halfpel_vertical<OP>(out, dc_out, src_plane, ref_plane, ystride, src_offs, ref_offs, K, nblocks, i);
    else halfpel_side<OP, 1, false>(out, dc_out, src_plane, ref_plane, ystride, src_offs, ref_offs, vecs, K, nblocks, i);
  }
}

}  // namespace thip
