/* theora_oracle.c -- TEST INFRASTRUCTURE ONLY (see theora_oracle.h).
 *
 * Plain-C restatement of the reference's per-fragment reconstruction path.
 * PARITY UNPINNED: not validated against a reference binary (the reference does not
 * build in this image: <ogg/ogg.h> is missing) and the reference ships no golden
 * vectors for this path.  Each function cites the reference lines it restates.
 * The inverse DCT, reconstruction, motion-vector, loop-filter and encoder parts are restructured
 * restatements; the out-of-loop post-processing part (orc_pp_*, near the end) follows
 * decode.c:1610-1957 statement by statement on purpose -- that filter is non-normative, has no
 * specification text to restate it from, and the point of this file is to be the checker.  It is
 * test infrastructure under oracle/: nothing in theora_amd/ or include/ uses it.
 */
#include "theora_oracle.h"
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* tables                                                                     */
/* ------------------------------------------------------------------------- */

/* zig-zag index -> natural coefficient position; indices past 63 land on the
   dump slot 64 (internal.c:24-45). Generated instead of tabulated. */
uint8_t const ORC_FZIG_ZAG[128] = {
  0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,
  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
  35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
  58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
  64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64,
  64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64,
  64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64,
  64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64};

/* cos(k*pi/16) in Q16, dct.h:23-29 */
enum { C1 = 64277, C2 = 60547, C3 = 54491, C4 = 46341, C5 = 36410, C6 = 25080, C7 = 12785 };

#define Q16(c, v) (((int32_t)(c) * (int32_t)(v)) >> 16)
#define S16(v) ((int16_t)(v))

static inline uint8_t clamp255(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); } /* ocintrin.h:59 */

/* ------------------------------------------------------------------------- */
/* inverse DCT, idct.c                                                        */
/* ------------------------------------------------------------------------- */

/* The four-stage output network shared by every 1-D variant (idct.c:57-81,
   :116-130, :158-172, :196-206): inputs are the stage-2 values. */
static void idct_tail(int16_t *y, int32_t t0, int32_t t1, int32_t t2, int32_t t3,
                      int32_t t4, int32_t t5, int32_t t6, int32_t t7) {
  int32_t r;
  /* stage 3 */
  r = t0 + t3; t3 = t0 - t3; t0 = r;
  r = t1 + t2; t2 = t1 - t2; t1 = r;
  r = t6 + t5; t5 = t6 - t5; t6 = r;
  /* stage 4; outputs go down a column (stride 8) */
  y[0 * 8] = S16(t0 + t7);
  y[1 * 8] = S16(t1 + t6);
  y[2 * 8] = S16(t2 + t5);
  y[3 * 8] = S16(t3 + t4);
  y[4 * 8] = S16(t3 - t4);
  y[5 * 8] = S16(t2 - t5);
  y[6 * 8] = S16(t1 - t6);
  y[7 * 8] = S16(t0 - t7);
}

/* idct.c:30-81 */
static void idct8_all(int16_t *y, const int16_t x[8]) {
  int32_t t0, t1, t2, t3, t4, t5, t6, t7, r;
  t0 = Q16(C4, S16(x[0] + x[4]));
  t1 = Q16(C4, S16(x[0] - x[4]));
  t2 = Q16(C6, x[2]) - Q16(C2, x[6]);
  t3 = Q16(C2, x[2]) + Q16(C6, x[6]);
  t4 = Q16(C7, x[1]) - Q16(C1, x[7]);
  t5 = Q16(C3, x[5]) - Q16(C5, x[3]);
  t6 = Q16(C5, x[5]) + Q16(C3, x[3]);
  t7 = Q16(C1, x[1]) + Q16(C7, x[7]);
  r = t4 + t5; t5 = Q16(C4, S16(t4 - t5)); t4 = r;
  r = t7 + t6; t6 = Q16(C4, S16(t7 - t6)); t7 = r;
  idct_tail(y, t0, t1, t2, t3, t4, t5, t6, t7);
}

/* idct.c:92-130: only x[0..3] are read */
static void idct8_first4(int16_t *y, const int16_t x[8]) {
  int32_t t0, t2, t3, t4, t5, t6, t7, r;
  t0 = Q16(C4, x[0]);
  t2 = Q16(C6, x[2]);
  t3 = Q16(C2, x[2]);
  t4 = Q16(C7, x[1]);
  t5 = -Q16(C5, x[3]);
  t6 = Q16(C3, x[3]);
  t7 = Q16(C1, x[1]);
  r = t4 + t5; t5 = Q16(C4, S16(t4 - t5)); t4 = r;
  r = t7 + t6; t6 = Q16(C4, S16(t7 - t6)); t7 = r;
  /* stage 3 in the reference uses t1=t0+t2,t2=t0-t2 directly: same as the tail with t1==t0 */
  idct_tail(y, t0, t0, t2, t3, t4, t5, t6, t7);
}

/* idct.c:141-172: only x[0..2] are read */
static void idct8_first3(int16_t *y, const int16_t x[8]) {
  int32_t t0, t2, t3, t4, t5, t6, t7;
  t0 = Q16(C4, x[0]);
  t2 = Q16(C6, x[2]);
  t3 = Q16(C2, x[2]);
  t4 = Q16(C7, x[1]);
  t7 = Q16(C1, x[1]);
  t5 = Q16(C4, t4);   /* NOTE: no int16 cast here, idct.c:151-152 */
  t6 = Q16(C4, t7);
  idct_tail(y, t0, t0, t2, t3, t4, t5, t6, t7);
}

/* idct.c:183-206: only x[0..1] are read */
static void idct8_first2(int16_t *y, const int16_t x[8]) {
  int32_t t0, t4, t5, t6, t7;
  t0 = Q16(C4, x[0]);
  t4 = Q16(C7, x[1]);
  t7 = Q16(C1, x[1]);
  t5 = Q16(C4, t4);
  t6 = Q16(C4, t7);
  idct_tail(y, t0, t0, 0, 0, t4, t5, t6, t7);
}

/* idct.c:214-217 */
static void idct8_first1(int16_t *y, const int16_t x[1]) {
  int16_t v = S16(Q16(C4, x[0]));
  int k;
  for (k = 0; k < 8; k++) y[k * 8] = v;
}

static void idct_descale(int16_t y[64]) {           /* idct.c:243,274,293 */
  int i;
  for (i = 0; i < 64; i++) y[i] = S16((y[i] + 8) >> 4);
}

/* idct.c:234-246 */
static void idct8x8_zz3(int16_t y[64], int16_t x[64]) {
  int16_t w[64];
  int i;
  memset(w, 0, sizeof(w));   /* the reference leaves the unused part of w unread */
  idct8_first2(w, x);
  idct8_first1(w + 1, x + 8);
  for (i = 0; i < 8; i++) idct8_first2(y + i, w + i * 8);
  idct_descale(y);
  x[0] = x[1] = x[8] = 0;
}

/* idct.c:263-277 */
static void idct8x8_zz10(int16_t y[64], int16_t x[64]) {
  int16_t w[64];
  int i;
  memset(w, 0, sizeof(w));
  idct8_first4(w, x);
  idct8_first3(w + 1, x + 8);
  idct8_first2(w + 2, x + 16);
  idct8_first1(w + 3, x + 24);
  for (i = 0; i < 8; i++) idct8_first4(y + i, w + i * 8);
  idct_descale(y);
  x[0] = x[1] = x[2] = x[3] = x[8] = x[9] = x[10] = x[16] = x[17] = x[24] = 0;
}

#ifdef ORC_SIMD
/* ------------------------------------------------------------------------------------------------------------------
   The SSE2 legs of this file (built only into _build/libtheora_oracle_simd.so, -DORC_SIMD): own intrinsics code for the three
   hot loops -- the full 8x8 inverse transform, the three reconstruction loops and the loop filter's two edge filters -- so
   that bench.py's cpu_baseline has a figure from a VECTORISED CPU path measured on the box, beside the scalar one (BASELINE.md
   section 4.2; the reference's own x86 path cannot be built here).  tests/test_oracle.py proves them equal to the scalar
   functions above and below, value for value; nothing else differs between the two libraries.
   ------------------------------------------------------------------------------------------------------------------ */
#include <emmintrin.h>
static void idct8_all(int16_t *y, const int16_t x[8]);
static void (*const orc_scalar_idct8_all)(int16_t *, const int16_t *) __attribute__((unused)) = idct8_all;   /* (the scalar pass stays compiled) */
/* (c * v) >> 16 on eight int16: pmulhw is signed x signed, so a constant beyond 32767 goes in as c - 65536 and v is added back
   (floor(c v / 65536) = floor((c - 65536) v / 65536) + v, v being an integer) */
static inline __m128i q16v(int c, __m128i v) {
  if (c >= 32768) return _mm_add_epi16(_mm_mulhi_epi16(v, _mm_set1_epi16((short)(c - 65536))), v);
  return _mm_mulhi_epi16(v, _mm_set1_epi16((short)c));
}
static inline void transpose8x8_epi16(__m128i r[8]) {
  __m128i a0 = _mm_unpacklo_epi16(r[0], r[1]), a1 = _mm_unpackhi_epi16(r[0], r[1]);
  __m128i a2 = _mm_unpacklo_epi16(r[2], r[3]), a3 = _mm_unpackhi_epi16(r[2], r[3]);
  __m128i a4 = _mm_unpacklo_epi16(r[4], r[5]), a5 = _mm_unpackhi_epi16(r[4], r[5]);
  __m128i a6 = _mm_unpacklo_epi16(r[6], r[7]), a7 = _mm_unpackhi_epi16(r[6], r[7]);
  __m128i b0 = _mm_unpacklo_epi32(a0, a2), b1 = _mm_unpackhi_epi32(a0, a2);
  __m128i b2 = _mm_unpacklo_epi32(a1, a3), b3 = _mm_unpackhi_epi32(a1, a3);
  __m128i b4 = _mm_unpacklo_epi32(a4, a6), b5 = _mm_unpackhi_epi32(a4, a6);
  __m128i b6 = _mm_unpacklo_epi32(a5, a7), b7 = _mm_unpackhi_epi32(a5, a7);
  r[0] = _mm_unpacklo_epi64(b0, b4); r[1] = _mm_unpackhi_epi64(b0, b4);
  r[2] = _mm_unpacklo_epi64(b1, b5); r[3] = _mm_unpackhi_epi64(b1, b5);
  r[4] = _mm_unpacklo_epi64(b2, b6); r[5] = _mm_unpackhi_epi64(b2, b6);
  r[6] = _mm_unpacklo_epi64(b3, b7); r[7] = _mm_unpackhi_epi64(b3, b7);
}
/* idct8_all (idct.c:30-81) on eight vectors at once: v[k] holds input k of eight independent 1-D transforms; on return v[k] is
   output k.  Additions wrap at 16 bits, which is what the scalar code's (ogg_int16_t) casts make of its int32 sums: between two
   casts it only adds and subtracts. */
static inline void idct8_all_v(__m128i v[8]) {
  __m128i t0 = q16v(C4, _mm_add_epi16(v[0], v[4]));
  __m128i t1 = q16v(C4, _mm_sub_epi16(v[0], v[4]));
  __m128i t2 = _mm_sub_epi16(q16v(C6, v[2]), q16v(C2, v[6]));
  __m128i t3 = _mm_add_epi16(q16v(C2, v[2]), q16v(C6, v[6]));
  __m128i t4 = _mm_sub_epi16(q16v(C7, v[1]), q16v(C1, v[7]));
  __m128i t5 = _mm_sub_epi16(q16v(C3, v[5]), q16v(C5, v[3]));
  __m128i t6 = _mm_add_epi16(q16v(C5, v[5]), q16v(C3, v[3]));
  __m128i t7 = _mm_add_epi16(q16v(C1, v[1]), q16v(C7, v[7]));
  __m128i r;
  r = _mm_add_epi16(t4, t5); t5 = q16v(C4, _mm_sub_epi16(t4, t5)); t4 = r;
  r = _mm_add_epi16(t7, t6); t6 = q16v(C4, _mm_sub_epi16(t7, t6)); t7 = r;
  r = _mm_add_epi16(t0, t3); t3 = _mm_sub_epi16(t0, t3); t0 = r;
  r = _mm_add_epi16(t1, t2); t2 = _mm_sub_epi16(t1, t2); t1 = r;
  r = _mm_add_epi16(t6, t5); t5 = _mm_sub_epi16(t6, t5); t6 = r;
  v[0] = _mm_add_epi16(t0, t7);
  v[1] = _mm_add_epi16(t1, t6);
  v[2] = _mm_add_epi16(t2, t5);
  v[3] = _mm_add_epi16(t3, t4);
  v[4] = _mm_sub_epi16(t3, t4);
  v[5] = _mm_sub_epi16(t2, t5);
  v[6] = _mm_sub_epi16(t1, t6);
  v[7] = _mm_sub_epi16(t0, t7);
}
/* idct.c:286-296: every 1-D pass reads rows and writes columns, i.e. transposes; with input k of eight transforms in v[k] the
   pass itself needs the rows transposed first and leaves output k of all eight rows in v[k] = row k of the transposed result */
void orc_idct8x8_full(int16_t y[64], int16_t x[64]) {
  __m128i v[8];
  const __m128i eight = _mm_set1_epi16(8);
  int i;
  for (i = 0; i < 8; i++) v[i] = _mm_loadu_si128((const __m128i *)(x + i * 8));
  transpose8x8_epi16(v);
  idct8_all_v(v);
  transpose8x8_epi16(v);
  idct8_all_v(v);
  /* (y + 8) >> 4 as the scalar code has it: the sum in int32, the result cast -- y + 8 can pass 32767, so the vector form
     shifts first: ((y >> 3) + 1) >> 1 is the same number for every int16 y */
  (void)eight;
  for (i = 0; i < 8; i++) {
    const __m128i h = _mm_add_epi16(_mm_srai_epi16(v[i], 3), _mm_set1_epi16(1));
    _mm_storeu_si128((__m128i *)(y + i * 8), _mm_srai_epi16(h, 1));
  }
  memset(x, 0, 64 * sizeof(*x));
}
#else
/* idct.c:286-296 */
void orc_idct8x8_full(int16_t y[64], int16_t x[64]) {
  int16_t w[64];
  int i;
  for (i = 0; i < 8; i++) idct8_all(w + i, x + i * 8);
  for (i = 0; i < 8; i++) idct8_all(y + i, w + i * 8);
  idct_descale(y);
  memset(x, 0, 64 * sizeof(*x));
}
#endif

/* idct.c:301-330 */
void orc_idct8x8(int16_t y[64], int16_t x[64], int last_zzi) {
  if (last_zzi <= 3) idct8x8_zz3(y, x);
  else if (last_zzi <= 10) idct8x8_zz10(y, x);
  else orc_idct8x8_full(y, x);
}

void orc_idct8x8_batch(int16_t *y, const int16_t *x, const int32_t *last_zzi, ptrdiff_t n) {
  ptrdiff_t i;
  int16_t tmp[64];
  for (i = 0; i < n; i++) {
    memcpy(tmp, x + i * 64, sizeof(tmp));
    orc_idct8x8(y + i * 64, tmp, last_zzi ? last_zzi[i] : 64);
  }
}

/* ------------------------------------------------------------------------- */
/* fragment copy / reconstruction, fragment.c                                 */
/* ------------------------------------------------------------------------- */

void orc_frag_copy(uint8_t *dst, const uint8_t *src, int ystride) {       /* fragment.c:20-27 */
  int i;
  for (i = 0; i < 8; i++) memcpy(dst + (ptrdiff_t)i * ystride, src + (ptrdiff_t)i * ystride, 8);
}

void orc_frag_copy_list(uint8_t *dst_frame, const uint8_t *src_frame, int ystride,
                        const ptrdiff_t *fragis, ptrdiff_t nfragis,
                        const ptrdiff_t *frag_buf_offs) {                 /* fragment.c:37-47 */
  ptrdiff_t k;
  for (k = 0; k < nfragis; k++) {
    ptrdiff_t off = frag_buf_offs[fragis[k]];
    orc_frag_copy(dst_frame + off, src_frame + off, ystride);
  }
}

#ifdef ORC_SIMD
/* clamp255(residue + predictor) for a row of eight: the sum saturates at 16 bits (a residue near 32767 still ends as 255, as the
   int sum does), packuswb clamps to 0..255 */
static inline void recon_row_v(uint8_t *dst, const int16_t *res, __m128i pred16) {
  const __m128i sum = _mm_adds_epi16(_mm_loadu_si128((const __m128i *)res), pred16);
  _mm_storel_epi64((__m128i *)dst, _mm_packus_epi16(sum, sum));
}
void orc_frag_recon_intra(uint8_t *dst, int ystride, const int16_t residue[64]) { /* fragment.c:49-57 */
  const __m128i p = _mm_set1_epi16(128);
  int i;
  for (i = 0; i < 8; i++, dst += ystride) recon_row_v(dst, residue + i * 8, p);
}
void orc_frag_recon_inter(uint8_t *dst, const uint8_t *src, int ystride, const int16_t residue[64]) { /* fragment.c:59-68 */
  const __m128i z = _mm_setzero_si128();
  int i;
  for (i = 0; i < 8; i++, dst += ystride, src += ystride)
    recon_row_v(dst, residue + i * 8, _mm_unpacklo_epi8(_mm_loadl_epi64((const __m128i *)src), z));
}
void orc_frag_recon_inter2(uint8_t *dst, const uint8_t *src1, const uint8_t *src2, int ystride,
                           const int16_t residue[64]) {                   /* fragment.c:70-80 */
  const __m128i z = _mm_setzero_si128();
  int i;
  for (i = 0; i < 8; i++, dst += ystride, src1 += ystride, src2 += ystride) {
    const __m128i a = _mm_unpacklo_epi8(_mm_loadl_epi64((const __m128i *)src1), z);
    const __m128i b = _mm_unpacklo_epi8(_mm_loadl_epi64((const __m128i *)src2), z);
    recon_row_v(dst, residue + i * 8, _mm_srli_epi16(_mm_add_epi16(a, b), 1));   /* (a + b) >> 1, truncating */
  }
}
#else
void orc_frag_recon_intra(uint8_t *dst, int ystride, const int16_t residue[64]) { /* fragment.c:49-57 */
  int i, j;
  for (i = 0; i < 8; i++, dst += ystride)
    for (j = 0; j < 8; j++) dst[j] = clamp255(residue[i * 8 + j] + 128);
}

void orc_frag_recon_inter(uint8_t *dst, const uint8_t *src, int ystride,
                          const int16_t residue[64]) {                    /* fragment.c:59-68 */
  int i, j;
  for (i = 0; i < 8; i++, dst += ystride, src += ystride)
    for (j = 0; j < 8; j++) dst[j] = clamp255(residue[i * 8 + j] + src[j]);
}

void orc_frag_recon_inter2(uint8_t *dst, const uint8_t *src1, const uint8_t *src2,
                           int ystride, const int16_t residue[64]) {      /* fragment.c:70-80 */
  int i, j;
  for (i = 0; i < 8; i++, dst += ystride, src1 += ystride, src2 += ystride)
    for (j = 0; j < 8; j++) dst[j] = clamp255(residue[i * 8 + j] + ((src1[j] + src2[j]) >> 1));
}
#endif

/* ------------------------------------------------------------------------- */
/* motion vector -> buffer offsets, state.c:846-957                           */
/* ------------------------------------------------------------------------- */

/* The reference tabulates (OC_MVMAP, OC_MVMAP2, state.c:901-928) what its own
   table-free variant (state.c:866-899) computes: the first offset is the
   vector divided by 2 (4 on a decimated axis) truncating toward zero, the
   second differs from it by sign(v) when any fractional bit is set. */
static void mv_axis(int v, int quarter, int *whole, int *frac) {
  int div = quarter ? 4 : 2;
  *whole = v / div;                           /* C division truncates toward zero */
  *frac = (v % div) ? (v < 0 ? -1 : 1) : 0;
}

int orc_mv_offsets(int offsets[2], int ystride, int qpx, int qpy, int dx, int dy) {
  int mx, my, mx2, my2, offs;
  mv_axis(dy, qpy, &my, &my2);
  mv_axis(dx, qpx, &mx, &mx2);
  offs = my * ystride + mx;
  offsets[0] = offs;
  if (mx2 || my2) {
    offsets[1] = offs + my2 * ystride + mx2;
    return 2;
  }
  return 1;
}

/* ------------------------------------------------------------------------- */
/* stream state                                                               */
/* ------------------------------------------------------------------------- */

orc_state *orc_state_new(int frame_width, int frame_height, int pixel_fmt) {
  orc_state *st;
  int pli, b;
  int yhstride, yheight, chstride, cheight, hpad, vpad;
  size_t yplane_sz, cplane_sz;
  ptrdiff_t yoffset, coffset, align, fragi;
  if ((frame_width & 15) || (frame_height & 15) || frame_width <= 0 || frame_height <= 0 ||
      pixel_fmt < 0 || pixel_fmt > 3 || pixel_fmt == ORC_PF_RSVD)
    return NULL;                                   /* state.c:712-727 */
  st = (orc_state *)calloc(1, sizeof(*st));
  st->frame_width = frame_width;
  st->frame_height = frame_height;
  st->pixel_fmt = pixel_fmt;
  st->hdec = !(pixel_fmt & 1);
  st->vdec = !(pixel_fmt & 2);
  /* fragment planes, state.c:443-475 */
  st->fplanes[0].nhfrags = frame_width >> 3;
  st->fplanes[0].nvfrags = frame_height >> 3;
  for (pli = 1; pli < 3; pli++) {
    st->fplanes[pli].nhfrags = (st->fplanes[0].nhfrags + st->hdec) >> st->hdec;
    st->fplanes[pli].nvfrags = (st->fplanes[0].nvfrags + st->vdec) >> st->vdec;
  }
  fragi = 0;
  for (pli = 0; pli < 3; pli++) {
    orc_plane_geom *g = st->fplanes + pli;
    g->froffset = fragi;
    g->nfrags = (ptrdiff_t)g->nhfrags * g->nvfrags;
    fragi += g->nfrags;
    g->width = pli ? frame_width >> st->hdec : frame_width;
    g->height = pli ? frame_height >> st->vdec : frame_height;
  }
  st->nfrags = fragi;
  st->coded = (uint8_t *)calloc(st->nfrags, 1);
  st->refi = (uint8_t *)calloc(st->nfrags, 1);
  st->mvs = (int16_t *)calloc(st->nfrags, sizeof(int16_t));
  st->dc = (int16_t *)calloc(st->nfrags, sizeof(int16_t));
  st->frag_buf_offs = (ptrdiff_t *)calloc(st->nfrags, sizeof(ptrdiff_t));
  /* reference frames, state.c:566-629 */
  yhstride = frame_width + 2 * ORC_UMV_PADDING;
  yheight = frame_height + 2 * ORC_UMV_PADDING;
  chstride = ((yhstride >> st->hdec) + 15) & ~15;
  cheight = yheight >> st->vdec;
  yplane_sz = (size_t)yhstride * yheight;
  cplane_sz = (size_t)chstride * cheight;
  hpad = ORC_UMV_PADDING >> st->hdec;
  vpad = ORC_UMV_PADDING >> st->vdec;
  yoffset = ORC_UMV_PADDING + ORC_UMV_PADDING * (ptrdiff_t)yhstride;
  coffset = hpad + vpad * (ptrdiff_t)chstride;
  align = -coffset & 15;
  st->ref_frame_sz = yplane_sz + 2 * cplane_sz + 16;
  st->ref_slab = (uint8_t *)calloc(3, st->ref_frame_sz);
  st->fplanes[0].stride = -yhstride;
  st->fplanes[1].stride = st->fplanes[2].stride = -chstride;
  for (b = 0; b < 3; b++) {
    uint8_t *p = st->ref_slab + (size_t)b * st->ref_frame_sz;
    uint8_t *top[3];
    top[0] = p + yoffset;
    p += yplane_sz + align;
    top[1] = p + coffset;
    p += cplane_sz;
    top[2] = p + coffset;
    /* flip: pixel (0,0) of the bitstream is the first pixel of the LAST memory row */
    for (pli = 0; pli < 3; pli++) {
      ptrdiff_t pos_stride = -st->fplanes[pli].stride;
      st->ref_plane_data[b][pli] = top[pli] + (ptrdiff_t)(st->fplanes[pli].height - 1) * pos_stride;
    }
  }
  for (pli = 0; pli < 3; pli++) st->plane_off[pli] = st->ref_plane_data[0][pli] - st->ref_plane_data[0][0];
  /* fragment offsets relative to the Y data pointer, state.c:631-656 */
  fragi = 0;
  for (pli = 0; pli < 3; pli++) {
    const orc_plane_geom *g = st->fplanes + pli;
    int fy, fx;
    for (fy = 0; fy < g->nvfrags; fy++)
      for (fx = 0; fx < g->nhfrags; fx++)
        st->frag_buf_offs[fragi++] = st->plane_off[pli] + (ptrdiff_t)fy * 8 * g->stride + fx * 8;
  }
  for (b = 0; b < 3; b++) {
    st->ref_frame_idx[b] = -1;                      /* state.c:658-669 */
    st->ref_frame_data[b] = NULL;
  }
  return st;
}

void orc_state_free(orc_state *st) {
  if (!st) return;
  free(st->coded); free(st->refi); free(st->mvs); free(st->dc);
  free(st->frag_buf_offs); free(st->ref_slab); free(st);
}

void orc_state_set_ref_idx(orc_state *st, int gold, int prev, int self) {
  int v[3], k;
  v[ORC_FRAME_GOLD] = gold; v[ORC_FRAME_PREV] = prev; v[ORC_FRAME_SELF] = self;
  for (k = 0; k < 3; k++) {
    st->ref_frame_idx[k] = v[k];
    st->ref_frame_data[k] = v[k] < 0 ? NULL : st->ref_plane_data[v[k]][0];
  }
}

void orc_state_get_plane(const orc_state *st, int slot, int pli, uint8_t *out) {
  const orc_plane_geom *g = st->fplanes + pli;
  const uint8_t *p = st->ref_plane_data[st->ref_frame_idx[slot]][pli];
  int y;
  for (y = 0; y < g->height; y++) memcpy(out + (size_t)y * g->width, p + (ptrdiff_t)y * g->stride, g->width);
}

void orc_state_set_plane(orc_state *st, int slot, int pli, const uint8_t *in) {
  const orc_plane_geom *g = st->fplanes + pli;
  int bufi = st->ref_frame_idx[slot];
  uint8_t *p = st->ref_plane_data[bufi][pli];
  int y;
  for (y = 0; y < g->height; y++) memcpy(p + (ptrdiff_t)y * g->stride, in + (size_t)y * g->width, g->width);
  orc_state_borders_fill_rows(st, bufi, pli, 0, g->height);
  orc_state_borders_fill_caps(st, bufi, pli);
}

/* state.c:770-793.  Rows are addressed in bitstream coordinates. */
void orc_state_borders_fill_rows(orc_state *st, int bufi, int pli, int y0, int yend) {
  const orc_plane_geom *g = st->fplanes + pli;
  int hpad = ORC_UMV_PADDING >> (pli != 0 && st->hdec);
  uint8_t *row = st->ref_plane_data[bufi][pli] + (ptrdiff_t)y0 * g->stride;
  int y;
  for (y = y0; y < yend; y++, row += g->stride) {
    memset(row - hpad, row[0], hpad);
    memset(row + g->width, row[g->width - 1], hpad);
  }
}

/* state.c:799-822: replicate the (already side-padded) first and last rows outwards. */
void orc_state_borders_fill_caps(orc_state *st, int bufi, int pli) {
  const orc_plane_geom *g = st->fplanes + pli;
  int hpad = ORC_UMV_PADDING >> (pli != 0 && st->hdec);
  int vpad = ORC_UMV_PADDING >> (pli != 0 && st->vdec);
  int fullw = g->width + 2 * hpad;
  uint8_t *a = st->ref_plane_data[bufi][pli] - hpad;                       /* row 0 */
  uint8_t *b = a + (ptrdiff_t)(g->height - 1) * g->stride;                 /* row height-1 */
  int k;
  for (k = 1; k <= vpad; k++) {
    memcpy(a - (ptrdiff_t)k * g->stride, a, fullw);
    memcpy(b + (ptrdiff_t)k * g->stride, b, fullw);
  }
}

/* state.c:959-1000 */
void orc_state_frag_recon(orc_state *st, ptrdiff_t fragi, int pli, int16_t dct_coeffs[128],
                          int last_zzi, uint16_t dc_quant) {
  uint8_t *dst;
  ptrdiff_t off;
  int ystride, refi;
  if (last_zzi < 2) {
    /* DC-only shortcut with its own rounding, state.c:967-975 */
    int16_t p = S16((dct_coeffs[0] * (int32_t)dc_quant + 15) >> 5);
    int ci;
    for (ci = 0; ci < 64; ci++) dct_coeffs[64 + ci] = p;
  } else {
    dct_coeffs[0] = S16(dct_coeffs[0] * (int)dc_quant);                    /* state.c:978 */
    orc_idct8x8(dct_coeffs + 64, dct_coeffs, last_zzi);
  }
  off = st->frag_buf_offs[fragi];
  refi = st->refi[fragi];
  ystride = st->fplanes[pli].stride;
  dst = st->ref_frame_data[ORC_FRAME_SELF] + off;
  if (refi == ORC_FRAME_SELF) orc_frag_recon_intra(dst, ystride, dct_coeffs + 64);
  else {
    const uint8_t *ref = st->ref_frame_data[refi] + off;
    int offs[2];
    int16_t mv = st->mvs[fragi];
    int dx = (int8_t)(mv & 0xFF);               /* OC_MV_X, state.h:233 */
    int dy = mv >> 8;                           /* OC_MV_Y, state.h:234 */
    if (orc_mv_offsets(offs, ystride, pli != 0 && st->hdec, pli != 0 && st->vdec, dx, dy) > 1)
      orc_frag_recon_inter2(dst, ref + offs[0], ref + offs[1], ystride, dct_coeffs + 64);
    else
      orc_frag_recon_inter(dst, ref + offs[0], ystride, dct_coeffs + 64);
  }
}

/* ------------------------------------------------------------------------- */
/* loop filter, state.c:1002-1105                                             */
/* ------------------------------------------------------------------------- */

void orc_loop_filter_init(int8_t bv[256], int flimit) {                   /* state.c:1036-1045 */
  int i;
  memset(bv, 0, 256);
  for (i = 0; i < flimit; i++) {
    if (127 - i - flimit >= 0) bv[127 - i - flimit] = (int8_t)(i - flimit);
    bv[127 - i] = (int8_t)(-i);
    bv[127 + i] = (int8_t)(i);
    if (127 + i + flimit < 256) bv[127 + i + flimit] = (int8_t)(flimit - i);
  }
}

#ifdef ORC_SIMD
/* The table of oc_loop_filter_init tabulates lflim(R) = sign(R) * min(|R|, max(2L - |R|, 0)) for R = (f + 4) >> 3 (state.c:1036-1045,
   spec 7.10); on vectors the function itself: with a = clamp(R, -L, L), lflim(R) = a - clamp(R - a, -L, L).  L is read back
   from the table (its largest entry: lflim(L) = L).  p2..p5: the four pixels across the edge as int16; returns the new p3, p4. */
static inline int lf_limit_of(const int8_t *bv127) {
  /* lflim(R) = R for 0 <= R <= L and falls from there: the first index whose entry is not its own number ends the ramp */
  int L = 0;
  while (L < 127 && bv127[L + 1] == L + 1) L++;
  return L;
}
static inline void lf_filter_v(__m128i p2, __m128i *p3, __m128i *p4, __m128i p5, int L) {
  const __m128i l = _mm_set1_epi16((short)L), nl = _mm_set1_epi16((short)-L);
  const __m128i d = _mm_sub_epi16(*p4, *p3);
  const __m128i f = _mm_add_epi16(_mm_sub_epi16(p2, p5), _mm_add_epi16(d, _mm_add_epi16(d, d)));
  const __m128i R = _mm_srai_epi16(_mm_add_epi16(f, _mm_set1_epi16(4)), 3);
  const __m128i a = _mm_max_epi16(_mm_min_epi16(R, l), nl);
  const __m128i b = _mm_max_epi16(_mm_min_epi16(_mm_sub_epi16(R, a), l), nl);
  const __m128i lf = _mm_sub_epi16(a, b);
  *p3 = _mm_add_epi16(*p3, lf);       /* clamped to 0..255 by the caller's packuswb */
  *p4 = _mm_sub_epi16(*p4, lf);
}
/* state.c:1002-1016: eight rows, four pixels each around the edge -- transposed into four vectors of eight, filtered, the two
   changed columns written back */
static void lf_across_vertical_edge(uint8_t *pix, int ystride, const int8_t *bv127) {
  const __m128i z = _mm_setzero_si128();
  const int L = lf_limit_of(bv127);
  __m128i r[8], p2, p3, p4, p5, o;
  uint8_t q3[16], q4[16];
  int y;
  for (y = 0; y < 8; y++) {
    int32_t w;
    memcpy(&w, pix + (ptrdiff_t)y * ystride - 2, 4);
    r[y] = _mm_unpacklo_epi8(_mm_cvtsi32_si128(w), z);      /* four int16: p2 p3 p4 p5 of row y */
  }
  {
    /* 8 rows x 4 columns of int16 -> 4 vectors of 8 */
    const __m128i a0 = _mm_unpacklo_epi16(r[0], r[1]), a1 = _mm_unpacklo_epi16(r[2], r[3]);
    const __m128i a2 = _mm_unpacklo_epi16(r[4], r[5]), a3 = _mm_unpacklo_epi16(r[6], r[7]);
    const __m128i b0 = _mm_unpacklo_epi32(a0, a1), b1 = _mm_unpackhi_epi32(a0, a1);
    const __m128i b2 = _mm_unpacklo_epi32(a2, a3), b3 = _mm_unpackhi_epi32(a2, a3);
    p2 = _mm_unpacklo_epi64(b0, b2);
    p3 = _mm_unpackhi_epi64(b0, b2);
    p4 = _mm_unpacklo_epi64(b1, b3);
    p5 = _mm_unpackhi_epi64(b1, b3);
  }
  lf_filter_v(p2, &p3, &p4, p5, L);
  o = _mm_packus_epi16(p3, p4);
  _mm_storeu_si128((__m128i *)q3, o);
  memcpy(q4, q3 + 8, 8);
  for (y = 0; y < 8; y++) {
    pix[(ptrdiff_t)y * ystride - 1] = q3[y];
    pix[(ptrdiff_t)y * ystride] = q4[y];
  }
}
/* state.c:1018-1031: eight columns at once, the four rows across the edge as they lie in memory */
static void lf_across_horizontal_edge(uint8_t *pix, int ystride, const int8_t *bv127) {
  const __m128i z = _mm_setzero_si128();
  const int L = lf_limit_of(bv127);
  const __m128i p2 = _mm_unpacklo_epi8(_mm_loadl_epi64((const __m128i *)(pix - 2 * (ptrdiff_t)ystride)), z);
  __m128i p3 = _mm_unpacklo_epi8(_mm_loadl_epi64((const __m128i *)(pix - (ptrdiff_t)ystride)), z);
  __m128i p4 = _mm_unpacklo_epi8(_mm_loadl_epi64((const __m128i *)pix), z);
  const __m128i p5 = _mm_unpacklo_epi8(_mm_loadl_epi64((const __m128i *)(pix + ystride)), z);
  lf_filter_v(p2, &p3, &p4, p5, L);
  _mm_storel_epi64((__m128i *)(pix - (ptrdiff_t)ystride), _mm_packus_epi16(p3, p3));
  _mm_storel_epi64((__m128i *)pix, _mm_packus_epi16(p4, p4));
}
#else
/* filter across a vertical edge: pix points at the first column right of it, state.c:1002-1016 */
static void lf_across_vertical_edge(uint8_t *pix, int ystride, const int8_t *bv127) {
  int y;
  for (y = 0; y < 8; y++, pix += ystride) {
    int f = pix[-2] - pix[1] + 3 * (pix[0] - pix[-1]);
    f = bv127[(f + 4) >> 3];
    pix[-1] = clamp255(pix[-1] + f);
    pix[0] = clamp255(pix[0] - f);
  }
}

/* filter across a horizontal edge: pix points at the first row past it, state.c:1018-1031 */
static void lf_across_horizontal_edge(uint8_t *pix, int ystride, const int8_t *bv127) {
  int x;
  for (x = 0; x < 8; x++) {
    uint8_t *p = pix + x;
    int f = p[-2 * (ptrdiff_t)ystride] - p[ystride] + 3 * (p[0] - p[-(ptrdiff_t)ystride]);
    f = bv127[(f + 4) >> 3];
    p[-(ptrdiff_t)ystride] = clamp255(p[-(ptrdiff_t)ystride] + f);
    p[0] = clamp255(p[0] - f);
  }
}
#endif

void orc_state_loop_filter_frag_rows(orc_state *st, int8_t bvarray[256], int slot, int pli,
                                     int fragy0, int fragy_end) {          /* state.c:1055-1105 */
  const orc_plane_geom *g = st->fplanes + pli;
  const int8_t *bv = bvarray + 127;
  uint8_t *frame = st->ref_frame_data[slot];
  int ystride = g->stride;
  int fy, fx;
  for (fy = fragy0; fy < fragy_end; fy++) {
    for (fx = 0; fx < g->nhfrags; fx++) {
      ptrdiff_t fragi = g->froffset + (ptrdiff_t)fy * g->nhfrags + fx;
      uint8_t *ref;
      if (!st->coded[fragi]) continue;
      ref = frame + st->frag_buf_offs[fragi];
      if (fx > 0) lf_across_vertical_edge(ref, ystride, bv);
      if (fy > 0) lf_across_horizontal_edge(ref, ystride, bv);
      if (fx + 1 < g->nhfrags && !st->coded[fragi + 1]) lf_across_vertical_edge(ref + 8, ystride, bv);
      if (fy + 1 < g->nvfrags && !st->coded[fragi + g->nhfrags])
        lf_across_horizontal_edge(ref + 8 * (ptrdiff_t)ystride, ystride, bv);
    }
  }
}

/* ------------------------------------------------------------------------- */
/* coded order, state.c:123-190                                               */
/* ------------------------------------------------------------------------- */

ptrdiff_t orc_sb_order(const orc_state *st, int pli, ptrdiff_t *out) {
  /* (row,col) of the k-th block along the 4x4 Hilbert curve.  Derived from the
     reference's SB_MAP[i][j]={quadrant,block} table (state.c:134-139): curve
     position = quadrant*4+block. */
  static const uint8_t HILBERT_RC[16][2] = {
    {0, 0}, {0, 1}, {1, 1}, {1, 0}, {2, 0}, {3, 0}, {3, 1}, {2, 1},
    {2, 2}, {3, 2}, {3, 3}, {2, 3}, {1, 3}, {1, 2}, {0, 2}, {0, 3}};
  const orc_plane_geom *g = st->fplanes + pli;
  ptrdiff_t n = 0;
  int sby, sbx, k;
  for (sby = 0; sby < g->nvfrags; sby += 4)
    for (sbx = 0; sbx < g->nhfrags; sbx += 4)
      for (k = 0; k < 16; k++) {
        int fy = sby + HILBERT_RC[k][0], fx = sbx + HILBERT_RC[k][1];
        if (fy < g->nvfrags && fx < g->nhfrags) out[n++] = g->froffset + (ptrdiff_t)fy * g->nhfrags + fx;
      }
  return n;
}

/* ------------------------------------------------------------------------- */
/* DC un-prediction, decode.c:1392-1500                                       */
/* ------------------------------------------------------------------------- */

ptrdiff_t orc_dc_unpredict_rows(orc_state *st, int pli, int fragy0, int fragy_end, int pred_last[3]) {
  const orc_plane_geom *g = st->fplanes + pli;
  int nh = g->nhfrags;
  ptrdiff_t ncoded = 0;
  int fy, fx;
  for (fy = fragy0; fy < fragy_end; fy++) {
    ptrdiff_t row = g->froffset + (ptrdiff_t)fy * nh;
    for (fx = 0; fx < nh; fx++) {
      ptrdiff_t fi = row + fx;
      int refi, pred, mask = 0;
      int l = 0, ul = 0, u = 0, ur = 0;
      if (!st->coded[fi]) continue;
      refi = st->refi[fi];
      /* a neighbour counts when it is coded and predicted from the same frame
         (uncoded fragments carry refi NONE in the reference, decode.c:1432-1441) */
      if (fx > 0 && st->coded[fi - 1] && st->refi[fi - 1] == refi) { mask |= 1; l = st->dc[fi - 1]; }
      if (fy > 0) {
        if (fx > 0 && st->coded[fi - nh - 1] && st->refi[fi - nh - 1] == refi) { mask |= 2; ul = st->dc[fi - nh - 1]; }
        if (st->coded[fi - nh] && st->refi[fi - nh] == refi) { mask |= 4; u = st->dc[fi - nh]; }
        if (fx + 1 < nh && st->coded[fi - nh + 1] && st->refi[fi - nh + 1] == refi) { mask |= 8; ur = st->dc[fi - nh + 1]; }
      }
      switch (mask) {                                       /* decode.c:1450-1485 */
        default: pred = pred_last[refi]; break;
        case 1: case 3: pred = l; break;
        case 2: pred = ul; break;
        case 4: case 6: case 12: pred = u; break;
        case 5: pred = (l + u) / 2; break;
        case 8: pred = ur; break;
        case 9: case 11: case 13: pred = (75 * l + 53 * ur) / 128; break;
        case 10: pred = (ul + ur) / 2; break;
        case 14: pred = (3 * (ul + ur) + 10 * u) / 16; break;
        case 7: case 15:
          pred = (29 * (l + u) - 26 * ul) / 32;
          if (abs(pred - u) > 128) pred = u;
          else if (abs(pred - l) > 128) pred = l;
          else if (abs(pred - ul) > 128) pred = ul;
          break;
      }
      /* dc is a signed 16-bit bitfield in the reference (state.h:321): the sum wraps */
      st->dc[fi] = S16(st->dc[fi] + pred);
      pred_last[refi] = st->dc[fi];
      ncoded++;
    }
  }
  return ncoded;
}

/* ------------------------------------------------------------------------- */
/* one frame, decode.c:2757-2962                                              */
/* ------------------------------------------------------------------------- */

static void init_dummy_frame(orc_state *st) {               /* decode.c:2053-2080 */
  memset(st->ref_slab, 0x80, st->ref_frame_sz);
  orc_state_set_ref_idx(st, 0, 0, 0);
}

int orc_decode_frame(orc_state *st, int frame_type, const ptrdiff_t *coded_fragis,
                     const ptrdiff_t ncoded[3], const int16_t *coeffs, const uint8_t *last_zzi,
                     const uint16_t *dc_quant, const ptrdiff_t *uncoded_fragis,
                     ptrdiff_t nuncoded, int flimit) {
  int8_t bv[256];
  int16_t block[128];
  ptrdiff_t done[3], base[3], k, ntotal;
  ptrdiff_t *copy_list;
  int pli, bufi, mcu_nvfrags, stripe, notstart, notdone;
  ntotal = ncoded[0] + ncoded[1] + ncoded[2];
  if (ntotal + nuncoded != st->nfrags) return -1;
  for (k = 0; k < ntotal; k++) st->coded[coded_fragis[k]] = 1;
  for (k = 0; k < nuncoded; k++) st->coded[uncoded_fragis[k]] = 0;
  if (frame_type != ORC_INTRA_FRAME &&
      (st->ref_frame_idx[ORC_FRAME_GOLD] < 0 || st->ref_frame_idx[ORC_FRAME_PREV] < 0))
    init_dummy_frame(st);                                    /* decode.c:2757-2762 */
  if (ntotal <= 0) return 1;                                 /* TH_DUPFRAME, decode.c:2764-2772 */
  for (bufi = 0; bufi == st->ref_frame_idx[ORC_FRAME_GOLD] || bufi == st->ref_frame_idx[ORC_FRAME_PREV]; bufi++);
  st->ref_frame_idx[ORC_FRAME_SELF] = bufi;                  /* decode.c:2790-2794 */
  st->ref_frame_data[ORC_FRAME_SELF] = st->ref_plane_data[bufi][0];
  if (flimit) orc_loop_filter_init(bv, flimit);              /* decode.c:1369-1371 */
  base[0] = 0; base[1] = ncoded[0]; base[2] = ncoded[0] + ncoded[1];
  done[0] = done[1] = done[2] = 0;
  copy_list = (ptrdiff_t *)malloc(sizeof(ptrdiff_t) * (size_t)(st->nfrags ? st->nfrags : 1));
  mcu_nvfrags = 4 << st->vdec;                               /* decode.c:1341 */
  notstart = 0; notdone = 1;
  for (stripe = 0; notdone; stripe += mcu_nvfrags) {         /* decode.c:2858 */
    notdone = stripe + mcu_nvfrags < st->fplanes[0].nvfrags;
    for (pli = 0; pli < 3; pli++) {
      const orc_plane_geom *g = st->fplanes + pli;
      int frag_shift = pli != 0 && st->vdec;
      int fragy0 = stripe >> frag_shift;
      int fragy_end = fragy0 + (mcu_nvfrags >> frag_shift);
      ptrdiff_t fi, fi_end, nc = 0, nu = 0;
      int sdelay, edelay;
      if (fragy_end > g->nvfrags) fragy_end = g->nvfrags;
      /* the reference counts the MCU's coded fragments during DC un-prediction
         (decode.c:1412-1498); DC values arrive here already un-predicted */
      fi = g->froffset + (ptrdiff_t)fragy0 * g->nhfrags;
      fi_end = g->froffset + (ptrdiff_t)fragy_end * g->nhfrags;
      for (; fi < fi_end; fi++) {
        if (st->coded[fi]) nc++;
        else copy_list[nu++] = fi;
      }
      /* decode.c:1530-1586: the next nc entries of this plane's coded list */
      for (k = 0; k < nc; k++) {
        ptrdiff_t slot = base[pli] + done[pli] + k;
        memcpy(block, coeffs + slot * 64, 64 * sizeof(int16_t));
        orc_state_frag_recon(st, coded_fragis[slot], pli, block, last_zzi[slot], dc_quant[slot]);
      }
      done[pli] += nc;
      if (nu > 0)                                            /* decode.c:1599-1606 */
        orc_frag_copy_list(st->ref_frame_data[ORC_FRAME_SELF], st->ref_frame_data[ORC_FRAME_PREV],
                           g->stride, copy_list, nu, st->frag_buf_offs);
      sdelay = edelay = 0;
      if (flimit) {                                          /* decode.c:2879-2884 */
        sdelay += notstart;
        edelay += notdone;
        orc_state_loop_filter_frag_rows(st, bv, ORC_FRAME_SELF, pli, fragy0 - sdelay, fragy_end - edelay);
      }
      orc_state_borders_fill_rows(st, bufi, pli, ((fragy0 - sdelay) << 3) - (sdelay << 1),
                                  ((fragy_end - edelay) << 3) - (edelay << 1));   /* decode.c:2890-2892 */
    }
    notstart = 1;
  }
  free(copy_list);
  for (pli = 0; pli < 3; pli++) orc_state_borders_fill_caps(st, bufi, pli);   /* decode.c:2945 */
  if (frame_type == ORC_INTRA_FRAME) orc_state_set_ref_idx(st, bufi, bufi, bufi);  /* decode.c:2947-2955 */
  else orc_state_set_ref_idx(st, st->ref_frame_idx[ORC_FRAME_GOLD], bufi, bufi);   /* decode.c:2956-2962 */
  return 0;
}

/* ------------------------------------------------------------------------- */
/* out-of-loop post-processing, decode.c:1608-1957                            */
/* ------------------------------------------------------------------------- */

#define ORC_MINI(a, b) ((a) < (b) ? (a) : (b))
#define ORC_CLAMPI(a, b, c) ((b) < (a) ? (a) : ((b) > (c) ? (c) : (b)))

/* decode.c:1610-1661 */
static void pp_filter_hedge(uint8_t *dst, int dst_ystride, const uint8_t *src, int src_ystride, int qstep, int flimit,
                            int *variance0, int *variance1) {
  uint8_t *rdst = dst;
  const uint8_t *rsrc = src;
  int bx, by;
  for (bx = 0; bx < 8; bx++) {
    uint8_t *cdst = rdst;
    const uint8_t *csrc = rsrc;
    int r[10], sum0 = 0, sum1 = 0;
    for (by = 0; by < 10; by++) {
      r[by] = *csrc;
      csrc += src_ystride;
    }
    for (by = 0; by < 4; by++) {
      sum0 += abs(r[by + 1] - r[by]);
      sum1 += abs(r[by + 5] - r[by + 6]);
    }
    *variance0 += ORC_MINI(255, sum0);
    *variance1 += ORC_MINI(255, sum1);
    if (sum0 < flimit && sum1 < flimit && r[5] - r[4] < qstep && r[4] - r[5] < qstep) {
      *cdst = (uint8_t)((r[0] * 3 + r[1] * 2 + r[2] + r[3] + r[4] + 4) >> 3);
      cdst += dst_ystride;
      *cdst = (uint8_t)((r[0] * 2 + r[1] + r[2] * 2 + r[3] + r[4] + r[5] + 4) >> 3);
      cdst += dst_ystride;
      for (by = 0; by < 4; by++) {
        *cdst = (uint8_t)((r[by] + r[by + 1] + r[by + 2] + r[by + 3] * 2 + r[by + 4] + r[by + 5] + r[by + 6] + 4) >> 3);
        cdst += dst_ystride;
      }
      *cdst = (uint8_t)((r[4] + r[5] + r[6] + r[7] * 2 + r[8] + r[9] * 2 + 4) >> 3);
      cdst += dst_ystride;
      *cdst = (uint8_t)((r[5] + r[6] + r[7] + r[8] * 2 + r[9] * 3 + 4) >> 3);
    } else {
      for (by = 1; by <= 8; by++) {
        *cdst = (uint8_t)r[by];
        cdst += dst_ystride;
      }
    }
    rdst++;
    rsrc++;
  }
}

/* decode.c:1663-1694 */
static void pp_filter_vedge(uint8_t *dst, int dst_ystride, int qstep, int flimit, int *variances) {
  uint8_t *cdst = dst;
  int bx, by;
  for (by = 0; by < 8; by++) {
    const uint8_t *rsrc = cdst - 1;
    uint8_t *rdst = cdst;
    int r[10], sum0 = 0, sum1 = 0;
    for (bx = 0; bx < 10; bx++) r[bx] = *rsrc++;
    for (bx = 0; bx < 4; bx++) {
      sum0 += abs(r[bx + 1] - r[bx]);
      sum1 += abs(r[bx + 5] - r[bx + 6]);
    }
    variances[0] += ORC_MINI(255, sum0);
    variances[1] += ORC_MINI(255, sum1);
    if (sum0 < flimit && sum1 < flimit && r[5] - r[4] < qstep && r[4] - r[5] < qstep) {
      *rdst++ = (uint8_t)((r[0] * 3 + r[1] * 2 + r[2] + r[3] + r[4] + 4) >> 3);
      *rdst++ = (uint8_t)((r[0] * 2 + r[1] + r[2] * 2 + r[3] + r[4] + r[5] + 4) >> 3);
      for (bx = 0; bx < 4; bx++)
        *rdst++ = (uint8_t)((r[bx] + r[bx + 1] + r[bx + 2] + r[bx + 3] * 2 + r[bx + 4] + r[bx + 5] + r[bx + 6] + 4) >> 3);
      *rdst++ = (uint8_t)((r[4] + r[5] + r[6] + r[7] * 2 + r[8] + r[9] * 2 + 4) >> 3);
      *rdst = (uint8_t)((r[5] + r[6] + r[7] + r[8] * 2 + r[9] * 3 + 4) >> 3);
    }
    cdst += dst_ystride;
  }
}

/* decode.c:1696-1786.  dst / src: pixel (0,0) of the plane in bitstream coordinates, any stride sign;
   variances and dc_qis: the plane's own rows (already offset by froffset). */
void orc_pp_deblock_frag_rows(uint8_t *dst_data, int dst_ystride, const uint8_t *src_data, int src_ystride, int width, int height,
                              int nhfrags, int nvfrags, int *variances, const uint8_t *dc_qis, const int pp_dc_scale[64],
                              int fragy0, int fragy_end) {
  int *variance = variances + (ptrdiff_t)fragy0 * nhfrags;
  const uint8_t *dc_qi = dc_qis + (ptrdiff_t)fragy0 * nhfrags;
  int notstart = fragy0 > 0, notdone = fragy_end < nvfrags;
  int flimit, qstep, y_end, y, x;
  uint8_t *dst;
  const uint8_t *src;
  /* "We want to clear an extra row of variances, except at the end." */
  if (fragy_end + notdone - fragy0 - notstart > 0)
    memset(variance + (nhfrags & -notstart), 0, (size_t)(fragy_end + notdone - fragy0 - notstart) * ((size_t)nhfrags * sizeof(variance[0])));
  y = (fragy0 << 3) + (notstart << 2);
  dst = dst_data + y * (ptrdiff_t)dst_ystride;
  src = src_data + y * (ptrdiff_t)src_ystride;
  for (; y < 4; y++) {
    memcpy(dst, src, (size_t)width);
    dst += dst_ystride;
    src += src_ystride;
  }
  y_end = (fragy_end - !notdone) << 3;
  for (; y < y_end; y += 8) {
    qstep = pp_dc_scale[*dc_qi];
    flimit = (qstep * 3) >> 2;
    pp_filter_hedge(dst, dst_ystride, src - src_ystride, src_ystride, qstep, flimit, variance, variance + nhfrags);
    variance++;
    dc_qi++;
    for (x = 8; x < width; x += 8) {
      qstep = pp_dc_scale[*dc_qi];
      flimit = (qstep * 3) >> 2;
      pp_filter_hedge(dst + x, dst_ystride, src + x - src_ystride, src_ystride, qstep, flimit, variance, variance + nhfrags);
      pp_filter_vedge(dst + x - (dst_ystride * 4) - 4, dst_ystride, qstep, flimit, variance - 1);
      variance++;
      dc_qi++;
    }
    dst += dst_ystride * 8;
    src += src_ystride * 8;
  }
  if (!notdone) {
    for (; y < height; y++) {
      memcpy(dst, src, (size_t)width);
      dst += dst_ystride;
      src += src_ystride;
    }
    dc_qi++;
    for (x = 8; x < width; x += 8) {
      qstep = pp_dc_scale[*dc_qi++];
      flimit = (qstep * 3) >> 2;
      pp_filter_vedge(dst + x - (dst_ystride * 8) - 4, dst_ystride, qstep, flimit, variance++);
    }
  }
}

/* decode.c:1788-1890 */
static void pp_dering_block(uint8_t *idata, int ystride, int b, int dc_scale, int sharp_mod, int strong) {
  static const unsigned char MOD_MAX[2] = {24, 32};
  static const unsigned char MOD_SHIFT[2] = {1, 0};
  const uint8_t *psrc, *src, *nsrc;
  uint8_t *dst;
  int vmod[72], hmod[72];
  int mod_hi, by, bx;
  mod_hi = ORC_MINI(3 * dc_scale, MOD_MAX[strong]);
  dst = idata;
  src = dst;
  psrc = src - (ystride & -!(b & 4));
  for (by = 0; by < 9; by++) {
    for (bx = 0; bx < 8; bx++) {
      int mod = 32 + dc_scale - (abs(src[bx] - psrc[bx]) << MOD_SHIFT[strong]);
      vmod[(by << 3) + bx] = mod < -64 ? sharp_mod : ORC_CLAMPI(0, mod, mod_hi);
    }
    psrc = src;
    src += ystride & -(!(b & 8) | (by < 7));
  }
  nsrc = dst;
  psrc = dst - !(b & 1);
  for (bx = 0; bx < 9; bx++) {
    src = nsrc;
    for (by = 0; by < 8; by++) {
      int mod = 32 + dc_scale - (abs(*src - *psrc) << MOD_SHIFT[strong]);
      hmod[(bx << 3) + by] = mod < -64 ? sharp_mod : ORC_CLAMPI(0, mod, mod_hi);
      psrc += ystride;
      src += ystride;
    }
    psrc = nsrc;
    nsrc += !(b & 2) | (bx < 7);
  }
  src = dst;
  psrc = src - (ystride & -!(b & 4));
  nsrc = src + ystride;
  for (by = 0; by < 8; by++) {
    int a, bb, w;
    a = 128; bb = 64;
    w = hmod[by]; a -= w; bb += w * *(src - !(b & 1));
    w = vmod[by << 3]; a -= w; bb += w * psrc[0];
    w = vmod[(by + 1) << 3]; a -= w; bb += w * nsrc[0];
    w = hmod[(1 << 3) + by]; a -= w; bb += w * src[1];
    dst[0] = (uint8_t)ORC_CLAMPI(0, (a * src[0] + bb) >> 7, 255);
    for (bx = 1; bx < 7; bx++) {
      a = 128; bb = 64;
      w = hmod[(bx << 3) + by]; a -= w; bb += w * src[bx - 1];
      w = vmod[(by << 3) + bx]; a -= w; bb += w * psrc[bx];
      w = vmod[((by + 1) << 3) + bx]; a -= w; bb += w * nsrc[bx];
      w = hmod[((bx + 1) << 3) + by]; a -= w; bb += w * src[bx + 1];
      dst[bx] = (uint8_t)ORC_CLAMPI(0, (a * src[bx] + bb) >> 7, 255);
    }
    a = 128; bb = 64;
    w = hmod[(7 << 3) + by]; a -= w; bb += w * src[6];
    w = vmod[(by << 3) + 7]; a -= w; bb += w * psrc[7];
    w = vmod[((by + 1) << 3) + 7]; a -= w; bb += w * nsrc[7];
    w = hmod[(8 << 3) + by]; a -= w; bb += w * src[7 + !(b & 2)];
    dst[7] = (uint8_t)ORC_CLAMPI(0, (a * src[7] + bb) >> 7, 255);
    dst += ystride;
    psrc = src;
    src = nsrc;
    nsrc += ystride & -(!(b & 8) | (by < 6));
  }
}

#define ORC_DERING_THRESH1 (384)
#define ORC_DERING_THRESH2 (4 * ORC_DERING_THRESH1)
#define ORC_DERING_THRESH3 (5 * ORC_DERING_THRESH1)
#define ORC_DERING_THRESH4 (10 * ORC_DERING_THRESH1)

/* decode.c:1897-1957.  frag_qi[f] = qis[frags[f].qii] (decode.c:1926), the plane's own rows. */
void orc_pp_dering_frag_rows(uint8_t *img_data, int ystride, int width, int height, int nhfrags, const int *variances,
                             const uint8_t *frag_qi, const int pp_dc_scale[64], const int pp_sharp_mod[64], int strong, int pli,
                             int fragy0, int fragy_end) {
  const int *variance = variances + (ptrdiff_t)fragy0 * nhfrags;
  const uint8_t *fq = frag_qi + (ptrdiff_t)fragy0 * nhfrags;
  int sthresh = pli ? ORC_DERING_THRESH4 : ORC_DERING_THRESH3;
  int y = fragy0 << 3, y_end = fragy_end << 3, x;
  uint8_t *idata = img_data + y * (ptrdiff_t)ystride;
  for (; y < y_end; y += 8) {
    for (x = 0; x < width; x += 8) {
      int qi = *fq, var = *variance;
      int b = (x <= 0) | (x + 8 >= width) << 1 | (y <= 0) << 2 | (y + 8 >= height) << 3;
      if (strong && var > sthresh) {
        pp_dering_block(idata + x, ystride, b, pp_dc_scale[qi], pp_sharp_mod[qi], 1);
        if (pli || (!(b & 1) && *(variance - 1) > ORC_DERING_THRESH4) || (!(b & 2) && variance[1] > ORC_DERING_THRESH4) ||
            (!(b & 4) && *(variance - nhfrags) > ORC_DERING_THRESH4) || (!(b & 8) && variance[nhfrags] > ORC_DERING_THRESH4)) {
          pp_dering_block(idata + x, ystride, b, pp_dc_scale[qi], pp_sharp_mod[qi], 1);
          pp_dering_block(idata + x, ystride, b, pp_dc_scale[qi], pp_sharp_mod[qi], 1);
        }
      } else if (var > ORC_DERING_THRESH2) {
        pp_dering_block(idata + x, ystride, b, pp_dc_scale[qi], pp_sharp_mod[qi], 1);
      } else if (var > ORC_DERING_THRESH1) {
        pp_dering_block(idata + x, ystride, b, pp_dc_scale[qi], pp_sharp_mod[qi], 0);
      }
      fq++;
      variance++;
    }
    idata += (ptrdiff_t)ystride * 8;
  }
}

/* Post-processing of the frame in reference slot `slot` as th_decode_packetin's MCU loop drives it
   (decode.c:2858-2945): per MCU and plane, de-blocking one fragment row behind the loop filter and
   de-ringing one row behind that (sdelay / edelay), into `out` (three tightly packed planes, bitstream row
   order).  level: decode.c:32-48 (2 de-block Y, 3 de-ring Y, 4 strong Y, 5-7 the same for chroma); planes
   below their level are copies of the reference frame's (decode.c:1319-1323).  dc_qis, frag_qi: per fragment. */
void orc_postprocess_frame(const orc_state *st, int slot, int level, int loop_filter, const uint8_t *dc_qis,
                           const uint8_t *frag_qi, const int pp_dc_scale[64], const int pp_sharp_mod[64], uint8_t *out,
                           int *variances) {
  int bufi = st->ref_frame_idx[slot];
  int mcu_nvfrags = 4 << st->vdec;
  int pli, stripe, notstart = 0, notdone = 1;
  uint8_t *pp[3];
  size_t off = 0;
  for (pli = 0; pli < 3; pli++) {
    const orc_plane_geom *g = st->fplanes + pli;
    int y;
    pp[pli] = out + off;
    off += (size_t)g->width * g->height;
    /* planes that are not post-processed: the reference's own (decode.c:1319-1323, :1380-1382) */
    if (level < 2 + 3 * (pli != 0))
      for (y = 0; y < g->height; y++)
        memcpy(pp[pli] + (size_t)y * g->width, st->ref_plane_data[bufi][pli] + (ptrdiff_t)y * g->stride, (size_t)g->width);
  }
  for (stripe = 0; notdone; stripe += mcu_nvfrags) {
    notdone = stripe + mcu_nvfrags < st->fplanes[0].nvfrags;
    for (pli = 0; pli < 3; pli++) {
      const orc_plane_geom *g = st->fplanes + pli;
      int frag_shift = pli != 0 && st->vdec;
      int fragy0 = stripe >> frag_shift;
      int fragy_end = fragy0 + (mcu_nvfrags >> frag_shift);
      int sdelay = 0, edelay = 0, pp_offset = 3 * (pli != 0);
      if (fragy_end > g->nvfrags) fragy_end = g->nvfrags;
      if (loop_filter) {
        sdelay += notstart;
        edelay += notdone;
      }
      if (level >= 2 + pp_offset) {                          /* decode.c:2895-2911 */
        sdelay += notstart;
        edelay += notdone;
        orc_pp_deblock_frag_rows(pp[pli], g->width, st->ref_plane_data[bufi][pli], g->stride, g->width, g->height,
                                 g->nhfrags, g->nvfrags, variances + g->froffset, dc_qis + g->froffset, pp_dc_scale,
                                 fragy0 - sdelay, fragy_end - edelay);
        if (level >= 3 + pp_offset) {
          sdelay += notstart;
          edelay += notdone;
          orc_pp_dering_frag_rows(pp[pli], g->width, g->width, g->height, g->nhfrags, variances + g->froffset,
                                  frag_qi + g->froffset, pp_dc_scale, pp_sharp_mod, level >= (pli ? 7 : 4), pli,
                                  fragy0 - sdelay, fragy_end - edelay);
        }
      }
    }
    notstart = 1;
  }
}

/* ------------------------------------------------------------------------- */
/* encoder block kernels, encfrag.c / fdct.c                                  */
/* ------------------------------------------------------------------------- */

void orc_enc_frag_sub(int16_t diff[64], const uint8_t *src, const uint8_t *ref, int ystride) {
  int i, j;
  for (i = 0; i < 8; i++, src += ystride, ref += ystride)
    for (j = 0; j < 8; j++) diff[i * 8 + j] = S16(src[j] - ref[j]);
}

void orc_enc_frag_sub_128(int16_t diff[64], const uint8_t *src, int ystride) {
  int i, j;
  for (i = 0; i < 8; i++, src += ystride)
    for (j = 0; j < 8; j++) diff[i * 8 + j] = S16(src[j] - 128);
}

unsigned orc_enc_frag_sad(const uint8_t *src, const uint8_t *ref, int ystride) {
  unsigned sad = 0;
  int i, j;
  for (i = 0; i < 8; i++, src += ystride, ref += ystride)
    for (j = 0; j < 8; j++) sad += abs(src[j] - ref[j]);
  return sad;
}

unsigned orc_enc_frag_sad_thresh(const uint8_t *src, const uint8_t *ref, int ystride, unsigned thresh) {
  unsigned sad = 0;
  int i, j;
  for (i = 0; i < 8; i++, src += ystride, ref += ystride) {
    for (j = 0; j < 8; j++) sad += abs(src[j] - ref[j]);
    if (sad > thresh) break;                                 /* row-granular early out, encfrag.c:64 */
  }
  return sad;
}

unsigned orc_enc_frag_sad2_thresh(const uint8_t *src, const uint8_t *ref1, const uint8_t *ref2,
                                  int ystride, unsigned thresh) {
  unsigned sad = 0;
  int i, j;
  for (i = 0; i < 8; i++, src += ystride, ref1 += ystride, ref2 += ystride) {
    for (j = 0; j < 8; j++) sad += abs(src[j] - ((ref1[j] + ref2[j]) >> 1));
    if (sad > thresh) break;
  }
  return sad;
}

unsigned orc_enc_frag_intra_sad(const uint8_t *src, int ystride) {
  const uint8_t *p = src;
  unsigned sad = 0;
  int dc = 0, i, j;
  for (i = 0; i < 8; i++, p += ystride)
    for (j = 0; j < 8; j++) dc += p[j];
  dc = (dc + 32) >> 6;
  for (i = 0; i < 8; i++, src += ystride)
    for (j = 0; j < 8; j++) sad += abs(src[j] - dc);
  return sad;
}

/* 8-point Hadamard butterfly network used by both passes (encfrag.c:122-154, :279-310):
   outputs in the reference's order (t0+t1, t0-t1, t2+t3, ...). */
static void hadamard8(int o[8], const int v[8]) {
  int t0 = v[0] + v[4], t4 = v[0] - v[4];
  int t1 = v[1] + v[5], t5 = v[1] - v[5];
  int t2 = v[2] + v[6], t6 = v[2] - v[6];
  int t3 = v[3] + v[7], t7 = v[3] - v[7];
  int r;
  r = t0; t0 += t2; t2 = r - t2;
  r = t1; t1 += t3; t3 = r - t3;
  r = t4; t4 += t6; t6 = r - t6;
  r = t5; t5 += t7; t7 = r - t7;
  o[0] = t0 + t1; o[1] = t0 - t1; o[2] = t2 + t3; o[3] = t2 - t3;
  o[4] = t4 + t5; o[5] = t4 - t5; o[6] = t6 + t7; o[7] = t6 - t7;
}

/* second pass + absolute sum, encfrag.c:264-315 */
static unsigned hadamard_sad(int *dc_out, const int16_t buf[64]) {
  unsigned sad = 0;
  int i, k;
  for (i = 0; i < 8; i++) {
    int v[8], o[8];
    for (k = 0; k < 8; k++) v[k] = buf[i * 8 + k];
    hadamard8(o, v);
    for (k = 0; k < 8; k++)
      if (i > 0 || k > 0) sad += abs(o[k]);                  /* the DC term is left out, encfrag.c:302 */
  }
  *dc_out = buf[0] + buf[1] + buf[2] + buf[3] + buf[4] + buf[5] + buf[6] + buf[7];
  return sad;
}

/* first pass over rows of (src - pred); row i's outputs go to column i, encfrag.c:109-158 */
static void row_hadamard(int16_t buf[64], int i, const int d[8]) {
  int o[8], k;
  hadamard8(o, d);
  for (k = 0; k < 8; k++) buf[k * 8 + i] = S16(o[k]);
}

unsigned orc_enc_frag_satd(int *dc, const uint8_t *src, const uint8_t *ref, int ystride) {
  int16_t buf[64];
  int i, j, d[8];
  for (i = 0; i < 8; i++, src += ystride, ref += ystride) {
    for (j = 0; j < 8; j++) d[j] = src[j] - ref[j];
    row_hadamard(buf, i, d);
  }
  return hadamard_sad(dc, buf);
}

unsigned orc_enc_frag_satd2(int *dc, const uint8_t *src, const uint8_t *ref1, const uint8_t *ref2,
                            int ystride) {
  int16_t buf[64];
  int i, j, d[8];
  for (i = 0; i < 8; i++, src += ystride, ref1 += ystride, ref2 += ystride) {
    for (j = 0; j < 8; j++) d[j] = src[j] - ((ref1[j] + ref2[j]) >> 1);
    row_hadamard(buf, i, d);
  }
  return hadamard_sad(dc, buf);
}

unsigned orc_enc_frag_intra_satd(int *dc, const uint8_t *src, int ystride) {
  int16_t buf[64];
  int i, j, d[8];
  for (i = 0; i < 8; i++, src += ystride) {
    for (j = 0; j < 8; j++) d[j] = src[j];
    row_hadamard(buf, i, d);
  }
  return hadamard_sad(dc, buf);
}

unsigned orc_enc_frag_ssd(const uint8_t *src, const uint8_t *ref, int ystride) {
  unsigned ret = 0;
  int y, x;
  for (y = 0; y < 8; y++, src += ystride, ref += ystride)
    for (x = 0; x < 8; x++) ret += (unsigned)((src[x] - ref[x]) * (src[x] - ref[x]));
  return ret;
}

unsigned orc_enc_frag_border_ssd(const uint8_t *src, const uint8_t *ref, int ystride, int64_t mask) {
  uint64_t m = (uint64_t)mask;
  unsigned ret = 0;
  int y, x;
  for (y = 0; y < 8; y++, src += ystride, ref += ystride)
    for (x = 0; x < 8; x++, m >>= 1)
      if (m & 1) ret += (unsigned)((src[x] - ref[x]) * (src[x] - ref[x]));
  return ret;
}

void orc_enc_frag_copy2(uint8_t *dst, const uint8_t *src1, const uint8_t *src2, int ystride) {
  int i, j;
  for (i = 0; i < 8; i++, dst += ystride, src1 += ystride, src2 += ystride)
    for (j = 0; j < 8; j++) dst[j] = (uint8_t)((src1[j] + src2[j]) >> 1);
}

/* fdct.c:28-120.  Reads x[0], x[8], ..., x[56]; writes y[0..7] (int16 stores). */
static void fdct8(int16_t y[8], const int16_t *x) {
  int t0, t1, t2, t3, t4, t5, t6, t7, r, s, u, v;
  t0 = x[0 * 8] + (int)x[7 * 8]; t7 = x[0 * 8] - (int)x[7 * 8];
  t1 = x[1 * 8] + (int)x[6 * 8]; t6 = x[1 * 8] - (int)x[6 * 8];
  t2 = x[2 * 8] + (int)x[5 * 8]; t5 = x[2 * 8] - (int)x[5 * 8];
  t3 = x[3 * 8] + (int)x[4 * 8]; t4 = x[3 * 8] - (int)x[4 * 8];
  r = t0 + t3; t3 = t0 - t3; t0 = r;
  r = t1 + t2; t2 = t1 - t2; t1 = r;
  r = t6 + t5; t5 = t6 - t5; t6 = r;
  /* stage 3, fdct.c:85-93 */
  s = (((27146 * t5 + 0xB500) >> 16) + t5 + (t5 != 0)) >> 1;
  r = t4 + s; t5 = t4 - s; t4 = r;
  s = (((27146 * t6 + 0xB500) >> 16) + t6 + (t6 != 0)) >> 1;
  r = t7 + s; t6 = t7 - s; t7 = r;
  /* stage 4, fdct.c:95-119 */
  r = ((27146 * t0 + 0x4000) >> 16) + t0 + (t0 != 0);
  s = ((27146 * t1 + 0xB500) >> 16) + t1 + (t1 != 0);
  u = (r + s) >> 1;
  v = r - u;
  y[0] = S16(u);
  y[4] = S16(v);
  u = ((C6 * t2 + C2 * t3 + 0x6CB7) >> 16) + (t3 != 0);
  s = ((C6 * u) >> 16) - t2;
  v = ((s * 21600 + 0x2800) >> 18) + s + (s != 0);
  y[2] = S16(u);
  y[6] = S16(v);
  u = ((C5 * t6 + C3 * t5 + 0x0E3D) >> 16) + (t5 != 0);
  s = t6 - ((C5 * u) >> 16);
  v = ((s * 26568 + 0x3400) >> 17) + s + (s != 0);
  y[5] = S16(u);
  y[3] = S16(v);
  u = ((C7 * t4 + C1 * t7 + 0x7B1B) >> 16) + (t7 != 0);
  s = ((C7 * u) >> 16) - t4;
  v = ((s * 20539 + 0x3000) >> 20) + s + (s != 0);
  y[1] = S16(u);
  y[7] = S16(v);
}

void orc_enc_fdct8x8(int16_t y[64], const int16_t x[64]) {               /* fdct.c:128-150 */
  int16_t w[64], z[64];
  int i;
  for (i = 0; i < 64; i++) w[i] = S16(x[i] << 2);
  w[0] = S16(w[0] + (w[0] != 0) + 1);
  w[1] = S16(w[1] + 1);
  w[8] = S16(w[8] - 1);
  for (i = 0; i < 8; i++) fdct8(z + i * 8, w + i);           /* columns of w -> rows of z */
  for (i = 0; i < 8; i++) fdct8(w + i * 8, z + i);           /* columns of z -> rows of w */
  for (i = 0; i < 64; i++) y[i] = S16((w[ORC_FZIG_ZAG[i]] + 2) >> 2);   /* zig-zag ordered output */
}

/* reciprocal of one quantiser step, enquant.c:183-191 */
void orc_enc_enquant_table_init(int16_t enquant[128], const uint16_t dequant[64]) {
  int zzi;
  for (zzi = 0; zzi < 64; zzi++) {
    uint32_t d = (uint32_t)dequant[zzi] << 1, t;
    int l = 0;
    while ((d >> (l + 1)) != 0) l++;          /* OC_ILOGNZ_32(d)-1 */
    t = 1 + ((uint32_t)1 << (16 + l)) / d;
    enquant[2 * zzi] = (int16_t)(t - 0x10000);
    enquant[2 * zzi + 1] = (int16_t)l;
  }
}

/* enquant.c:219-248 */
int orc_enc_quantize(int16_t qdct[64], const int16_t dct[64], const uint16_t dequant[64], const int16_t enquant[128]) {
  int nonzero = 0, zzi;
  for (zzi = 0; zzi < 64; zzi++) {
    int val = dct[zzi], d = dequant[zzi], s;
    val = val << 1;
    if (abs(val) >= d) {
      s = -(val < 0);
      val += (d + s) ^ s;                      /* +-d: ties round away from zero */
      val = (((((int32_t)enquant[2 * zzi] * (int32_t)val) >> 16) + val) >> enquant[2 * zzi + 1]) - s;
      qdct[zzi] = S16(val);
      nonzero = zzi;
    } else qdct[zzi] = 0;
  }
  return nonzero;
}

void orc_enc_quantize_batch(int16_t *qdct, int32_t *nonzero, const int16_t *dct, const uint16_t dequant[64], ptrdiff_t n) {
  int16_t enq[128];
  ptrdiff_t i;
  orc_enc_enquant_table_init(enq, dequant);
  for (i = 0; i < n; i++) nonzero[i] = orc_enc_quantize(qdct + i * 64, dct + i * 64, dequant, enq);
}

void orc_enc_fdct8x8_batch(int16_t *y, const int16_t *x, ptrdiff_t n) {
  ptrdiff_t i;
  for (i = 0; i < n; i++) orc_enc_fdct8x8(y + i * 64, x + i * 64);
}

void orc_enc_metric_batch(int op, uint32_t *out, int32_t *dc_out, const uint8_t *src_plane,
                          const uint8_t *ref_plane, int ystride, const int32_t *src_offs,
                          const int32_t *ref_offs, const int32_t *ref2_offs, unsigned thresh,
                          ptrdiff_t n) {
  ptrdiff_t i;
  for (i = 0; i < n; i++) {
    const uint8_t *s = src_plane + src_offs[i];
    const uint8_t *r1 = ref_offs ? ref_plane + ref_offs[i] : NULL;
    const uint8_t *r2 = ref2_offs ? ref_plane + ref2_offs[i] : NULL;
    int dc = 0;
    unsigned v = 0;
    switch (op) {
      case 0: v = orc_enc_frag_sad(s, r1, ystride); break;
      case 1: v = orc_enc_frag_sad_thresh(s, r1, ystride, thresh); break;
      case 2: v = orc_enc_frag_sad2_thresh(s, r1, r2, ystride, thresh); break;
      case 3: v = orc_enc_frag_intra_sad(s, ystride); break;
      case 4: v = orc_enc_frag_satd(&dc, s, r1, ystride); break;
      case 5: v = orc_enc_frag_satd2(&dc, s, r1, r2, ystride); break;
      case 6: v = orc_enc_frag_intra_satd(&dc, s, ystride); break;
      case 7: v = orc_enc_frag_ssd(s, r1, ystride); break;
      default: break;
    }
    out[i] = v;
    if (dc_out) dc_out[i] = dc;
  }
}

/* ------------------------------------------------------------------------- */
/* encoder analysis: per-macro-block cost maps     */
/* ------------------------------------------------------------------------- */
/* mathops.c:294-314 (oc_bexp32_q10, oc_blog32_q10) and mathops.h's OC_ILOGNZ_32 (the position of the top bit, one-based) */
static uint32_t bexp32_q10(int z) {
  unsigned n;
  int ipart = z >> 10;
  n = (unsigned)(z & ((1 << 10) - 1)) << 4;
  n = (n * ((n * ((n * ((n * 3548 >> 15) + 6817) >> 15) + 15823) >> 15) + 22708) >> 15) + 16384;
  return 14 - ipart > 0 ? (n + (1u << (13 - ipart))) >> (14 - ipart) : n << (ipart - 14);
}
static int blog32_q10(uint32_t w) {
  int n, ipart, fpart;
  if (w <= 0) return -1;
  ipart = 32 - __builtin_clz(w);
  n = (int)(ipart - 16 > 0 ? w >> (ipart - 16) : w << (16 - ipart)) - 32768 - 16384;
  fpart = (n * ((n * ((n * ((n * -1402 >> 15) + 2546) >> 15) - 5216) >> 15) + 15745) >> 15) - 6793;
  return (ipart << 10) + (fpart >> 4);
}

/* One plane of the encoder's input frame as the reference keeps it: replicated 16-pixel (luma) borders around it
   (encode.c:1735-1744: oc_img_plane_copy_pad, oc_state_borders_fill_rows, _caps), so that the edge test of oc_mb_activity may
   read one pixel beyond a block on every side.  Built here from the unpadded plane the caller hands over. */
typedef struct {
  uint8_t *buf, *origin;   /* origin = pixel (0, 0) */
  int stride, w, h;
} padded_plane;
static int pad_plane(padded_plane *p, const uint8_t *src, int src_stride, int w, int h) {
  const int B = 16;
  int x, y;
  p->stride = w + 2 * B;
  p->w = w;
  p->h = h;
  p->buf = (uint8_t *)malloc((size_t)p->stride * (h + 2 * B));
  if (!p->buf) return -1;
  p->origin = p->buf + (size_t)B * p->stride + B;
  for (y = -B; y < h + B; y++) {
    const int yy = y < 0 ? 0 : y >= h ? h - 1 : y;
    for (x = -B; x < w + B; x++) {
      const int xx = x < 0 ? 0 : x >= w ? w - 1 : x;
      p->origin[(ptrdiff_t)y * p->stride + x] = src[(ptrdiff_t)yy * src_stride + xx];
    }
  }
  return 0;
}

/* oc_mb_activity's body for one luma block (analyze.c:1167-1234); returns the block's pixel sum through *x_out */
static unsigned block_activity(const uint8_t *blk, int ystride, unsigned *x_out) {
  const uint8_t *s = blk;
  unsigned x = 0, x2 = 0, act;
  int i, j;
  for (i = 0; i < 8; i++, s += ystride)
    for (j = 0; j < 8; j++) {
      const unsigned c = s[j];
      x += c;
      x2 += c * c;
    }
  *x_out = x;
  act = (x2 << 6) - x * x;
  if (act < 8u << 12) return act < 5u << 12 ? act : 5u << 12;   /* the region is flat */
  {
    unsigned e1 = 0, e2 = 0, e3 = 0, e4 = 0, emax;
    s = blk - 1;
    for (i = 0; i < 8; i++, s += ystride)
      for (j = 0; j < 8; j++) {
        const uint8_t *u = s - ystride, *d = s + ystride;
        e1 += (unsigned)abs(((s[j + 2] - s[j]) << 1) + u[j + 2] - u[j] + d[j + 2] - d[j]);
        e2 += (unsigned)abs(((d[j + 1] - u[j + 1]) << 1) + d[j] - u[j] + d[j + 2] - u[j + 2]);
        e3 += (unsigned)abs(((d[j + 2] - u[j]) << 1) + d[j + 1] - s[j] + s[j + 2] - u[j + 1]);
        e4 += (unsigned)abs(((d[j] - u[j + 2]) << 1) + d[j + 1] - s[j + 2] + s[j] - u[j + 1]);
      }
    emax = e1 > e2 ? e1 : e2;
    if (e3 > emax) emax = e3;
    if (e4 > emax) emax = e4;
    /* an edge block: act = act_th * (act / act_th)^0.7, 0x394A = oc_blog32_q10(5 << 12) */
    if (5 * emax > 2 * (e1 + e2 + e3 + e4)) act = bexp32_q10(0x394A + (7 * (blog32_q10(act) - 0x394A + 5) / 10));
  }
  return act;
}

/* oc_mb_intra_satd (analyze.c:1360-1403), oc_mb_activity (:1152-1237) and oc_mb_activity_fast (:1239-1251) for EVERY macro block
   of a frame, in the reference's macro-block numbering: mbi = super block << 2 | quadrant (state.c:300-330), luma blocks of a
   macro block in the order of sb_maps (state.c:123-190), chroma blocks in the order of OC_MB_MAP_IDXS (internal.c:67-76).
   planes: the input frame, unpadded, bitstream row order.  Macro blocks outside the frame: zeros. */
int orc_mb_cost_maps(const uint8_t *const planes[3], const int strides[3], int frame_width, int frame_height, int pixel_fmt,
                     uint32_t *intra_satd /* [nmbs][12] */, uint32_t *luma /* [nmbs] */, uint32_t *activity /* [nmbs][4] */,
                     uint32_t *activity_fast /* [nmbs][4] */) {
  static const int SB_MAP[4][4][2] = {{{0, 0}, {0, 1}, {3, 2}, {3, 3}}, {{0, 3}, {0, 2}, {3, 1}, {3, 0}},
                                      {{1, 0}, {1, 3}, {2, 0}, {2, 3}}, {{1, 1}, {1, 2}, {2, 1}, {2, 2}}};   /* state.c:134-139 */
  static const int MB_MAP[2][2] = {{0, 3}, {1, 2}};                                                          /* internal.c:63 */
  static const int MAP_IDXS[4][12] = {{0, 1, 2, 3, 4, 8}, {0, 1, 2, 3, 4, 5, 8, 9}, {0, 1, 2, 3, 4, 6, 8, 10},
                                      {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11}};                               /* internal.c:67-72 */
  static const int MAP_NIDXS[4] = {6, 8, 8, 12};
  const int hdec = !(pixel_fmt & 1), vdec = !(pixel_fmt & 2);
  const int nh = frame_width >> 3, nv = frame_height >> 3;
  const int nsbw = (nh + 3) >> 2, nsbh = (nv + 3) >> 2;
  const int nmbs = 4 * nsbw * nsbh;
  padded_plane pp[3];
  int pli, sby, sbx, ymb, xmb, rc = 0;
  memset(intra_satd, 0, sizeof(uint32_t) * 12 * (size_t)nmbs);
  memset(luma, 0, sizeof(uint32_t) * (size_t)nmbs);
  memset(activity, 0, sizeof(uint32_t) * 4 * (size_t)nmbs);
  memset(activity_fast, 0, sizeof(uint32_t) * 4 * (size_t)nmbs);
  memset(pp, 0, sizeof(pp));
  for (pli = 0; pli < 3; pli++)
    rc |= pad_plane(pp + pli, planes[pli], strides[pli], pli ? frame_width >> hdec : frame_width, pli ? frame_height >> vdec : frame_height);
  if (!rc)
    for (sby = 0; sby < nsbh; sby++)
      for (sbx = 0; sbx < nsbw; sbx++)
        for (ymb = 0; ymb < 2; ymb++)
          for (xmb = 0; xmb < 2; xmb++) {
            const int quad = MB_MAP[ymb][xmb];
            const unsigned mbi = (unsigned)(sby * nsbw + sbx) << 2 | (unsigned)quad;
            const int mbx = sbx * 4 + xmb * 2, mby = sby * 4 + ymb * 2;   /* luma fragment coordinates of the macro block */
            int i, j, mapii;
            if (mbx >= nh || mby >= nv) continue;                        /* OC_MODE_INVALID, state.c:316-319 */
            /* the four luma blocks in sb_maps order: block (row i, column j) of the super block is entry SB_MAP[i][j][1] of
               quadrant SB_MAP[i][j][0] */
            for (i = 0; i < 4; i++)
              for (j = 0; j < 4; j++)
                if (SB_MAP[i][j][0] == quad) {
                  const int bi = SB_MAP[i][j][1], fx = sbx * 4 + j, fy = sby * 4 + i;
                  const uint8_t *blk = pp[0].origin + (ptrdiff_t)(fy * 8) * pp[0].stride + fx * 8;
                  int dc;
                  unsigned x, satd;
                  satd = orc_enc_frag_intra_satd(&dc, blk, pp[0].stride);
                  intra_satd[mbi * 12 + bi] = satd;
                  luma[mbi] += (uint32_t)dc;
                  activity[mbi * 4 + bi] = block_activity(blk, pp[0].stride, &x);
                  {
                    uint32_t act = (11 * satd >> 8) * satd;                 /* analyze.c:1244 */
                    if (act < 8u << 12 && act > 5u << 12) act = 5u << 12;
                    activity_fast[mbi * 4 + bi] = act;
                  }
                }
            /* the chroma blocks through mb_maps (state.c:200-290) */
            for (mapii = 4; mapii < MAP_NIDXS[pixel_fmt]; mapii++) {
              const int mapi = MAP_IDXS[pixel_fmt][mapii], p = mapi >> 2, bi = mapi & 3;
              const int ci = bi >> 1, cj = bi & 1;                           /* mb_map[p][i << 1 | j] */
              const int fx = (mbx >> hdec) + cj, fy = (mby >> vdec) + ci;
              int dc;
              intra_satd[mbi * 12 + mapii] =
                  orc_enc_frag_intra_satd(&dc, pp[p].origin + (ptrdiff_t)(fy * 8) * pp[p].stride + fx * 8, pp[p].stride);
            }
          }
  for (pli = 0; pli < 3; pli++) free(pp[pli].buf);
  return rc;
}
