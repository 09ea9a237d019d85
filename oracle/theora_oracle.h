/* theora_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C99) of libtheora's per-fragment reconstruction path
 * and of the encoder block kernels.  It is the checker the HIP path is compared
 * against; it is never linked into, imported by, or called from the product
 * (theora_amd/ + include/).  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may use it.
 *
 * PARITY UNPINNED: the reference (/root/reference, libtheora 1.2.0) cannot be
 * built in this image -- every translation unit includes <ogg/ogg.h> through
 * include/theora/codec.h:66 and libogg is not installed -- and the reference's
 * tests/ hold no golden vectors for this path.  This
 * restatement therefore follows the reference source line by line (citations
 * below are file:line into /root/reference) and is cross-checked against an
 * independent restatement of the normative specification text
 * (oracle/spec_model.py, doc/spec/spec.tex) and against a third-party decoder
 * (FFmpeg's, inside the Chromium of the kaleido package: tests/test_thirdparty_decoder.py),
 * but it has not been run against the reference binary.
 */
#ifndef THEORA_ORACLE_H
#define THEORA_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Frame classification indices, lib/state.h:170-176. */
#define ORC_FRAME_GOLD 0
#define ORC_FRAME_PREV 1
#define ORC_FRAME_SELF 2
#define ORC_FRAME_NONE 3
/* Frame types, lib/state.h:155-158. */
#define ORC_INTRA_FRAME 0
#define ORC_INTER_FRAME 1
/* Pixel formats, include/theora/codec.h (th_pixel_fmt). */
#define ORC_PF_420 0
#define ORC_PF_RSVD 1
#define ORC_PF_422 2
#define ORC_PF_444 3
/* lib/state.h:167 */
#define ORC_UMV_PADDING 16

/* ---- single-block kernels (the ten oc_base_opt_vtable slots, state.h:352-370) ---- */
void orc_idct8x8(int16_t y[64], int16_t x[64], int last_zzi);            /* idct.c:301 */
/* same transform but always through the full ("slow") variant, idct.c:289 */
void orc_idct8x8_full(int16_t y[64], int16_t x[64]);
void orc_frag_copy(uint8_t *dst, const uint8_t *src, int ystride);        /* fragment.c:20 */
void orc_frag_copy_list(uint8_t *dst_frame, const uint8_t *src_frame, int ystride,
                        const ptrdiff_t *fragis, ptrdiff_t nfragis,
                        const ptrdiff_t *frag_buf_offs);                   /* fragment.c:37 */
void orc_frag_recon_intra(uint8_t *dst, int ystride, const int16_t residue[64]);   /* fragment.c:49 */
void orc_frag_recon_inter(uint8_t *dst, const uint8_t *src, int ystride,
                          const int16_t residue[64]);                      /* fragment.c:59 */
void orc_frag_recon_inter2(uint8_t *dst, const uint8_t *src1, const uint8_t *src2,
                           int ystride, const int16_t residue[64]);        /* fragment.c:70 */
void orc_loop_filter_init(int8_t bv[256], int flimit);                     /* state.c:1036 */
/* MV -> one or two byte offsets; state.c:846-957.  qpx/qpy say whether the axis
   is decimated for this plane (quarter-pel vectors). Returns 1 or 2. */
int orc_mv_offsets(int offsets[2], int ystride, int qpx, int qpy, int dx, int dy);

/* ---- stream state (the part of oc_theora_state the path reads, state.h:380-468) ---- */
typedef struct orc_plane_geom {
  int nhfrags, nvfrags;      /* state.h:327-344 */
  ptrdiff_t froffset, nfrags;
  int width, height;         /* plane size in pixels */
  int stride;                /* NEGATIVE: frames are stored flipped, state.c:622-629 */
} orc_plane_geom;

typedef struct orc_state {
  int frame_width, frame_height, pixel_fmt;
  int hdec, vdec;
  orc_plane_geom fplanes[3];
  ptrdiff_t nfrags;
  /* per-fragment data (oc_fragment, state.h:297-322, and frag_mvs) */
  uint8_t *coded;            /* [nfrags] */
  uint8_t *refi;             /* [nfrags] ORC_FRAME_* */
  int16_t *mvs;              /* [nfrags] packed as OC_MV: x in the low byte, y in the high, state.h:232-240 */
  int16_t *dc;               /* [nfrags] */
  ptrdiff_t *frag_buf_offs;  /* [nfrags], state.c:631-656 */
  /* reference frame ring, state.c:545-629: three padded frames in one slab */
  uint8_t *ref_slab;
  size_t ref_frame_sz;
  uint8_t *ref_plane_data[3][3];  /* [buffer][plane] -> pixel (0,0) in BITSTREAM coordinates (bottom-left) */
  int ref_frame_idx[3];           /* ORC_FRAME_GOLD/PREV/SELF -> buffer index, or -1 */
  uint8_t *ref_frame_data[3];     /* = ref_plane_data[idx][0]; frag_buf_offs are relative to it */
  ptrdiff_t plane_off[3];         /* ref_plane_data[b][p]-ref_plane_data[b][0], identical for every b */
} orc_state;

orc_state *orc_state_new(int frame_width, int frame_height, int pixel_fmt);
void orc_state_free(orc_state *st);
/* Copy the picture area of plane pli of buffer slot (ORC_FRAME_*) out to / in from a
   tightly packed buffer in bitstream row order (row 0 = bottom row of the picture).
   _set also refills the UMV borders of that plane (state.c:770-835). */
void orc_state_get_plane(const orc_state *st, int slot, int pli, uint8_t *out);
void orc_state_set_plane(orc_state *st, int slot, int pli, const uint8_t *in);
/* Force the ring to a given assignment (tests). */
void orc_state_set_ref_idx(orc_state *st, int gold, int prev, int self);

void orc_state_frag_recon(orc_state *st, ptrdiff_t fragi, int pli, int16_t dct_coeffs[128],
                          int last_zzi, uint16_t dc_quant);                /* state.c:959 */
void orc_state_loop_filter_frag_rows(orc_state *st, int8_t bv[256], int slot, int pli,
                                     int fragy0, int fragy_end);           /* state.c:1055 */
void orc_state_borders_fill_rows(orc_state *st, int bufi, int pli, int y0, int yend); /* state.c:770 */
void orc_state_borders_fill_caps(orc_state *st, int bufi, int pli);                    /* state.c:799 */

/* Coded-order traversal: writes the fragment indices of plane pli in super-block
   raster / Hilbert-inside order (state.c:123-190, the order of coded_fragis for a
   fully coded frame).  Returns the count (= fplanes[pli].nfrags). */
ptrdiff_t orc_sb_order(const orc_state *st, int pli, ptrdiff_t *out);

/* Serial DC un-prediction of rows [fragy0,fragy_end) of one plane, decode.c:1392-1500.
   pred_last[3] carries across calls within a frame (reset to 0 per frame, decode.c:1367).
   Uses st->coded, st->refi, st->dc.  Returns the number of coded fragments visited. */
ptrdiff_t orc_dc_unpredict_rows(orc_state *st, int pli, int fragy0, int fragy_end,
                                int pred_last[3]);

/* One frame through the reference's MCU-pipelined loop, decode.c:2790-2962, with the
   token expansion already done:
     coded_fragis[ntotal]  coded order, planes concatenated, ncoded[3] per plane
     coeffs[ntotal][64]    natural-order dequantised AC, raw DC in [0]   (decode.c:1573-1581)
     last_zzi[ntotal], dc_quant[ntotal]
     uncoded_fragis[nuncoded] any order (every fragment is in exactly one list)
   st->refi / st->mvs must be filled for the coded fragments.  flimit==0 disables the
   loop filter (decode.c:1369-1371).  Rotates the reference ring afterwards. */
int orc_decode_frame(orc_state *st, int frame_type, const ptrdiff_t *coded_fragis,
                     const ptrdiff_t ncoded[3], const int16_t *coeffs, const uint8_t *last_zzi,
                     const uint16_t *dc_quant, const ptrdiff_t *uncoded_fragis,
                     ptrdiff_t nuncoded, int flimit);

/* ---- encoder block kernels (oc_enc_opt_vtable, encint.h:292-326) ---- */
/* out-of-loop post-processing (decode.c:1608-1957), test infrastructure like the rest */
void orc_pp_deblock_frag_rows(uint8_t *dst_data, int dst_ystride, const uint8_t *src_data, int src_ystride, int width, int height,
                              int nhfrags, int nvfrags, int *variances, const uint8_t *dc_qis, const int pp_dc_scale[64],
                              int fragy0, int fragy_end);
void orc_pp_dering_frag_rows(uint8_t *img_data, int ystride, int width, int height, int nhfrags, const int *variances,
                             const uint8_t *frag_qi, const int pp_dc_scale[64], const int pp_sharp_mod[64], int strong, int pli,
                             int fragy0, int fragy_end);
void orc_postprocess_frame(const orc_state *st, int slot, int level, int loop_filter, const uint8_t *dc_qis,
                           const uint8_t *frag_qi, const int pp_dc_scale[64], const int pp_sharp_mod[64], uint8_t *out,
                           int *variances);
void orc_enc_frag_sub(int16_t diff[64], const uint8_t *src, const uint8_t *ref, int ystride);   /* encfrag.c:21 */
void orc_enc_frag_sub_128(int16_t diff[64], const uint8_t *src, int ystride);                   /* encfrag.c:32 */
unsigned orc_enc_frag_sad(const uint8_t *src, const uint8_t *ref, int ystride);                 /* encfrag.c:42 */
unsigned orc_enc_frag_sad_thresh(const uint8_t *src, const uint8_t *ref, int ystride, unsigned thresh); /* :56 */
unsigned orc_enc_frag_sad2_thresh(const uint8_t *src, const uint8_t *ref1, const uint8_t *ref2,
                                  int ystride, unsigned thresh);                                /* :71 */
unsigned orc_enc_frag_intra_sad(const uint8_t *src, int ystride);                               /* :88 */
unsigned orc_enc_frag_satd(int *dc, const uint8_t *src, const uint8_t *ref, int ystride);       /* :317 */
unsigned orc_enc_frag_satd2(int *dc, const uint8_t *src, const uint8_t *ref1, const uint8_t *ref2,
                            int ystride);                                                       /* :324 */
unsigned orc_enc_frag_intra_satd(int *dc, const uint8_t *src, int ystride);                     /* :331 */
unsigned orc_enc_frag_ssd(const uint8_t *src, const uint8_t *ref, int ystride);                 /* :338 */
unsigned orc_enc_frag_border_ssd(const uint8_t *src, const uint8_t *ref, int ystride, int64_t mask); /* :352 */
void orc_enc_frag_copy2(uint8_t *dst, const uint8_t *src1, const uint8_t *src2, int ystride);   /* :368 */
void orc_enc_fdct8x8(int16_t y[64], const int16_t x[64]);                                       /* fdct.c:128 */
/* oc_iquant_init + oc_enc_quantize_c (enquant.c:183-248): enquant is 64 {m,l} int16 pairs */
void orc_enc_enquant_table_init(int16_t enquant[128], const uint16_t dequant[64]);
int orc_enc_quantize(int16_t qdct[64], const int16_t dct[64], const uint16_t dequant[64], const int16_t enquant[128]);
void orc_enc_quantize_batch(int16_t *qdct, int32_t *nonzero, const int16_t *dct, const uint16_t dequant[64], ptrdiff_t n);

/* ---- batch drivers used by the tests and by bench.py's cpu_baseline leg ---- */
void orc_idct8x8_batch(int16_t *y, const int16_t *x, const int32_t *last_zzi, ptrdiff_t n);
void orc_enc_fdct8x8_batch(int16_t *y, const int16_t *x, ptrdiff_t n);
/* op: 0 sad, 1 sad_thresh, 2 sad2_thresh, 3 intra_sad, 4 satd, 5 satd2, 6 intra_satd, 7 ssd */
void orc_enc_metric_batch(int op, uint32_t *out, int32_t *dc_out, const uint8_t *src_plane,
                          const uint8_t *ref_plane, int ystride, const int32_t *src_offs,
                          const int32_t *ref_offs, const int32_t *ref2_offs, unsigned thresh,
                          ptrdiff_t n);

/* oc_mb_intra_satd / oc_mb_activity / oc_mb_activity_fast (analyze.c:1360-1403, 1152-1251) for every macro block of a frame,
   in the reference's macro-block numbering; planes unpadded, bitstream row order.  Returns 0, or -1 when out of memory. */
int orc_mb_cost_maps(const uint8_t *const planes[3], const int strides[3], int frame_width, int frame_height, int pixel_fmt,
                     uint32_t *intra_satd, uint32_t *luma, uint32_t *activity, uint32_t *activity_fast);

extern const uint8_t ORC_FZIG_ZAG[128];   /* internal.c:27 */

#ifdef __cplusplus
}
#endif
#endif
