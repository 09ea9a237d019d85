/* Compile-only check (tests/test_abi.py::test_integration_shim_compiles): the glue INTEGRATION.md
 * section 2 asks a libtheora maintainer to add in lib/hip/hipstate.c, written against a stand-in
 * for the few oc_theora_state fields it touches, must compile against include/theora_hip.h as it
 * is.  Nothing here is linked or run; the real struct lives in the reference's lib/state.h. */
#include <stddef.h>
#include <stdint.h>

#include "../../include/theora_hip.h"

typedef int16_t ogg_int16_t;
typedef uint16_t ogg_uint16_t;
typedef int16_t oc_mv;

typedef struct {
  unsigned coded : 1, invalid : 1, qii : 4, refi : 2, mb_mode : 3;
  signed borderi : 5, dc : 16;
} oc_fragment; /* shape of state.h:297-322 */

typedef struct oc_theora_state oc_theora_state;
typedef struct {
  void (*frag_copy_list)(unsigned char *, const unsigned char *, int, const ptrdiff_t *, ptrdiff_t, const ptrdiff_t *);
  void (*state_frag_recon)(const oc_theora_state *, ptrdiff_t, int, ogg_int16_t[128], int, ogg_uint16_t);
  void (*loop_filter_init)(signed char[256], int);
  void (*state_loop_filter_frag_rows)(const oc_theora_state *, signed char[256], int, int, int, int);
} oc_base_opt_vtable; /* the four slots the backend replaces, state.h:352-370 */

struct oc_theora_state {
  oc_base_opt_vtable opt_vtable;
  oc_fragment *frags;
  oc_mv *frag_mvs;
  unsigned char loop_filter_limits[64];
  unsigned char qis[3];
  int frame_type;
  struct thip_state *hip; /* the one field the integration adds */
};

static struct thip_state *g_current; /* note (a) of INTEGRATION.md: frag_copy_list has no state argument */

static void oc_state_frag_recon_hip(const oc_theora_state *_state, ptrdiff_t _fragi, int _pli, ogg_int16_t _dct_coeffs[128],
                                    int _last_zzi, ogg_uint16_t _dc_quant) {
  thip_state_frag_recon(_state->hip, _fragi, _pli, _dct_coeffs, _last_zzi, _dc_quant, _state->frags[_fragi].refi,
                        _state->frag_mvs[_fragi]);
}
static void oc_frag_copy_list_hip(unsigned char *_dst_frame, const unsigned char *_src_frame, int _ystride,
                                  const ptrdiff_t *_fragis, ptrdiff_t _nfragis, const ptrdiff_t *_frag_buf_offs) {
  (void)_dst_frame; (void)_src_frame; (void)_ystride; (void)_frag_buf_offs;
  thip_frag_copy_list(g_current, _fragis, _nfragis);
}
static void oc_loop_filter_init_hip(signed char _bv[256], int _flimit) { thip_loop_filter_init(_bv, _flimit); }
static void oc_state_loop_filter_frag_rows_hip(const oc_theora_state *_state, signed char _bv[256], int _refi, int _pli,
                                               int _fragy0, int _fragy_end) {
  (void)_bv;
  thip_state_loop_filter_frag_rows(_state->hip, _state->loop_filter_limits[_state->qis[0]], _refi, _pli, _fragy0, _fragy_end);
}

void oc_state_accel_init_hip(oc_theora_state *_state) {
  _state->opt_vtable.state_frag_recon = oc_state_frag_recon_hip;
  _state->opt_vtable.frag_copy_list = oc_frag_copy_list_hip;
  _state->opt_vtable.loop_filter_init = oc_loop_filter_init_hip;
  _state->opt_vtable.state_loop_filter_frag_rows = oc_state_loop_filter_frag_rows_hip;
}

/* the three frame-scope calls of INTEGRATION.md section 3 */
int oc_hip_frame(oc_theora_state *_state, unsigned char *planes[3], const int32_t strides[3]) {
  g_current = _state->hip;
  if (thip_frame_begin(_state->hip, _state->frame_type) < 0) return -1;
  /* ... the MCU loop calls the slots above ... */
  if (thip_frame_flush(_state->hip) < 0) return -1;
  return thip_state_ycbcr_out(_state->hip, planes, strides);
}

/* th_decode_alloc / th_decode_ycbcr_out glue of INTEGRATION.md section 3: pointers instead of a copy */
int oc_hip_map(oc_theora_state *_state, const unsigned char *planes[3], int32_t strides[3]) {
  if (thip_state_set_eager_output(_state->hip, 1) < 0) return -1;
  return thip_state_ycbcr_map(_state->hip, planes, strides);
}

/* ---- lib/hip/hipenc.c: oc_enc_accel_init_hip (cf. lib/x86/x86enc.c:21-62) ----
   stand-in for the encoder context: the vtable and data of encint.h:292-338 */
typedef int64_t ogg_int64_t;
typedef struct {
  void (*frag_sub)(ogg_int16_t _diff[64], const unsigned char *_src, const unsigned char *_ref, int _ystride);
  void (*frag_sub_128)(ogg_int16_t _diff[64], const unsigned char *_src, int _ystride);
  unsigned (*frag_sad)(const unsigned char *_src, const unsigned char *_ref, int _ystride);
  unsigned (*frag_sad_thresh)(const unsigned char *_src, const unsigned char *_ref, int _ystride, unsigned _thresh);
  unsigned (*frag_sad2_thresh)(const unsigned char *_src, const unsigned char *_ref1, const unsigned char *_ref2, int _ystride,
                               unsigned _thresh);
  unsigned (*frag_intra_sad)(const unsigned char *_src, int _ystride);
  unsigned (*frag_satd)(int *_dc, const unsigned char *_src, const unsigned char *_ref, int _ystride);
  unsigned (*frag_satd2)(int *_dc, const unsigned char *_src, const unsigned char *_ref1, const unsigned char *_ref2, int _ystride);
  unsigned (*frag_intra_satd)(int *_dc, const unsigned char *_src, int _ystride);
  unsigned (*frag_ssd)(const unsigned char *_src, const unsigned char *_ref, int _ystride);
  unsigned (*frag_border_ssd)(const unsigned char *_src, const unsigned char *_ref, int _ystride, ogg_int64_t _mask);
  void (*frag_copy2)(unsigned char *_dst, const unsigned char *_src1, const unsigned char *_src2, int _ystride);
  void (*enquant_table_init)(void *_enquant, const ogg_uint16_t _dequant[64]);
  void (*enquant_table_fixup)(void *_enquant[3][3][2], int _nqis);
  int (*quantize)(ogg_int16_t _qdct[64], const ogg_int16_t _dct[64], const ogg_uint16_t _dequant[64], const void *_enquant);
  void (*frag_recon_intra)(unsigned char *_dst, int _ystride, const ogg_int16_t _residue[64]);
  void (*frag_recon_inter)(unsigned char *_dst, const unsigned char *_src, int _ystride, const ogg_int16_t _residue[64]);
  void (*fdct8x8)(ogg_int16_t _y[64], const ogg_int16_t _x[64]);
} oc_enc_opt_vtable;
typedef struct {
  size_t enquant_table_size;
  int enquant_table_alignment;
} oc_enc_opt_data;
typedef struct {
  oc_enc_opt_vtable opt_vtable;
  oc_enc_opt_data opt_data;
} oc_enc_ctx;

void oc_enc_accel_init_hip(oc_enc_ctx *_enc) {
  /* every slot keeps the reference's signature: the function pointers are assigned without a cast */
  _enc->opt_vtable.frag_sub = thip_enc1_frag_sub;
  _enc->opt_vtable.frag_sub_128 = thip_enc1_frag_sub_128;
  _enc->opt_vtable.frag_sad = thip_enc1_frag_sad;
  _enc->opt_vtable.frag_sad_thresh = thip_enc1_frag_sad_thresh;
  _enc->opt_vtable.frag_sad2_thresh = thip_enc1_frag_sad2_thresh;
  _enc->opt_vtable.frag_intra_sad = thip_enc1_frag_intra_sad;
  _enc->opt_vtable.frag_satd = thip_enc1_frag_satd;
  _enc->opt_vtable.frag_satd2 = thip_enc1_frag_satd2;
  _enc->opt_vtable.frag_intra_satd = thip_enc1_frag_intra_satd;
  _enc->opt_vtable.frag_ssd = thip_enc1_frag_ssd;
  _enc->opt_vtable.frag_border_ssd = thip_enc1_frag_border_ssd;
  _enc->opt_vtable.frag_copy2 = thip_enc1_frag_copy2;
  _enc->opt_vtable.enquant_table_init = thip_enc_enquant_table_init;
  _enc->opt_vtable.enquant_table_fixup = thip_enc_enquant_table_fixup;
  _enc->opt_vtable.quantize = thip_enc1_quantize;
  _enc->opt_vtable.frag_recon_intra = thip_enc1_frag_recon_intra;
  _enc->opt_vtable.frag_recon_inter = thip_enc1_frag_recon_inter;
  _enc->opt_vtable.fdct8x8 = thip_enc1_fdct8x8;
  thip_enc_opt_data(&_enc->opt_data.enquant_table_size, &_enc->opt_data.enquant_table_alignment);
}
