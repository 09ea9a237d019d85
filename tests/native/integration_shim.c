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
