"""Boundary evidence against the REAL reference headers (CPU only; skipped where /root/reference does not exist, i.e. on the
GPU box): every slot function this backend offers is assigned into the real `oc_base_opt_vtable` / `oc_dec_opt_vtable` /
`oc_enc_opt_vtable` of lib/state.h, lib/decint.h and lib/encint.h with -Werror=incompatible-pointer-types, so a drift between
include/theora_hip.h and the reference's slot signatures fails here -- which the hand-copied stand-in structs of
tests/native/integration_shim.c cannot notice.  The reference's headers reach <ogg/ogg.h> (codec.h:66), which this image does not
have; a TYPES-ONLY stand-in is written into a temporary directory for this syntax check (never committed, nothing is linked or
run: this is evidence about the boundary, not a reference build and not parity)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

OGG_TYPES = """
#ifndef OGG_TYPES_ONLY_H
#define OGG_TYPES_ONLY_H
#include <stdint.h>
#include <stdlib.h>
typedef int16_t ogg_int16_t; typedef uint16_t ogg_uint16_t; typedef int32_t ogg_int32_t; typedef uint32_t ogg_uint32_t;
typedef int64_t ogg_int64_t; typedef uint64_t ogg_uint64_t;
#define _ogg_malloc malloc
#define _ogg_calloc calloc
#define _ogg_realloc realloc
#define _ogg_free free
typedef struct { long endbyte; int endbit; unsigned char *buffer; unsigned char *ptr; long storage; } oggpack_buffer;
typedef struct { unsigned char *packet; long bytes; long b_o_s; long e_o_s; ogg_int64_t granulepos; ogg_int64_t packetno; } ogg_packet;
#endif
"""

SHIM = r"""
#include "theora_hip.h"
#define OC_STATE_USE_VTABLE 1
#define OC_DEC_USE_VTABLE 1
#define OC_ENC_USE_VTABLE 1
#include "encint.h"   /* pulls in state.h */
#include "decint.h"

/* ---- lib/hip/hipstate.c as INTEGRATION.md section 2 has it, against the real oc_theora_state ---- */
static struct thip_state *g_current;
static struct thip_state *hip_of(const oc_theora_state *_state) { (void)_state; return g_current; }

static void oc_state_frag_recon_hip(const oc_theora_state *_state, ptrdiff_t _fragi, int _pli, ogg_int16_t _dct_coeffs[128],
                                    int _last_zzi, ogg_uint16_t _dc_quant) {
  thip_state_frag_recon(hip_of(_state), _fragi, _pli, _dct_coeffs, _last_zzi, _dc_quant, _state->frags[_fragi].refi,
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
  thip_state_loop_filter_frag_rows(hip_of(_state), _state->loop_filter_limits[_state->qis[0]], _refi, _pli, _fragy0, _fragy_end);
}
void oc_state_accel_init_hip(oc_theora_state *_state) {
  oc_state_accel_init_c(_state);                               /* the slots a device backend leaves alone */
  _state->opt_vtable.state_frag_recon = oc_state_frag_recon_hip;
  _state->opt_vtable.frag_copy_list = oc_frag_copy_list_hip;
  _state->opt_vtable.loop_filter_init = oc_loop_filter_init_hip;
  _state->opt_vtable.state_loop_filter_frag_rows = oc_state_loop_filter_frag_rows_hip;
}

/* ---- lib/hip/hipenc.c: all 18 slots of oc_enc_opt_vtable, assigned without a cast ---- */
void oc_enc_accel_init_hip(oc_enc_ctx *_enc) {
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

/* ---- lib/hip/hipdec.c: the decoder's one slot (decint.h:72-75); the device form works on whole planes ---- */
static void oc_dec_dc_unpredict_mcu_plane_hip(oc_dec_ctx *_dec, oc_dec_pipeline_state *_pipe, int _pli) {
  (void)_dec; (void)_pipe; (void)_pli;   /* thip_state_set_device_dc(hip, 1): the flush un-predicts (INTEGRATION.md section 4) */
}
void oc_dec_accel_init_hip(oc_dec_ctx *_dec) { _dec->opt_vtable.dc_unpredict_mcu_plane = oc_dec_dc_unpredict_mcu_plane_hip; }

/* constants the two sides must agree on */
_Static_assert(THIP_FRAME_GOLD == OC_FRAME_GOLD && THIP_FRAME_PREV == OC_FRAME_PREV && THIP_FRAME_SELF == OC_FRAME_SELF, "state.h:170-176");
_Static_assert(THIP_INTRA_FRAME == OC_INTRA_FRAME && THIP_INTER_FRAME == OC_INTER_FRAME, "state.h:155-158");
_Static_assert(THIP_EFAULT == TH_EFAULT && THIP_EINVAL == TH_EINVAL && THIP_EIMPL == TH_EIMPL && THIP_DUPFRAME == TH_DUPFRAME, "codec.h:77-93");
"""


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "lib", "encint.h")), reason="the reference's headers are not on this box")
def test_slots_fit_the_reference_vtables(tmp_path):
    os.makedirs(tmp_path / "ogg")
    (tmp_path / "ogg" / "ogg.h").write_text(OGG_TYPES)
    (tmp_path / "shim.c").write_text(SHIM)
    cmd = ["gcc", "-std=gnu11", "-fsyntax-only", "-Wall", "-Werror=incompatible-pointer-types", "-Werror=implicit-function-declaration",
           "-Werror=int-conversion", "-I" + str(tmp_path), "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(REF, "include"),
           "-I" + os.path.join(REF, "lib"), str(tmp_path / "shim.c")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "lib", "encint.h")), reason="the reference's headers are not on this box")
def test_a_wrong_signature_is_noticed(tmp_path):
    """Negative control: the same file with one slot given a wrong prototype must NOT compile."""
    os.makedirs(tmp_path / "ogg")
    (tmp_path / "ogg" / "ogg.h").write_text(OGG_TYPES)
    bad = SHIM.replace("_enc->opt_vtable.frag_satd = thip_enc1_frag_satd;", "_enc->opt_vtable.frag_satd = thip_enc1_frag_sad;")
    (tmp_path / "shim.c").write_text(bad)
    cmd = ["gcc", "-std=gnu11", "-fsyntax-only", "-Werror=incompatible-pointer-types", "-I" + str(tmp_path), "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(REF, "include"), "-I" + os.path.join(REF, "lib"), str(tmp_path / "shim.c")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode != 0 and "incompatible" in r.stderr
