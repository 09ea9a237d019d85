"""ctypes binding of libtheora_hip.so (include/theora_hip.h).

There is no fallback: if the HIP library has not been built, or a call fails, this
raises.  Nothing here (or anywhere in theora_amd/) touches the CPU checker used by the tests.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# THIP_LIB: developer override to A/B a differently compiled copy of the same library (tools/)
SO_PATH = os.environ.get("THIP_LIB") or os.path.join(_HERE, "libtheora_hip.so")

OK, EFAULT, EINVAL, EIMPL, DUPFRAME = 0, -1, -10, -23, 1
FRAME_GOLD, FRAME_PREV, FRAME_SELF = 0, 1, 2
INTRA_FRAME, INTER_FRAME = 0, 1
MAX_BATCH = 8
TILE_FRAGS = 64
SLOT_GROUP, SLOT_GROUP_BYTES = 64, 8192
INFO_CODED, INFO_DC_ONLY, INFO_QII_SHIFT = 0x1, 0x8, 4
COEFFS_DEQUANT16, COEFFS_LEVELS = 0, 1
UNIT_BYTES, UNIT_GROUP_BYTES, SLOT_WIDE = 64, 4096, 0x80000000
KERNEL_RECON, KERNEL_LOOPFILTER, NKERNELS = 0, 1, 2

ENC_OPS = dict(sad=0, sad_thresh=1, sad2_thresh=2, intra_sad=3, satd=4, satd2=5, intra_satd=6, ssd=7)


class PlaneGeom(C.Structure):
    _fields_ = [("nhfrags", C.c_int32), ("nvfrags", C.c_int32), ("froffset", C.c_int32),
                ("nfrags", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
                ("stride", C.c_int32), ("plane_off", C.c_int32)]


class FrameDesc(C.Structure):
    _fields_ = [("frag_info", C.c_void_p), ("coeffs", C.c_void_p), ("tile_slot0", C.c_void_p),
                ("nslots", C.c_int32), ("ncoded", C.c_int32), ("frame_type", C.c_int32),
                ("flimit", C.c_int32), ("dc_tokens", C.c_void_p), ("coeff_format", C.c_int32), ("dequant", C.c_void_p)]


class TileGeom(C.Structure):
    _fields_ = [("tiles_x", C.c_int32 * 3), ("tiles_y", C.c_int32 * 3), ("tile_off", C.c_int32 * 3),
                ("ntiles", C.c_int32)]


class TheoraHipError(RuntimeError):
    pass


# every symbol include/theora_hip.h declares: (name, restype, argtypes)
_P, _I, _I64, _U32 = C.c_void_p, C.c_int, C.c_int64, C.c_uint32
SYMBOLS = [
    ("thip_state_create", _I, [C.POINTER(_P), _I, _I, _I]),
    ("thip_state_create_on", _I, [C.POINTER(_P), _I, _I, _I, _I]),
    ("thip_state_device", _I, [_P]),
    ("thip_device_count", _I, []),
    ("thip_state_free", None, [_P]),
    ("thip_state_get_geom", _I, [_P, C.POINTER(PlaneGeom), C.POINTER(_I64), C.POINTER(_I64)]),
    ("thip_state_get_tiles", _I, [_P, C.POINTER(TileGeom)]),
    ("thip_state_frag_pos", _I64, [_P, _I64]),
    ("thip_state_ref_idx", _I, [_P, _I]),
    ("thip_state_set_ref_idx", _I, [_P, _I, _I, _I]),
    ("thip_state_frame_ptr", _P, [_P, _I]),
    ("thip_state_read_plane", _I, [_P, _I, _I, _P]),
    ("thip_state_write_plane", _I, [_P, _I, _I, _P]),
    ("thip_state_ycbcr_out", _I, [_P, C.POINTER(_P), C.POINTER(C.c_int32)]),
    ("thip_state_ycbcr_map", _I, [_P, C.POINTER(_P), C.POINTER(C.c_int32)]),
    ("thip_state_ycbcr_map_begin", _I, [_P]),
    ("thip_state_ycbcr_map_end", _I, [_P, C.POINTER(_P), C.POINTER(C.c_int32)]),
    ("thip_state_check_fault", _I, [_P]),
    ("thip_state_ring_mark", _I, [_P, C.POINTER(_I64)]),
    ("thip_state_ring_rewind", _I, [_P, C.POINTER(_I64)]),
    ("thip_state_set_eager_output", _I, [_P, _I]),
    ("thip_state_postprocess", _I, [_P, _I, _P, _P, _P, _P]),
    ("thip_state_decode_token_lists", _I, [_P, _P]),
    ("thip_state_token_lists_begin", _I, [_P, _P]),
    ("thip_state_token_lists_begin_assigned", _I, [_P, _P, _P, _P]),
    ("thip_state_token_lists_finish", _I, [_P, _P]),
    ("thip_state_token_lists_open", _I, [_P, _P]),
    ("thip_state_token_lists_append", _I, [_P, _I, _I, _P, C.c_int64, _P, _P, _P, _P]),
    ("thip_state_token_lists_abort", _I, [_P]),
    ("thip_state_token_lists_staging", _I, [_P, _P]),
    ("thip_state_read_pp_plane", _I, [_P, _I, _P]),
    ("thip_decode_frames", _I, [C.POINTER(_P), C.POINTER(FrameDesc), _I, _P, C.POINTER(C.c_int32)]),
    ("thip_synchronize", _I, []),
    ("thip_frame_begin", _I, [_P, _I]),
    ("thip_state_frag_recon", _I, [_P, C.c_ssize_t, _I, _P, _I, C.c_uint16, _I, C.c_int16]),
    ("thip_state_frag_recon_levels", _I, [_P, C.c_ssize_t, _I, _P, _I, C.c_uint16, _I, _I, C.c_int16]),
    ("thip_frag_copy_list", _I, [_P, _P, C.c_ssize_t]),
    ("thip_loop_filter_init", None, [_P, _I]),
    ("thip_state_loop_filter_frag_rows", _I, [_P, _I, _I, _I, _I, _I]),
    ("thip_frame_flush", _I, [_P]),
    ("thip_idct8x8_batch", _I, [_P, _P, _P, _I64]),
    ("thip_frag_recon_batch", _I, [_P, _P, _I, _I, _P, _P, _P, _P, _I64]),
    ("thip_frag_copy_list_batch", _I, [_P, _P, _I, _P, _I64, _P]),
    ("thip_loop_filter_plane", _I, [_P, _I, _I, _I, _P, _I, _I, _I]),
    ("thip_dc_unpredict_plane", _I, [_P, _P, _I, _I]),
    ("thip_state_set_device_dc", _I, [_P, _I]),
    ("thip_frame_dequant_table", _I, [_P, _I, _P]),
    ("thip_pack_dequant_table", None, [_P, _P]),
    ("thip_state_frag_recon_tokens", _I, [_P, C.c_ssize_t, _I, _P, _I, C.c_int16, _I, C.c_uint16, _I, _I, C.c_int16]),
    ("thip_enc_frag_metric_batch", _I, [_I, _P, _P, _P, _P, _I, _P, _P, _P, _U32, _I64]),
    ("thip_enc_frag_metric_sites_batch", _I, [_I, _P, _P, _P, _P, _I, _P, _P, _P, _P, _I, _I64]),
    ("thip_enc_frag_metric_halfpel_batch", _I, [_I, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _I, _I64]),
    ("thip_enc_mb_count", _I, [_I, _I]),
    ("thip_enc_mb_cost_maps", _I, [C.POINTER(_P), C.POINTER(C.c_int32), _I, _I, _I, _P, _P, _P, _P]),
    ("thip_enc_frag_border_ssd_batch", _I, [_P, _P, _P, _I, _P, _P, _P, _I64]),
    ("thip_enc_frag_sub_batch", _I, [_P, _P, _P, _I, _P, _P, _I64]),
    ("thip_enc_frag_copy2_batch", _I, [_P, _P, _I, _P, _P, _P, _I64]),
    ("thip_set_batch_stream", _I, [_P, _I]),
    ("thip_enc_fdct8x8_batch", _I, [_P, _P, _I64]),
    ("thip_enc_enquant_table_init", None, [_P, _P]),
    ("thip_enc_enquant_table_fixup", None, [_P, _I]),
    ("thip_enc_opt_data", None, [C.POINTER(C.c_size_t), C.POINTER(_I)]),
    ("thip_enc_quantize_tab_batch", _I, [_P, _P, _P, _P, _P, _I64]),
    ("thip_enc1_frag_sub", None, [_P, _P, _P, _I]),
    ("thip_enc1_frag_sub_128", None, [_P, _P, _I]),
    ("thip_enc1_frag_sad", _U32, [_P, _P, _I]),
    ("thip_enc1_frag_sad_thresh", _U32, [_P, _P, _I, _U32]),
    ("thip_enc1_frag_sad2_thresh", _U32, [_P, _P, _P, _I, _U32]),
    ("thip_enc1_frag_intra_sad", _U32, [_P, _I]),
    ("thip_enc1_frag_satd", _U32, [C.POINTER(_I), _P, _P, _I]),
    ("thip_enc1_frag_satd2", _U32, [C.POINTER(_I), _P, _P, _P, _I]),
    ("thip_enc1_frag_intra_satd", _U32, [C.POINTER(_I), _P, _I]),
    ("thip_enc1_frag_ssd", _U32, [_P, _P, _I]),
    ("thip_enc1_frag_border_ssd", _U32, [_P, _P, _I, _I64]),
    ("thip_enc1_frag_copy2", None, [_P, _P, _P, _I]),
    ("thip_enc1_quantize", _I, [_P, _P, _P, _P]),
    ("thip_enc1_frag_recon_intra", None, [_P, _I, _P]),
    ("thip_enc1_frag_recon_inter", None, [_P, _P, _I, _P]),
    ("thip_enc1_fdct8x8", None, [_P, _P]),
    ("thip_enc_quantize_batch", _I, [_P, _P, _P, _P, _I64]),
    ("thip_enc_fdct_quantize_batch", _I, [_P, _P, _P, _P, _P, _P, _I64]),
    ("thip_profile_enable", _I, [_I]),
    ("thip_profile_read", _I, [C.POINTER(_I64), C.POINTER(C.c_double)]),
    ("thip_profile_reset", _I, []),
    ("thip_version_string", C.c_char_p, []),
    ("thip_set_option", _I, [C.c_char_p, _I]),
    ("thip_get_option", _I, [C.c_char_p, C.POINTER(_I)]),
    ("thip_option_name", C.c_char_p, [_I, C.POINTER(C.c_char_p)]),
    ("thip_option", _I, [C.c_char_p]),
    ("thip_option_add", None, [C.c_char_p, _I]),
]

# ---- th_decode_* API (include/theoradec_hip.h) ----------------------------------------------
class ThInfo(C.Structure):
    _fields_ = [("version_major", C.c_ubyte), ("version_minor", C.c_ubyte), ("version_subminor", C.c_ubyte),
                ("frame_width", C.c_uint32), ("frame_height", C.c_uint32), ("pic_width", C.c_uint32),
                ("pic_height", C.c_uint32), ("pic_x", C.c_uint32), ("pic_y", C.c_uint32),
                ("fps_numerator", C.c_uint32), ("fps_denominator", C.c_uint32),
                ("aspect_numerator", C.c_uint32), ("aspect_denominator", C.c_uint32),
                ("colorspace", C.c_int), ("pixel_fmt", C.c_int), ("target_bitrate", C.c_int),
                ("quality", C.c_int), ("keyframe_granule_shift", C.c_int)]


class ThComment(C.Structure):
    _fields_ = [("user_comments", C.POINTER(C.c_char_p)), ("comment_lengths", C.POINTER(C.c_int)),
                ("comments", C.c_int), ("vendor", C.c_char_p)]


class OggPacket(C.Structure):
    _fields_ = [("packet", C.c_void_p), ("bytes", C.c_long), ("b_o_s", C.c_long), ("e_o_s", C.c_long),
                ("granulepos", C.c_int64), ("packetno", C.c_int64)]


class ThImgPlane(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("stride", C.c_int), ("data", C.POINTER(C.c_ubyte))]


DEC_SYMBOLS = [
    ("th_info_init", None, [C.POINTER(ThInfo)]),
    ("th_info_clear", None, [C.POINTER(ThInfo)]),
    ("th_comment_init", None, [C.POINTER(ThComment)]),
    ("th_comment_clear", None, [C.POINTER(ThComment)]),
    ("th_decode_headerin", _I, [C.POINTER(ThInfo), C.POINTER(ThComment), C.POINTER(_P), C.POINTER(OggPacket)]),
    ("th_decode_alloc", _P, [C.POINTER(ThInfo), _P]),
    ("th_decode_alloc_on", _P, [C.POINTER(ThInfo), _P, _I]),
    ("th_setup_free", None, [_P]),
    ("th_decode_ctl", _I, [_P, _I, _P, C.c_size_t]),
    ("th_decode_packetin", _I, [_P, C.POINTER(OggPacket), C.POINTER(_I64)]),
    ("th_decode_ycbcr_out", _I, [_P, C.POINTER(ThImgPlane)]),
    ("th_decode_free", None, [_P]),
    ("th_granule_frame", _I64, [_P, _I64]),
    ("th_granule_time", C.c_double, [_P, _I64]),
    ("th_version_string", C.c_char_p, []),
    ("th_version_number", C.c_uint32, []),
    ("th_packet_isheader", _I, [C.POINTER(OggPacket)]),
    ("th_packet_iskeyframe", _I, [C.POINTER(OggPacket)]),
    ("th_comment_add", None, [C.POINTER(ThComment), C.c_char_p]),
    ("th_comment_add_tag", None, [C.POINTER(ThComment), C.c_char_p, C.c_char_p]),
    ("th_comment_query", C.c_void_p, [C.POINTER(ThComment), C.c_char_p, _I]),
    ("th_comment_query_count", _I, [C.POINTER(ThComment), C.c_char_p]),
]

# include/thip_ogg.h
OGG_SYMBOLS = [
    ("thip_ogg_open_memory", _P, [_P, C.c_size_t]),
    ("thip_ogg_open_file", _P, [C.c_char_p]),
    ("thip_ogg_next_packet", _I, [_P, C.POINTER(OggPacket), C.POINTER(C.c_uint32)]),
    ("thip_ogg_stats", None, [_P, C.POINTER(_I64), C.POINTER(_I64)]),
    ("thip_ogg_close", None, [_P]),
]

_lib = None


def load():
    """dlopen the HIP library; raises if it is missing (no CPU fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP/HSA runtime per process: torch ships its own libamdhip64.so (SONAME
    # libamdhip64.so.7) and looks it up by the unversioned name, so it must be loaded
    # BEFORE this library; our DT_NEEDED libamdhip64.so.7 then binds to the copy already in
    # the process.  The other order loads two runtimes and the second sees no device.
    import torch  # noqa: F401
    if not os.path.exists(SO_PATH):
        raise TheoraHipError(
            "%s not found: build it with `python -m theora_amd.build` (hipcc, gfx950). "
            "theora_amd has no CPU fallback." % SO_PATH)
    L = C.CDLL(SO_PATH)
    for name, restype, argtypes in SYMBOLS + DEC_SYMBOLS + OGG_SYMBOLS:
        fn = getattr(L, name)   # AttributeError if the library does not export it
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = L
    return L


def check(rc, what):
    if rc < 0:
        raise TheoraHipError("%s failed with TH error %d" % (what, rc))
    return rc
