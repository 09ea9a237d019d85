/* theora_hip.h -- C ABI of the MI355X (gfx950) backend for libtheora's per-fragment
 * reconstruction path.
 *
 * This is the boundary a maintainer binds from lib/hip/hipstate.c:
 * the ten oc_base_opt_vtable slots (lib/state.h:352-370) and the encoder's block
 * kernels (oc_enc_opt_vtable, lib/encint.h:292-326), plus the frame-scope entry points
 * a device backend needs where the reference has its MCU loop (lib/decode.c:2858-2962).
 * Plain C types only; every pointer marked "device" is HBM memory of the current HIP
 * device, everything else is host memory.  All functions return 0 or a negative TH_E*
 * code (include/theora/codec.h:77-93); none of them ever falls back to a CPU path.
 *
 * Coordinates: everything is in BITSTREAM coordinates (row 0 = bottom row of the
 * picture, as inside the reference: lib/state.c:622-629).  Device planes are stored
 * unpadded with a positive pitch; motion-compensated reads clamp their coordinates,
 * which is bit-identical to the reference's replicated UMV border
 * (lib/state.c:770-835).
 */
#ifndef THEORA_HIP_H
#define THEORA_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* error codes, include/theora/codec.h:77-93 */
#define THIP_OK 0
#define THIP_EFAULT (-1)
#define THIP_EINVAL (-10)
#define THIP_EIMPL (-23)
#define THIP_DUPFRAME 1 /* TH_DUPFRAME, codec.h:91 */

/* lib/state.h:170-176 */
#define THIP_FRAME_GOLD 0
#define THIP_FRAME_PREV 1
#define THIP_FRAME_SELF 2
/* lib/state.h:155-158 */
#define THIP_INTRA_FRAME 0
#define THIP_INTER_FRAME 1

#define THIP_MAX_BATCH 8 /* streams per kernel launch; larger batches are chunked */

/* ------------------------------------------------------------------------------------
 * Stream state: the device side of one oc_theora_state (lib/state.h:380-468):
 * three resident reference frames, the SELF/PREV/GOLD ring, the per-frame coded map.
 * ---------------------------------------------------------------------------------- */
typedef struct thip_state thip_state;

typedef struct thip_plane_geom {
  int32_t nhfrags, nvfrags; /* oc_fragment_plane, state.h:327-344 */
  int32_t froffset, nfrags;
  int32_t width, height; /* pixels */
  int32_t stride;        /* device pitch in bytes (positive) */
  int32_t plane_off;     /* byte offset of the plane inside one device frame */
} thip_plane_geom;

/* Replaces oc_state_init + oc_state_ref_bufs_init (state.c:698, :545) for the device
   side.  frame_width/height are the coded size (multiples of 16), pixel_fmt is
   th_pixel_fmt (codec.h: 0=4:2:0, 2=4:2:2, 3=4:4:4). */
int thip_state_create(thip_state **out, int frame_width, int frame_height, int pixel_fmt);
/* The same on HIP device `device` (0 .. hipGetDeviceCount()-1; -1 = the calling thread's current
   device, which is what thip_state_create uses).  A state lives on its device for life: its frames,
   staging buffers and library-owned streams are created there and every entry point that takes the
   state switches to that device for the duration of the call and back, so the states of one process
   may be spread over all GPUs of a node (a single stream stays on one GPU; streams are sharded whole).
   thip_decode_frames accepts states of different devices in one call when `stream` is NULL.  */
int thip_state_create_on(thip_state **out, int device, int frame_width, int frame_height, int pixel_fmt);
int thip_state_device(const thip_state *st);
int thip_device_count(void); /* hipGetDeviceCount, 0 when there is none */
void thip_state_free(thip_state *st);
int thip_state_get_geom(const thip_state *st, thip_plane_geom geom[3], int64_t *nfrags,
                        int64_t *frame_bytes);
/* Buffer index (0..2) currently holding THIP_FRAME_*; -1 before the first frame. */
int thip_state_ref_idx(const thip_state *st, int which);
/* Force the ring (tests / seeking; mirrors the bookkeeping of decode.c:2947-2962). */
int thip_state_set_ref_idx(thip_state *st, int gold, int prev, int self);
/* device pointer of buffer bufi (0..2); planes at +geom[pli].plane_off.  For reading (encoders,
   on-device consumers of decoded frames).  The library keeps track of which decoded frame every
   buffer holds so that it can leave unchanged blocks where they are; a caller that WRITES a buffer
   through this pointer must say so with thip_state_set_ref_idx (which forgets that bookkeeping), as
   thip_state_write_plane does by itself. */
uint8_t *thip_state_frame_ptr(const thip_state *st, int bufi);
/* Synchronous copies of one plane of buffer bufi, tightly packed, bitstream row order. */
int thip_state_read_plane(thip_state *st, int bufi, int pli, uint8_t *host_out);
int thip_state_write_plane(thip_state *st, int bufi, int pli, const uint8_t *host_in);
/* th_decode_ycbcr_out (decode.c:2988): copies the most recently decoded frame to host
   planes in DISPLAY order (top row first), dst_stride[pli] bytes per row. */
int thip_state_ycbcr_out(thip_state *st, uint8_t *const dst[3], const int32_t dst_stride[3]);
/* The same without the copy: pointers to the library's own pinned image of the most recently
   decoded frame (display order, strides[pli] == plane width).  Like the reference's buffers
   (theoradec.h:283-299) they belong to the decoder and must not be written; they stay intact until
   the SECOND following frame has been decoded (two images alternate). */
int thip_state_ycbcr_map(thip_state *st, const uint8_t *planes[3], int32_t strides[3]);
/* The same in two halves: _begin names the picture of the frame decoded last (launching its copy to the host if need be), the
   caller may then hand the NEXT frame over -- it goes to the other of the state's two host images -- and _end waits for the named
   picture and hands it out (what th_decode_ycbcr_out does with option fe_pipeline).  THIP_EINVAL without a _begin; THIP_EFAULT
   from _end if the named frame's hand-over failed: with the next frame on the device it cannot be repeated any more. */
int thip_state_ycbcr_map_begin(thip_state *st);
int thip_state_ycbcr_map_end(thip_state *st, const uint8_t *planes[3], int32_t strides[3]);
/* A frame decoded AHEAD of its turn can be taken back: _mark notes the reference ring (which buffer is GOLD / PREV / the frame
   decoded last: oc_theora_state.ref_frame_idx, state.h:433; rotated at decode.c:2947-2962) before such a frame is handed over, _rewind puts the ring back --
   the frames decoded since the mark never happened as far as references and pictures go: the next frame is decoded against the
   marked references, thip_state_ycbcr_map shows the marked frame again (the picture a _map_begin named before the mark stays
   valid).  What the discarded frame left in its buffer, its half of the coded map and the host image it went to is treated as
   unknown (the next frame takes no static-block shortcut).  Work already queued is not cancelled: it runs, in order, ahead of
   whatever is queued next.  th_decode_packetin uses this when, with option fe_pipeline, another packet arrives than the one
   th_decode_ycbcr_out decoded ahead.  mark: eight words owned by the caller. */
int thip_state_ring_mark(thip_state *st, int64_t mark[8]);
int thip_state_ring_rewind(thip_state *st, const int64_t mark[8]);
/* on != 0: every decoded frame of this state is sent to its pinned host image by the launch that
   decodes it (a kernel behind the loop filter writes it across PCIe), so that
   thip_state_ycbcr_map / _out only wait.  Off by default: a caller that keeps frames on the
   device (bench.py, transcoding) pays nothing. */
int thip_state_set_eager_output(thip_state *st, int on);

/* Out-of-loop post-processing of the most recently decoded frame (decode.c:1608-1957: oc_dec_deblock_frag_rows,
   oc_dec_dering_frag_rows, driven by the MCU loop at decode.c:2893-2911): level as TH_DECCTL_SET_PPLEVEL
   (theoradec.h:40-75; decode.c:32-48: 2 de-block Y, 3 de-ring Y, 4 strong de-ring Y, 5-7 the same for chroma;
   0 and 1 change no pixel).  dc_qis[f]: the quantiser index in force when fragment f was last coded
   (decode.c:1220-1243); frag_qi[f] = qis[frags[f].qii] (decode.c:1926); pp_dc_scale / pp_sharp_mod: the
   decoder's tables (quant.c:88, decode.c:398-409).  All four are HOST arrays.  The decoded frame itself is not
   touched (it stays the reference for the next frames); thip_state_ycbcr_map / _out hand out the post-processed
   picture until the next frame is decoded. */
int thip_state_postprocess(thip_state *st, int level, const uint8_t *dc_qis, const uint8_t *frag_qi,
                           const int32_t pp_dc_scale[64], const int32_t pp_sharp_mod[64]);
/* Synchronous copy of one plane of that picture, tightly packed, bitstream row order (tests). */
int thip_state_read_pp_plane(thip_state *st, int pli, uint8_t *host_out);

/* ------------------------------------------------------------------------------------
 * Work tiles.  The device walks a frame in the reference's CODED ORDER (state.c:123-190):
 * a tile is 4 consecutive super blocks of one super-block row of one plane -- 16x4
 * fragments, 128x32 pixels, one wavefront -- and lane = 16*(super block within the tile) + (position of the
 * fragment on the 4x4 Hilbert curve, state.c:134-139).  Tiles are numbered plane by plane,
 * super-block row by super-block row, left to right.  Lanes that fall outside the plane
 * (ragged right/top edge) exist in the arrays and are ignored.
 * ---------------------------------------------------------------------------------- */
#define THIP_TILE_FRAGS 64
typedef struct thip_tile_geom {
  int32_t tiles_x[3], tiles_y[3]; /* tiles across / super-block rows, per plane */
  int32_t tile_off[3];            /* index of the plane's first tile */
  int32_t ntiles;
} thip_tile_geom;
int thip_state_get_tiles(const thip_state *st, thip_tile_geom *out);
/* Position (tile*64 + lane) of raster fragment index fragi in the tile-ordered arrays. */
int64_t thip_state_frag_pos(const thip_state *st, int64_t fragi);

/* ------------------------------------------------------------------------------------
 * Fragment command stream: what oc_dec_frags_recon_mcu_plane (decode.c:1511-1607) hands to
 * oc_state_frag_recon / oc_frag_copy_list for one frame, laid out for HBM.
 *
 *  frag_info  two 32-bit words per tile position (ntiles*64 positions):
 *               word0  coded       bit  0     0 = uncoded: copied PREV->SELF (fragment.c:37)
 *                      refi        bits 1-2   THIP_FRAME_* (SELF == intra, state.h:215-217)
 *                      dc_only     bit  3     last_zzi<2 (state.c:967-975)
 *                      last_zzi    bits 8-14  0..64 (idct.c:301-330 picks its variant on it)
 *                      mv x, mv y  bits 16-23, 24-31 signed (oc_mv, state.h:232-240)
 *               word1  dc_quant    bits 16-31 the _dc_quant argument of oc_state_frag_recon
 *                      dc          bits 0-15  dc_only blocks: the raw (un-predicted, not yet
 *                                             dequantised) DC coefficient; the kernel forms
 *                                             p=(dc*dc_quant+15)>>5 (state.c:972)
 *  coeffs     one 128-byte slot per coded fragment that is NOT dc_only, slots numbered in
 *             tile/lane order (== the order the reference reconstructs them in).  Slot s
 *             lives in group-of-64 s/64, lane s%64: the block is eight 16-byte pieces, piece
 *             q = 2*j+h (j = row pair 0..3, h = column half 0..1) at
 *             (s/64)*8192 + q*1024 + (s%64)*16, holding for columns c = 4h..4h+3 the int16
 *             pairs { x[2j][c], x[2j+1][c] } (x = the _dct_coeffs argument of
 *             oc_state_frag_recon: natural order, AC dequantised by the caller as in
 *             decode.c:1573, x[0][0] the RAW DC -- the kernel multiplies it by dc_quant,
 *             state.c:978).  This is the backend's
 *             counterpart of the per-backend dct_fzig_zag table (state.h:374-376,
 *             x86state.c:44-64).  Allocate whole groups of 64 slots.
 *  tile_slot0 per tile: slot number of its first non-dc_only coded fragment; the kernel
 *             finds the others with a ballot / prefix count over the tile's lanes.
 *
 * The LEVELS form of the same stream (coeff_format = THIP_COEFFS_LEVELS): what decode.c:1573-1574 does between the
 * tokens and the slot -- `(ogg_int16_t)(coeff*ac_quant[zzi])` -- moves into the reconstruction kernel, and the
 * coefficient slots shrink to the quantised LEVELS the tokens carry (eight bits cover almost every stream; the largest term
 * of a dense frame's traffic halves):
 *  frag_info  word0 additionally carries qii (bits 4-5: the block's index into the frame's qi list, oc_fragment.qii,
 *             state.h:307); the table a block is dequantised with is dequant[pli][qii][qti], qti = (refi != SELF)
 *             (decode.c:1537-1538).  word1 bits 0-15 carry the raw DC of EVERY coded block (no DC in the slots).
 *  dequant    device, 18 tables of 64 uint16, table (pli*3 + qii)*2 + qti, each in SLOT ORDER: entry (j*8 + c)*2 + p
 *             is ac_quant for natural position (2j+p, c) (thip_pack_dequant_table converts a zig-zag-ordered table);
 *             the DC entry is ignored.
 *  coeffs     counted in 64-byte UNITS, groups of 64 units (4096 bytes): piece q (0..3) of unit u at
 *             (u/64)*4096 + q*1024 + (u%64)*16.  A block of a NARROW tile owns one unit of int8 levels: piece j is row
 *             pair j, its dword d the bytes { x[2j][2d], x[2j][2d+1], x[2j+1][2d], x[2j+1][2d+1] }.  A tile in which
 *             some level does not fit eight bits is WIDE: each of its blocks owns two consecutive units holding the
 *             eight int16 pieces described above (pieces 0-3 in the first unit, 4-7 in the second) -- int16 LEVELS,
 *             still multiplied by the table.
 *  tile_slot0 per tile: its first unit | THIP_SLOT_WIDE for a wide tile.
 *  nslots     units in total.
 * ---------------------------------------------------------------------------------- */
#define THIP_INFO_CODED 0x1u
#define THIP_INFO_REFI_SHIFT 1
#define THIP_INFO_DC_ONLY 0x8u
#define THIP_INFO_QII_SHIFT 4 /* levels form: oc_fragment.qii, two bits */
#define THIP_INFO_LAST_ZZI_SHIFT 8
#define THIP_INFO_MVX_SHIFT 16
#define THIP_INFO_MVY_SHIFT 24
#define THIP_SLOT_GROUP 64
#define THIP_SLOT_GROUP_BYTES 8192
#define THIP_COEFFS_DEQUANT16 0 /* slots of dequantised int16 coefficients: what oc_state_frag_recon receives */
#define THIP_COEFFS_LEVELS 1    /* units of quantised levels + the frame's dequantisation tables */
#define THIP_UNIT_BYTES 64
#define THIP_UNIT_GROUP_BYTES 4096
#define THIP_SLOT_WIDE 0x80000000u

typedef struct thip_frame_desc {
  const uint32_t *frag_info;  /* device, 2*64*ntiles words */
  const int16_t *coeffs;      /* device, ceil(nslots/64) groups of 8192 bytes */
  const uint32_t *tile_slot0; /* device, ntiles words */
  int32_t nslots;             /* coded fragments carrying coefficients */
  int32_t ncoded;             /* coded fragments in total; 0 => TH_DUPFRAME (decode.c:2764) */
  int32_t frame_type;         /* THIP_INTRA_FRAME / THIP_INTER_FRAME */
  int32_t flimit;             /* loop_filter_limits[qis[0]], 0 = filter off (decode.c:1369-1371) */
  /* Optional (NULL = the DC values in frag_info / coeffs are final, as oc_state_frag_recon receives
     them).  Non-NULL: device int16[nfrags], fragment-index (raster, plane by plane) order: the DC value of
     every coded fragment AS DECODED FROM THE TOKENS, i.e. before oc_dec_dc_unpredict_mcu_plane
     (decode.c:1392-1500).  The launch then undoes the DC prediction on the device (k_dc_unpredict, an
     anti-diagonal wavefront per plane) and the reconstruction takes every block's DC from its result; the
     DC fields of frag_info / coeffs are then ignored.  Planes of more than 1024 fragment rows: TH_EIMPL. */
  const int16_t *dc_tokens;
  int32_t coeff_format;       /* THIP_COEFFS_DEQUANT16 (0) or THIP_COEFFS_LEVELS */
  const uint16_t *dequant;    /* THIP_COEFFS_LEVELS: device, 18 x 64 entries in slot order (see above) */
} thip_frame_desc;
/* One AC dequantisation table from the reference's order (zig-zag index, dequant[pli][qii][qti], decode.c:1537) into the
   order the kernels multiply a slot with: out[(j*8 + c)*2 + p] = zz[zig-zag index of natural position (2j+p, c)]. */
void thip_pack_dequant_table(uint16_t out[64], const uint16_t zz[64]);

/* One frame of each of nstreams independent streams, all inputs resident in HBM:
   reconstruct coded fragments (oc_state_frag_recon, state.c:959), copy uncoded ones
   (fragment.c:37), run the in-loop filter over the whole frame (state.c:1055) and rotate
   each stream's reference ring (decode.c:2790-2794, 2947-2962).  Asynchronous on
   `stream` (a hipStream_t, or NULL for the library's own stream).
   results[i] (optional, host) receives 0 or THIP_DUPFRAME per stream. */
int thip_decode_frames(thip_state *const *states, const thip_frame_desc *descs, int nstreams,
                       void *stream, int32_t *results);
/* Wait for everything submitted on the library's own streams.  k_recon_lf hands tile edges between concurrently running
   work groups and relies on their being dispatched in order; every wait is bounded, and one that gives up sets a pinned word
   of the state (it has never been observed outside the test that forces it).  The calls that synchronise with a state --
   thip_state_ycbcr_map / _out, thip_state_read_plane -- look at the word: the state's most recent frame is decoded
   AGAIN with the two-pass kernels when its command stream lives in the state's own staging buffers (frames that came through
   the enqueue slots or the token lists, i.e. every th_decode_* frame) and nothing has been decoded since; the call then
   succeeds and option "faults_recovered" counts.  A frame given to thip_decode_frames by descriptor can be decoded again only
   if the caller has promised (option "redo_descs" = 1) that the buffers its descriptors point to stay as they are until the
   state's next synchronising call; otherwise, and whenever a frame was decoded on top of the failed one before anything looked
   (the kernels record WHICH launch gave up), the call returns THIP_EFAULT, once, and the state's pictures are wrong until its
   next key frame.  thip_synchronize itself only reports (THIP_EFAULT while some state's words are set): a state belongs to
   one thread at a time, so decoding again is left to the state's own synchronising calls -- or to thip_state_check_fault,
   for a caller that keeps its frames on the device and brackets its work with thip_synchronize only: it waits for the state's
   stream and does what those calls do about a set word.  0: nothing was wrong; 1: the newest frame was decoded again and is
   right; THIP_EFAULT: see above (reported once -- the words are cleared). */
int thip_synchronize(void);
int thip_state_check_fault(thip_state *st);

/* ------------------------------------------------------------------------------------
 * Host-enqueue form of the same path: the vtable slots as the reference calls them, one
 * fragment at a time from decode.c:1584 / :1601 / :2882.  They only stage into pinned
 * memory; thip_frame_flush launches the kernels, which read the pinned staging in place
 * across PCIe (THIP_ZEROCOPY=0 in the environment copies it to device memory first).
 * ---------------------------------------------------------------------------------- */
/* Where th_decode_packetin picks the SELF buffer (decode.c:2790-2794). */
int thip_frame_begin(thip_state *st, int frame_type);
/* oc_state_frag_recon slot (state.h:365-366).  refi and mv are what the reference reads
   from state->frags[fragi].refi and state->frag_mvs[fragi].  Like the reference's iDCT
   (idct.c:245,276,295) it leaves dct_coeffs[0..63] zeroed. */
int thip_state_frag_recon(thip_state *st, ptrdiff_t fragi, int pli, int16_t dct_coeffs[128],
                          int last_zzi, uint16_t dc_quant, int refi, int16_t mv);
/* oc_frag_copy_list slot (state.h:355-357); frame pointers and offsets are implied. */
int thip_frag_copy_list(thip_state *st, const ptrdiff_t *fragis, ptrdiff_t nfragis);
/* oc_loop_filter_init slot (state.h:367): the bounding-value table of state.c:1036. */
void thip_loop_filter_init(signed char bv[256], int flimit);
/* oc_state_loop_filter_frag_rows slot (state.h:368): records the request; the filter
   itself runs over the recorded rows at flush.  Only refi==THIP_FRAME_SELF is valid. */
int thip_state_loop_filter_frag_rows(thip_state *st, int flimit, int refi, int pli, int fragy0,
                                     int fragy_end);
/* Upload, launch, rotate the ring.  Returns 0 or THIP_DUPFRAME. */
int thip_frame_flush(thip_state *st);
/* Token form of the oc_state_frag_recon slot: what decode.c:1540-1581 does between the token lists and
   the call -- zero fill, scatter through the zig-zag table, `(ogg_int16_t)(coeff*ac_quant[zzi])`
   (decode.c:1573) -- moves to the device (k_expand_tokens).  The caller hands over the fragment's tokens
   as they delimit it: toks[k] = zig-zag position (1..63) << 16 | quantised value (low 16 bits, two's
   complement), ntoks <= 63, in any order, zero-valued ones left out; `dc` is the fragment's DC exactly as
   dct_coeffs[0] would carry it; dqsel names one of the frame's AC dequantisation tables
   (thip_frame_dequant_table: 64 entries in zig-zag order, the reference's dequant[pli][qii][qti],
   decode.c:1537-1538; sel 0..17, valid until the flush).  4 bytes per non-zero coefficient cross PCIe
   instead of 128 per block.  One form per frame for the blocks that need a coefficient slot (last_zzi >= 2): once a
   frame has received such a block through one entry point, the other refuses them with THIP_EINVAL and leaves
   the frame as it was; DC-only blocks (last_zzi < 2) own no slot and may arrive through either. */
int thip_frame_dequant_table(thip_state *st, int sel, const uint16_t dequant[64]);
/* Levels form of the oc_state_frag_recon slot: dct_coeffs[1..63] hold the QUANTISED levels as the tokens carry them (the caller
   leaves out the multiplication of decode.c:1573-1574; natural order, dct_coeffs[0] the raw DC as ever, zeroed on return like the
   other slot), qii = frags[fragi].qii (state.h:307).  The frame's tables go over with thip_frame_dequant_table (sel = (pli*3 + qii)*2
   + qti) before the flush and `(ogg_int16_t)(coeff*ac_quant[zzi])` is done by the reconstruction kernel: 64 bytes of staging a
   block instead of 128 (int16 units for the tiles in which a level does not fit eight bits).  One form per frame for the blocks
   that own a slot, as above. */
int thip_state_frag_recon_levels(thip_state *st, ptrdiff_t fragi, int pli, int16_t dct_coeffs[128], int last_zzi, uint16_t dc_quant,
                                 int qii, int refi, int16_t mv);
int thip_state_frag_recon_tokens(thip_state *st, ptrdiff_t fragi, int pli, const uint32_t *toks, int ntoks,
                                 int16_t dc, int last_zzi, uint16_t dc_quant, int dqsel, int refi, int16_t mv);

/* The whole step between the entropy decoder and the pixel path on the device (SURVEY section 8f rank 1 in its
   parallel form; decode.c:1511-1587 with the lists of decode.c:993-1139 as input, DC un-prediction
   decode.c:1392-1500 included): the caller hands over the frame's DCT tokens as the entropy decoder leaves them
   -- one list per (plane, zig-zag index) -- plus one word per coded fragment, and the backend finds each
   fragment's tokens (two prefix sums per index over the arrivals and over what the tokens consume,
   k_tok_assign), expands and dequantises them, un-predicts the DC values (k_dc_unpredict), builds the command
   stream and decodes the frame.  Stands for thip_frame_begin ... thip_frame_flush of one frame.
   tokens: 32-bit words, all lists concatenated: bits 0-15 the value (two's complement; 0 for a pure zero run),
   bits 16-22 the zeros before it, bit 23 set for an EOB token, whose run length is bits 0-15 | bits 24-31 << 16
   (the part of a run that reaches past its list is the later lists' eob_carry).
   list_off / list_len: first token and length of list [plane][zzi].  eob_carry[plane][zzi]: fragments ended at
   that index by a run from an earlier list.  arrivals[plane][zzi]: fragments open at that index (carry included).
   coded: the coded fragments in coded order, plane after plane (ncoded per plane); frag_meta per coded fragment:
   refi | dequantisation table << 2 | (mvx & 255) << 8 | (mvy & 255) << 16 | plane << 24, table = (plane * 3 + qii)
   * 2 + qti into dequant[18][64] (zig-zag order, decode.c:1537-1538).  dc_quant[plane][qti] as the slot's _dc_quant.
   dc: NULL -- the DC token values are un-predicted on the device (one wave per plane walks the chain of decode.c:1392-1500:
   exact, and slow when neighbouring fragments differ in their reference frames) -- or the UN-PREDICTED DC value of every coded
   fragment, in the order of `coded`: the caller has run oc_dec_dc_unpredict_mcu_plane itself (a few nanoseconds a fragment on
   a host core) and the device does everything else.
   Returns 0, THIP_DUPFRAME (nothing coded), or THIP_EIMPL when a plane has more than 147456 coded fragments (beyond 4K) or
   more than 1024 fragment rows (the caller falls back to the slots); all pointers are host memory, read before return. */
typedef struct thip_token_lists {
  int32_t frame_type;          /* THIP_INTRA_FRAME / THIP_INTER_FRAME */
  int32_t flimit;              /* loop_filter_limits[qis[0]] */
  const uint32_t *tokens;
  int64_t ntokens;
  uint32_t list_off[3][64], list_len[3][64], eob_carry[3][64], arrivals[3][64];
  const int32_t *coded;
  const uint32_t *frag_meta;
  int32_t ncoded[3];
  const uint16_t *dequant;     /* [18][64] */
  uint16_t dc_quant[3][2];
  const int16_t *dc;           /* NULL, or the un-predicted DC of every coded fragment (order of `coded`) */
} thip_token_lists;
int thip_state_decode_token_lists(thip_state *st, const thip_token_lists *tl);
/* The same in two steps, for a caller that un-predicts the DC values itself: _begin takes everything but `dc` (which it
   ignores) and starts the device on the token lists; the caller runs oc_dec_dc_unpredict_mcu_plane meanwhile -- the device
   needs the DC values last, when it writes the command words -- and hands them to _finish (NULL: un-predict on the device),
   which returns what thip_state_decode_token_lists returns.  Nothing else may be done with the state in between (THIP_EINVAL
   from _begin and from the enqueue slots while a frame is pending).  thip_state_decode_token_lists(st, tl) is
   _begin(st, tl) followed by _finish(st, tl->dc). */
int thip_state_token_lists_begin(thip_state *st, const thip_token_lists *tl);
int thip_state_token_lists_finish(thip_state *st, const int16_t *dc);
/* _begin for a caller that has done the walk of decode.c:1540-1581 itself -- which token belongs to which fragment is 64 dependent
   rounds per plane, one compute unit's work on the device (k_tok_assign) and nothing at all on a host thread that is not on
   anybody's critical path (th_decode_*'s look-ahead parses announced packets on threads of their own).  `assign`: one word per
   entry of tl->tokens: bits 0-17 the index, in the order of tl->coded, of the fragment the token belongs to, bits 18-24 the
   zig-zag position its value lands at (the list's index + the zeros before the value; positions beyond 63 are dropped like the
   reference's dump slot, decint.h:96), 0xFFFFFFFF for a token no fragment consumes (EOB tokens' words are not read);
   `last_zzi`: for every fragment of tl->coded the index at which it met its end (decode.c:1545).  The device pairs nothing: one
   thread per token stores the (dequantised) value (k_tok_scatter), the rest is _begin's.  tl->list_off / list_len / eob_carry /
   arrivals are not read.  Returns what _begin returns, or THIP_EIMPL when the two arrays do not fit behind the tokens in the
   state's staging buffer (a frame of nearly 32 tokens a fragment) or the frame has more than 262143 coded fragments: call
   _begin then -- the state is untouched.  Followed by _finish, as _begin is. */
int thip_state_token_lists_begin_assigned(thip_state *st, const thip_token_lists *tl, const uint32_t *assign, const uint8_t *last_zzi);
/* The lists in GROUPS of zig-zag indices, as the entropy decoder finishes them (decode.c:1164-1205 reads index after index, all
   three planes of one before the next): _open takes the frame's description without its tokens (tl->tokens, ntokens, list_off,
   list_len, eob_carry, arrivals and dc are ignored) and prepares the device; every _append hands over the lists of the indices
   [z0, z1) -- `tokens` holds just those, list_off[p][z] counts from its start, the four tables are read at columns z0..z1-1 only --
   and the device walks them while the caller decodes the next indices; the groups come in order and without gaps (the first
   z0 is 0, each z0 the z1 before it, THIP_EINVAL otherwise); after the group that ends at 64, _finish as above.  _abort gives up a
   frame that was opened (a group failed its checks: the state is as it was before _open).  _begin(st, tl) is _open(st, tl)
   followed by _append(st, 0, 64, tl->tokens, tl->ntokens, tl->list_off, ...).  All pointers are host memory, read before return;
   one thread at a time per state (the calls of one frame may come from different threads, one after the other). */
int thip_state_token_lists_open(thip_state *st, const thip_token_lists *tl);
/* The pinned buffer the device reads a frame's arrays from, for a caller that writes them there itself: `coded`, `frag_meta`
   (one entry per fragment of the frame at most), `dequant` ([18][64]) and `tokens` (token_capacity entries).  Pointers handed to
   _open / _append that ARE these are not copied: tl->coded == coded, tl->frag_meta == frag_meta, tl->dequant == dequant, and the
   tokens of a group written at tokens + n, n = the tokens handed over before it rounded up to a multiple of 4.  The call waits
   until the device is done with the previous frame's contents; the pointers hold until the frame's _finish or _abort.
   THIP_EIMPL where _open would return it for the frame size. */
typedef struct thip_token_staging {
  int32_t *coded;
  uint32_t *frag_meta;
  uint16_t *dequant;
  uint32_t *tokens;
  int64_t token_capacity;
} thip_token_staging;
int thip_state_token_lists_staging(thip_state *st, thip_token_staging *out);
int thip_state_token_lists_append(thip_state *st, int z0, int z1, const uint32_t *tokens, int64_t ntokens,
                                  const uint32_t (*list_off)[64], const uint32_t (*list_len)[64],
                                  const uint32_t (*eob_carry)[64], const uint32_t (*arrivals)[64]);
int thip_state_token_lists_abort(thip_state *st);
/* on != 0: the DC coefficient handed to thip_state_frag_recon (dct_coeffs[0]) is the value decoded from
   the tokens, NOT yet un-predicted: the caller skips its oc_dec_dc_unpredict_mcu_plane calls
   (decode.c:2869) and thip_frame_flush undoes the prediction on the device before reconstructing
   (thip_frame_desc.dc_tokens).  Takes effect from the next thip_frame_begin. */
int thip_state_set_device_dc(thip_state *st, int on);

/* ------------------------------------------------------------------------------------
 * Batched forms of the individual slots (device pointers).  These exist so each vtable
 * entry can be parity-tested on its own; the frame path above fuses them.
 * ---------------------------------------------------------------------------------- */
/* oc_idct8x8 (state.h:364, idct.c:301): n blocks of 64 natural-order coefficients. */
int thip_idct8x8_batch(int16_t *y, const int16_t *x, const int32_t *last_zzi, int64_t n);
/* oc_frag_recon_intra / _inter / _inter2 (state.h:358-363; fragment.c:49-80), selected by
   nsrc = 0 / 1 / 2.  dst_offs/src*_offs are byte offsets of each block's first row. */
int thip_frag_recon_batch(uint8_t *dst_frame, const uint8_t *src_frame, int ystride, int nsrc,
                          const int32_t *dst_offs, const int32_t *src1_offs,
                          const int32_t *src2_offs, const int16_t *residue, int64_t n);
/* oc_frag_copy_list (fragment.c:37) with explicit frames and offsets. */
int thip_frag_copy_list_batch(uint8_t *dst_frame, const uint8_t *src_frame, int ystride,
                              const int32_t *fragis, int64_t nfragis, const int32_t *frag_buf_offs);
/* oc_dec_dc_unpredict_mcu_plane (the oc_dec_opt_vtable slot, decint.h:72-75; decode.c:1392-1500) over one
   whole plane, in place on the device: dc = int16 per fragment (raster) holding the token DC values of the
   coded fragments on entry and their un-predicted values on return; flags = 1 byte per fragment, bit 0
   coded, bits 1-2 the fragment's reference frame index (state.h:170-176).  pred_last starts at 0
   (decode.c:1367).  nvfrags <= 1024. */
int thip_dc_unpredict_plane(int16_t *dc, const uint8_t *flags, int nhfrags, int nvfrags);
/* oc_state_loop_filter_frag_rows (state.c:1055) on one plane: coded = 1 byte per fragment
   of that plane (raster), rows [fragy0,fragy_end). */
int thip_loop_filter_plane(uint8_t *plane, int ystride, int nhfrags, int nvfrags,
                           const uint8_t *coded, int flimit, int fragy0, int fragy_end);

/* ------------------------------------------------------------------------------------
 * Encoder block kernels (oc_enc_opt_vtable, encint.h:292-326), batched: element i works on
 * the 8x8 block at src_plane+src_offs[i] (and ref_plane+ref_offs[i], +ref2_offs[i]).
 * ---------------------------------------------------------------------------------- */
enum {
  THIP_ENC_SAD = 0,         /* oc_enc_frag_sad,         encfrag.c:42  */
  THIP_ENC_SAD_THRESH = 1,  /* oc_enc_frag_sad_thresh,  encfrag.c:56  */
  THIP_ENC_SAD2_THRESH = 2, /* oc_enc_frag_sad2_thresh, encfrag.c:71  */
  THIP_ENC_INTRA_SAD = 3,   /* oc_enc_frag_intra_sad,   encfrag.c:88  */
  THIP_ENC_SATD = 4,        /* oc_enc_frag_satd,        encfrag.c:317 */
  THIP_ENC_SATD2 = 5,       /* oc_enc_frag_satd2,       encfrag.c:324 */
  THIP_ENC_INTRA_SATD = 6,  /* oc_enc_frag_intra_satd,  encfrag.c:331 */
  THIP_ENC_SSD = 7          /* oc_enc_frag_ssd,         encfrag.c:338 */
};
int thip_enc_frag_metric_batch(int op, uint32_t *out, int32_t *dc_out, const uint8_t *src_plane,
                               const uint8_t *ref_plane, int ystride, const int32_t *src_offs,
                               const int32_t *ref_offs, const int32_t *ref2_offs,
                               uint32_t thresh, int64_t n);
/* The motion-search form of oc_enc_frag_sad (encfrag.c:42) / oc_enc_frag_satd (encfrag.c:317): every block i against
   nsites candidate positions around ONE reference position, candidate c = ref_plane + ref_offs[i] + site_dy[c]*ystride
   + site_dx[c] with site_dx, site_dy in {-1, 0, 1} (the square pattern of mcenc.c:50-53, any subset, any order, no
   position twice).  What the reference does with nsites calls per block -- oc_mcenc_ysad_check_mbcandidate_fullpel /
   oc_mcenc_ysatd_check_mbcandidate_fullpel, mcenc.c:267-330 -- is one launch here: the source block and the ten
   reference rows a column of candidates shares are fetched and prepared once.  Results candidate-major:
   out[c*nblocks + i] (and dc_out, SATD only, may be null).  op: THIP_ENC_SAD or THIP_ENC_SATD. */
int thip_enc_frag_metric_sites_batch(int op, uint32_t *out, int32_t *dc_out, const uint8_t *src_plane,
                                     const uint8_t *ref_plane, int ystride, const int32_t *src_offs,
                                     const int32_t *ref_offs, const int8_t *site_dx, const int8_t *site_dy,
                                     int nsites, int64_t nblocks);
/* The half-pel refinement's form of oc_enc_frag_satd2 (encfrag.c:323-328) / oc_enc_frag_sad2_thresh (encfrag.c:62-86, without a
   threshold: every row is added): every block i against nsites of the eight half-pel vectors 2 * vec[i] + (site_dx[c],
   site_dy[c]) around its whole-pel vector vec[i] (vecs[i] = x & 0xFF | y << 8, the reference's oc_mv, state.h:232-240), as
   oc_mcenc_ysatd_halfpel_mbrefine / oc_mcenc_ysad_halfpel_mbrefine do it (mcenc.c:551-657): the source block against the
   truncating average of the two whole-pel blocks ref_plane + ref_offs[i] + mvoffset0 / mvoffset1 of mcenc.c:644-647 (ref_offs[i]
   is the block at the whole-pel vector, the reference's frag_offs + mvoffset_base).  One launch instead of nsites calls per
   block: the ten rows around the whole-pel position are fetched once per column, and the two vertical sites share their nine
   averaged rows' horizontal Hadamard levels.  (0, 0) is not a half-pel site (THIP_EINVAL).  Results site-major: out[c*nblocks
   + i] (and dc_out, SATD2 only, may be null).  op: THIP_ENC_SATD2 or THIP_ENC_SAD2_THRESH.
   Footprint: a row is fetched as one 12-byte window that starts at column -1 or 0 of the whole-pel block, ten rows from the row
   above it -- up to three bytes beyond the 9 x 10 samples the refinement can need.  ref_plane must be readable that far: one
   pixel of margin around every whole-pel block (what the refinement itself needs) plus FOUR more bytes behind the last row's
   last needed sample.  The reference's frames have a 16-pixel border (state.c:545-671), which covers it. */
int thip_enc_frag_metric_halfpel_batch(int op, uint32_t *out, int32_t *dc_out, const uint8_t *src_plane,
                                       const uint8_t *ref_plane, int ystride, const int32_t *src_offs,
                                       const int32_t *ref_offs, const int16_t *vecs, const int8_t *site_dx,
                                       const int8_t *site_dy, int nsites, int64_t nblocks);
/* The encoder's per-macro-block cost maps for a whole frame, computed up front in one launch: what
   oc_mb_intra_satd (analyze.c:1360-1403), oc_mb_activity (analyze.c:1152-1237) and oc_mb_activity_fast (analyze.c:1239-1251)
   return for every macro block -- the reference calls them macro block by macro block from its mode-decision loop
   (analyze.c:1697-1718, 2372-2398), but they depend on the input picture alone.  planes: the frame to be encoded (device, unpadded,
   bitstream row order like everything here; reads beyond a plane's edge are clamped, which is the replicated border of the
   reference's input frame, encode.c:1735-1744); strides in bytes; frame size the coded size (multiples of 16).
   Macro blocks are numbered as in the reference: mbi = luma super block << 2 | quadrant (state.c:300-330; thip_enc_mb_count of
   them, those outside the frame are zero in every array).  Outputs (device, each may be NULL):
     intra_satd[mbi*12 + k]   _frag_satd[k] of oc_mb_intra_satd: k = 0..3 the luma blocks in sb_maps order (state.c:134-139),
                              then the macro block's Cb and Cr blocks in OC_MB_MAP_IDXS order (internal.c:67-76: 1+1, 2+2 or 4+4
                              by pixel format), unused entries 0
     luma[mbi]                its return value (the sum of the four luma blocks' pixels)
     activity[mbi*4 + k]      _activity[k] of oc_mb_activity (edge classification and its 0.7 power included)
     activity_fast[mbi*4 + k] _activity[k] of oc_mb_activity_fast
   Stream and synchronisation as thip_set_batch_stream says. */
int thip_enc_mb_count(int frame_width, int frame_height);
int thip_enc_mb_cost_maps(const uint8_t *const planes[3], const int32_t strides[3], int frame_width, int frame_height, int pixel_fmt,
                          uint32_t *intra_satd, uint32_t *luma, uint32_t *activity, uint32_t *activity_fast);
/* oc_enc_frag_border_ssd (encfrag.c:352): per-block 64-bit pixel masks (state.h:285-292). */
int thip_enc_frag_border_ssd_batch(uint32_t *out, const uint8_t *src_plane,
                                   const uint8_t *ref_plane, int ystride,
                                   const int32_t *src_offs, const int32_t *ref_offs,
                                   const int64_t *masks, int64_t n);
/* oc_enc_frag_sub / _sub_128 (encfrag.c:21,32): ref_offs==NULL selects sub_128. */
int thip_enc_frag_sub_batch(int16_t *diff, const uint8_t *src_plane, const uint8_t *ref_plane,
                            int ystride, const int32_t *src_offs, const int32_t *ref_offs,
                            int64_t n);
/* oc_enc_frag_copy2 (encfrag.c:368) */
int thip_enc_frag_copy2_batch(uint8_t *dst_plane, const uint8_t *src_plane, int ystride,
                              const int32_t *dst_offs, const int32_t *src1_offs,
                              const int32_t *src2_offs, int64_t n);
/* The batched single-slot and encoder entry points of this header (thip_idct8x8_batch ...
   thip_enc_quantize_batch) run on the null stream and return when the work is done, like the C
   slots they stand for.  An encoder that chains them (sub -> fdct -> quantize ...) sets a stream
   of its own and synchronous = 0: the calls then only enqueue, in order, on that stream (a
   hipStream_t; NULL = the null stream) and the caller synchronises when it needs the results.
   Process-wide setting. */
int thip_set_batch_stream(void *stream, int synchronous);
/* oc_enc_fdct8x8 (fdct.c:128): natural-order int16 in, ZIG-ZAG-ordered int16 out. */
int thip_enc_fdct8x8_batch(int16_t *y, const int16_t *x, int64_t n);
/* oc_enc_quantize (enquant.c:219) with the reciprocals of oc_enc_enquant_table_init
   (enquant.c:193) derived on the device: n blocks of 64 zig-zag-ordered coefficients against
   one 64-entry dequantisation table (device, zig-zag order); nonzero[i] = index of the last
   non-zero quantised coefficient (the function's return value). */
int thip_enc_quantize_batch(int16_t *qdct, int32_t *nonzero, const int16_t *dct, const uint16_t *dequant,
                            int64_t n);

/* oc_enc_enquant_table_init / _fixup (encint.h:316-318, enquant.c:194-217): HOST functions, as in the
   reference -- the tables are built when the quantisation parameters change.  `enquant` is 64 entries
   of {int16 m, int16 l} (oc_iquant), THIP_ENQUANT_TABLE_SIZE bytes; thip_enc_opt_data reports what
   oc_enc_opt_data carries (encint.h:331-338). */
#define THIP_ENQUANT_TABLE_SIZE 256
void thip_enc_enquant_table_init(void *enquant, const uint16_t dequant[64]);
void thip_enc_enquant_table_fixup(void *enquant[3][3][2], int nqis);
void thip_enc_opt_data(size_t *enquant_table_size, int *enquant_table_alignment);
/* oc_enc_quantize with the table the slot receives (encint.h:319-320): dequant and enquant are DEVICE
   copies of one 64-entry table each, built once and reused by every launch. */
int thip_enc_quantize_tab_batch(int16_t *qdct, int32_t *nonzero, const int16_t *dct, const uint16_t *dequant,
                                const void *enquant, int64_t n);

/* oc_enc_fdct8x8 and oc_enc_quantize on the same blocks in one pass (what the encoder does with every residual block: fdct.c:128,
   then enquant.c:219): x = n blocks of 64 natural-order int16 (the residual), qdct / nonzero as thip_enc_quantize_tab_batch
   returns them for the transform of x; dct (may be NULL) additionally receives the unquantised zig-zag-ordered coefficients,
   as thip_enc_fdct8x8_batch would; enquant NULL: the reciprocals are derived on the device as in thip_enc_quantize_batch.
   The coefficients do not travel to memory and back between the two steps, and there is one launch instead of two. */
int thip_enc_fdct_quantize_batch(int16_t *qdct, int32_t *nonzero, int16_t *dct, const int16_t *x, const uint16_t *dequant,
                                 const void *enquant, int64_t n);

/* ------------------------------------------------------------------------------------
 * The single-block slots of oc_enc_opt_vtable (encint.h:292-326) with the reference's signatures:
 * HOST pointers, one 8x8 block per call, each bound to a one-element batch of the kernels above.
 * oc_enc_accel_init_hip fills the vtable with these; they exist so that every
 * slot can be compared with its C original at the encoder's own call sites -- throughput comes from the
 * batch entry points.  (enquant_table_init / _fixup: above.)
 * ---------------------------------------------------------------------------------- */
void thip_enc1_frag_sub(int16_t diff[64], const unsigned char *src, const unsigned char *ref, int ystride);
void thip_enc1_frag_sub_128(int16_t diff[64], const unsigned char *src, int ystride);
unsigned thip_enc1_frag_sad(const unsigned char *src, const unsigned char *ref, int ystride);
unsigned thip_enc1_frag_sad_thresh(const unsigned char *src, const unsigned char *ref, int ystride, unsigned thresh);
unsigned thip_enc1_frag_sad2_thresh(const unsigned char *src, const unsigned char *ref1, const unsigned char *ref2,
                                    int ystride, unsigned thresh);
unsigned thip_enc1_frag_intra_sad(const unsigned char *src, int ystride);
unsigned thip_enc1_frag_satd(int *dc, const unsigned char *src, const unsigned char *ref, int ystride);
unsigned thip_enc1_frag_satd2(int *dc, const unsigned char *src, const unsigned char *ref1, const unsigned char *ref2,
                              int ystride);
unsigned thip_enc1_frag_intra_satd(int *dc, const unsigned char *src, int ystride);
unsigned thip_enc1_frag_ssd(const unsigned char *src, const unsigned char *ref, int ystride);
unsigned thip_enc1_frag_border_ssd(const unsigned char *src, const unsigned char *ref, int ystride, int64_t mask);
void thip_enc1_frag_copy2(unsigned char *dst, const unsigned char *src1, const unsigned char *src2, int ystride);
int thip_enc1_quantize(int16_t qdct[64], const int16_t dct[64], const uint16_t dequant[64], const void *enquant);
void thip_enc1_frag_recon_intra(unsigned char *dst, int ystride, const int16_t residue[64]);
void thip_enc1_frag_recon_inter(unsigned char *dst, const unsigned char *src, int ystride, const int16_t residue[64]);
void thip_enc1_fdct8x8(int16_t y[64], const int16_t x[64]);

/* ------------------------------------------------------------------------------------
 * Measurement support for bench.py: HIP-event timing of the kernels of
 * thip_decode_frames on the stream they run on.
 * ---------------------------------------------------------------------------------- */
#define THIP_KERNEL_RECON 0      /* k_recon: reconstruction + uncoded copy */
#define THIP_KERNEL_LOOPFILTER 1 /* k_loopfilter: in-loop deblocking */
#define THIP_NKERNELS 2
int thip_profile_enable(int on);
/* Sums since the last reset: launches and milliseconds per kernel (synchronises). */
int thip_profile_read(int64_t launches[THIP_NKERNELS], double ms[THIP_NKERNELS]);
int thip_profile_reset(void);

const char *thip_version_string(void);

/* ------------------------------------------------------------------------------------
 * Run-time options.  One table inside the library: the environment is read ONCE, when the first
 * option is looked up (THIP_<NAME> in upper case = integer value; THIP_DEVICE also takes "rr"), and
 * thip_set_option changes a value from then on.  Options are read where they are used -- a change
 * takes effect with the next call -- except "lanes" and "ctx_lanes", which size stream pools created when
 * a device is first used.  thip_option_name enumerates the table (NULL past its end) with a one-line
 * description of each entry:
 *   fuse         3 (default) k_recon_lf: reconstruction + loop filter in one pass; 0 the two passes
 *   sb_tiles     launches of fewer tiles than this (all their streams together; default 600: a 720p frame has 345, a 1080p frame 782 and
 *                gains nothing) take k_recon_lf_sb -- one
 *                super block per wave, four lanes per block -- instead of k_recon_lf's one tile per wave (0: never)
 *   half_tiles   launches of at least sb_tiles and fewer tiles than this take k_recon_lf_h -- two super blocks per wave, two lanes per
 *                block (round 6) -- instead of k_recon_lf (default 0: never; measured: nothing for one 1080p stream, slower elsewhere)
 *   lanes        library-owned HIP streams per device for thip_decode_frames (default 2)
 *   ctx_lanes    HIP streams shared by the enqueue-fed states, i.e. th_decode_* contexts (default 8; 0 = the lanes)
 *   chunk        streams per kernel launch (default THIP_MAX_BATCH)
 *   skip_static  leave static blocks in place: 0 never, 1 when most of the frame is uncoded (default), 2 whenever possible
 *   lf_sparse    k_loopfilter reads the coded flags first: -1 by coded fraction (default), 0 never, 1 always
 *   zerocopy     enqueue path: kernels read the pinned staging across PCIe (default 1)
 *   wait_spin    wait for a frame in hipEventSynchronize instead of polling with short sleeps (default 0)
 *   dc_global    DC un-prediction through memory even where the LDS kernel fits (default 0)
 *   debug        k_recon ablation switches (profiling); 256: k_recon_lf's cells copy without filtering; 512: tile 1 of every
 *                stream mis-tags its edge units, so that a hand-over fails and the recovery below can be tested
 *   faults_recovered   (counter) frames decoded a second time with the two passes because a bounded wait of k_recon_lf ran out
 *   tl_dc_copy     thip_state_token_lists_finish: 0 (default) a kernel copies the DC values out of the pinned staging buffer, 1 a copy engine
 *   spec_coeffs    k_recon_lf, levels form: 1 (default) frames with a unit for every block get their coefficients asked for at wave start (address from the tile number, checked later), 0 never
 *   enc_sites_lds  thip_enc_frag_metric_sites_batch, SATD: 1 (default) the source block shared by a block's three lanes through LDS, 0 every lane its own
 *   enc_fdct_lanes thip_enc_fdct8x8_batch: 4 (default) four lanes per block, 1 one block per lane
 *   enc_fq_lanes   thip_enc_fdct_quantize_batch: 4 (default) four lanes per block, 1 one block per lane
 *   enc_halfpel_lanes  thip_enc_frag_metric_halfpel_batch: 2 (default) a lane per side with four sites each, 3 a lane per dx
 *   redo_descs   thip_decode_frames on the caller's descriptors: 1 = the caller promises that the buffers a descriptor points to stay
 *                as they are until the state's next synchronising call, so a frame whose hand-over failed is decoded again (default 0:
 *                THIP_EFAULT, see thip_synchronize)
 *   fe_device_dc, fe_device_tokens   th_decode_*: front-end stages on the device (default 0; also TH_DECCTL_THIP_SET_DEVICE_*
 *                per context)
 *   fe_device_lists   th_decode_*: the token lists go to the device as the entropy decoder leaves them: 1 on, 0 off, -1 (default)
 *                on while at most four decoder contexts are alive in the process and neither of the two above is set (one to
 *                four streams decode a fifth faster that way, sixteen slower); TH_DECCTL_THIP_SET_DEVICE_LISTS per context
 *   tl_algo      thip_state_token_lists_*: which kernels pair tokens and fragments (thip_tokens.h): 1 k_tok_assign, 2 k_tok_rank +
 *                k_tok_walk, 0 (default) the second for planes of more than 36 864 coded fragments (4K), the first otherwise -- equal
 *                end to end at every size, the second 22 % less kernel time at 4K; tl_walk_threads: k_tok_walk's work-group size
 *                (256, 512, 1024; 0 = by plane size)
 *   tl_levels    thip_state_token_lists_* / thip_state_decode_token_lists: 1 (default): the device builds the coefficient slots in
 *                the levels form (THIP_COEFFS_LEVELS: int8 units, wide tiles where a level needs more; the reconstruction kernel
 *                dequantises); 0: dequantised int16 slots, as in round 3
 *   fe_groups    th_decode_*, token-list path: how many groups of zig-zag indices a frame's lists go to the device in WHILE the packet is
 *                being decoded (thip_state_token_lists_open / _append; boundaries in thip_frontend.cpp, kFeGroupEnd*): 6 (default since round 6:
 *                {3, 10, 28, 48, 64}: what the device still walks behind the packet's last bit is sixteen indices instead of thirty-six,
 *                + 6 % at 1080p, the same at 720p), 4 ({3, 10, 28, 64}), 7 ({3, 10, 28, 44, 56, 64}), 9, 5, 3, 2; 1: in one piece after
 *                the packet's last bit (thip_state_token_lists_begin).  More groups start the device earlier and cost a pair of
 *                launches each
 *   fe_worker    th_decode_*, token-list path: 1: the context has a second thread that undoes the DC prediction (spec 7.8;
 *                decode.c:1392-1500) while th_decode_packetin's caller decodes the tokens of indices 1..63; 0: the caller does it
 *                behind the tokens, while the device still walks the last group of indices (it needs the values last); 2 (default):
 *                1 for frames of more than 32 768 fragments, 0 otherwise.  Round 5, one stream, plain loop: 1080p + 11..13 % with
 *                the thread, 720p 0 (dense) .. - 7 % (typical content) with it -- up to 720p the chain fits under the device's walk
 *   fe_worker_pin   fe_worker on: 1 (default): that thread is kept (pthread_setaffinity_np) on the CPUs that share a last-level
 *                cache with the thread that calls th_decode_packetin (the two hand each other a frame's flags, lists and DC values);
 *                0: left to the scheduler (measured 5-10 % SLOWER than no second thread on a two-socket host)
 *   fe_lookahead   th_decode_*: how many packets a caller may announce ahead of their th_decode_packetin
 *                (th_decode_ctl TH_DECCTL_THIP_PREFETCH_PACKET, include/theoradec_hip.h): each is parsed -- entropy decoder and DC
 *                chain -- by a thread of its own; 8 (default), up to 16; 0: announcements are not taken (read when a context's
 *                first packet is announced)
 *   fe_assign    th_decode_*, announced packets on the token-list path: 1: the parser pairs tokens and fragments
 *                (which token belongs to which fragment, decode.c:1540-1581) as it decodes the tokens and the frame goes to
 *                thip_state_token_lists_begin_assigned -- the device pairs nothing; 0: thip_state_token_lists_begin, the
 *                device walks the lists (k_tok_assign / k_tok_walk); 2 (default): measured per stream -- the time between
 *                adopted frames, 24 frames each way, the better rule for the next 1024 (pairing moves 3.5-4.3 ns a token
 *                from the device's critical path to the parser threads: right when they have room, wrong when they are the bound)
 *   fe_pipeline  th_decode_*, with packets announced ahead: 1 (default since round 6): th_decode_ycbcr_out(N) hands frame N + 1 -- the
 *                oldest announced packet, if its parser is done -- to the device BEFORE it waits for picture N, so that the device never
 *                waits for the caller's thread.  An announcement stays a hint: if the next th_decode_packetin brings another packet the
 *                frame decoded ahead is taken back (thip_state_ring_rewind: reference ring, counters, qi indices as they were; zero-byte
 *                packets in between are fine) and the packet is decoded the ordinary way -- the pictures are the same either way.  What
 *                is left of the price: a failed tile hand-over of frame N, noticed when frame N + 1 is already on the device, is
 *                THIP_EFAULT instead of a frame decoded again.  0: off.  fe_pipelined: (counter) frames handed over ahead;
 *                fe_pipeline_taken_back: (counter) those of them that were taken back
 *   fe_lists_rule   th_decode_*, fe_device_lists = -1: 1 (default): lists on the device or the host's own walk, measured per context
 *                (the time between its th_decode_packetin calls, 16 inter frames each way, the faster for fe_assign_settle frames, and
 *                again); 0: the count of contexts alive decides (rounds 3 and 4).  fe_lists_to_device, fe_lists_to_host: (counters) how
 *                often a context changed sides
 *   fe_lookahead_adopted, fe_lookahead_missed   (counters) announced packets taken over by their th_decode_packetin / parsed for
 *                nothing (a different packet came -- every announcement outstanding is then dropped -- or the parser refused it; a
 *                zero-byte packet, i.e. a dropped frame, leaves the announcements where they are)
 *   fe_assign_settle   fe_assign = 2: adopted frames a stream keeps the rule that measured faster before it measures again (default 1024)
 *   fe_assign_to_device, fe_assign_to_parsers   (counters) fe_assign = 2: how often a stream's rule changed from the parsers pairing to
 *                the device walking (every measurement begins with one such change) and back
 *   fe_levels    th_decode_*: 1: the host's own token walk feeds thip_state_frag_recon_levels; 0 (default): thip_state_frag_recon --
 *                measured equal within 2 % end to end (the walk is bound by the tokens, not by the 64 bytes a block saved)
 *   fe_trace_backend, fe_prof   th_decode_*: record slot calls instead of running them (tests); per-stage host timing
 *   device       th_decode_alloc: -1 the calling thread's current device (default), n that device, -2 round robin
 * Returns THIP_EINVAL for a name the table does not have.
 * ---------------------------------------------------------------------------------- */
int thip_set_option(const char *name, int value);
int thip_get_option(const char *name, int *value);
const char *thip_option_name(int index, const char **help);
/* (internal shorthand of thip_get_option for the library's own translation units: 0 for an unknown name) */
int thip_option(const char *name);
/* (internal: adds to a counter of the table) */
void thip_option_add(const char *name, int delta);

#ifdef __cplusplus
}
#endif
#endif
