/* theoradec_hip.h -- libtheora's decoder API (include/theora/codec.h, theoradec.h) served by
 * the MI355X backend: the same function names, argument meaning and TH_E* return codes, so
 * a program written against libtheoradec relinks against libtheora_hip.so.
 *
 * The structs below restate the PUBLIC layout of the reference's types (they are API
 * facts: codec.h:144-299, theoradec.h:131-215, libogg's ogg_packet); nothing else of the
 * reference is reproduced.  The front end behind these calls -- bit reader, header and
 * frame parsing, Huffman decode, DC un-prediction, token expansion and dequantisation --
 * is written from the Theora specification (doc/spec/spec.tex); the pixel path is the HIP
 * backend of theora_hip.h, driven through the accel-vtable slots exactly where
 * lib/decode.c:2858-2962 drives oc_state_frag_recon / oc_frag_copy_list /
 * oc_state_loop_filter_frag_rows.
 *
 * Not provided (out of scope): the telemetry requests, the legacy theora_* API,
 * th_granule_* helpers beyond th_granule_frame and th_granule_time, the encoder.
 */
#ifndef THEORADEC_HIP_H
#define THEORADEC_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* codec.h:77-93 */
#define TH_EFAULT (-1)
#define TH_EINVAL (-10)
#define TH_EBADHEADER (-20)
#define TH_ENOTFORMAT (-21)
#define TH_EVERSION (-22)
#define TH_EIMPL (-23)
#define TH_EBADPACKET (-24)
#define TH_DUPFRAME (1)

typedef enum { TH_CS_UNSPECIFIED, TH_CS_ITU_REC_470M, TH_CS_ITU_REC_470BG, TH_CS_NSPACES } th_colorspace;
typedef enum { TH_PF_420, TH_PF_RSVD, TH_PF_422, TH_PF_444, TH_PF_NFORMATS } th_pixel_fmt;

typedef struct {
  int width, height, stride;
  unsigned char *data;
} th_img_plane; /* codec.h:144-153 */
typedef th_img_plane th_ycbcr_buffer[3];

typedef struct { /* codec.h:206-299 */
  unsigned char version_major, version_minor, version_subminor;
  uint32_t frame_width, frame_height;
  uint32_t pic_width, pic_height, pic_x, pic_y;
  uint32_t fps_numerator, fps_denominator;
  uint32_t aspect_numerator, aspect_denominator;
  th_colorspace colorspace;
  th_pixel_fmt pixel_fmt;
  int target_bitrate;
  int quality;
  int keyframe_granule_shift;
} th_info;

typedef struct { /* codec.h:326-335 */
  char **user_comments;
  int *comment_lengths;
  int comments;
  char *vendor;
} th_comment;

/* libogg's ogg_packet (ogg/ogg.h): only packet, bytes and b_o_s are read here. */
typedef struct {
  unsigned char *packet;
  long bytes;
  long b_o_s;
  long e_o_s;
  int64_t granulepos;
  int64_t packetno;
} ogg_packet;

typedef struct th_dec_ctx th_dec_ctx;
typedef struct th_setup_info th_setup_info;

void th_info_init(th_info *info);   /* info.c */
void th_info_clear(th_info *info);
void th_comment_init(th_comment *tc);
void th_comment_clear(th_comment *tc);

/* theoradec.h:234-322 */
int th_decode_headerin(th_info *info, th_comment *tc, th_setup_info **setup, ogg_packet *op);
th_dec_ctx *th_decode_alloc(const th_info *info, const th_setup_info *setup);
/* Extension (not in theoradec.h): the same with the GPU chosen -- device 0 .. thip_device_count()-1 (theora_hip.h), or -1 for
   what th_decode_alloc does: the THIP_DEVICE environment variable if set (a device number, or "rr": the
   contexts of the process take the visible devices in turn), else the calling thread's current device.
   A context stays on its GPU (a single stream is not split); contexts of one process may use all GPUs. */
th_dec_ctx *th_decode_alloc_on(const th_info *info, const th_setup_info *setup, int device);
void th_setup_free(th_setup_info *setup);
int th_decode_ctl(th_dec_ctx *dec, int req, void *buf, size_t buf_sz);
int th_decode_packetin(th_dec_ctx *dec, const ogg_packet *op, int64_t *granpos);
int th_decode_ycbcr_out(th_dec_ctx *dec, th_ycbcr_buffer ycbcr);
void th_decode_free(th_dec_ctx *dec);
int64_t th_granule_frame(void *encdec, int64_t granpos);
double th_granule_time(void *encdec, int64_t granpos);                      /* state.c:1259 */

/* codec.h "Basic shared functions" and comment helpers (internal.c:189-210, info.c) */
const char *th_version_string(void);
uint32_t th_version_number(void);                                          /* bitstream 3.2.1 -> 0x030201 */
int th_packet_isheader(ogg_packet *op);
int th_packet_iskeyframe(ogg_packet *op);                                  /* 1 key, 0 delta (or empty), -1 header */
void th_comment_add(th_comment *tc, const char *comment);
void th_comment_add_tag(th_comment *tc, const char *tag, const char *value);
char *th_comment_query(th_comment *tc, const char *tag, int count);        /* value of the count-th TAG=, or NULL */
int th_comment_query_count(th_comment *tc, const char *tag);

/* th_decode_ctl requests that are honoured (theoradec.h:40-105).  TH_DECCTL_GET_PPLEVEL_MAX answers 7 and
   TH_DECCTL_SET_PPLEVEL takes 0..7 as the reference does (decode.c:32-48, :1989-1997): the de-blocking and
   de-ringing filters of decode.c:1608-1957 run on the GPU (thip_state_postprocess) and th_decode_ycbcr_out
   hands out the post-processed picture; like the reference, a level set between two key frames takes
   effect at the next key frame (decode.c:1221-1223). */
#define TH_DECCTL_GET_PPLEVEL_MAX (1)
#define TH_DECCTL_SET_PPLEVEL (3)
#define TH_DECCTL_SET_GRANPOS (5)
/* theoradec.h:79-92,141-151.  The reference calls back after every few fragment rows while the
   frame is still being decoded; a frame is one batch of GPU work here, so the callback is made
   ONCE per decoded frame, after it is complete and copied to the host, with the whole range of
   fragment rows [0, frame_height/8) -- what the reference's telemetry build does
   (decode.c:2974-2977).  The buffer is the one th_decode_ycbcr_out returns. */
#define TH_DECCTL_SET_STRIPE_CB (7)
typedef void (*th_stripe_decoded_func)(void *ctx, th_ycbcr_buffer buf, int yfrag0, int yfrag_end);
typedef struct {
  void *ctx;
  th_stripe_decoded_func stripe_decoded;
} th_stripe_callback;
/* Extension (not in theoradec.h): a context allocated while THIP_FE_TRACE_BACKEND=1 is set owns no
   device state; th_decode_packetin parses the packet completely and RECORDS the accel-vtable slot
   calls of the frame (state_frag_recon per coded fragment in coded order, frag_copy_list, the loop
   filter limit) instead of making them.  This request fills a thip_slot_trace describing the most
   recent frame (pointers stay valid until the next packet).  It exists so that the host logic can be
   tested without a GPU; on a normal context it returns TH_EINVAL. */
#define TH_DECCTL_THIP_GET_SLOT_TRACE (0x7101)
/* Extension: buf = int.  Non-zero: from the next packet on the DC prediction (spec 7.8, decode.c:1392-1500)
   is undone by the backend on the GPU (thip_state_set_device_dc: an anti-diagonal wavefront per plane)
   instead of by th_decode_packetin on the host; the pictures are the same.  TH_EIMPL for planes of more
   than 1024 fragment rows.  The environment variable THIP_FE_DEVICE_DC=1 sets it for every new context. */
#define TH_DECCTL_THIP_SET_DEVICE_DC (0x7102)
/* Extension: buf = int.  Non-zero: th_decode_packetin only delimits each fragment's DCT tokens; their
   expansion into 64 coefficients and the AC dequantisation (decode.c:1540-1581) happen on the GPU
   (thip_state_frag_recon_tokens / k_expand_tokens), and 4 bytes per non-zero coefficient cross PCIe
   instead of 128 per block.  THIP_FE_DEVICE_TOKENS=1 sets it for every new context. */
#define TH_DECCTL_THIP_SET_DEVICE_TOKENS (0x7103)
/* Extension: buf = int.  Non-zero: th_decode_packetin stops behind the entropy decoder -- the frame's token lists
   (one per plane and zig-zag index, decode.c:993-1139), the coded-fragment list and one word per fragment go to the
   GPU, which finds each fragment's tokens, expands and dequantises them and decodes the frame
   (thip_state_decode_token_lists).  The DC prediction is still undone in th_decode_packetin (a chain through the plane in
   raster order: nanoseconds a fragment on a host core) unless TH_DECCTL_THIP_SET_DEVICE_DC asks for the GPU there too.  Frames with a plane of more than 147456 coded fragments (beyond
   4K) keep the host path.  Zero: the host's own token walk.  Without this call a context follows the option
   fe_device_lists (THIP_FE_DEVICE_LISTS in the environment): 1 / 0, or -1, the default: this path while at most four
   decoder contexts are alive in the process. */
#define TH_DECCTL_THIP_SET_DEVICE_LISTS (0x7104)
/* Extension: buf = ogg_packet (a data packet the caller will hand to th_decode_packetin LATER, announced in decode order).
   What th_decode_packetin reads out of a packet -- coded flags, modes, vectors, qi indices, DCT tokens, and the DC prediction it
   undoes (decode.c:2790-2823 up to the MCU loop) -- depends on the headers and on nothing an earlier frame left behind: only the
   pictures form a chain.  The packet is copied and parsed on a thread of its own (up to option fe_lookahead, default 8, at a
   time); the th_decode_packetin that gets the same bytes adopts the result and does only the hand-over to the GPU, so one
   stream is no longer bound by one host thread.  Returns 0: announced; 1: not taken (no slot free, an empty packet, a context with
   TH_DECCTL_THIP_SET_DEVICE_DC / _DEVICE_TOKENS on, a process confined to one CPU) -- harmless, the packet is parsed in its
   th_decode_packetin as ever.  A th_decode_packetin whose packet is not the oldest announced one drops everything announced and
   parses the ordinary way: announcing is a hint, never a requirement, and the pictures are the same either way.
   (That holds under option fe_pipeline too, include/theora_hip.h -- there th_decode_ycbcr_out hands the oldest announced packet's
   frame to the device before it waits for its own picture: a th_decode_packetin that brings another packet takes that frame back.) */
#define TH_DECCTL_THIP_PREFETCH_PACKET (0x7105)
/* Which GPU the context's device state lives on (th_decode_alloc_on, option "device" / THIP_DEVICE): buf = int, receives the
   device index (thip_state_device); TH_EINVAL for a context without device state (slot-trace mode). */
#define TH_DECCTL_THIP_GET_DEVICE (0x7106)
typedef struct thip_slot_trace {
  int64_t ncoded;           /* state_frag_recon calls, in call (= coded) order */
  const int32_t *fragi;     /* _fragi */
  const uint8_t *pli;       /* _pli */
  const uint8_t *last_zzi;  /* _last_zzi */
  const uint8_t *refi;      /* frags[_fragi].refi */
  const uint16_t *dc_quant; /* _dc_quant */
  const int16_t *mv;        /* frag_mvs[_fragi], OC_MV packing */
  const int16_t *coeffs;    /* _dct_coeffs[0..63] of each call: natural order, AC dequantised, raw DC */
  int64_t nuncoded;         /* fragments handed to frag_copy_list */
  const int64_t *uncoded;
  int32_t flimit;           /* loop_filter_limits[qis[0]] */
  int32_t frame_type;
} thip_slot_trace;

#ifdef __cplusplus
}
#endif
#endif
