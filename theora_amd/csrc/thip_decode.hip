// thip_decode.hip -- frame-scope reconstruction path for gfx950 (MI355X): host side and C ABI.
// The kernels themselves are in thip_kernels.h.
//
// Two kernels per batch of frames (one 8x8 block per lane, wave64):
//
//   k_recon       K1+K2.  One wave (= one workgroup) per TILE (4 super blocks = 16x4 fragments
//                 = 128x32 pixels), lanes in coded order (super block by super block, Hilbert
//                 inside).  Round trip 1: the 64 command words + the tile's first slot number.
//                 Round trip 2: coefficients (slot found by ballot / prefix count over the
//                 tile's mask) and predictor windows, all issued before anything waits;
//                 when few lanes of a tile own coefficients, 4 or 2 lanes share a block's
//                 inverse DCT.  Then dequantised coefficients -> iDCT -> predictor
//                 -> pixels (state.c:959, idct.c:301, fragment.c:49-80); an uncoded fragment
//                 (fragment.c:37) is the zero-vector predictor plus a zero residual; DC-only
//                 fragments carry their value in the command word and own no slot.
//   k_loopfilter  K3.  Whole-frame in-loop deblocking (state.c:1055-1105), one 8x8 "corner
//                 cell" per lane; cells are independent, so the pass is fully parallel and a
//                 cell is one memory round trip.
//
// What makes these kernels fast or slow on this chip is not arithmetic but (a) how many
// DEPENDENT memory round trips a wave makes (kernel arguments indexed by a value the compiler
// cannot prove uniform become vector loads; a register copy behind a load is a hidden wait),
// and (b) straggler waves.  tools/wave_trace.py shows both; DESIGN.md section 5 has the story.
//
// K4 (UMV border fill, state.c:770-835) does not exist: device frames are unpadded and
// motion-compensated reads clamp their coordinates, which is bit-identical.
//
// Host side (C ABI, include/theora_hip.h): stream state = three device frames + ring,
// batched launch over up to THIP_MAX_BATCH independent streams per kernel, pinned staging
// for the one-fragment-at-a-time vtable slots.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <atomic>
#include <mutex>
#include <vector>

#include "../../include/theora_hip.h"
#include "thip_device.h"

using namespace thip;

#define HIP_TRY(expr)                                                                    \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess) {                                                              \
      fprintf(stderr, "theora_hip: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), \
              __FILE__, __LINE__);                                                       \
      return THIP_EFAULT;                                                                \
    }                                                                                    \
  } while (0)

// ---------------------------------------------------------------------------------------
// run-time options: one table, one parser (include/theora_hip.h: thip_set_option)
// ---------------------------------------------------------------------------------------
// a spinning thread's pause: the x86 hint where there is one (the unguarded builtin kept other hosts from compiling)
static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#elif defined(__aarch64__)
  asm volatile("yield" ::: "memory");
#else
  std::this_thread::yield();
#endif
}
namespace {
struct Option {
  const char *name;
  std::atomic<int> value;   // written by thip_set_option, read per frame by the decoding threads (relaxed: a change takes effect "with the next call")
  const char *help;
};
Option g_options[] = {
    {"fuse", 3, "3: k_recon_lf, reconstruction + loop filter in one pass (default); 0: the two passes k_recon + k_loopfilter"},
    {"lanes", 2, "library-owned HIP streams per device for thip_decode_frames (1..4); read when a device is first used"},
    {"ctx_lanes", 8, "HIP streams shared by the enqueue-fed states (th_decode_* contexts), 0 = use the lanes; read at first use"},
    {"chunk", THIP_MAX_BATCH, "streams per kernel launch (1..THIP_MAX_BATCH)"},
    {"skip_static", 1, "leave static blocks where they are: 0 never, 1 when most of the frame is uncoded, 2 whenever possible (tests)"},
    {"lf_sparse", -1, "k_loopfilter reads the coded flags first: -1 by coded fraction, 0 never, 1 always"},
    {"zerocopy", 1, "enqueue path: kernels read the pinned staging buffers across PCIe (0: hipMemcpyAsync into device copies)"},
    {"wait_spin", 0, "wait for a frame with hipEventSynchronize instead of polling with 20 us sleeps"},
    {"dc_global", 0, "DC un-prediction with the work-group kernel that goes through memory (planes beyond k_dc_wave's LDS take it anyway)"},
    {"debug", 0, "k_recon ablation switches for profiling"},
    {"fe_device_dc", 0, "th_decode_*: DC un-prediction on the device"},
    {"fe_device_tokens", 0, "th_decode_*: token expansion + dequantisation on the device (host-delimited tokens)"},
    {"fe_device_lists", -1, "th_decode_*: the token lists themselves on the device (1 on, 0 off, -1 on while at most four decoder contexts are alive)"},
    {"tl_dc_copy", 0, "token lists on the device, thip_state_token_lists_finish: 0 (default): a kernel copies the caller's DC values out of the pinned staging buffer (k_tok_copy); 1: a copy engine does (rounds 3-4)"},
    {"spec_coeffs", 1, "k_recon_lf, levels form: 1 (default): for a frame with a coefficient unit for every block (nslots == its fragments) the waves of whole tile rows ask for their tile's units when they start, at the address that follows from the tile's place in its plane, and check it against the first-slot word; 0: always after the command words"},
    {"tl_algo", 0, "token lists on the device: 1: k_tok_assign (rank -> fragment map in LDS, or in memory for planes beyond 36 864 coded fragments); 2: k_tok_rank + k_tok_walk (every fragment looked after by one thread, one barrier per index, one byte of LDS per fragment); 0 (default): 2 where 1 would keep its map in memory (4K), 1 otherwise"},
    {"tl_walk_threads", 0, "k_tok_walk: threads of the work group (256, 512, 1024); 0 (default): by plane size"},
    {"tl_levels", 1, "token lists on the device (thip_state_token_lists_*): 1 (default): the device writes the coefficient slots in the levels form (int8 units, the reconstruction kernel dequantises); 0: dequantised int16 slots"},
    {"fe_groups", 6, "th_decode_*, token-list path: the groups of zig-zag indices a frame's lists are handed over in while the packet is still being decoded: 6 (default since round 6: {3, 10, 28, 48, 64} -- what the device walks behind the packet's last bit is sixteen indices instead of thirty-six), 4 ({3, 10, 28, 64}, rounds 4-5), 7 ({3, 10, 28, 44, 56, 64}), 9, 5, 3, 2, or 1: in one piece after the packet's last bit"},
    {"fe_worker", 2, "th_decode_*, token-list path: 1: a second thread per context undoes the DC prediction while the caller decodes the tokens of indices 1..63; 0: the caller does it behind the tokens, while the device walks the last indices; 2 (default): 1 for frames of more than 32 768 fragments (beyond 720p), 0 otherwise"},
    {"fe_worker_pin", 1, "th_decode_*, fe_worker on: 1 (default): the second thread is kept on the CPUs that share a last-level cache with the caller's; 0: left to the scheduler"},
    {"fe_lookahead", 8, "th_decode_*: packets a caller may announce ahead of their th_decode_packetin (TH_DECCTL_THIP_PREFETCH_PACKET), each parsed by a thread of its own on a parser context: 8 (default), up to 16; 0: announcements are not taken"},
    {"fe_assign", 2, "th_decode_*, announced packets on the token-list path: 1: the parser pairs tokens and fragments while it decodes the tokens and the frame goes to thip_state_token_lists_begin_assigned (k_tok_scatter: the device pairs nothing); 0: the device walks the lists (thip_state_token_lists_begin); 2 (default): whichever measures faster for this stream (24 frames each way, the better for 1024, and again)"},
    {"fe_pipeline", 1, "th_decode_*: 1 (default since round 6): th_decode_ycbcr_out hands the next ANNOUNCED packet's frame to the device before it waits for its own picture; if another packet than the announced one comes next, the frame decoded ahead is taken back (reference ring and counters restored, fe_pipeline_taken_back counts it); a failed tile hand-over of a frame behind which the next one is already on the device is THIP_EFAULT instead of a frame decoded again; 0: off"},
    {"fe_pipelined", 0, "(counter) frames handed to the device ahead of their th_decode_packetin (fe_pipeline)"},
    {"fe_pipeline_taken_back", 0, "(counter) frames decoded ahead that were taken back because another packet came (fe_pipeline)"},
    {"fe_lists_rule", 1, "th_decode_*, fe_device_lists = -1: 1 (default): whether a context's token lists go to the device or the host walks them is measured per context (the time between its th_decode_packetin calls, 16 inter frames each way, the faster for fe_assign_settle frames); 0: by the count of contexts alive (at most four: the device)"},
    {"fe_lists_to_device", 0, "(counter) fe_lists_rule: times a context went from the host's walk to the lists on the device"},
    {"fe_lists_to_host", 0, "(counter) fe_lists_rule: times a context went from the lists on the device to the host's walk"},
    {"fe_lookahead_adopted", 0, "(counter) announced packets whose th_decode_packetin found them parsed and took the frame over"},
    {"fe_lookahead_missed", 0, "(counter) announced packets that were parsed for nothing: another packet came instead (everything announced is then dropped), or the parser refused the packet"},
    {"fe_assign_settle", 1024, "th_decode_*, fe_assign = 2: adopted frames a stream keeps the rule that measured faster before it measures again (default 1024; tests shorten it)"},
    {"fe_assign_to_device", 0, "(counter) fe_assign = 2: times a stream's rule went from the parsers pairing to the device walking (every measurement begins with one)"},
    {"fe_assign_to_parsers", 0, "(counter) fe_assign = 2: times it went back to the parsers pairing (they measured faster, or a new measurement began)"},
    {"fe_levels", 0, "th_decode_*: 1: the host's own token walk hands the slots quantised levels (thip_state_frag_recon_levels: the kernel dequantises); 0 (default): dequantised coefficients"},
    {"fe_trace_backend", 0, "th_decode_*: record the slot calls instead of running them (tests)"},
    {"fe_prof", 0, "th_decode_*: per-stage host timing"},
    {"device", -1, "th_decode_alloc: -1 the current device, n that device, -2 round robin over the node's devices (THIP_DEVICE=rr)"},
    {"half_tiles", 0, "k_recon_lf_h (two super blocks per wave, two lanes per block) for launches of at least sb_tiles and fewer tiles than this (0: never)"},
    {"sb_tiles", 600, "k_recon_lf_sb (one super block per wave, four lanes per block) instead of k_recon_lf for launches of fewer tiles than this (0: never)"},
    {"enc_halfpel_lanes", 2, "thip_enc_frag_metric_halfpel_batch: 2 (default): a lane per side (dx = -1 / +1), four sites each; 3: a lane per dx (two or three sites)"},
    {"enc_sites_lds", 1, "thip_enc_frag_metric_sites_batch, SATD: 1 (default, round 6): k_enc_sites_satd, the source block shared by a block's three lanes through LDS; 0: k_enc_sites<SATD> (rounds 4-5)"},
    {"enc_fdct_lanes", 4, "thip_enc_fdct8x8_batch: 4 (default): four lanes per block (k_enc_fdct4); 1: one block per lane (k_enc_fdct, rounds 1-5)"},
    {"enc_fq_lanes", 4, "thip_enc_fdct_quantize_batch: 4 (default): four lanes per block (k_enc_fdct_quantize4); 1: one block per lane (round 4's kernel)"},
    {"redo_descs", 0, "thip_decode_frames on the caller's descriptors: 1: the caller promises that the buffers a descriptor points to stay as they are until the state's next synchronising call, so a frame whose hand-over failed can be decoded again like a th_decode_* frame; 0 (default): such a frame gets THIP_EFAULT"},
    {"faults_recovered", 0, "(counter) frames decoded a second time with the two passes because a bounded wait of k_recon_lf had run out"},
};
constexpr int kNumOptions = (int)(sizeof(g_options) / sizeof(g_options[0]));
std::once_flag g_options_once;
// The environment is read ONCE, here: THIP_<NAME> (upper case) = integer; THIP_DEVICE also takes "rr".
void options_from_env() {
  for (int i = 0; i < kNumOptions; i++) {
    char env[64] = "THIP_";
    size_t k = 5;
    for (const char *c = g_options[i].name; *c && k + 1 < sizeof(env); c++) env[k++] = (char)(*c >= 'a' && *c <= 'z' ? *c - 32 : *c);
    env[k] = 0;
    const char *v = getenv(env);
    if (!v || !*v) continue;
    if (!strcmp(g_options[i].name, "device") && !strcmp(v, "rr")) g_options[i].value.store(-2, std::memory_order_relaxed);
    else g_options[i].value.store(atoi(v), std::memory_order_relaxed);
  }
}
Option *find_option(const char *name) {
  std::call_once(g_options_once, options_from_env);
  if (!name) return nullptr;
  for (int i = 0; i < kNumOptions; i++)
    if (!strcmp(g_options[i].name, name)) return &g_options[i];
  return nullptr;
}
}  // namespace
// (internal accessor by name, for thip_frontend.cpp's per-context lookups)
extern "C" int thip_option(const char *name) {
  const Option *o = find_option(name);
  return o ? o->value.load(std::memory_order_relaxed) : 0;
}
// ... and for this translation unit's per-frame reads: the entry is looked up once per call site, a read is one relaxed load
#define THIP_OPT(name) ([]() -> int { static const Option *const o_ = find_option(name); return o_ ? o_->value.load(std::memory_order_relaxed) : 0; }())
// (internal, for the counters thip_frontend.cpp keeps in the table)
extern "C" void thip_option_add(const char *name, int delta) {
  Option *o = find_option(name);
  if (o) o->value.fetch_add(delta, std::memory_order_relaxed);
}
extern "C" int thip_set_option(const char *name, int value) {
  Option *o = find_option(name);
  if (!o) return THIP_EINVAL;
  o->value.store(value, std::memory_order_relaxed);
  return THIP_OK;
}
extern "C" int thip_get_option(const char *name, int *value) {
  const Option *o = find_option(name);
  if (!o || !value) return o ? THIP_EFAULT : THIP_EINVAL;
  *value = o->value.load(std::memory_order_relaxed);
  return THIP_OK;
}
extern "C" const char *thip_option_name(int index, const char **help) {
  if (index < 0 || index >= kNumOptions) return nullptr;
  if (help) *help = g_options[index].help;
  return g_options[index].name;
}

#include "thip_kernels.h"
#include "thip_fused.h"
#include "thip_fused_sb.h"
#include "thip_dc.h"
#include "thip_postproc.h"
#include "thip_tokens.h"

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
struct thip_state {
  int device;             // HIP device the frames, the staging buffers and the launches of this state live on
  int frame_width, frame_height, pixel_fmt, hdec, vdec;
  thip_plane_geom geom[3];
  thip_tile_geom tiles;
  int64_t nfrags;
  size_t frame_bytes;
  uint8_t *frames[3];   // device
  uint8_t *coded_map;   // device, 2 x nfrags bytes: the coded flags of the current and of the previous frame
  // Which decoded frame (counted by frame_serial) each buffer holds, -1 if unknown, and which frame's
  // flags sit in each half of coded_map: what makes leaving static blocks in place safe (launch_chunk).
  int64_t buf_serial[3], map_serial[2];
  int ref_idx[3];       // THIP_FRAME_* -> buffer index
  int last_decoded;     // buffer index of the most recently completed frame, -1 if none
  int lane;             // library-owned HIP stream this state is bound to, -1 until first use
  int ctx_lane;         // ... when it is fed through the enqueue slots: its context stream, -1 until first use
  int32_t *frag_pos;    // host: raster fragment index -> tile*256+lane
  // host-enqueue staging (allocated on first use)
  int staging_ready;
  uint32_t *h_info, *d_info;
  int16_t *h_coeffs, *d_coeffs;
  uint32_t *h_slot0, *d_slot0;
  hipStream_t ctx_stream_cached;   // context_stream()'s answer, once it has given one
  hipEvent_t ev_staging;     // recorded behind the kernels that read the staging buffers (enqueue path)
  int64_t staging_serial;    // frame_serial of the frame ev_staging stands behind
  int64_t out_done_serial;   // frame_serial of the newest frame whose output copy a host wait has seen complete (ev_out)
  hipEvent_t ev_order;       // orders a frame behind the previous one when the two go down different streams
  hipStream_t last_stream;   // stream of the most recent launch for this state (ycbcr_out copies on it)
  int order_recorded;        // ev_order already marks the end of that launch (it went down a caller-owned stream)
  // Output to the host: k_frame_out writes the finished frame, top row first and tightly packed,
  // straight into one of two pinned images (the kernel's stores cross PCIe; no DMA call, no flip
  // on the host).  frame_serial counts finished frames, out_serial is the frame h_out[out_cur] holds.
  uint8_t *h_out[2];
  int out_cur, eager_out;
  int64_t frame_serial, out_serial;
  hipEvent_t ev_out2[2];     // recorded behind k_frame_out, one per host image
  int held_img;              // thip_state_ycbcr_map_begin: the image (and its event) thip_state_ycbcr_map_end hands out, -1 none
  int64_t held_serial;       // ... and the frame_serial of the frame that image holds (its output copy done = that frame is done)
  int32_t *enq_last_lane;   // per tile: last lane that received a slot (arrival-order check)
  int enq_ncoded, enq_nuncoded, enq_nslots, enq_frame_type, enq_flimit, enq_active, enq_last_tile;
  int enq_lf_y0[3], enq_lf_y1[3], enq_lf_any;
  int lf_y0[3], lf_y1[3], lf_rows_custom;
  // DC un-prediction on the device (thip_frame_desc.dc_tokens / thip_state_set_device_dc)
  int16_t *d_dc;        // device, nfrags: un-predicted DC values of the frame being decoded
  uint4 *d_dc_ent;      // device, nfrags: k_dc_prepare's per-fragment entries
  uint8_t *d_dc_rowhas; // device, one byte per fragment row of every plane
  uint32_t *fault;      // pinned host words (16): [0] set to 1 by k_recon_lf / k_recon_lf_sb when a bounded wait ran out, [1 + id % 8] = the
                        // id of the launch that did (its edge serial number, | 0x1000 for the super-block kernel), [9] set to 2 by k_pp_dering
  // the state's most recent frame, kept so that it can be decoded again with the two passes if its hand-over failed
  // (recover_fault); only frames whose command stream lives in the state's own staging buffers (enqueue slots, token lists)
  struct {
    int valid;
    thip_frame_desc d;
    int ring[3];            // ref_idx before the frame
    int lf_custom, lf_y0[3], lf_y1[3];
    int flush_flags;
    int64_t serial;         // frame_serial after the frame
    uint32_t launch_id;     // what its fused launch writes behind the fault flag (0: the frame took the two passes)
  } redo;
  int redo_owned;       // set by the callers whose descriptors point into the state's own buffers, around their thip_decode_frames call
  uint8_t *d_edge;      // device, k_recon_lf: kTfRec bytes per tile (the tiles' edges for their neighbours)
  uint32_t edge_epoch;  // serial number of the last k_recon_lf launch for this state (0 = never: the records are zero); 12 bits
  uint8_t *d_edge_h;    // ... k_recon_lf_h's (a record per half tile; 12-bit serial numbers of its own like the other two)
  uint32_t edge_epoch_h;
  uint8_t *d_edge_sb;   // ... and k_recon_lf_sb's: a record per super block (a buffer and a serial number of its own: a tag vouches for
  uint32_t edge_epoch_sb;   //  the bytes in front of it only while every launch that uses the buffer rewrites every unit of it)
  int device_dc, enq_device_dc;
  int16_t *h_dc, *d_dc_in;   // enqueue path: token DC values staged per fragment (pinned) and their device copy
  uint8_t *h_flags, *d_flags;   // ... and the fragments' coded | refi << 1 in fragment-index order (the staged command
                                // words stay in host memory: the wavefront kernel must not poll them across PCIe)
  int flush_flags;              // set while thip_frame_flush runs: d_flags describes the frame being launched
  // token form of the slot (thip_state_frag_recon_tokens): tokens, per-slot {first token, count | table << 8}, tables
  uint32_t *h_tok, *d_tok;
  uint32_t *h_slot_tok, *d_slot_tok;
  uint16_t *h_dq, *d_dq;
  uint16_t *h_dqp, *d_dqp;   // the same 18 tables in slot order (thip_pack_dequant_table): the levels form's thip_frame_desc.dequant
  size_t tok_cap;
  int tok_ready;             // all six token staging buffers exist
  int enq_ntok, enq_tok_slots, enq_dense_slots;
  int enq_levels_hint;       // a DC-only block came through thip_state_frag_recon_levels: a frame without any unit is in the levels form too
  int enq_level_slots;       // blocks that came through thip_state_frag_recon_levels (the frame is then in the levels form)
  int enq_tile_blocks;       // ... of them in the tile being filled (enq_last_tile)
  // token lists expanded on the device (thip_state_decode_token_lists): pinned staging, its device copy, work arrays
  uint32_t *h_tl, *d_tl;     // (h_tl: the staging buffer of the frame being handed over, one of h_tl_buf)
  // Two pinned staging buffers by turns: the host may write frame N + 1's lists while frame N's kernels still read theirs (the
  // th_decode_* front end with option fe_pipeline hands frame N + 1 over before it has waited for frame N).  The device copy d_tl
  // and the work arrays exist once: what uses them runs on one stream, in order.
  uint32_t *h_tl_buf[2];
  int tl_buf;                // which of the two h_tl is
  int64_t tl_buf_serial[2];  // frame_serial of the frame that last used each (its kernels have read it once that frame is done)
  size_t tl_cap;            // bytes of each
  int tl_ready;             // all of the buffers below exist and d_frag_pos is filled
  int16_t *d_tl_tmp;        // [nfrags][64]
  uint8_t *d_tl_last;       // [nfrags]
  uint32_t *d_tl_slot;      // [nfrags]
  uint32_t *d_tl_arr;       // [nfrags] k_tok_assign's rank -> fragment map for planes beyond its LDS
  // a frame handed over with thip_state_token_lists_begin and not yet finished
  int tl_pending;
  TlK tl_K;
  thip_frame_desc tl_desc;
  int64_t tl_ncoded;
  size_t tl_o_dcv;          // where the caller's DC values go in h_tl / d_tl (dwords)
  size_t tl_o_tok;          // where the tokens start
  int tl_z;                 // indices handed over so far (thip_state_token_lists_append)
  int64_t tl_ntok;          // tokens staged so far
  uint8_t *d_tl_pos;        // k_tok_assign's fragment positions between the launches of a frame
  int tl_claimed;           // thip_state_token_lists_staging has handed the staging buffer out for the frame to come
  uint32_t *d_tl_wide;      // [ntiles] levels form: the tiles whose levels do not all fit eight bits
  uint32_t *d_tl_rank;      // [64 nfrags] k_tok_rank's output (option tl_algo = 2), allocated with its first use
  int64_t tl_rank_at;       // entries of it handed out to the lists of the frame so far
  hipStream_t tl_stream;
  int32_t *d_frag_pos;      // [nfrags], uploaded once
  // out-of-loop post-processing (thip_state_postprocess): the post-processed picture, the per-fragment
  // variances and quantiser indices, which planes of which decoded frame the picture holds
  uint8_t *pp_frame;
  int *pp_var;
  uint8_t *pp_qis;          // device, 2 * nfrags: dc_qis, then frag_qi
  uint8_t *h_pp_qis;        // pinned staging of the same
  uint32_t *pp_done;        // de-ringing: per group of blocks, the launch that finished it (k_pp_dering)
  uint32_t pp_run;          // serial number of the last de-ringing launch
  hipEvent_t ev_pp;         // recorded behind the copy out of h_pp_qis
  int64_t pp_serial;        // frame_serial of the frame pp_frame was made from, -1 if none
  int pp_active[3];
};

// 0: the state's fault word is clear; 1: a failed hand-over was noticed and the frame has been decoded again with the two
// passes; THIP_EFAULT: a bounded wait ran out and the frame could not be decoded again (reported once).  The state's device
// must be current.  (Defined behind launch_chunk.)
static int check_fault(thip_state *st);

namespace {
std::mutex g_mu;
// Library-owned HIP streams ("lanes").  Every thip_state is bound to one lane for life, so
// the frames of a stream stay ordered; different lanes let one group's loop filter overlap
// another group's reconstruction (dependent kernels of one group cannot overlap).
constexpr int kMaxLanes = 4;
constexpr int kMaxDevices = 16;
hipStream_t g_lanes[kMaxDevices][kMaxLanes];   // per device, created on first use of that device
int g_lanes_ready[kMaxDevices];
int g_nlanes = 0;                              // lanes per device (THIP_LANES, default 2)
int g_next_lane[kMaxDevices];
// Streams for the states that are fed through the enqueue slots (one th_decode_* context each): their GPU work is a
// chain of small dependent launches per frame (longer with post-processing or the device-side front-end stages), so
// contexts on different host threads want chains of their own rather than two lanes between them.  THIP_CTX_LANES
// (default 8, 0 = the lanes above).
constexpr int kCtxLanes = 16;
hipStream_t g_ctx_lanes[kMaxDevices][kCtxLanes];
int g_ctx_ready[kMaxDevices], g_next_ctx[kMaxDevices];
// thip_synchronize waits for the streams that got work since it last did: ten hipStreamSynchronize calls on idle streams are
// tens of microseconds of host time, which a caller that brackets 0.8 ms of work with it (bench.py's blocks) would book as GPU time.
std::atomic<uint8_t> g_lane_dirty[kMaxDevices][kMaxLanes], g_ctx_dirty[kMaxDevices][kCtxLanes];

// Makes `device` current for the calling host thread for the lifetime of the object (HIP's current
// device is per thread) and puts the previous one back: a state may live on any GPU of the node
// whatever the caller's current device is.
struct DeviceGuard {
  int prev = -1, want;
  explicit DeviceGuard(int device) : want(device) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != want) (void)hipSetDevice(want);
  }
  ~DeviceGuard() {
    if (prev >= 0 && prev != want) (void)hipSetDevice(prev);
  }
};
int g_profile = 0;
// Every state has a pinned host word that kernels set when a bounded wait ran out (k_recon_lf's hand-over: bit 0; the
// de-ringing's: bit 1).  The calls that synchronise with a state look at it (check_fault, below launch_chunk): a frame whose
// hand-over failed is decoded again with the two passes where that is possible, else the call returns THIP_EFAULT once.
std::mutex g_states_mu;
std::vector<thip_state *> g_states;   // every live state (thip_synchronize looks at all of them)
struct EvPair { hipEvent_t a, b; int kernel; };
std::vector<EvPair> g_events;
std::vector<hipEvent_t> g_pool;

std::mutex g_lanes_mu;
int ensure_lanes(int device) {   // the device must be current; callable from any host thread
  std::lock_guard<std::mutex> lk(g_lanes_mu);
  if (device < 0 || device >= kMaxDevices) return THIP_EINVAL;
  if (g_lanes_ready[device]) return 0;
  int n = THIP_OPT("lanes");
  if (n < 1) n = 1;
  if (n > kMaxLanes) n = kMaxLanes;
  for (int i = 0; i < n; i++) HIP_TRY(hipStreamCreateWithFlags(&g_lanes[device][i], hipStreamNonBlocking));
  g_nlanes = n;
  g_lanes_ready[device] = 1;
  return 0;
}

std::mutex g_prof_mu;   // guards g_events / g_pool (touched from launch paths only while profiling)
hipEvent_t take_event() {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (!g_pool.empty()) {
    hipEvent_t e = g_pool.back();
    g_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

struct ScopedTimer {
  hipStream_t s;
  EvPair p;
  bool on;
  ScopedTimer(hipStream_t s_, int kernel) : s(s_), on(g_profile != 0) {
    if (on) {
      p.a = take_event();
      p.b = take_event();
      p.kernel = kernel;
      on = p.a && p.b;
      if (on) (void)hipEventRecord(p.a, s);
    }
  }
  ~ScopedTimer() {
    if (on) {
      (void)hipEventRecord(p.b, s);
      std::lock_guard<std::mutex> lk(g_prof_mu);
      g_events.push_back(p);
    }
  }
};

// The stream of an enqueue-fed state (device current).
int context_stream(thip_state *st, hipStream_t *out) {
  if (st->ctx_stream_cached) {   // (a context keeps its stream: no locks on the per-frame path)
    *out = st->ctx_stream_cached;
    if (st->ctx_lane >= 0) g_ctx_dirty[st->device][st->ctx_lane].store(1, std::memory_order_relaxed);
    return THIP_OK;
  }
  static const int nctx = [] {   // (streams are created once: read at first use)
    const int v = THIP_OPT("ctx_lanes");
    return v < 0 ? 0 : (v > kCtxLanes ? kCtxLanes : v);
  }();
  int rc = ensure_lanes(st->device);
  if (rc) return rc;
  std::lock_guard<std::mutex> lk(g_mu);
  if (nctx == 0) {
    if (st->lane < 0) st->lane = g_next_lane[st->device]++ % g_nlanes;
    *out = g_lanes[st->device][st->lane];
    g_lane_dirty[st->device][st->lane].store(1, std::memory_order_relaxed);
    return THIP_OK;
  }
  if (!g_ctx_ready[st->device]) {
    for (int i = 0; i < nctx; i++) HIP_TRY(hipStreamCreateWithFlags(&g_ctx_lanes[st->device][i], hipStreamNonBlocking));
    g_ctx_ready[st->device] = 1;
  }
  if (st->ctx_lane < 0) st->ctx_lane = g_next_ctx[st->device]++ % nctx;
  *out = g_ctx_lanes[st->device][st->ctx_lane];
  st->ctx_stream_cached = *out;
  g_ctx_dirty[st->device][st->ctx_lane].store(1, std::memory_order_relaxed);
  return THIP_OK;
}

// Is `s` one of the library's own streams of this device (never destroyed)?  (lanes / context streams)
bool is_library_stream(int device, hipStream_t s) {
  if (device < 0 || device >= kMaxDevices) return false;
  for (int i = 0; i < kMaxLanes; i++)
    if (g_lanes[device][i] == s && s) return true;
  for (int i = 0; i < kCtxLanes; i++)
    if (g_ctx_lanes[device][i] == s && s) return true;
  return false;
}
// A state whose previous frame went down another stream (frame calls and enqueue calls mixed, or a caller-owned stream):
// everything queued on `s` from here on waits for that frame.  Called before the FIRST enqueue of a frame on `s`.  When the
// previous stream was the caller's, the event was recorded right behind that frame's launch (order_mark): the stream may be
// gone by now.
int order_behind_previous(thip_state *st, hipStream_t s) {
  if (!st->last_stream || st->last_stream == s) return THIP_OK;
  if (!st->ev_order) HIP_TRY(hipEventCreateWithFlags(&st->ev_order, hipEventDisableTiming));
  if (!st->order_recorded) HIP_TRY(hipEventRecord(st->ev_order, st->last_stream));
  HIP_TRY(hipStreamWaitEvent(s, st->ev_order, 0));
  return THIP_OK;
}
// A live stream for work that follows the state's last frame (output copy, post-processing): the stream that frame went
// down if it is the library's, else the state's context stream, ordered behind the frame through ev_order.
int followup_stream(thip_state *st, hipStream_t *out);

// ... and behind a frame's launch on `s`: a caller-owned stream gets its event now (a library stream lives as long as the library).
static void mark_dirty(int device, hipStream_t s) {   // (thip_synchronize waits for the library streams that got work)
  if (!s || device < 0 || device >= kMaxDevices) return;
  for (int i = 0; i < kMaxLanes; i++)
    if (g_lanes[device][i] == s) g_lane_dirty[device][i].store(1, std::memory_order_relaxed);
  for (int i = 0; i < kCtxLanes; i++)
    if (g_ctx_lanes[device][i] == s) g_ctx_dirty[device][i].store(1, std::memory_order_relaxed);
}
int order_mark(thip_state *st, hipStream_t s) {
  st->last_stream = s;
  st->order_recorded = 0;
  mark_dirty(st->device, s);
  if (!is_library_stream(st->device, s)) {
    if (!st->ev_order) HIP_TRY(hipEventCreateWithFlags(&st->ev_order, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(st->ev_order, s));
    st->order_recorded = 1;
  }
  return THIP_OK;
}

int followup_stream(thip_state *st, hipStream_t *out) {
  if (st->last_stream && is_library_stream(st->device, st->last_stream)) {
    *out = st->last_stream;
    mark_dirty(st->device, st->last_stream);
    return THIP_OK;
  }
  hipStream_t s;
  int rc = context_stream(st, &s);
  if (rc) return rc;
  rc = order_behind_previous(st, s);
  if (rc) return rc;
  *out = s;
  return THIP_OK;
}

// k_recon_lf hands tile edges from work group to work group through the L2 of ONE XCD, which is only right if
// work group i of a launch runs on XCD i mod 8 (how the eight XCDs of this chip share a dispatch).  Checked once
// per device with a probe launch; on any other answer the fused kernel is not used.
__global__ void k_xcc_probe(uint32_t *out) {
  uint32_t xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if (threadIdx.x == 0) out[blockIdx.x] = xcc & 15u;
}
bool xcd_round_robin(int device) {   // the device must be current
  static std::mutex mu;
  static int known[kMaxDevices];     // 0 = not probed, 1 = round robin over 8, -1 = anything else
  std::lock_guard<std::mutex> lk(mu);
  if (device < 0 || device >= kMaxDevices) return false;
  if (!known[device]) {
    known[device] = -1;
    constexpr int kN = 2048;
    uint32_t *d = nullptr;
    std::vector<uint32_t> h(kN, 0xFFu);
    if (hipMalloc((void **)&d, kN * sizeof(uint32_t)) == hipSuccess) {
      hipLaunchKernelGGL(k_xcc_probe, dim3(kN), dim3(64), 0, 0, d);
      if (hipMemcpy(h.data(), d, kN * sizeof(uint32_t), hipMemcpyDeviceToHost) == hipSuccess) {
        bool ok = true;
        for (int i = 0; i < kN && ok; i++) ok = h[i] == h[i & 7];
        for (int i = 0; i < 8 && ok; i++)
          for (int j = 0; j < i; j++) ok = ok && h[i] != h[j];
        known[device] = ok ? 1 : -1;
      }
      (void)hipFree(d);
    }
  }
  return known[device] == 1;
}

// k_recon_lf: the stream's tile rows dealt to 8 bands of (nearly) equal tile counts; returns the longest band.
int fill_bands(StreamK &K, const thip_state *st) {
  const int total = K.tile_end[2];
  int row_start[3 * 4096 + 1], nrows = 0, u = 0;
  for (int pli = 0; pli < 3; pli++)
    for (int y = 0; y < st->tiles.tiles_y[pli] && nrows < 3 * 4096; y++) {
      row_start[nrows++] = u;
      u += st->tiles.tiles_x[pli];
    }
  row_start[nrows] = total;
  K.band_u0[0] = 0;
  K.band_u0[8] = total;
  int r = 0;
  for (int b = 1; b < 8; b++) {
    const long long target = (long long)b * total / 8;
    while (r < nrows && row_start[r + 1] <= target) r++;           // row_start[r] <= target < row_start[r+1]
    const int lo = row_start[r], hi = row_start[r + 1];
    K.band_u0[b] = (target - lo <= hi - target) ? lo : hi;          // the nearer row boundary
    if (K.band_u0[b] < K.band_u0[b - 1]) K.band_u0[b] = K.band_u0[b - 1];
  }
  int longest = 0;
  for (int b = 0; b < 8; b++) longest = std::max(longest, K.band_u0[b + 1] - K.band_u0[b]);
  return longest;
}

// Fills the per-plane kernel geometry and the cumulative tile / cell counts.
void fill_stream_geom(StreamK &K, const thip_state *st) {
  int tiles = 0, cells = 0;
  for (int pli = 0; pli < 3; pli++) {
    const thip_plane_geom &g = st->geom[pli];
    PlaneK &k = K.pl[pli];
    k.nh = g.nhfrags;
    k.nv = g.nvfrags;
    k.stride = g.stride;
    k.off = g.plane_off;
    k.tiles_x = st->tiles.tiles_x[pli];
    k.tile_off = st->tiles.tile_off[pli];
    k.fro = g.froffset;
    k.rcp_cx = 1.0f / (float)(g.nhfrags + 1);
    tiles += st->tiles.tiles_x[pli] * st->tiles.tiles_y[pli];
    K.tile_end[pli] = tiles;
    cells += ((g.nhfrags + 1) * (g.nvfrags + 1) + 63) & ~63;   // whole waves per plane (k_loopfilter)
    K.cell_end[pli] = cells;
    k.tiles_y = st->tiles.tiles_y[pli];
  }
}
}  // namespace

extern "C" {

const char *thip_version_string(void) { return "theora_hip 0.2 (gfx950; libtheora 1.2.0 fragment path)"; }

int thip_state_create(thip_state **out, int frame_width, int frame_height, int pixel_fmt) {
  return thip_state_create_on(out, -1, frame_width, frame_height, pixel_fmt);
}

int thip_device_count(void) {
  int n = 0;
  return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

int thip_state_create_on(thip_state **out, int device, int frame_width, int frame_height, int pixel_fmt) {
  if (!out) return THIP_EFAULT;
  *out = nullptr;
  // state.c:712-727: coded size must be a positive multiple of 16, format not reserved
  if (frame_width <= 0 || frame_height <= 0 || (frame_width & 15) || (frame_height & 15) ||
      frame_width >= 0x100000 || frame_height >= 0x100000 || pixel_fmt < 0 || pixel_fmt > 3 ||
      pixel_fmt == 1)
    return THIP_EINVAL;
  thip_state *st = (thip_state *)calloc(1, sizeof(*st));
  if (!st) return THIP_EFAULT;
  st->frame_width = frame_width;
  st->frame_height = frame_height;
  st->pixel_fmt = pixel_fmt;
  st->hdec = !(pixel_fmt & 1);
  st->vdec = !(pixel_fmt & 2);
  int64_t fro = 0, ntiles = 0;
  size_t off = 0;
  for (int pli = 0; pli < 3; pli++) {
    thip_plane_geom &g = st->geom[pli];
    const int yh = frame_width >> 3, yv = frame_height >> 3;
    g.nhfrags = pli ? (yh + st->hdec) >> st->hdec : yh;   // state.c:443-449
    g.nvfrags = pli ? (yv + st->vdec) >> st->vdec : yv;
    const int64_t nf = (int64_t)g.nhfrags * g.nvfrags;
    g.froffset = (int32_t)fro;
    g.nfrags = (int32_t)nf;
    fro += nf;
    g.width = g.nhfrags * 8;
    g.height = g.nvfrags * 8;
    g.stride = g.width;
    g.plane_off = (int32_t)off;
    off += ((size_t)g.stride * g.height + 255) & ~(size_t)255;
    st->tiles.tiles_x[pli] = (g.nhfrags + 15) / 16;
    st->tiles.tiles_y[pli] = (g.nvfrags + 3) / 4;
    st->tiles.tile_off[pli] = (int32_t)ntiles;
    ntiles += (int64_t)st->tiles.tiles_x[pli] * st->tiles.tiles_y[pli];
    // 32-bit positions and byte offsets on the device (the reference has the same kind of
    // overflow guard, state.c:476-487, 583-590)
    if (fro >= (1 << 28) || ntiles * THIP_TILE_FRAGS >= (1ll << 30) || off >= 0x7FFFFFFFull) {
      free(st);
      return THIP_EIMPL;
    }
  }
  st->tiles.ntiles = (int32_t)ntiles;
  st->nfrags = fro;
  st->frame_bytes = off;
  st->frag_pos = (int32_t *)malloc(sizeof(int32_t) * (size_t)st->nfrags);
  if (!st->frag_pos) {
    free(st);
    return THIP_EFAULT;
  }
  for (int pli = 0; pli < 3; pli++) {
    const thip_plane_geom &g = st->geom[pli];
    for (int by = 0; by < g.nvfrags; by++)
      for (int bx = 0; bx < g.nhfrags; bx++)
        st->frag_pos[g.froffset + by * g.nhfrags + bx] =
            (st->tiles.tile_off[pli] + (by >> 2) * st->tiles.tiles_x[pli] + (bx >> 4)) * THIP_TILE_FRAGS +
            ((bx >> 2) & 3) * 16 + hilb_inv(by & 3, bx & 3);
  }
  // (everything above is host arithmetic: argument errors are reported with or without a GPU)
  if (device < 0) {   // the calling thread's current device
    if (hipGetDevice(&device) != hipSuccess) device = -1;
  } else if (device >= thip_device_count()) {
    free(st->frag_pos);
    free(st);
    return THIP_EINVAL;
  }
  if (device < 0 || device >= kMaxDevices) {
    free(st->frag_pos);
    free(st);
    return device < 0 ? THIP_EFAULT : THIP_EIMPL;
  }
  st->device = device;
  DeviceGuard dg(device);
  hipError_t err = hipSuccess;
  // +256: aligned 12-byte predictor windows may read 3 bytes past a row end
  for (int b = 0; b < 3 && err == hipSuccess; b++) err = hipMalloc((void **)&st->frames[b], st->frame_bytes + 256);
  if (err == hipSuccess) err = hipMalloc((void **)&st->coded_map, 2 * (size_t)st->nfrags);
  if (err == hipSuccess) err = hipMemset(st->coded_map, 0, 2 * (size_t)st->nfrags);
  if (err == hipSuccess) err = hipHostMalloc((void **)&st->fault, 64, hipHostMallocMapped);
  if (err == hipSuccess) memset(st->fault, 0, 64);
  if (err != hipSuccess) {
    fprintf(stderr, "theora_hip: thip_state_create: device allocation failed: %s\n", hipGetErrorString(err));
    thip_state_free(st);
    return THIP_EFAULT;
  }
  st->ref_idx[0] = st->ref_idx[1] = st->ref_idx[2] = -1;   // state.c:658-663
  st->last_decoded = -1;
  st->lane = -1;
  st->ctx_lane = -1;
  st->out_cur = -1;
  st->held_img = -1;
  st->out_serial = -1;
  st->buf_serial[0] = st->buf_serial[1] = st->buf_serial[2] = -1;
  st->map_serial[0] = st->map_serial[1] = -1;
  st->pp_serial = -1;
  {
    std::lock_guard<std::mutex> lk(g_states_mu);
    g_states.push_back(st);
  }
  *out = st;
  return THIP_OK;
}

int thip_state_device(const thip_state *st) { return st ? st->device : THIP_EFAULT; }

void thip_state_free(thip_state *st) {
  if (!st) return;
  {
    std::lock_guard<std::mutex> lk(g_states_mu);
    for (size_t i = 0; i < g_states.size(); i++)
      if (g_states[i] == st) {
        g_states[i] = g_states.back();
        g_states.pop_back();
        break;
      }
  }
  DeviceGuard dg(st->device);
  (void)hipDeviceSynchronize();
  if (st->fault) (void)hipHostFree(st->fault);
  for (int b = 0; b < 3; b++)
    if (st->frames[b]) (void)hipFree(st->frames[b]);
  if (st->coded_map) (void)hipFree(st->coded_map);
  if (st->h_info) (void)hipHostFree(st->h_info);
  if (st->h_coeffs) (void)hipHostFree(st->h_coeffs);
  if (st->h_slot0) (void)hipHostFree(st->h_slot0);
  for (int b = 0; b < 2; b++)
    if (st->h_out[b]) (void)hipHostFree(st->h_out[b]);
  for (int k = 0; k < 2; k++)
    if (st->ev_out2[k]) (void)hipEventDestroy(st->ev_out2[k]);
  if (st->ev_staging) (void)hipEventDestroy(st->ev_staging);
  if (st->d_info) (void)hipFree(st->d_info);
  if (st->d_coeffs) (void)hipFree(st->d_coeffs);
  if (st->d_slot0) (void)hipFree(st->d_slot0);
  if (st->d_dc) (void)hipFree(st->d_dc);
  if (st->d_dc_ent) (void)hipFree(st->d_dc_ent);
  if (st->d_dc_rowhas) (void)hipFree(st->d_dc_rowhas);
  if (st->ev_order) (void)hipEventDestroy(st->ev_order);
  if (st->d_edge) (void)hipFree(st->d_edge);
  if (st->d_edge_sb) (void)hipFree(st->d_edge_sb);
  if (st->d_edge_h) (void)hipFree(st->d_edge_h);
  for (int k = 0; k < 2; k++)
    if (st->h_tl_buf[k]) (void)hipHostFree(st->h_tl_buf[k]);
  if (st->d_tl) (void)hipFree(st->d_tl);
  if (st->d_tl_tmp) (void)hipFree(st->d_tl_tmp);
  if (st->d_tl_last) (void)hipFree(st->d_tl_last);
  if (st->d_tl_slot) (void)hipFree(st->d_tl_slot);
  if (st->d_tl_arr) (void)hipFree(st->d_tl_arr);
  if (st->d_tl_pos) (void)hipFree(st->d_tl_pos);
  if (st->d_tl_wide) (void)hipFree(st->d_tl_wide);
  if (st->d_tl_rank) (void)hipFree(st->d_tl_rank);
  if (st->d_frag_pos) (void)hipFree(st->d_frag_pos);
  if (st->pp_frame) (void)hipFree(st->pp_frame);
  if (st->pp_var) (void)hipFree(st->pp_var);
  if (st->pp_qis) (void)hipFree(st->pp_qis);
  if (st->pp_done) (void)hipFree(st->pp_done);
  if (st->h_pp_qis) (void)hipHostFree(st->h_pp_qis);
  if (st->ev_pp) (void)hipEventDestroy(st->ev_pp);
  if (st->d_dc_in) (void)hipFree(st->d_dc_in);
  if (st->h_dc) (void)hipHostFree(st->h_dc);
  if (st->h_flags) (void)hipHostFree(st->h_flags);
  if (st->h_tok) (void)hipHostFree(st->h_tok);
  if (st->h_slot_tok) (void)hipHostFree(st->h_slot_tok);
  if (st->h_dq) (void)hipHostFree(st->h_dq);
  if (st->d_tok) (void)hipFree(st->d_tok);
  if (st->d_slot_tok) (void)hipFree(st->d_slot_tok);
  if (st->d_dq) (void)hipFree(st->d_dq);
  if (st->h_dqp) (void)hipHostFree(st->h_dqp);
  if (st->d_dqp) (void)hipFree(st->d_dqp);
  if (st->d_flags) (void)hipFree(st->d_flags);
  free(st->enq_last_lane);
  free(st->frag_pos);
  free(st);
}

int thip_state_get_geom(const thip_state *st, thip_plane_geom geom[3], int64_t *nfrags,
                        int64_t *frame_bytes) {
  if (!st) return THIP_EFAULT;
  if (geom) memcpy(geom, st->geom, sizeof(st->geom));
  if (nfrags) *nfrags = st->nfrags;
  if (frame_bytes) *frame_bytes = (int64_t)st->frame_bytes;
  return THIP_OK;
}

int thip_state_get_tiles(const thip_state *st, thip_tile_geom *out) {
  if (!st || !out) return THIP_EFAULT;
  *out = st->tiles;
  return THIP_OK;
}

int64_t thip_state_frag_pos(const thip_state *st, int64_t fragi) {
  if (!st) return THIP_EFAULT;
  if (fragi < 0 || fragi >= st->nfrags) return THIP_EINVAL;
  return st->frag_pos[fragi];
}

int thip_state_ref_idx(const thip_state *st, int which) {
  if (!st || which < 0 || which > 2) return THIP_EINVAL;
  return st->ref_idx[which];
}

int thip_state_set_ref_idx(thip_state *st, int gold, int prev, int self) {
  if (!st) return THIP_EFAULT;
  if (gold < -1 || gold > 2 || prev < -1 || prev > 2 || self < -1 || self > 2) return THIP_EINVAL;
  st->ref_idx[THIP_FRAME_GOLD] = gold;
  st->ref_idx[THIP_FRAME_PREV] = prev;
  st->ref_idx[THIP_FRAME_SELF] = self;
  if (self >= 0) st->last_decoded = self;
  st->frame_serial++;
  st->buf_serial[0] = st->buf_serial[1] = st->buf_serial[2] = -1;   // the caller re-labelled the buffers
  return THIP_OK;
}

uint8_t *thip_state_frame_ptr(const thip_state *st, int bufi) {
  if (!st || bufi < 0 || bufi > 2) return nullptr;
  return st->frames[bufi];
}

int thip_state_read_plane(thip_state *st, int bufi, int pli, uint8_t *host_out) {
  if (!st || !host_out) return THIP_EFAULT;
  if (bufi < 0 || bufi > 2 || pli < 0 || pli > 2) return THIP_EINVAL;
  const thip_plane_geom &g = st->geom[pli];
  DeviceGuard dg(st->device);
  HIP_TRY(hipDeviceSynchronize());
  const int frc = check_fault(st);   // (a frame decoded again is complete when this returns)
  if (frc < 0) return frc;
  HIP_TRY(hipMemcpy2D(host_out, g.width, st->frames[bufi] + g.plane_off, g.stride, g.width, g.height,
                      hipMemcpyDeviceToHost));
  return THIP_OK;
}

int thip_state_write_plane(thip_state *st, int bufi, int pli, const uint8_t *host_in) {
  if (!st || !host_in) return THIP_EFAULT;
  if (bufi < 0 || bufi > 2 || pli < 0 || pli > 2) return THIP_EINVAL;
  const thip_plane_geom &g = st->geom[pli];
  DeviceGuard dg(st->device);
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy2D(st->frames[bufi] + g.plane_off, g.stride, host_in, g.width, g.width, g.height,
                      hipMemcpyHostToDevice));
  st->frame_serial++;
  st->buf_serial[bufi] = -1;
  return THIP_OK;
}

// The finished frame -> pinned host memory in display order.  One thread moves 8 bytes (plane
// widths are multiples of 8); a wave writes 512 contiguous bytes.
struct OutK {
  const uint8_t *src[3];   // per plane: the decoded frame, or the post-processed picture (thip_state_postprocess)
  uint8_t *dst;
  int width[3], height[3], stride[3], src_off[3], dst_off[3], unit_end[3];
};
__global__ __launch_bounds__(256) void k_frame_out(const OutK K) {
  const int u = (int)(blockIdx.x * 256 + threadIdx.x);
  if (u >= K.unit_end[2]) return;
  const int p = (u >= K.unit_end[0]) + (u >= K.unit_end[1]);
  const int v = u - (p ? K.unit_end[p - 1] : 0);
  const int wu = K.width[p] >> 3;
  const int y = v / wu, x = v - y * wu;
  // the device keeps row 0 at the bottom of the picture (decode.c:2988-2992 flips pointers instead)
  const uint2 val = *reinterpret_cast<const uint2 *>(K.src[p] + K.src_off[p] + (size_t)(K.height[p] - 1 - y) * K.stride[p] + x * 8);
  *reinterpret_cast<uint2 *>(K.dst + K.dst_off[p] + (size_t)y * K.width[p] + x * 8) = val;
}

// Wait for an event of this state only.  A few immediate polls, then 20 us sleeps between polls:
// a spinning wait (hipEventSynchronize, THIP_WAIT_SPIN=1) is 2 % faster for one decoder thread per
// core, but with more threads than cores the spinning threads eat the parsing threads' CPU time --
// measured on 16 cores with 16 / 32 / 64 decoder threads: 11.6 / 9.6 / 6.1 k frames/s spinning,
// 11.7 / 10.9 / 10.0 k sleeping (tools/wait_modes.sh).
static int wait_event(hipEvent_t ev) {
  if (THIP_OPT("wait_spin")) {
    HIP_TRY(hipEventSynchronize(ev));
    return THIP_OK;
  }
  // Look continuously for the first 60 microseconds -- what is left of a frame's kernels when the host gets here is usually
  // shorter than the shortest sleep the kernel grants (a 20-us nanosleep returns after 70) --, then in short sleeps: a context
  // that waits longer must not hold a core other contexts' entropy decoders want.
  timespec t0;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int polls = 0;; polls++) {
    const hipError_t e = hipEventQuery(ev);
    if (e == hipSuccess) return THIP_OK;
    if (e != hipErrorNotReady) {
      fprintf(stderr, "theora_hip: hipEventQuery: %s\n", hipGetErrorString(e));
      return THIP_EFAULT;
    }
    if (polls >= 8) {
      timespec t1;
      clock_gettime(CLOCK_MONOTONIC, &t1);
      const long long ns = (long long)(t1.tv_sec - t0.tv_sec) * 1000000000ll + (t1.tv_nsec - t0.tv_nsec);
      if (ns > 60000) {
        timespec ts = {0, 20000};
        nanosleep(&ts, nullptr);
      } else {
        cpu_relax();
      }
    }
  }
}

// Enqueue the copy of the most recently finished frame on stream s (the one that produced it).
static int launch_frame_out(thip_state *st, hipStream_t s) {
  const int nb = st->out_cur < 0 ? 0 : st->out_cur ^ 1;   // the other image: the previous one stays intact
  if (!st->h_out[nb]) HIP_TRY(hipHostMalloc((void **)&st->h_out[nb], st->frame_bytes, hipHostMallocDefault));
  if (!st->ev_out2[nb]) HIP_TRY(hipEventCreateWithFlags(&st->ev_out2[nb], hipEventDisableTiming));
  OutK K;
  const bool pp = st->pp_serial == st->frame_serial;
  for (int p = 0; p < 3; p++) K.src[p] = pp && st->pp_active[p] ? st->pp_frame : st->frames[st->last_decoded];
  K.dst = st->h_out[nb];
  int units = 0, off = 0;
  for (int p = 0; p < 3; p++) {
    const thip_plane_geom &g = st->geom[p];
    K.width[p] = g.width;
    K.height[p] = g.height;
    K.stride[p] = g.stride;
    K.src_off[p] = g.plane_off;
    K.dst_off[p] = off;
    off += g.width * g.height;
    units += (g.width >> 3) * g.height;
    K.unit_end[p] = units;
  }
  hipLaunchKernelGGL(k_frame_out, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, s, K);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(st->ev_out2[nb], s));
  st->out_cur = nb;
  st->out_serial = st->frame_serial;
  mark_dirty(st->device, s);   // (AFTER the work is queued too: a thip_synchronize of another thread may have cleared the mark in between)
  return THIP_OK;
}

int thip_state_set_eager_output(thip_state *st, int on) {
  if (!st) return THIP_EFAULT;
  st->eager_out = on ? 1 : 0;
  return THIP_OK;
}

static int ensure_frame_out(thip_state *st) {
  if (st->out_serial != st->frame_serial) {   // not copied yet (no eager output, or the frame came from write_plane)
    hipStream_t fs;
    int rc = followup_stream(st, &fs);
    if (rc < 0) return rc;
    rc = launch_frame_out(st, fs);
    if (rc < 0) return rc;
    if (fs != st->last_stream) {
      rc = order_mark(st, fs);
      if (rc < 0) return rc;
    }
  }
  return THIP_OK;
}

int thip_state_ycbcr_map(thip_state *st, const uint8_t *planes[3], int32_t strides[3]) {
  if (!st || !planes || !strides) return THIP_EFAULT;
  if (st->last_decoded < 0) return THIP_EINVAL;
  DeviceGuard dg(st->device);
  st->held_img = -1;
  for (int attempt = 0; attempt < 2; attempt++) {
    const int rc = ensure_frame_out(st);
    if (rc < 0) return rc;
    if (wait_event(st->ev_out2[st->out_cur]) < 0) return THIP_EFAULT;
    st->out_done_serial = st->out_serial;
    const int frc = check_fault(st);
    if (frc < 0) return frc;
    if (frc == 0) break;
    // the frame has been decoded again: once more, for the right picture
  }
  int off = 0;
  for (int p = 0; p < 3; p++) {
    planes[p] = st->h_out[st->out_cur] + off;
    strides[p] = st->geom[p].width;
    off += st->geom[p].width * st->geom[p].height;
  }
  return THIP_OK;
}

// The same in two halves, for a caller that wants the NEXT frame on the device before it waits for this one's picture (the
// th_decode_* front end with option fe_pipeline): _begin names the picture (the frame decoded last; its copy to the host is
// launched if it was not), the caller hands the next frame over -- which goes to the OTHER host image -- and _end waits for the
// named picture and hands it out.  A hand-over of the named frame that failed cannot be repaired any more once the next frame is
// on the device (the kernels record which launch gave up: check_fault): _end then returns THIP_EFAULT.
int thip_state_ycbcr_map_begin(thip_state *st) {
  if (!st) return THIP_EFAULT;
  if (st->last_decoded < 0) return THIP_EINVAL;
  DeviceGuard dg(st->device);
  const int rc = ensure_frame_out(st);
  if (rc < 0) return rc;
  st->held_img = st->out_cur;
  st->held_serial = st->out_serial;
  return THIP_OK;
}
int thip_state_ycbcr_map_end(thip_state *st, const uint8_t *planes[3], int32_t strides[3]) {
  if (!st || !planes || !strides) return THIP_EFAULT;
  if (st->held_img < 0) return THIP_EINVAL;
  DeviceGuard dg(st->device);
  if (wait_event(st->ev_out2[st->held_img]) < 0) return THIP_EFAULT;
  // the held frame's copy is behind all of its kernels: whatever staging buffer that frame read is free (without this
  // tl_take_staging waited for ev_staging -- recorded behind the NEXT frame -- from the second pipelined frame on)
  if (st->held_serial > st->out_done_serial) st->out_done_serial = st->held_serial;
  const int frc = check_fault(st);
  if (frc < 0) return frc;
  if (frc > 0 && st->held_img == st->out_cur) return THIP_EFAULT;   // (cannot be: a repeated frame is the newest one, and that is not the held one)
  int off = 0;
  for (int p = 0; p < 3; p++) {
    planes[p] = st->h_out[st->held_img] + off;
    strides[p] = st->geom[p].width;
    off += st->geom[p].width * st->geom[p].height;
  }
  return THIP_OK;
}

// The reference ring noted and put back (include/theora_hip.h): what a frame decoded ahead of its turn changed on the HOST side of
// the state.  frame_serial stays where it is -- it only ever grows, the events and staging buffers that are labelled with it keep
// their order -- so the discarded frame has used a number up, and whatever it wrote is labelled "unknown".
int thip_state_ring_mark(thip_state *st, int64_t mark[8]) {
  if (!st || !mark) return THIP_EFAULT;
  mark[0] = st->ref_idx[0];
  mark[1] = st->ref_idx[1];
  mark[2] = st->ref_idx[2];
  mark[3] = st->last_decoded;
  mark[4] = st->frame_serial;
  mark[5] = mark[6] = 0;
  mark[7] = 0x7468697052696e67ll;   // (a mark is a mark)
  return THIP_OK;
}
int thip_state_ring_rewind(thip_state *st, const int64_t mark[8]) {
  if (!st || !mark) return THIP_EFAULT;
  if (mark[7] != 0x7468697052696e67ll || mark[4] > st->frame_serial) return THIP_EINVAL;
  for (int k = 0; k < 3; k++)
    if (mark[k] < -1 || mark[k] > 2) return THIP_EINVAL;
  if (mark[3] < -1 || mark[3] > 2) return THIP_EINVAL;
  if (mark[4] == st->frame_serial) return THIP_OK;   // nothing was decoded since
  for (int k = 0; k < 3; k++) st->ref_idx[k] = (int)mark[k];
  st->last_decoded = (int)mark[3];
  // every buffer that is not one of the marked references may have been written by a discarded frame; so may either half of the
  // coded map (a half keeps its label only if it is older than the mark)
  for (int b = 0; b < 3; b++)
    if (b != st->ref_idx[THIP_FRAME_GOLD] && b != st->ref_idx[THIP_FRAME_PREV]) st->buf_serial[b] = -1;
  for (int h = 0; h < 2; h++)
    if (st->map_serial[h] > mark[4]) st->map_serial[h] = -1;
  // the references' labels are serial numbers of the past: "the previous frame is the PREV reference" (launch_chunk's static-block
  // test) compares them with frame_serial, which has moved on -- no shortcut for the next frame, as promised
  st->out_serial = -1;      // the host images: the newest one holds a discarded picture (thip_state_ycbcr_map copies the marked frame again)
  st->pp_serial = -1;
  st->redo.valid = 0;       // the frame that could be decoded again is not the state's newest any more
  return THIP_OK;
}

int thip_state_ycbcr_out(thip_state *st, uint8_t *const dst[3], const int32_t dst_stride[3]) {
  if (!st || !dst || !dst_stride) return THIP_EFAULT;
  if (st->last_decoded < 0) return THIP_EINVAL;
  for (int pli = 0; pli < 3; pli++)
    if (!dst[pli] || dst_stride[pli] < st->geom[pli].width) return THIP_EINVAL;
  const uint8_t *src[3];
  int32_t stride[3];
  const int rc = thip_state_ycbcr_map(st, src, stride);
  if (rc < 0) return rc;
  for (int pli = 0; pli < 3; pli++) {
    const thip_plane_geom &g = st->geom[pli];
    for (int y = 0; y < g.height; y++) memcpy(dst[pli] + (size_t)y * dst_stride[pli], src[pli] + (size_t)y * stride[pli], g.width);
  }
  return THIP_OK;
}

int thip_synchronize(void) {
  int rc = THIP_OK;
  for (int d = 0; d < kMaxDevices; d++) {
    if (!g_lanes_ready[d] && !g_ctx_ready[d]) continue;
    DeviceGuard dg(d);
    for (int i = 0; i < g_nlanes && g_lanes_ready[d]; i++)
      if (g_lane_dirty[d][i].exchange(0, std::memory_order_relaxed)) HIP_TRY(hipStreamSynchronize(g_lanes[d][i]));
    for (int i = 0; i < kCtxLanes; i++)
      if (g_ctx_ready[d] && g_ctx_lanes[d][i] && g_ctx_dirty[d][i].exchange(0, std::memory_order_relaxed)) HIP_TRY(hipStreamSynchronize(g_ctx_lanes[d][i]));
  }
  // Every state's fault words, whichever stream its frames went down (a caller-owned stream is the caller's to synchronise).  This
  // call only REPORTS: a state belongs to one thread at a time (include/theora_hip.h) and this may not be that thread, so the
  // words stay set and the state's own next synchronising call (thip_state_ycbcr_map / _out, thip_state_read_plane) decodes the
  // frame again or returns THIP_EFAULT itself.  The list is read under the lock thip_state_free takes before a state goes away.
  {
    std::lock_guard<std::mutex> lk(g_states_mu);
    for (thip_state *st : g_states) {
      volatile uint32_t *const fw = st->fault;
      if (fw && (fw[0] | fw[9])) rc = THIP_EFAULT;
    }
  }
  return rc;
}

#ifdef THIP_TRACE
int thip_debug_trace_buffer(void *dev_buf) {
  HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_trace_buf), &dev_buf, sizeof(dev_buf)));
  return THIP_OK;
}
#endif

int thip_profile_enable(int on) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_profile = on;
  return THIP_OK;
}

int thip_profile_reset(void) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  HIP_TRY(hipDeviceSynchronize());
  for (auto &p : g_events) {
    g_pool.push_back(p.a);
    g_pool.push_back(p.b);
  }
  g_events.clear();
  return THIP_OK;
}

int thip_profile_read(int64_t launches[THIP_NKERNELS], double ms[THIP_NKERNELS]) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  HIP_TRY(hipDeviceSynchronize());
  for (int k = 0; k < THIP_NKERNELS; k++) {
    launches[k] = 0;
    ms[k] = 0.0;
  }
  for (auto &p : g_events) {
    float t = 0.f;
    HIP_TRY(hipEventElapsedTime(&t, p.a, p.b));
    launches[p.kernel]++;
    ms[p.kernel] += (double)t;
  }
  return THIP_OK;
}

// More than 64 KB of dynamic LDS has to be allowed per kernel and per device, once.
static hipError_t set_dynamic_lds(const void *kernel, int bytes, int which) {
  static std::mutex mu;
  static bool done[8][kMaxDevices];
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lk(mu);
  if (dev >= 0 && dev < kMaxDevices && done[which][dev]) return hipSuccess;
  e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess && dev >= 0 && dev < kMaxDevices) done[which][dev] = true;
  return e;
}

// Launch one chunk of <= THIP_MAX_BATCH streams.
static int launch_chunk(thip_state *const *states, const thip_frame_desc *descs, int n, hipStream_t s,
                        int32_t *results, bool two_passes = false) {
  BatchK B;
  memset(&B, 0, sizeof(B));
  int max_wg = 0, max_seam_wg = 0, any_lf = 0, nlive = 0;
  int any_skip = 0;
  int live_state[THIP_MAX_BATCH];
  const bool levels = n > 0 && descs[0].coeff_format == THIP_COEFFS_LEVELS;   // (the callers cut chunks where the form changes)
  for (int i = 0; i < n; i++) {
    thip_state *st = states[i];
    const thip_frame_desc &d = descs[i];
    if (!st) return THIP_EFAULT;
    if (d.coeff_format != THIP_COEFFS_DEQUANT16 && d.coeff_format != THIP_COEFFS_LEVELS) return THIP_EINVAL;
    if ((d.coeff_format == THIP_COEFFS_LEVELS) != levels) return THIP_EINVAL;
    // (nslots counts slots, or units in the levels form: at most two per coded fragment)
    if (d.ncoded < 0 || d.nslots < 0 || d.nslots > (d.coeff_format == THIP_COEFFS_LEVELS ? 2 : 1) * (int64_t)d.ncoded || d.ncoded > st->nfrags) return THIP_EINVAL;
    if (d.ncoded && (!d.frag_info || !d.tile_slot0 || (d.nslots && !d.coeffs))) return THIP_EFAULT;
    if (d.ncoded && d.coeff_format == THIP_COEFFS_LEVELS && !d.dequant) return THIP_EFAULT;
    if (d.flimit < 0 || d.flimit > 127) return THIP_EINVAL;
    if (d.frame_type != THIP_INTRA_FRAME && d.frame_type != THIP_INTER_FRAME) return THIP_EINVAL;
    if (d.frame_type == THIP_INTRA_FRAME && d.ncoded != st->nfrags) return THIP_EINVAL;
    // a state whose previous frame went down another stream (frame calls and enqueue calls mixed): this one waits for it
    {
      const int orc = order_behind_previous(st, s);
      if (orc < 0) return orc;
    }
    // decode.c:2757-2762: an inter frame without references decodes against mid-grey
    if (d.frame_type != THIP_INTRA_FRAME &&
        (st->ref_idx[THIP_FRAME_GOLD] < 0 || st->ref_idx[THIP_FRAME_PREV] < 0)) {
      HIP_TRY(hipMemsetAsync(st->frames[0], 0x80, st->frame_bytes, s));
      st->ref_idx[0] = st->ref_idx[1] = st->ref_idx[2] = 0;
      st->last_decoded = 0;
      st->last_stream = s;
      st->order_recorded = 0;
      st->frame_serial++;
      st->buf_serial[0] = st->buf_serial[1] = st->buf_serial[2] = -1;
    }
    if (d.ncoded == 0) {  // decode.c:2764-2772
      if (results) results[i] = THIP_DUPFRAME;
      continue;
    }
    if (results) results[i] = THIP_OK;
    st->redo.valid = 0;
    if (st->redo_owned || THIP_OPT("redo_descs")) {   // (the descriptor points into the state's own buffers, or into buffers the caller keeps still: the frame can be decoded again, check_fault)
      st->redo.d = d;
      for (int k = 0; k < 3; k++) st->redo.ring[k] = st->ref_idx[k];
      st->redo.lf_custom = st->lf_rows_custom;
      for (int k = 0; k < 3; k++) {
        st->redo.lf_y0[k] = st->lf_y0[k];
        st->redo.lf_y1[k] = st->lf_y1[k];
      }
      st->redo.flush_flags = st->flush_flags;
      st->redo.serial = st->frame_serial + 1;
      st->redo.launch_id = 0;
      st->redo.valid = 1;
    }
    int bufi = 0;  // decode.c:2790-2794
    while (bufi == st->ref_idx[THIP_FRAME_GOLD] || bufi == st->ref_idx[THIP_FRAME_PREV]) bufi++;
    st->ref_idx[THIP_FRAME_SELF] = bufi;
    StreamK &K = B.s[nlive];
    K.fault = st->fault;
    K.info = reinterpret_cast<const uint2 *>(d.frag_info);
    K.coeffs = reinterpret_cast<const int4 *>(d.coeffs);
    K.tile_slot0 = d.tile_slot0;
    K.dequant = d.coeff_format == THIP_COEFFS_LEVELS ? reinterpret_cast<const uint4 *>(d.dequant) : nullptr;
    K.self = st->frames[bufi];
    K.prev = st->ref_idx[THIP_FRAME_PREV] >= 0 ? st->frames[st->ref_idx[THIP_FRAME_PREV]] : st->frames[bufi];
    K.gold = st->ref_idx[THIP_FRAME_GOLD] >= 0 ? st->frames[st->ref_idx[THIP_FRAME_GOLD]] : st->frames[bufi];
    // The flags of this frame go to the half of coded_map that does not hold the previous frame's.
    // A block that is uncoded now can stay where it is if the destination already holds it: the
    // buffer was last written two frames ago (the usual rotation between two key frames), the
    // previous frame is the PREV reference, and the previous frame's flags are at hand to tell that
    // it did not touch the block (k_recon).  THIP_SKIP_STATIC=0 switches the elision off, 2 applies it to
    // every frame with an uncoded block (tests).
    const int skip_static = THIP_OPT("skip_static");
    const int64_t serial = st->frame_serial + 1;   // of the frame being decoded
    const int pm = st->map_serial[0] == serial - 1 ? 0 : (st->map_serial[1] == serial - 1 ? 1 : -1);
    const int cm = pm == 0 ? 1 : 0;
    K.coded_map = st->coded_map + (size_t)cm * st->nfrags;
    K.coded_prev = st->coded_map + (size_t)(pm < 0 ? cm : pm) * st->nfrags;
    // (only when most of the frame is uncoded: the test costs every tile five byte loads per block, and
    //  with scattered uncoded blocks -- the smooth class, 34 % -- hardly a 64-byte line is saved)
    K.skip_ok = skip_static && pm >= 0 && ((int64_t)d.ncoded * 2 < st->nfrags || (skip_static == 2 && d.ncoded < st->nfrags)) && serial >= 2 && st->buf_serial[bufi] == serial - 2 &&
                st->ref_idx[THIP_FRAME_PREV] >= 0 && st->buf_serial[st->ref_idx[THIP_FRAME_PREV]] == serial - 1;
    st->map_serial[cm] = serial;
    st->buf_serial[bufi] = serial;
    K.flimit2 = 2 * d.flimit;
    K.debug = THIP_OPT("debug");
    // a unit for every block of the frame: the first unit of a tile in the whole tile rows of a plane is the plane's first +
    // 4 nhfrags * its tile row + 64 * its place in the row (StreamK::spec_*)
    K.spec_on = d.coeff_format == THIP_COEFFS_LEVELS && d.nslots == st->nfrags && d.ncoded == st->nfrags && THIP_OPT("spec_coeffs") != 0;
    K.spec_last = d.nslots > 0 ? (uint32_t)d.nslots - 1u : 0u;
    {
      uint32_t base = 0;
      for (int pli = 0; pli < 3; pli++) {
        K.spec_base[pli] = base;
        K.spec_rows[pli] = st->geom[pli].nvfrags / 4;
        K.spec_rowunits[pli] = 4 * st->geom[pli].nhfrags;
        K.spec_tx[pli] = st->tiles.tiles_x[pli];
        base += (uint32_t)st->geom[pli].nhfrags * (uint32_t)st->geom[pli].nvfrags;
      }
    }
    // flags-first loop filter when at least a tenth of the frame is uncoded (THIP_LF_SPARSE=0/1 forces)
    const int lf_sparse_env = THIP_OPT("lf_sparse");
    K.lf_sparse = lf_sparse_env >= 0 ? lf_sparse_env : (int64_t)d.ncoded * 10 < (int64_t)st->nfrags * 9;
    K.qpx = st->hdec;
    K.qpy = st->vdec;
    fill_stream_geom(K, st);
    for (int pli = 0; pli < 3; pli++) {
      K.lf_y0[pli] = st->lf_rows_custom ? st->lf_y0[pli] : 0;
      K.lf_y1[pli] = st->lf_rows_custom ? st->lf_y1[pli] : st->geom[pli].nvfrags;
    }
    const int wgs = ((K.tile_end[2] + THIP_RECON_WG_WAVES - 1) / THIP_RECON_WG_WAVES + 7) & ~7;   // 8 XCD bands
    if (wgs > max_wg) max_wg = wgs;
    if (d.flimit) {
      any_lf = 1;
      const int swg = ((K.cell_end[2] + THIP_LF_WG - 1) / THIP_LF_WG + 7) & ~7;   // 8 XCD bands
      if (swg > max_seam_wg) max_seam_wg = swg;
    }
    if (K.skip_ok) any_skip = 1;
    live_state[nlive++] = i;
  }
  if (!nlive) return THIP_OK;
  // ---- DC un-prediction on the device for the streams that ask for it (decode.c:1392-1500) ----------
  {
    DcBatchK D;
    memset(&D, 0, sizeof(D));
    int ndc = 0, max_rows = 0, max_nh = 0;
    for (int j = 0; j < nlive; j++) {
      thip_state *st = states[live_state[j]];
      const thip_frame_desc &d = descs[live_state[j]];
      if (!d.dc_tokens) continue;
      if (!st->d_dc) HIP_TRY(hipMalloc((void **)&st->d_dc, sizeof(int16_t) * (size_t)st->nfrags));
      if (!st->d_dc_ent) HIP_TRY(hipMalloc((void **)&st->d_dc_ent, sizeof(uint4) * (size_t)st->nfrags));
      const int rows_total = st->geom[0].nvfrags + st->geom[1].nvfrags + st->geom[2].nvfrags;
      if (!st->d_dc_rowhas) HIP_TRY(hipMalloc((void **)&st->d_dc_rowhas, (size_t)rows_total + 16));
      B.s[j].dc = st->d_dc;          // k_recon / k_recon_lf take every block's DC from here
      int row0 = 0;
      for (int pli = 0; pli < 3; pli++) {
        const thip_plane_geom &g = st->geom[pli];
        DcPlaneK &p = D.p[ndc][pli];
        p.ent = st->d_dc_ent + g.froffset;
        p.rowhas = st->d_dc_rowhas + row0;
        row0 += g.nvfrags;
        max_nh = std::max(max_nh, g.nhfrags);
        p.in = d.dc_tokens + g.froffset;
        p.out = st->d_dc + g.froffset;
        p.flags = st->flush_flags ? st->d_flags + g.froffset : nullptr;
        p.info = d.frag_info;
        p.nh = g.nhfrags;
        p.nv = g.nvfrags;
        p.tiles_x = st->tiles.tiles_x[pli];
        p.tile_base = st->tiles.tile_off[pli];
        if (g.nvfrags > max_rows) max_rows = g.nvfrags;
      }
      ndc++;
    }
    if (ndc) {
      // one wave per plane with the rows in flight in LDS (thip_dc.h); planes too large for that -- beyond 4K -- keep the
      // work-group version that goes through memory
      int lds = 0;
      bool fits = true;
      for (int j = 0; j < ndc && fits; j++)
        for (int pli = 0; pli < 3 && fits; pli++) {
          fits = dcw_fits(D.p[j][pli].nh, D.p[j][pli].nv);
          if (fits) lds = std::max(lds, dcw_layout(D.p[j][pli].nh, D.p[j][pli].nv).bytes);
        }
      const int dc_global = THIP_OPT("dc_global");
      if (fits && !dc_global) {
        HIP_TRY(set_dynamic_lds(reinterpret_cast<const void *>(k_dc_wave), kDcwLdsMax, 3));
        hipLaunchKernelGGL(k_dc_prepare, dim3(max_rows, 3, ndc), dim3((max_nh + 63) & ~63), 0, s, D);
        hipLaunchKernelGGL(k_dc_wave, dim3(3, ndc), dim3(64), (size_t)lds, s, D);
      } else {
        hipLaunchKernelGGL(k_dc_unpredict, dim3(3, ndc), dim3((max_rows + 63) & ~63), 0, s, D);
      }
      HIP_TRY(hipGetLastError());
    }
  }
  // Default (option "fuse" = 3): k_recon_lf, reconstruction and the whole loop filter in one pass (thip_fused.h); 0: the two
  // passes k_recon + k_loopfilter.  Frames that leave static blocks in place (skip_ok) and frames without a loop filter
  // always take the two passes, whose first kernel knows how to skip whole tiles.
  const int fuse = two_passes ? 0 : THIP_OPT("fuse");
  if (fuse == 3 && any_lf && !any_skip && xcd_round_robin(states[live_state[0]]->device)) {
    // one wave per tile, reconstruction and every filter cell in one pass (thip_fused.h) -- or, for a launch that would leave
    // the chip empty, one wave per super block (thip_fused_sb.h)
    int longest = 1, total_tiles = 0;
    for (int j = 0; j < nlive; j++) total_tiles += B.s[j].tile_end[2];
    const int sb_tiles = THIP_OPT("sb_tiles"), half_tiles = THIP_OPT("half_tiles");
    const bool small = sb_tiles > 0 && total_tiles < sb_tiles;
    const bool half = !small && half_tiles > 0 && total_tiles < half_tiles;   // two super blocks a wave, two lanes a block (k_recon_lf_h)
    for (int j = 0; j < nlive; j++) {
      thip_state *st = states[live_state[j]];
      StreamK &K = B.s[j];
      if (st->tiles.tiles_y[0] > 4096) return THIP_EIMPL;
      longest = std::max(longest, fill_bands(K, st));
      if (small) {
        if (!st->d_edge_sb) {
          HIP_TRY(hipMalloc((void **)&st->d_edge_sb, (size_t)4 * K.tile_end[2] * Tf4::kRec));
          HIP_TRY(hipMemsetAsync(st->d_edge_sb, 0, (size_t)4 * K.tile_end[2] * Tf4::kRec, s));
        }
        K.edge = st->d_edge_sb;
        st->edge_epoch_sb = st->edge_epoch_sb % 4095u + 1u;
        K.epoch = st->edge_epoch_sb;
        st->redo.launch_id = K.epoch | 0x1000u;
        continue;
      }
      if (half) {
        if (!st->d_edge_h) {
          HIP_TRY(hipMalloc((void **)&st->d_edge_h, (size_t)2 * K.tile_end[2] * Tf8::kRec));
          HIP_TRY(hipMemsetAsync(st->d_edge_h, 0, (size_t)2 * K.tile_end[2] * Tf8::kRec, s));
        }
        K.edge = st->d_edge_h;
        st->edge_epoch_h = st->edge_epoch_h % 4095u + 1u;
        K.epoch = st->edge_epoch_h;
        st->redo.launch_id = K.epoch | 0x2000u;
        continue;
      }
      if (!st->d_edge) {
        HIP_TRY(hipMalloc((void **)&st->d_edge, (size_t)K.tile_end[2] * kTfRec));
        HIP_TRY(hipMemsetAsync(st->d_edge, 0, (size_t)K.tile_end[2] * kTfRec, s));
      }
      K.edge = st->d_edge;
      st->edge_epoch = st->edge_epoch % 4095u + 1u;   // 12 bits in a record's flag word, never 0
      K.epoch = st->edge_epoch;
      st->redo.launch_id = K.epoch;
    }
    ScopedTimer t(s, THIP_KERNEL_RECON);
    if (small) {
      if (levels) hipLaunchKernelGGL(k_recon_lf_sb<true>, dim3(8 * longest, nlive), dim3(256), 0, s, B);
      else hipLaunchKernelGGL(k_recon_lf_sb<false>, dim3(8 * longest, nlive), dim3(256), 0, s, B);
    } else if (half) {
      if (levels) hipLaunchKernelGGL(k_recon_lf_h<true>, dim3(8 * longest, nlive), dim3(128), 0, s, B);
      else hipLaunchKernelGGL(k_recon_lf_h<false>, dim3(8 * longest, nlive), dim3(128), 0, s, B);
    } else if (levels) hipLaunchKernelGGL(k_recon_lf<true>, dim3(8 * longest, nlive), dim3(64), 0, s, B);
    else hipLaunchKernelGGL(k_recon_lf<false>, dim3(8 * longest, nlive), dim3(64), 0, s, B);
  } else {
    {
      ScopedTimer t(s, THIP_KERNEL_RECON);
      if (levels) hipLaunchKernelGGL(k_recon<true>, dim3(max_wg, nlive), dim3(64 * THIP_RECON_WG_WAVES), 0, s, B);
      else hipLaunchKernelGGL(k_recon<false>, dim3(max_wg, nlive), dim3(64 * THIP_RECON_WG_WAVES), 0, s, B);
    }
    if (any_lf) {
      ScopedTimer t(s, THIP_KERNEL_LOOPFILTER);
      hipLaunchKernelGGL(k_loopfilter, dim3(max_seam_wg, nlive), dim3(THIP_LF_WG), 0, s, B);
    }
  }
  HIP_TRY(hipGetLastError());
  // decode.c:2947-2962
  for (int j = 0; j < nlive; j++) {
    thip_state *st = states[live_state[j]];
    const int self = st->ref_idx[THIP_FRAME_SELF];
    if (descs[live_state[j]].frame_type == THIP_INTRA_FRAME) st->ref_idx[THIP_FRAME_GOLD] = self;
    st->ref_idx[THIP_FRAME_PREV] = self;
    st->last_decoded = self;
    st->frame_serial++;
    if (st->eager_out) {
      const int rc = launch_frame_out(st, s);
      if (rc < 0) return rc;
    }
    const int orc = order_mark(st, s);   // (behind the frame's last launch on s)
    if (orc < 0) return orc;
  }
  return THIP_OK;
}

static int check_fault(thip_state *st) {
  if (!st->fault) return THIP_OK;
  volatile uint32_t *const fw = st->fault;
  if (!(fw[0] | fw[9])) return THIP_OK;
  // everything queued for this state has to be over before the frame is launched again
  if (st->last_stream && is_library_stream(st->device, st->last_stream)) HIP_TRY(hipStreamSynchronize(st->last_stream));
  else HIP_TRY(hipDeviceSynchronize());
  const uint32_t w = fw[0] | fw[9];
  // Which launches reported a failed wait (tf_fetch_unit).  The frame that can be repeated is the state's most recent one; if an
  // EARLIER launch failed as well -- the caller decoded on without a synchronising call in between -- repeating the last frame
  // would put a right picture on top of a wrong reference and call it recovered.
  bool only_last = st->redo.valid && st->redo.launch_id != 0;
  for (int k = 1; k <= 8; k++) {
    if (fw[k] && fw[k] != st->redo.launch_id) only_last = false;
    fw[k] = 0;
  }
  if (fw[10]) only_last = false;   // two failing launches shared an id word (tf_fetch_unit): which ones is not known
  fw[10] = 0;
  fw[0] = 0;
  fw[9] = 0;
  if ((w & 1u) && !(w & 2u) && only_last && st->redo.serial == st->frame_serial) {
    fprintf(stderr, "theora_hip: a bounded wait of k_recon_lf for a neighbouring tile ran out (device %d); decoding the frame again with "
                    "the two passes\n", st->device);
    const thip_frame_desc d = st->redo.d;
    for (int k = 0; k < 3; k++) st->ref_idx[k] = st->redo.ring[k];
    st->lf_rows_custom = st->redo.lf_custom;
    for (int k = 0; k < 3; k++) {
      st->lf_y0[k] = st->redo.lf_y0[k];
      st->lf_y1[k] = st->redo.lf_y1[k];
    }
    st->flush_flags = st->redo.flush_flags;
    if (st->out_cur >= 0 && st->out_serial == st->frame_serial) st->out_cur ^= 1;   // the wrong picture's host image is the one to overwrite
    hipStream_t s;
    int rc = followup_stream(st, &s);
    if (rc < 0) return rc;
    thip_state *sp = st;
    int32_t res = 0;
    st->redo_owned = 1;
    rc = launch_chunk(&sp, &d, 1, s, &res, true);
    st->redo_owned = 0;
    st->lf_rows_custom = 0;
    st->flush_flags = 0;
    if (rc < 0) return rc;
    HIP_TRY(hipStreamSynchronize(s));
    if (!(fw[0] | fw[9])) {
      static Option *const counter = find_option("faults_recovered");
      if (counter) counter->value.fetch_add(1, std::memory_order_relaxed);
      return 1;
    }
    for (int k = 0; k <= 10; k++) fw[k] = 0;
  }
  fprintf(stderr, "theora_hip: a kernel's bounded wait ran out on device %d (%s) and the frame could not be decoded again: the frames "
                  "of this state since its last synchronisation are not to be trusted (option fuse = 0 takes the two-pass path)\n",
          st->device, (w & 2u) ? "de-ringing" : "tile hand-over");
  return THIP_EFAULT;
}

extern "C" int thip_state_check_fault(thip_state *st) {
  if (!st) return THIP_EFAULT;
  DeviceGuard dg(st->device);
  if (st->last_stream && is_library_stream(st->device, st->last_stream)) HIP_TRY(hipStreamSynchronize(st->last_stream));
  return check_fault(st);
}

// What launch_chunk would refuse, checked for the whole call before anything is launched or any
// state is advanced: a bad descriptor in stream 11 must not leave streams 0..7 one frame ahead.
static int validate_frames(thip_state *const *states, const thip_frame_desc *descs, int n) {
  for (int i = 0; i < n; i++) {
    const thip_state *st = states[i];
    const thip_frame_desc &d = descs[i];
    if (!st) return THIP_EFAULT;
    if (d.coeff_format != THIP_COEFFS_DEQUANT16 && d.coeff_format != THIP_COEFFS_LEVELS) return THIP_EINVAL;
    // (nslots counts slots, or units in the levels form: at most two per coded fragment)
    if (d.ncoded < 0 || d.nslots < 0 || d.nslots > (d.coeff_format == THIP_COEFFS_LEVELS ? 2 : 1) * (int64_t)d.ncoded || d.ncoded > st->nfrags) return THIP_EINVAL;
    if (d.ncoded && (!d.frag_info || !d.tile_slot0 || (d.nslots && !d.coeffs))) return THIP_EFAULT;
    if (d.ncoded && d.coeff_format == THIP_COEFFS_LEVELS && !d.dequant) return THIP_EFAULT;
    if (d.flimit < 0 || d.flimit > 127) return THIP_EINVAL;
    if (d.frame_type != THIP_INTRA_FRAME && d.frame_type != THIP_INTER_FRAME) return THIP_EINVAL;
    if (d.frame_type == THIP_INTRA_FRAME && d.ncoded != st->nfrags) return THIP_EINVAL;
    if (d.dc_tokens)
      for (int pli = 0; pli < 3; pli++)
        if (st->geom[pli].nvfrags > kDcMaxRows) return THIP_EIMPL;
    for (int j = 0; j < i; j++)
      if (states[j] == st) return THIP_EINVAL;   // one frame per stream per call
  }
  return THIP_OK;
}

int thip_decode_frames(thip_state *const *states, const thip_frame_desc *descs, int nstreams,
                       void *stream, int32_t *results) {
  if (!states || !descs) return THIP_EFAULT;
  if (nstreams < 0) return THIP_EINVAL;
  int rc = validate_frames(states, descs, nstreams);
  if (rc < 0) return rc;
  // No library-wide lock around the launches: a state belongs to the calling thread, the HIP
  // runtime is thread safe, and contexts on different host threads must not queue behind each
  // other here.  Only the lane assignment is shared.
  if (stream) {   // caller-owned stream: everything in submission order on it, all states on its device
    hipStream_t s = (hipStream_t)stream;
    for (int i = 1; i < nstreams; i++)
      if (states[i]->device != states[0]->device) return THIP_EINVAL;
    if (!nstreams) return THIP_OK;
    DeviceGuard dg(states[0]->device);
    for (int i = 0; i < nstreams;) {
      int n = 1;   // (a chunk holds one coefficient form: k_recon / k_recon_lf exist once per form)
      while (i + n < nstreams && n < THIP_MAX_BATCH && descs[i + n].coeff_format == descs[i].coeff_format) n++;
      rc = launch_chunk(states + i, descs + i, n, s, results ? results + i : nullptr);
      if (rc < 0) return rc;
      i += n;
    }
    return THIP_OK;
  }
  // group by device, then by lane (order inside a lane preserved), launch chunk by chunk
  // (THIP_CHUNK: streams per launch)
  const int chunk_max = [] {
    const int v = THIP_OPT("chunk");
    return v < 1 ? 1 : (v > THIP_MAX_BATCH ? THIP_MAX_BATCH : v);
  }();
  uint32_t devmask = 0;
  for (int i = 0; i < nstreams; i++) devmask |= 1u << states[i]->device;
  for (int dev = 0; dev < kMaxDevices; dev++) {
    if (!(devmask >> dev & 1u)) continue;
    DeviceGuard dg(dev);
    rc = ensure_lanes(dev);
    if (rc) return rc;
    for (int i = 0; i < nstreams; i++)
      if (states[i]->device == dev && states[i]->lane < 0) {
        std::lock_guard<std::mutex> lk(g_mu);
        states[i]->lane = g_next_lane[dev]++ % g_nlanes;
      }
    for (int lane = 0; lane < g_nlanes; lane++)
     for (int form = 0; form < 2; form++) {   // (a chunk holds one coefficient form)
      thip_state *ls[THIP_MAX_BATCH];
      thip_frame_desc ld[THIP_MAX_BATCH];
      int32_t lr[THIP_MAX_BATCH];
      int li[THIP_MAX_BATCH];
      int n = 0;
      for (int i = 0; i <= nstreams; i++) {
        if (i < nstreams && states[i]->device == dev && states[i]->lane == lane && (descs[i].coeff_format == THIP_COEFFS_LEVELS) == (form == 1)) {
          ls[n] = states[i];
          ld[n] = descs[i];
          li[n] = i;
          n++;
        }
        if (n == chunk_max || (i == nstreams && n > 0)) {
          g_lane_dirty[dev][lane].store(1, std::memory_order_relaxed);
          rc = launch_chunk(ls, ld, n, g_lanes[dev][lane], lr);
          if (rc < 0) return rc;
          if (results)
            for (int k = 0; k < n; k++) results[li[k]] = lr[k];
          n = 0;
        }
      }
    }
  }
  return THIP_OK;
}

int thip_dc_unpredict_plane(int16_t *dc, const uint8_t *flags, int nhfrags, int nvfrags) {
  if (!dc || !flags) return THIP_EFAULT;
  if (nhfrags <= 0 || nvfrags <= 0) return THIP_EINVAL;
  if (nvfrags > kDcMaxRows) return THIP_EIMPL;
  DcBatchK D;
  memset(&D, 0, sizeof(D));
  DcPlaneK &p = D.p[0][0];
  p.in = dc;
  p.out = dc;
  p.flags = flags;
  p.nh = nhfrags;
  p.nv = nvfrags;
  void *scratch = nullptr;
  if (dcw_fits(nhfrags, nvfrags) && !THIP_OPT("dc_global")) {
    const size_t nf = (size_t)nhfrags * nvfrags;
    HIP_TRY(hipMalloc(&scratch, nf * sizeof(uint4) + (size_t)nvfrags + 16));
    p.ent = (uint4 *)scratch;
    p.rowhas = (uint8_t *)scratch + nf * sizeof(uint4);
    HIP_TRY(set_dynamic_lds(reinterpret_cast<const void *>(k_dc_wave), kDcwLdsMax, 3));
    hipLaunchKernelGGL(k_dc_prepare, dim3(nvfrags, 1, 1), dim3((nhfrags + 63) & ~63), 0, 0, D);
    hipLaunchKernelGGL(k_dc_wave, dim3(1, 1), dim3(64), (size_t)dcw_layout(nhfrags, nvfrags).bytes, 0, D);
  } else {
    hipLaunchKernelGGL(k_dc_unpredict, dim3(1, 1), dim3((nvfrags + 63) & ~63), 0, 0, D);
  }
  const hipError_t le = hipGetLastError(), se = hipDeviceSynchronize();
  if (scratch) (void)hipFree(scratch);
  HIP_TRY(le);
  HIP_TRY(se);
  return THIP_OK;
}

int thip_state_set_device_dc(thip_state *st, int on) {
  if (!st) return THIP_EFAULT;
  if (on)
    for (int pli = 0; pli < 3; pli++)
      if (st->geom[pli].nvfrags > kDcMaxRows) return THIP_EIMPL;
  st->device_dc = on ? 1 : 0;
  return THIP_OK;
}

// th_decode_packetin's out-of-loop post-processing (decode.c:2893-2911) of the frame just decoded.
int thip_state_postprocess(thip_state *st, int level, const uint8_t *dc_qis, const uint8_t *frag_qi, const int32_t pp_dc_scale[64],
                           const int32_t pp_sharp_mod[64]) {
  if (!st || !dc_qis || !frag_qi || !pp_dc_scale || !pp_sharp_mod) return THIP_EFAULT;
  if (level < 0 || level > 7 || st->last_decoded < 0) return THIP_EINVAL;
  if (level < 2) {           // levels 0 and 1 change no pixel (decode.c:1208-1253)
    st->pp_serial = -1;
    return THIP_OK;
  }
  DeviceGuard dg(st->device);
  if (!st->pp_frame) HIP_TRY(hipMalloc((void **)&st->pp_frame, st->frame_bytes + 256));
  if (!st->pp_var) HIP_TRY(hipMalloc((void **)&st->pp_var, sizeof(int) * (size_t)st->nfrags));
  if (!st->pp_qis) HIP_TRY(hipMalloc((void **)&st->pp_qis, 2 * (size_t)st->nfrags));
  hipStream_t s;
  {
    const int rc = followup_stream(st, &s);
    if (rc < 0) return rc;
  }
  // The caller's arrays are pageable and rewritten by the next packet: they are staged through a pinned buffer of the state's
  // (a copy from pageable memory makes the call a host synchronisation point), guarded by an event behind the copy.
  if (!st->h_pp_qis) HIP_TRY(hipHostMalloc((void **)&st->h_pp_qis, 2 * (size_t)st->nfrags, hipHostMallocDefault));
  if (!st->ev_pp) HIP_TRY(hipEventCreateWithFlags(&st->ev_pp, hipEventDisableTiming));
  else if (wait_event(st->ev_pp) < 0) return THIP_EFAULT;   // the previous frame's copy has read the buffer
  memcpy(st->h_pp_qis, dc_qis, (size_t)st->nfrags);
  memcpy(st->h_pp_qis + st->nfrags, frag_qi, (size_t)st->nfrags);
  HIP_TRY(hipMemcpyAsync(st->pp_qis, st->h_pp_qis, 2 * (size_t)st->nfrags, hipMemcpyHostToDevice, s));
  HIP_TRY(hipEventRecord(st->ev_pp, s));
  HIP_TRY(hipMemsetAsync(st->pp_var, 0, sizeof(int) * (size_t)st->nfrags, s));
  PpK K;
  memset(&K, 0, sizeof(K));
  int max_w = 0, max_nv = 0, max_h = 0;
  for (int pli = 0; pli < 3; pli++) {
    const thip_plane_geom &g = st->geom[pli];
    PpPlaneK &p = K.p[pli];
    p.src = st->frames[st->last_decoded] + g.plane_off;
    p.dst = st->pp_frame + g.plane_off;
    p.stride = g.stride;
    p.width = g.width;
    p.height = g.height;
    p.nh = g.nhfrags;
    p.nv = g.nvfrags;
    p.variances = st->pp_var + g.froffset;
    p.dc_qis = st->pp_qis + g.froffset;
    p.frag_qi = st->pp_qis + st->nfrags + g.froffset;
    const int off = 3 * (pli != 0);   // decode.c:2894: chroma takes the same steps three levels later
    K.active[pli] = level >= 2 + off;
    K.dering[pli] = level >= 3 + off;
    K.strong[pli] = level >= 4 + off;
    st->pp_active[pli] = K.active[pli];
    if (K.active[pli]) {
      if (g.width > max_w) max_w = g.width;
      if (g.nvfrags > max_nv) max_nv = g.nvfrags;
      if (g.height > max_h) max_h = g.height;
    }
  }
  for (int i = 0; i < 64; i++) {
    K.dc_scale[i] = pp_dc_scale[i];
    K.sharp_mod[i] = pp_sharp_mod[i];
  }
  hipLaunchKernelGGL(k_pp_hedge, dim3((unsigned)((max_w / 4 + 63) / 64), (unsigned)(max_nv + 1), 3), dim3(64), 0, s, K);
  hipLaunchKernelGGL(k_pp_vedge, dim3((unsigned)((max_h + 63) / 64), 1, 3), dim3(64), 0, s, K);
  {
    // ONE launch: the anti-diagonals of the group grids along blockIdx.z (dispatched in that order), the three planes side
    // by side along blockIdx.y; a group waits for its left and upper neighbours' entries in pp_done (thip_postproc.h)
    int nd = 0, gmax = 0;
    size_t ngroups = 0, goff[3];
    for (int pli = 0; pli < 3; pli++) {
      const int gnx = (K.p[pli].nh + kPpGroup - 1) / kPpGroup, gny = (K.p[pli].nv + kPpGroup - 1) / kPpGroup;
      goff[pli] = ngroups;
      ngroups += (size_t)gnx * gny;
      if (!K.dering[pli]) continue;
      nd = std::max(nd, gnx + gny - 1);
      gmax = std::max(gmax, std::min(gnx, gny));
    }
    if (nd > 65535) return THIP_EIMPL;
    if (nd) {
      if (!st->pp_done) {
        HIP_TRY(hipMalloc((void **)&st->pp_done, ngroups * sizeof(uint32_t)));
        HIP_TRY(hipMemsetAsync(st->pp_done, 0, ngroups * sizeof(uint32_t), s));
      }
      for (int pli = 0; pli < 3; pli++) K.done[pli] = st->pp_done + goff[pli];
      st->pp_run = st->pp_run == 0xFFFFFFFFu ? 1u : st->pp_run + 1u;
      K.serial = st->pp_run;
      K.fault = st->fault;
      hipLaunchKernelGGL(k_pp_dering, dim3((unsigned)gmax, 3, (unsigned)nd), dim3(64 * kPpGroup), 0, s, K);
    }
  }
  HIP_TRY(hipGetLastError());
  st->pp_serial = st->frame_serial;
  st->out_serial = -1;       // the host image (if any) shows the frame before post-processing
  if (s != st->last_stream) {   // (the frame went down a caller-owned stream: the next one waits for these kernels too)
    const int rc = order_mark(st, s);
    if (rc < 0) return rc;
  }
  return THIP_OK;
}

// The post-processed picture's planes (device, bitstream row order), for tests and on-device consumers.
int thip_state_read_pp_plane(thip_state *st, int pli, uint8_t *host_out) {
  if (!st || !host_out) return THIP_EFAULT;
  if (pli < 0 || pli > 2 || st->pp_serial != st->frame_serial) return THIP_EINVAL;
  const thip_plane_geom &g = st->geom[pli];
  DeviceGuard dg(st->device);
  HIP_TRY(hipDeviceSynchronize());
  const uint8_t *src = (st->pp_active[pli] ? st->pp_frame : st->frames[st->last_decoded]) + g.plane_off;
  HIP_TRY(hipMemcpy2D(host_out, g.width, src, g.stride, g.width, g.height, hipMemcpyDeviceToHost));
  return THIP_OK;
}

int thip_loop_filter_plane(uint8_t *plane, int ystride, int nhfrags, int nvfrags, const uint8_t *coded,
                           int flimit, int fragy0, int fragy_end) {
  if (!plane || !coded) return THIP_EFAULT;
  if (nhfrags <= 0 || nvfrags <= 0 || flimit < 0 || flimit > 127 || ystride < nhfrags * 8 ||
      (int64_t)(nhfrags + 1) * (nvfrags + 1) >= (1 << 24))
    return THIP_EINVAL;
  if (flimit == 0) return THIP_OK;
  const int64_t cells = (int64_t)(nhfrags + 1) * (nvfrags + 1);
  hipLaunchKernelGGL(k_loopfilter_plane, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, 0, plane, ystride,
                     nhfrags, nvfrags, coded, 2 * flimit, fragy0, fragy_end, 1.0f / (float)(nhfrags + 1));
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  return THIP_OK;
}

// ---------------------------------------------------------------------------------------
// host-enqueue form (the vtable slots)
// ---------------------------------------------------------------------------------------
static int ensure_staging(thip_state *st) {
  if (st->staging_ready) return THIP_OK;
  const size_t npos = (size_t)st->tiles.ntiles * THIP_TILE_FRAGS;
  const size_t ngroups = ((size_t)st->nfrags + THIP_SLOT_GROUP - 1) / THIP_SLOT_GROUP;
  HIP_TRY(hipHostMalloc((void **)&st->h_info, npos * 8, hipHostMallocDefault));
  HIP_TRY(hipHostMalloc((void **)&st->h_coeffs, ngroups * THIP_SLOT_GROUP_BYTES, hipHostMallocDefault));
  HIP_TRY(hipHostMalloc((void **)&st->h_slot0, (size_t)st->tiles.ntiles * 4, hipHostMallocDefault));
  HIP_TRY(hipMalloc((void **)&st->d_info, npos * 8));
  HIP_TRY(hipMalloc((void **)&st->d_coeffs, ngroups * THIP_SLOT_GROUP_BYTES));
  HIP_TRY(hipMalloc((void **)&st->d_slot0, (((size_t)st->tiles.ntiles + 3) & ~(size_t)3) * 4));   // (zero-filled 16 bytes at a time)
  st->enq_last_lane = (int32_t *)malloc(sizeof(int32_t) * (size_t)st->tiles.ntiles);
  if (!st->enq_last_lane) return THIP_EFAULT;
  st->staging_ready = 1;
  return THIP_OK;
}

// The staging buffers are free again once the kernels of the frame that used them have read them.  The event that says so is
// the LAST thing queued for that frame, and the runtime submits a trailing marker lazily: asking for it costs a flush and one
// or two hundred microseconds of polling even when the work finished long ago.  The caller has usually waited for that very
// frame's output copy already (thip_state_ycbcr_map), which stands behind the same kernels on the same stream: then no question
// is asked at all.
static int wait_staging_free(thip_state *st) {
  if (!st->ev_staging) return THIP_OK;
  if (st->out_done_serial >= st->staging_serial) return THIP_OK;
  return wait_event(st->ev_staging);
}
// ... for the token lists: the OTHER of the two buffers is taken, which the frame before the previous one used -- a caller that
// looks at its pictures has waited for that frame long ago, so nothing is asked even with the previous frame still on the device.
static int tl_take_staging(thip_state *st) {
  const int nb = st->tl_buf ^ 1;
  if (st->ev_staging && st->out_done_serial < st->tl_buf_serial[nb] && wait_event(st->ev_staging) < 0) return THIP_EFAULT;
  st->tl_buf = nb;
  st->h_tl = st->h_tl_buf[nb];
  return THIP_OK;
}

int thip_frame_begin(thip_state *st, int frame_type) {
  if (!st) return THIP_EFAULT;
  if (frame_type != THIP_INTRA_FRAME && frame_type != THIP_INTER_FRAME) return THIP_EINVAL;
  if (st->tl_pending) return THIP_EINVAL;   // a frame handed over with thip_state_token_lists_begin waits for its finish
  DeviceGuard dg(st->device);
  int rc = ensure_staging(st);
  if (rc) return rc;
  // the previous frame's kernels must have read the staging buffers before they are reused;
  // only this stream's own work is waited for, so contexts on other host threads keep going
  if (wait_staging_free(st) < 0) return THIP_EFAULT;
  memset(st->h_info, 0, (size_t)st->tiles.ntiles * THIP_TILE_FRAGS * 8);   // everything uncoded
  memset(st->h_slot0, 0, (size_t)st->tiles.ntiles * 4);
  for (int t = 0; t < st->tiles.ntiles; t++) st->enq_last_lane[t] = -1;
  st->enq_ncoded = st->enq_nuncoded = st->enq_nslots = 0;
  st->enq_last_tile = -1;
  st->enq_frame_type = frame_type;
  st->enq_flimit = 0;
  st->enq_lf_any = 0;
  for (int p = 0; p < 3; p++) {
    st->enq_lf_y0[p] = 0x7FFFFFFF;
    st->enq_lf_y1[p] = -1;
  }
  st->enq_ntok = st->enq_tok_slots = st->enq_dense_slots = st->enq_level_slots = st->enq_tile_blocks = st->enq_levels_hint = 0;
  st->enq_device_dc = st->device_dc;
  if (st->enq_device_dc) {
    if (!st->h_dc) HIP_TRY(hipHostMalloc((void **)&st->h_dc, sizeof(int16_t) * (size_t)st->nfrags, hipHostMallocDefault));
    if (!st->d_dc_in) HIP_TRY(hipMalloc((void **)&st->d_dc_in, sizeof(int16_t) * (((size_t)st->nfrags + 7) & ~(size_t)7)));
    if (!st->h_flags) HIP_TRY(hipHostMalloc((void **)&st->h_flags, (size_t)st->nfrags, hipHostMallocDefault));
    if (!st->d_flags) HIP_TRY(hipMalloc((void **)&st->d_flags, (size_t)st->nfrags));
    memset(st->h_flags, 0, (size_t)st->nfrags);   // everything uncoded
  }
  st->enq_active = 1;
  return THIP_OK;
}

int thip_state_frag_recon(thip_state *st, ptrdiff_t fragi, int pli, int16_t dct_coeffs[128],
                          int last_zzi, uint16_t dc_quant, int refi, int16_t mv) {
  if (!st || !dct_coeffs) return THIP_EFAULT;
  if (!st->enq_active || fragi < 0 || fragi >= st->nfrags || pli < 0 || pli > 2 || refi < 0 || refi > 2 ||
      last_zzi < 0 || last_zzi > 64 || (int64_t)st->enq_ncoded + st->enq_nuncoded >= st->nfrags)
    return THIP_EINVAL;
  const int32_t pos = st->frag_pos[fragi];
  if (st->h_info[2 * (size_t)pos] & THIP_INFO_CODED) return THIP_EINVAL;   // fragment enqueued twice
  uint32_t flags = THIP_INFO_CODED | ((uint32_t)refi << THIP_INFO_REFI_SHIFT) |
                   ((uint32_t)last_zzi << THIP_INFO_LAST_ZZI_SHIFT) |
                   ((uint32_t)(uint8_t)(mv & 0xFF) << THIP_INFO_MVX_SHIFT) |
                   ((uint32_t)(uint8_t)((mv >> 8) & 0xFF) << THIP_INFO_MVY_SHIFT);
  // The DC coefficient travels raw; k_recon applies state.c:967-979 (the rounded DC-only form or the
  // 16-bit product) with the dc_quant carried in the upper half of command word 1.
  uint32_t word1 = (uint32_t)dc_quant << 16;
  if (last_zzi < 2) {
    flags |= THIP_INFO_DC_ONLY;   // no coefficient slot
    word1 |= (uint32_t)(uint16_t)dct_coeffs[0];
  } else {
    // Slots are handed out in arrival order, which for the reference's caller is coded
    // order == tile/lane order (decode.c:1530-1586); the kernel re-derives a lane's slot
    // from the tile's first slot and a prefix count, so arrival must not jump backwards
    // inside a tile.
    const int tile = pos / THIP_TILE_FRAGS, lane = pos % THIP_TILE_FRAGS;
    if (st->enq_tok_slots || st->enq_level_slots) return THIP_EINVAL;   // the frame's coefficient slots come in another form (tokens, levels)
    if (lane <= st->enq_last_lane[tile]) return THIP_EINVAL;
    if (st->enq_last_lane[tile] >= 0 && st->enq_last_tile != tile) return THIP_EINVAL;   // a tile's slots must be contiguous
    // (every refusal is above this line: a refused call leaves the tile bookkeeping as it was)
    if (st->enq_last_lane[tile] < 0) st->h_slot0[tile] = (uint32_t)st->enq_nslots;
    st->enq_last_lane[tile] = lane;
    st->enq_last_tile = tile;
    st->enq_dense_slots++;
    const int slot = st->enq_nslots++;
    // piece q = 2*j+h of the block: columns c = 4h..4h+3 as pairs { x[2j][c], x[2j+1][c] }, i.e. the
    // low (h = 0) and high (h = 1) halves of rows 2j and 2j+1 interleaved: two shuffles per row pair
    typedef int16_t v8s __attribute__((vector_size(16)));
    int16_t *blk = st->h_coeffs + (size_t)(slot >> 6) * (THIP_SLOT_GROUP_BYTES / 2) + (size_t)(slot & 63) * 8;
    for (int j = 0; j < 4; j++) {
      v8s a, b;
      memcpy(&a, dct_coeffs + (2 * j) * 8, 16);
      memcpy(&b, dct_coeffs + (2 * j + 1) * 8, 16);
      const v8s lo = __builtin_shufflevector(a, b, 0, 8, 1, 9, 2, 10, 3, 11);
      const v8s hi = __builtin_shufflevector(a, b, 4, 12, 5, 13, 6, 14, 7, 15);
      memcpy(blk + (size_t)(2 * j) * 512, &lo, 16);
      memcpy(blk + (size_t)(2 * j + 1) * 512, &hi, 16);
    }
  }
  if (st->enq_device_dc) {   // the token value: un-predicted on the device at flush
    st->h_dc[fragi] = dct_coeffs[0];
    st->h_flags[fragi] = (uint8_t)(1u | (uint32_t)refi << 1);
  }
  memset(dct_coeffs, 0, 64 * sizeof(int16_t));   // idct.c:245,276,295
  st->h_info[2 * (size_t)pos] = flags;
  st->h_info[2 * (size_t)pos + 1] = word1;
  st->enq_ncoded++;
  return THIP_OK;
}

// the frame's AC dequantisation tables (thip_frame_dequant_table): zig-zag order for k_expand_tokens, slot order for the levels form
static int ensure_dq_staging(thip_state *st) {
  if (!st->h_dq) {
    HIP_TRY(hipHostMalloc((void **)&st->h_dq, 18 * 64 * 2, hipHostMallocDefault));
    memset(st->h_dq, 0, 18 * 64 * 2);
  }
  if (!st->d_dq) HIP_TRY(hipMalloc((void **)&st->d_dq, 18 * 64 * 2));
  if (!st->h_dqp) {
    HIP_TRY(hipHostMalloc((void **)&st->h_dqp, 18 * 64 * 2, hipHostMallocDefault));
    memset(st->h_dqp, 0, 18 * 64 * 2);
  }
  if (!st->d_dqp) HIP_TRY(hipMalloc((void **)&st->d_dqp, 18 * 64 * 2));
  return THIP_OK;
}

static int ensure_token_staging(thip_state *st) {
  if (st->tok_ready) return THIP_OK;
  // A slot stages the raw DC plus up to 63 AC tokens: 64 words per block when every coefficient of every
  // block is non-zero (the slot refuses a token that would not fit, below).
  st->tok_cap = (size_t)st->nfrags * 64;
  const size_t ngroups = ((size_t)st->nfrags + THIP_SLOT_GROUP - 1) / THIP_SLOT_GROUP;
  // each buffer on its own: a failed allocation leaves the others to a later call, and nothing is used before all exist
  if (!st->h_tok) HIP_TRY(hipHostMalloc((void **)&st->h_tok, st->tok_cap * 4 + 64 * 4, hipHostMallocDefault));
  if (!st->h_slot_tok) HIP_TRY(hipHostMalloc((void **)&st->h_slot_tok, ngroups * 64 * 8, hipHostMallocDefault));
  {
    const int rc = ensure_dq_staging(st);
    if (rc) return rc;
  }
  if (!st->d_tok) HIP_TRY(hipMalloc((void **)&st->d_tok, st->tok_cap * 4 + 64 * 4));
  if (!st->d_slot_tok) HIP_TRY(hipMalloc((void **)&st->d_slot_tok, ngroups * 64 * 8));
  st->tok_ready = 1;
  return THIP_OK;
}

void thip_pack_dequant_table(uint16_t out[64], const uint16_t zz[64]) {
  uint16_t nat[64];
  for (int i = 0; i < 64; i++) nat[fzig_zag(i)] = zz[i];
  for (int j = 0; j < 4; j++)
    for (int c = 0; c < 8; c++)
      for (int p = 0; p < 2; p++) out[(j * 8 + c) * 2 + p] = nat[(2 * j + p) * 8 + c];
}

int thip_frame_dequant_table(thip_state *st, int sel, const uint16_t dequant[64]) {
  if (!st || !dequant) return THIP_EFAULT;
  if (!st->enq_active || sel < 0 || sel >= 18) return THIP_EINVAL;
  if (!st->h_dq || !st->d_dq || !st->h_dqp || !st->d_dqp) {
    DeviceGuard dg(st->device);
    const int rc = ensure_dq_staging(st);
    if (rc) return rc;
  }
  memcpy(st->h_dq + sel * 64, dequant, 128);
  thip_pack_dequant_table(st->h_dqp + sel * 64, dequant);
  return THIP_OK;
}

// The levels form of the oc_state_frag_recon slot: dct_coeffs holds the quantised LEVELS as the tokens carry them (natural order,
// [0] the raw DC as ever), qii = frags[fragi].qii; the frame's tables come through thip_frame_dequant_table and
// `(ogg_int16_t)(coeff*ac_quant[zzi])` (decode.c:1573) happens in the reconstruction kernel.  64 bytes of staging per block
// instead of 128 (the kernels read the staging across PCIe); a tile turns wide -- int16 units -- with its first level beyond
// eight bits, and the units it has been given so far are rewritten in place (tiles arrive contiguously, so nothing lies behind them).
static inline uint8_t *unit_piece_host(int16_t *base, uint32_t unit, int q) {
  return reinterpret_cast<uint8_t *>(base) + (size_t)(unit >> 6) * THIP_UNIT_GROUP_BYTES + (size_t)q * 1024 + (size_t)(unit & 63) * 16;
}
static void pack_wide_block(int16_t *base, uint32_t unit, const int16_t lv[64]) {
  typedef int16_t v8s __attribute__((vector_size(16)));
  for (int j = 0; j < 4; j++) {
    v8s a, b;
    memcpy(&a, lv + (2 * j) * 8, 16);
    memcpy(&b, lv + (2 * j + 1) * 8, 16);
    const v8s lo = __builtin_shufflevector(a, b, 0, 8, 1, 9, 2, 10, 3, 11);
    const v8s hi = __builtin_shufflevector(a, b, 4, 12, 5, 13, 6, 14, 7, 15);
    memcpy(unit_piece_host(base, unit + (uint32_t)((2 * j) >> 2), (2 * j) & 3), &lo, 16);
    memcpy(unit_piece_host(base, unit + (uint32_t)((2 * j + 1) >> 2), (2 * j + 1) & 3), &hi, 16);
  }
}
static void pack_narrow_block(int16_t *base, uint32_t unit, const int16_t lv[64]) {
  typedef int16_t v8s __attribute__((vector_size(16)));
  typedef int8_t v16c __attribute__((vector_size(16)));
  for (int j = 0; j < 4; j++) {
    v8s a, b;
    memcpy(&a, lv + (2 * j) * 8, 16);
    memcpy(&b, lv + (2 * j + 1) * 8, 16);
    const v16c ab = __builtin_convertvector(__builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15), v16c);   // a0..a7 b0..b7 (the levels fit)
    // dword d: a[2d], a[2d+1], b[2d], b[2d+1]
    const v16c o = __builtin_shufflevector(ab, ab, 0, 1, 8, 9, 2, 3, 10, 11, 4, 5, 12, 13, 6, 7, 14, 15);
    memcpy(unit_piece_host(base, unit, j), &o, 16);
  }
}
static void unpack_narrow_block(const int16_t *base, uint32_t unit, int16_t lv[64]) {
  for (int j = 0; j < 4; j++) {
    const int8_t *p = reinterpret_cast<const int8_t *>(unit_piece_host(const_cast<int16_t *>(base), unit, j));
    for (int d = 0; d < 4; d++) {
      lv[(2 * j) * 8 + 2 * d] = p[4 * d];
      lv[(2 * j) * 8 + 2 * d + 1] = p[4 * d + 1];
      lv[(2 * j + 1) * 8 + 2 * d] = p[4 * d + 2];
      lv[(2 * j + 1) * 8 + 2 * d + 1] = p[4 * d + 3];
    }
  }
}

int thip_state_frag_recon_levels(thip_state *st, ptrdiff_t fragi, int pli, int16_t dct_coeffs[128], int last_zzi, uint16_t dc_quant,
                                 int qii, int refi, int16_t mv) {
  if (!st || !dct_coeffs) return THIP_EFAULT;
  if (!st->enq_active || fragi < 0 || fragi >= st->nfrags || pli < 0 || pli > 2 || refi < 0 || refi > 2 || qii < 0 || qii > 2 ||
      last_zzi < 0 || last_zzi > 64 || (int64_t)st->enq_ncoded + st->enq_nuncoded >= st->nfrags)
    return THIP_EINVAL;
  const int32_t pos = st->frag_pos[fragi];
  if (st->h_info[2 * (size_t)pos] & THIP_INFO_CODED) return THIP_EINVAL;   // fragment enqueued twice
  uint32_t flags = THIP_INFO_CODED | ((uint32_t)refi << THIP_INFO_REFI_SHIFT) | ((uint32_t)qii << THIP_INFO_QII_SHIFT) |
                   ((uint32_t)last_zzi << THIP_INFO_LAST_ZZI_SHIFT) |
                   ((uint32_t)(uint8_t)(mv & 0xFF) << THIP_INFO_MVX_SHIFT) |
                   ((uint32_t)(uint8_t)((mv >> 8) & 0xFF) << THIP_INFO_MVY_SHIFT);
  const uint32_t word1 = (uint32_t)dc_quant << 16 | (uint32_t)(uint16_t)dct_coeffs[0];   // the raw DC of every block rides here
  if (last_zzi < 2) {
    flags |= THIP_INFO_DC_ONLY;   // no unit
    st->enq_levels_hint = 1;
  } else {
    if (st->enq_dense_slots || st->enq_tok_slots) return THIP_EINVAL;   // one form per frame for the blocks that own a slot
    const int tile = pos / THIP_TILE_FRAGS, lane = pos % THIP_TILE_FRAGS;
    if (lane <= st->enq_last_lane[tile]) return THIP_EINVAL;
    if (st->enq_last_lane[tile] >= 0 && st->enq_last_tile != tile) return THIP_EINVAL;   // a tile's units must be contiguous
    // (every refusal is above this line)
    if (st->enq_last_lane[tile] < 0) {
      st->h_slot0[tile] = (uint32_t)st->enq_nslots;
      st->enq_tile_blocks = 0;
    }
    st->enq_last_lane[tile] = lane;
    st->enq_last_tile = tile;
    st->enq_level_slots++;
    int16_t keep0 = dct_coeffs[0];
    dct_coeffs[0] = 0;
    bool big;
    {
      typedef int16_t v8s __attribute__((vector_size(16)));
      typedef uint16_t v8u __attribute__((vector_size(16)));
      v8u worst = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int r = 0; r < 8; r++) {
        v8s a;
        memcpy(&a, dct_coeffs + 8 * r, 16);
        const v8u b = (v8u)(a + 127);          // 0..254 for the levels that fit eight bits
        worst = worst > b ? worst : b;
      }
      uint16_t w8[8];
      memcpy(w8, &worst, 16);
      uint16_t m = 0;
      for (int i = 0; i < 8; i++) m = m > w8[i] ? m : w8[i];
      big = m > 254;
    }
    bool wide = (st->h_slot0[tile] & THIP_SLOT_WIDE) != 0;
    const uint32_t u0 = st->h_slot0[tile] & ~THIP_SLOT_WIDE;
    if (big && !wide) {   // the tile turns wide: its narrow units so far become pairs of int16 units, last block first
      for (int i = st->enq_tile_blocks - 1; i >= 0; i--) {
        int16_t lv[64];
        unpack_narrow_block(st->h_coeffs, u0 + (uint32_t)i, lv);
        pack_wide_block(st->h_coeffs, u0 + 2u * (uint32_t)i, lv);
      }
      st->h_slot0[tile] |= THIP_SLOT_WIDE;
      st->enq_nslots = (int)(u0 + 2u * (uint32_t)st->enq_tile_blocks);
      wide = true;
    }
    if (wide) {
      pack_wide_block(st->h_coeffs, (uint32_t)st->enq_nslots, dct_coeffs);
      st->enq_nslots += 2;
    } else {
      pack_narrow_block(st->h_coeffs, (uint32_t)st->enq_nslots, dct_coeffs);
      st->enq_nslots += 1;
    }
    st->enq_tile_blocks++;
    dct_coeffs[0] = keep0;
  }
  if (st->enq_device_dc) {
    st->h_dc[fragi] = dct_coeffs[0];
    st->h_flags[fragi] = (uint8_t)(1u | (uint32_t)refi << 1);
  }
  memset(dct_coeffs, 0, 64 * sizeof(int16_t));   // idct.c:245,276,295
  st->h_info[2 * (size_t)pos] = flags;
  st->h_info[2 * (size_t)pos + 1] = word1;
  st->enq_ncoded++;
  return THIP_OK;
}

int thip_state_frag_recon_tokens(thip_state *st, ptrdiff_t fragi, int pli, const uint32_t *toks, int ntoks, int16_t dc,
                                 int last_zzi, uint16_t dc_quant, int dqsel, int refi, int16_t mv) {
  if (!st || (!toks && ntoks)) return THIP_EFAULT;
  if (!st->enq_active || fragi < 0 || fragi >= st->nfrags || pli < 0 || pli > 2 || refi < 0 || refi > 2 ||
      last_zzi < 0 || last_zzi > 64 || ntoks < 0 || ntoks > 63 || dqsel < 0 || dqsel >= 18 ||
      (int64_t)st->enq_ncoded + st->enq_nuncoded >= st->nfrags)
    return THIP_EINVAL;
  const int32_t pos = st->frag_pos[fragi];
  if (st->h_info[2 * (size_t)pos] & THIP_INFO_CODED) return THIP_EINVAL;   // fragment enqueued twice
  uint32_t flags = THIP_INFO_CODED | ((uint32_t)refi << THIP_INFO_REFI_SHIFT) |
                   ((uint32_t)last_zzi << THIP_INFO_LAST_ZZI_SHIFT) |
                   ((uint32_t)(uint8_t)(mv & 0xFF) << THIP_INFO_MVX_SHIFT) |
                   ((uint32_t)(uint8_t)((mv >> 8) & 0xFF) << THIP_INFO_MVY_SHIFT);
  uint32_t word1 = (uint32_t)dc_quant << 16;
  if (last_zzi < 2) {
    flags |= THIP_INFO_DC_ONLY;
    word1 |= (uint32_t)(uint16_t)dc;
  } else {
    if (!st->tok_ready) {
      DeviceGuard dg(st->device);
      const int rc = ensure_token_staging(st);
      if (rc) return rc;
    }
    if (st->enq_dense_slots || st->enq_level_slots) return THIP_EINVAL;   // (see thip_state_frag_recon: one form per frame for the blocks that own a coefficient slot)
    if ((size_t)st->enq_ntok + (size_t)ntoks + 1 > st->tok_cap) return THIP_EINVAL;   // more tokens than the frame has coefficients
    const int tile = pos / THIP_TILE_FRAGS, lane = pos % THIP_TILE_FRAGS;
    if (lane <= st->enq_last_lane[tile]) return THIP_EINVAL;
    if (st->enq_last_lane[tile] >= 0 && st->enq_last_tile != tile) return THIP_EINVAL;   // a tile's slots must be contiguous
    if (st->enq_last_lane[tile] < 0) st->h_slot0[tile] = (uint32_t)st->enq_nslots;
    st->enq_last_lane[tile] = lane;
    st->enq_last_tile = tile;
    const int slot = st->enq_nslots++;
    st->enq_tok_slots++;
    uint32_t *t = st->h_tok + st->enq_ntok;
    t[0] = (uint32_t)(uint16_t)dc;                       // position 0: the raw DC
    for (int k = 0; k < ntoks; k++) t[1 + k] = toks[k];
    st->h_slot_tok[2 * (size_t)slot] = (uint32_t)st->enq_ntok;
    st->h_slot_tok[2 * (size_t)slot + 1] = (uint32_t)(ntoks + 1) | (uint32_t)dqsel << 8;
    st->enq_ntok += ntoks + 1;
  }
  if (st->enq_device_dc) {
    st->h_dc[fragi] = dc;
    st->h_flags[fragi] = (uint8_t)(1u | (uint32_t)refi << 1);
  }
  st->h_info[2 * (size_t)pos] = flags;
  st->h_info[2 * (size_t)pos + 1] = word1;
  st->enq_ncoded++;
  return THIP_OK;
}

int thip_frag_copy_list(thip_state *st, const ptrdiff_t *fragis, ptrdiff_t nfragis) {
  if (!st || (!fragis && nfragis)) return THIP_EFAULT;
  if (!st->enq_active || nfragis < 0 || (int64_t)st->enq_ncoded + st->enq_nuncoded + nfragis > st->nfrags)
    return THIP_EINVAL;
  for (ptrdiff_t k = 0; k < nfragis; k++) {
    if (fragis[k] < 0 || fragis[k] >= st->nfrags) return THIP_EINVAL;
    // an uncoded fragment is the default state of its command word; nothing to stage
    if (st->h_info[2 * (size_t)st->frag_pos[fragis[k]]] & THIP_INFO_CODED) return THIP_EINVAL;
  }
  st->enq_nuncoded += (int)nfragis;
  return THIP_OK;
}

void thip_loop_filter_init(signed char bv[256], int flimit) {
  // state.c:1036-1045 tabulates lflim(R,flimit) for R in [-127,128]
  for (int i = 0; i < 256; i++) {
    const int R = i - 127;
    const int a = R < 0 ? -R : R;
    int m = 2 * flimit - a;
    if (m < 0) m = 0;
    if (m > a) m = a;
    bv[i] = (signed char)(R < 0 ? -m : m);
  }
}

int thip_state_loop_filter_frag_rows(thip_state *st, int flimit, int refi, int pli, int fragy0,
                                     int fragy_end) {
  if (!st) return THIP_EFAULT;
  if (!st->enq_active || refi != THIP_FRAME_SELF || pli < 0 || pli > 2 || flimit < 0 || flimit > 127)
    return THIP_EINVAL;
  if (fragy0 < 0) fragy0 = 0;
  if (fragy_end > st->geom[pli].nvfrags) fragy_end = st->geom[pli].nvfrags;
  if (fragy_end <= fragy0) return THIP_OK;
  // successive calls extend the row range exactly as the MCU loop does (decode.c:2879-2884)
  if (st->enq_lf_y1[pli] >= 0 && fragy0 != st->enq_lf_y1[pli]) return THIP_EINVAL;
  if (fragy0 < st->enq_lf_y0[pli]) st->enq_lf_y0[pli] = fragy0;
  st->enq_lf_y1[pli] = fragy_end;
  st->enq_flimit = flimit;
  st->enq_lf_any = 1;
  return THIP_OK;
}

int thip_frame_flush(thip_state *st) {
  if (!st) return THIP_EFAULT;
  if (!st->enq_active) return THIP_EINVAL;
  st->enq_active = 0;
  // every fragment must have been reconstructed or copied exactly once
  if (st->enq_ncoded && (int64_t)st->enq_ncoded + st->enq_nuncoded != st->nfrags) return THIP_EINVAL;
  DeviceGuard dg(st->device);
  hipStream_t s;
  int rc = context_stream(st, &s);
  if (rc) return rc;
  rc = order_behind_previous(st, s);   // before the first copy or kernel of this frame goes onto s
  if (rc) return rc;
  const bool levels_frame = st->enq_level_slots != 0 || (st->enq_ncoded && !st->enq_dense_slots && !st->enq_tok_slots && st->h_dqp && st->enq_levels_hint);
  const size_t ngroups = ((size_t)st->enq_nslots + THIP_SLOT_GROUP - 1) / THIP_SLOT_GROUP;   // (groups of 64 slots, or of 64 units)
  const size_t group_bytes = levels_frame ? THIP_UNIT_GROUP_BYTES : THIP_SLOT_GROUP_BYTES;
  const int zerocopy = THIP_OPT("zerocopy");
  if (st->enq_ncoded && !zerocopy) {
    HIP_TRY(hipMemcpyAsync(st->d_info, st->h_info, (size_t)st->tiles.ntiles * THIP_TILE_FRAGS * 8,
                           hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(st->d_slot0, st->h_slot0, (size_t)st->tiles.ntiles * 4, hipMemcpyHostToDevice, s));
    if (ngroups)
      HIP_TRY(hipMemcpyAsync(st->d_coeffs, st->h_coeffs, ngroups * group_bytes, hipMemcpyHostToDevice, s));
  }
  thip_frame_desc d;
  memset(&d, 0, sizeof(d));
  // zero-copy: the kernels read the pinned staging buffers across PCIe themselves (every
  // byte exactly once, coalesced); hipMemcpyAsync of ~1 MB costs more host time than that
  d.frag_info = zerocopy ? st->h_info : st->d_info;
  d.coeffs = zerocopy ? st->h_coeffs : st->d_coeffs;
  d.tile_slot0 = zerocopy ? st->h_slot0 : st->d_slot0;
  d.nslots = st->enq_nslots;
  d.ncoded = st->enq_ncoded;
  d.frame_type = st->enq_frame_type;
  d.flimit = st->enq_lf_any ? st->enq_flimit : 0;
  if (levels_frame) {   // the frame's tables to the device (2.3 KB; every wave of the kernel reads its plane's six)
    HIP_TRY(hipMemcpyAsync(st->d_dqp, st->h_dqp, 18 * 64 * 2, hipMemcpyHostToDevice, s));
    d.coeff_format = THIP_COEFFS_LEVELS;
    d.dequant = st->d_dqp;
  }
  if (st->enq_tok_slots) {
    // tokens -> device, expanded there into the coefficient slots k_recon reads (decode.c:1540-1581)
    HIP_TRY(hipMemcpyAsync(st->d_tok, st->h_tok, (size_t)st->enq_ntok * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(st->d_slot_tok, st->h_slot_tok, (size_t)st->enq_nslots * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(st->d_dq, st->h_dq, 18 * 64 * 2, hipMemcpyHostToDevice, s));
    TokK T;
    T.tok = st->d_tok;
    T.slot_tok = reinterpret_cast<const uint2 *>(st->d_slot_tok);
    T.dq = st->d_dq;
    T.coeffs = reinterpret_cast<int4 *>(st->d_coeffs);
    T.nslots = st->enq_nslots;
    hipLaunchKernelGGL(k_expand_tokens, dim3((unsigned)((ngroups + 3) / 4)), dim3(256), 0, s, T);
    HIP_TRY(hipGetLastError());
    d.coeffs = st->d_coeffs;
  }
  if (st->enq_device_dc && st->enq_ncoded) {
    // (the wavefront kernel reads a value per thread per step: a device copy, not reads across PCIe)
    HIP_TRY(hipMemcpyAsync(st->d_dc_in, st->h_dc, sizeof(int16_t) * (size_t)st->nfrags, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(st->d_flags, st->h_flags, (size_t)st->nfrags, hipMemcpyHostToDevice, s));
    d.dc_tokens = st->d_dc_in;
    st->flush_flags = 1;
  }
  st->lf_rows_custom = st->enq_lf_any;
  for (int p = 0; p < 3; p++) {
    st->lf_y0[p] = st->enq_lf_y1[p] < 0 ? 0 : st->enq_lf_y0[p];
    st->lf_y1[p] = st->enq_lf_y1[p] < 0 ? 0 : st->enq_lf_y1[p];
  }
  int32_t res = 0;
  thip_state *sp = st;
  st->redo_owned = 1;   // (the descriptor points into this state's staging buffers, intact until the next thip_frame_begin)
  rc = thip_decode_frames(&sp, &d, 1, (void *)s, &res);
  st->redo_owned = 0;
  st->lf_rows_custom = 0;
  st->flush_flags = 0;
  if (rc < 0) return rc;
  if (!st->ev_staging) HIP_TRY(hipEventCreateWithFlags(&st->ev_staging, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(st->ev_staging, s));
  st->staging_serial = st->frame_serial;
  return res;
}

// ---- the frame's token lists, expanded on the device (thip_tokens.h) ---------------------------------------
// staging layout of the token-list path (dwords; 16-byte sections): header tables | coded list | fragment words | the caller's DC values |
// dequantisation tables | tokens
struct TlLayout {
  size_t nf, o_cl, o_meta, o_dcv, o_dq, o_tok;
  int pos_pitch;
};
static TlLayout tl_layout(const thip_state *st) {
  TlLayout y;
  y.nf = ((size_t)st->nfrags + 7) & ~(size_t)7;   // (whole 16-byte units of every element size used)
  y.o_cl = THIP_TL_HDR;
  y.o_meta = y.o_cl + y.nf;
  y.o_dcv = y.o_meta + y.nf;
  y.o_dq = y.o_dcv + y.nf / 2;
  y.o_tok = y.o_dq + 18 * 64 / 2;
  y.pos_pitch = (int)(((size_t)std::min<int64_t>(st->nfrags, kTlMaxFrags) + 31 + 15) & ~(size_t)15);
  return y;
}
static int64_t tl_token_capacity(const thip_state *st) { return (int64_t)st->nfrags * 64 + 192 + 192; }   // (+ the groups' padding to 16-byte units)
static int tl_ensure(thip_state *st) {
  if (st->tl_ready) return THIP_OK;
  const TlLayout y = tl_layout(st);
  // (each buffer on its own: a failed allocation is retried by the next call, nothing is used before all exist)
  st->tl_cap = (y.o_tok + (size_t)tl_token_capacity(st) + 8) * 4;
  for (int k = 0; k < 2; k++)
    if (!st->h_tl_buf[k]) HIP_TRY(hipHostMalloc((void **)&st->h_tl_buf[k], st->tl_cap, hipHostMallocDefault));
  st->h_tl = st->h_tl_buf[st->tl_buf];
  if (!st->d_tl) HIP_TRY(hipMalloc((void **)&st->d_tl, st->tl_cap));
  if (!st->d_tl_tmp) HIP_TRY(hipMalloc((void **)&st->d_tl_tmp, (size_t)st->nfrags * 128));
  if (!st->d_tl_last) HIP_TRY(hipMalloc((void **)&st->d_tl_last, y.nf));
  if (!st->d_tl_slot) HIP_TRY(hipMalloc((void **)&st->d_tl_slot, y.nf * 4));
  if (!st->d_tl_arr) HIP_TRY(hipMalloc((void **)&st->d_tl_arr, y.nf * 4));
  if (!st->d_tl_pos) HIP_TRY(hipMalloc((void **)&st->d_tl_pos, (size_t)y.pos_pitch * 3));
  if (!st->d_tl_wide)   // (+ k_tok_slots_count's sums, one per kTlSlotChunk fragments)
    HIP_TRY(hipMalloc((void **)&st->d_tl_wide, ((((size_t)st->tiles.ntiles + 3) & ~(size_t)3) + (size_t)st->nfrags / kTlSlotChunk + 4) * 4));
  if (!st->d_frag_pos) HIP_TRY(hipMalloc((void **)&st->d_frag_pos, y.nf * 4));
  if (!st->d_dc_in) HIP_TRY(hipMalloc((void **)&st->d_dc_in, sizeof(int16_t) * y.nf));
  HIP_TRY(hipMemcpy(st->d_frag_pos, st->frag_pos, (size_t)st->nfrags * 4, hipMemcpyHostToDevice));
  st->tl_ready = 1;
  return THIP_OK;
}

// The state's own staging buffer, for a caller that writes the frame's arrays where the device will read them instead of having
// them copied there: valid from this call -- which waits until the device is done with the previous frame's contents -- to the
// frame's _finish / _abort.
int thip_state_token_lists_staging(thip_state *st, thip_token_staging *out) {
  if (!st || !out) return THIP_EFAULT;
  if (st->enq_active || st->tl_pending) return THIP_EINVAL;
  for (int p = 0; p < 3; p++)
    if (st->geom[p].nvfrags > kDcMaxRows) return THIP_EIMPL;
  DeviceGuard dg(st->device);
  int rc = tl_ensure(st);
  if (rc) return rc;
  if (tl_take_staging(st) < 0) return THIP_EFAULT;
  const TlLayout y = tl_layout(st);
  out->coded = reinterpret_cast<int32_t *>(st->h_tl + y.o_cl);
  out->frag_meta = st->h_tl + y.o_meta;
  out->dequant = reinterpret_cast<uint16_t *>(st->h_tl + y.o_dq);
  out->tokens = st->h_tl + y.o_tok;
  out->token_capacity = tl_token_capacity(st);
  st->tl_claimed = 1;
  return THIP_OK;
}

// _open: everything about the frame but its tokens -- checked, staged, the device's work arrays zeroed; _append: the lists of
// the indices [z0, z1), which k_tok_assign walks at once; _finish: the DC values, the command words, the reconstruction.
int thip_state_token_lists_open(thip_state *st, const thip_token_lists *tl) {
  if (!st || !tl) return THIP_EFAULT;
  if (st->enq_active || st->tl_pending) return THIP_EINVAL;   // a frame is being enqueued through the slots, or one is waiting for its finish
  const int claimed = st->tl_claimed;   // (thip_state_token_lists_staging has waited for the buffer already; one frame's worth)
  st->tl_claimed = 0;
  if (tl->frame_type != THIP_INTRA_FRAME && tl->frame_type != THIP_INTER_FRAME) return THIP_EINVAL;
  if (tl->flimit < 0 || tl->flimit > 127) return THIP_EINVAL;
  int64_t ncoded = 0;
  for (int p = 0; p < 3; p++) {
    const thip_plane_geom &g = st->geom[p];
    if (tl->ncoded[p] < 0 || tl->ncoded[p] > g.nhfrags * g.nvfrags) return THIP_EINVAL;
    if (tl->ncoded[p] > kTlMaxFrags || g.nvfrags > kDcMaxRows) return THIP_EIMPL;
    ncoded += tl->ncoded[p];
  }
  if (ncoded && (!tl->coded || !tl->frag_meta || !tl->dequant)) return THIP_EFAULT;
  {
    int64_t c = 0;
    for (int p = 0; p < 3; p++) {
      const thip_plane_geom &g = st->geom[p];
      const int64_t lo = g.froffset, hi = lo + (int64_t)g.nhfrags * g.nvfrags;
      int32_t prev_pos = -1;
      for (int64_t i = 0; i < tl->ncoded[p]; i++, c++) {
        const int32_t f = tl->coded[c];
        if (f < lo || f >= hi) return THIP_EINVAL;
        const int32_t pos = st->frag_pos[f];
        if (pos <= prev_pos) return THIP_EINVAL;                 // coded order == tile/lane order, no fragment twice
        prev_pos = pos;
        const uint32_t m = tl->frag_meta[c];
        if ((m & 3u) > 2u || ((m >> 2) & 31u) >= 18u || ((m >> 24) & 3u) != (uint32_t)p) return THIP_EINVAL;
      }
    }
  }
  DeviceGuard dg(st->device);
  hipStream_t s;
  int rc = context_stream(st, &s);
  if (rc) return rc;
  rc = order_behind_previous(st, s);   // before the first copy or kernel of this frame goes onto s
  if (rc) return rc;
  rc = ensure_staging(st);
  if (rc) return rc;
  thip_frame_desc d;
  memset(&d, 0, sizeof(d));
  d.frame_type = tl->frame_type;
  d.flimit = tl->flimit;
  if (ncoded) {
    const TlLayout y = tl_layout(st);
    const size_t nf = y.nf, o_cl = y.o_cl, o_meta = y.o_meta, o_dcv = y.o_dcv, o_dq = y.o_dq, o_tok = y.o_tok;
    const int pos_pitch = y.pos_pitch;
    rc = tl_ensure(st);
    if (rc) return rc;
    // the previous frame's kernels must have read the staging buffer before it is reused (thip_state_token_lists_staging has
    // seen to that if the caller went through it)
    if (!claimed && tl_take_staging(st) < 0) return THIP_EFAULT;
    uint32_t *h = st->h_tl;
    memset(h, 0, THIP_TL_HDR * 4);
    for (int p = 0; p < 3; p++)
      for (int q = 0; q < 2; q++) h[THIP_TL_DCQ + p * 2 + q] = tl->dc_quant[p][q];
    // (arrays the caller wrote into the staging buffer itself are where they belong)
    if ((const void *)tl->coded != (const void *)(h + o_cl)) memcpy(h + o_cl, tl->coded, (size_t)ncoded * 4);
    if ((const void *)tl->frag_meta != (const void *)(h + o_meta)) memcpy(h + o_meta, tl->frag_meta, (size_t)ncoded * 4);
    const int levels = THIP_OPT("tl_levels") != 0;
    if (levels) {   // the tables in the order the reconstruction kernel reads them (thip_pack_dequant_table), where the zig-zag ones would go
      uint16_t zz[18 * 64];
      memcpy(zz, tl->dequant, sizeof(zz));   // (the caller may have written them into the staging buffer itself)
      uint16_t *hp = reinterpret_cast<uint16_t *>(h + o_dq);
      for (int t = 0; t < 18; t++) thip_pack_dequant_table(hp + t * 64, zz + t * 64);
    } else if ((const void *)tl->dequant != (const void *)(h + o_dq)) {
      memcpy(h + o_dq, tl->dequant, 18 * 64 * 2);
    }
    const size_t npos = (size_t)st->tiles.ntiles * THIP_TILE_FRAGS;
    TlPrepK P;
    P.src = reinterpret_cast<const int4 *>(st->h_tl);
    P.dst = reinterpret_cast<int4 *>(st->d_tl);
    P.ncopy = o_tok / 4;
    P.z[0] = reinterpret_cast<int4 *>(st->d_tl_tmp);
    P.nz[0] = (size_t)ncoded * 8;
    P.z[1] = reinterpret_cast<int4 *>(st->d_info);
    P.nz[1] = npos / 2;
    P.z[2] = reinterpret_cast<int4 *>(st->d_slot0);
    P.nz[2] = ((size_t)st->tiles.ntiles + 3) / 4;
    P.z[3] = reinterpret_cast<int4 *>(st->d_dc_in);
    P.nz[3] = nf / 8;
    P.z[4] = reinterpret_cast<int4 *>(st->d_tl_wide);
    P.nz[4] = ((size_t)st->tiles.ntiles + 3) / 4;
    hipLaunchKernelGGL(k_tok_prepare, dim3(256), dim3(256), 0, s, P);
    TlK K;
    memset(&K, 0, sizeof(K));
    K.hdr = st->d_tl;
    K.clist = reinterpret_cast<const int32_t *>(st->d_tl + o_cl);
    K.meta = st->d_tl + o_meta;
    K.dq = reinterpret_cast<const uint16_t *>(st->d_tl + o_dq);
    K.tok = st->d_tl + o_tok;
    K.frag_pos = st->d_frag_pos;
    K.tmp = st->d_tl_tmp;
    K.last_zzi = st->d_tl_last;
    K.slot = st->d_tl_slot;
    K.arr = st->d_tl_arr;
    K.dc_in = st->d_dc_in;
    K.dc_host = nullptr;   // (thip_state_token_lists_finish sets it)
    K.info = st->d_info;
    K.slot0 = st->d_slot0;
    K.coeffs = reinterpret_cast<int4 *>(st->d_coeffs);
    K.ncoded = (int)ncoded;
    K.pos_save = st->d_tl_pos;
    K.pos_pitch = pos_pitch;
    K.levels = levels;
    K.wide = st->d_tl_wide;
    int c0 = 0;
    for (int p = 0; p < 3; p++) {
      K.p[p].n = tl->ncoded[p];
      K.p[p].c0 = c0;
      c0 += tl->ncoded[p];
    }
    HIP_TRY(hipGetLastError());
    d.frag_info = st->d_info;
    d.coeffs = st->d_coeffs;
    d.tile_slot0 = st->d_slot0;
    d.nslots = (int)ncoded * (levels ? 2 : 1);   // (an upper bound: the device knows the number)
    d.ncoded = (int)ncoded;
    if (levels) {
      d.coeff_format = THIP_COEFFS_LEVELS;
      d.dequant = reinterpret_cast<const uint16_t *>(st->d_tl + o_dq);
    }
    st->tl_K = K;
    st->tl_o_dcv = o_dcv;
    st->tl_o_tok = o_tok;
  }
  st->tl_desc = d;
  st->tl_ncoded = ncoded;
  st->tl_stream = s;
  st->tl_pending = 1;
  st->tl_z = 0;
  st->tl_ntok = 0;
  st->tl_rank_at = 0;
  return THIP_OK;
}

int thip_state_token_lists_append(thip_state *st, int z0, int z1, const uint32_t *tokens, int64_t ntokens, const uint32_t (*list_off)[64],
                                  const uint32_t (*list_len)[64], const uint32_t (*eob_carry)[64], const uint32_t (*arrivals)[64]) {
  if (!st) return THIP_EFAULT;
  if (!st->tl_pending || z0 != st->tl_z || z1 <= z0 || z1 > 64 || ntokens < 0) return THIP_EINVAL;   // the groups come in order, without gaps
  if (!list_off || !list_len || !eob_carry || !arrivals || (ntokens && !tokens)) return THIP_EFAULT;
  const int64_t ncoded = st->tl_ncoded;
  if (!ncoded) {
    st->tl_z = z1;
    return THIP_OK;
  }
  const TlK &K0 = st->tl_K;
  // the lists lie inside the token array; what they consume is the device's business (a list that asks for more
  // than it has finds its fragments ended, a longer one has its surplus ignored)
  if ((((int64_t)st->tl_ntok + 3) & ~(int64_t)3) + ntokens > tl_token_capacity(st)) return THIP_EINVAL;
  for (int p = 0; p < 3; p++)
    for (int z = z0; z < z1; z++) {
      if ((int64_t)list_off[p][z] + list_len[p][z] > ntokens) return THIP_EINVAL;
      if (arrivals[p][z] > (uint32_t)K0.p[p].n || eob_carry[p][z] > arrivals[p][z]) return THIP_EINVAL;
    }
  DeviceGuard dg(st->device);
  hipStream_t s = st->tl_stream;
  uint32_t *h = st->h_tl;
  const size_t at = ((size_t)st->tl_ntok + 3) & ~(size_t)3;   // the group starts on a 16-byte unit of the token area
  int algo = THIP_OPT("tl_algo");
  if (algo != 1 && algo != 2) algo = std::max(std::max(K0.p[0].n, K0.p[1].n), K0.p[2].n) > kTlLdsFrags ? 2 : 1;
  if (algo == 2) {
    int64_t need = st->tl_rank_at;
    for (int p = 0; p < 3; p++)
      for (int z = z0; z < z1; z++) need += arrivals[p][z];
    if (need > (int64_t)st->nfrags * 64) return THIP_EINVAL;   // (a fragment arrives at an index once)
    if (!st->d_tl_rank) HIP_TRY(hipMalloc((void **)&st->d_tl_rank, (size_t)st->nfrags * 64 * 4));
    for (int z = z0; z < z1; z++)
      for (int p = 0; p < 3; p++) {
        h[THIP_TL_ROFF + p * 64 + z] = (uint32_t)st->tl_rank_at;
        st->tl_rank_at += arrivals[p][z];
      }
  }
  for (int p = 0; p < 3; p++)
    for (int z = z0; z < z1; z++) {
      h[THIP_TL_OFF + p * 64 + z] = (uint32_t)at + list_off[p][z];
      h[THIP_TL_LEN + p * 64 + z] = list_len[p][z];
      h[THIP_TL_CARRY + p * 64 + z] = eob_carry[p][z];
      h[THIP_TL_ARRIVE + p * 64 + z] = arrivals[p][z];
    }
  if (ntokens && tokens != h + st->tl_o_tok + at) memcpy(h + st->tl_o_tok + at, tokens, (size_t)ntokens * 4);   // (not if the caller wrote them there)
  // INVARIANT: the whole header travels with every group, and the host goes on to write the NEXT group's columns
  // [z1, ...) of the same pinned header while this copy may still be reading it.  That is harmless only because (a) a group's kernels
  // read nothing but their own columns [z0, z1) of the tables OFF / LEN / CARRY / ARRIVE / ROFF and the frame-constant words, which
  // are final when this launch is issued, and (b) the next group's copy brings the header again, its own columns final by then.
  // A kernel that reads another group's column must not be added without copying columns per group (or a second header).
  TlCopyK C;
  C.src[0] = reinterpret_cast<const int4 *>(h);
  C.dst[0] = reinterpret_cast<int4 *>(st->d_tl);
  C.n[0] = THIP_TL_HDR / 4;
  C.src[1] = reinterpret_cast<const int4 *>(h + st->tl_o_tok + at);
  C.dst[1] = reinterpret_cast<int4 *>(st->d_tl + st->tl_o_tok + at);
  C.n[1] = ((size_t)ntokens + 3) / 4;
  const unsigned cgroups = (unsigned)std::min<size_t>(256, (C.n[1] + 255) / 256 + 1);
  hipLaunchKernelGGL(k_tok_copy, dim3(cgroups), dim3(256), 0, s, C);
  TlK K = K0;
  K.z0 = z0;
  K.z1 = z1;
  int nmax = 0;
  for (int p = 0; p < 3; p++) nmax = std::max(nmax, K.p[p].n);
  if (algo == 2) {
    K.rank = st->d_tl_rank;
    hipLaunchKernelGGL(k_tok_rank, dim3(3 * (unsigned)(z1 - z0)), dim3(kTlRankThreads), 0, s, K);
    int T = THIP_OPT("tl_walk_threads");
    if (T != 256 && T != 512 && T != 1024) T = nmax <= 2048 ? 256 : (nmax <= 8192 ? 512 : 1024);   // (a round is one barrier: waves are cheap, arrivals per thread are not)
    while (T < 1024 && ((((nmax + 31) & ~31) >> 5) + T - 1) / T > kTlGroups) T *= 2;   // (a thread looks after kTlGroups x 32 fragments at most)
    const size_t lds = (size_t)((nmax + 31) & ~31) + 16;
    if (T == 256) {
      HIP_TRY(set_dynamic_lds(reinterpret_cast<const void *>(k_tok_walk<256>), kTlMaxFrags + 32, 4));
      hipLaunchKernelGGL(k_tok_walk<256>, dim3(3), dim3(256), lds, s, K);
    } else if (T == 512) {
      HIP_TRY(set_dynamic_lds(reinterpret_cast<const void *>(k_tok_walk<512>), kTlMaxFrags + 32, 5));
      hipLaunchKernelGGL(k_tok_walk<512>, dim3(3), dim3(512), lds, s, K);
    } else {
      HIP_TRY(set_dynamic_lds(reinterpret_cast<const void *>(k_tok_walk<1024>), kTlMaxFrags + 32, 6));
      hipLaunchKernelGGL(k_tok_walk<1024>, dim3(3), dim3(1024), lds, s, K);
    }
  } else if (nmax <= kTlLdsFrags) {
    const int lds = 2 * ((nmax + 31) & ~31) + 2 * nmax + 16;
    HIP_TRY(set_dynamic_lds(reinterpret_cast<const void *>(k_tok_assign<false>), 2 * ((kTlLdsFrags + 31) & ~31) + 2 * kTlLdsFrags + 16, 2));
    hipLaunchKernelGGL(k_tok_assign<false>, dim3(3), dim3(tl_threads(nmax)), (size_t)lds, s, K);
  } else {   // (4K luma: the rank -> fragment map in memory, the positions alone in LDS)
    const int lds = ((nmax + 31) & ~31) + 16;
    HIP_TRY(set_dynamic_lds(reinterpret_cast<const void *>(k_tok_assign<true>), kTlMaxFrags + 32, 1));
    hipLaunchKernelGGL(k_tok_assign<true>, dim3(3), dim3(tl_threads(nmax)), (size_t)lds, s, K);
  }
  if (z1 == 64) {
    if (K.levels) hipLaunchKernelGGL(k_tok_widths, dim3((unsigned)((ncoded + 255) / 256)), dim3(256), 0, s, K);
    if (ncoded <= 4 * kTlSlotChunk) {
      hipLaunchKernelGGL(k_tok_slots, dim3(1), dim3(1024), 0, s, K);
    } else {   // (large frames: many groups, two launches)
      const unsigned ng = (unsigned)((ncoded + kTlSlotChunk - 1) / kTlSlotChunk);
      uint32_t *part = st->d_tl_wide + (((size_t)st->tiles.ntiles + 3) & ~(size_t)3);   // (behind the tiles' words)
      hipLaunchKernelGGL(k_tok_slots_count, dim3(ng), dim3(1024), 0, s, K, part);
      hipLaunchKernelGGL(k_tok_slots_assign, dim3(ng), dim3(1024), 0, s, K, (const uint32_t *)part);
    }
  }
  HIP_TRY(hipGetLastError());
  st->tl_z = z1;
  st->tl_ntok = (int64_t)at + ntokens;
  mark_dirty(st->device, s);   // (behind the launches as well as in front of them: see launch_frame_out)
  return THIP_OK;
}

// A frame opened and not to be finished (a later group failed its checks): the state is where it was before _open.
int thip_state_token_lists_abort(thip_state *st) {
  if (!st) return THIP_EFAULT;
  if (!st->tl_pending) return THIP_EINVAL;
  st->tl_pending = 0;
  if (st->tl_ncoded) {   // the launches so far read the staging buffer
    DeviceGuard dg(st->device);
    (void)hipStreamSynchronize(st->tl_stream);
  }
  return THIP_OK;
}

int thip_state_token_lists_begin(thip_state *st, const thip_token_lists *tl) {
  if (!st || !tl) return THIP_EFAULT;
  if (tl->ntokens < 0) return THIP_EINVAL;
  int64_t ncoded = 0;
  for (int p = 0; p < 3; p++) ncoded += tl->ncoded[p] > 0 ? tl->ncoded[p] : 0;
  if (ncoded && !tl->tokens) return THIP_EFAULT;
  int rc = thip_state_token_lists_open(st, tl);
  if (rc < 0) return rc;
  rc = thip_state_token_lists_append(st, 0, 64, tl->tokens, tl->ntokens, tl->list_off, tl->list_len, tl->eob_carry, tl->arrivals);
  if (rc < 0) (void)thip_state_token_lists_abort(st);
  return rc;
}

// _begin for a caller that has walked the lists itself (k_tok_scatter, thip_tokens.h): `assign` one word per token, `last_zzi` one byte
// per coded fragment.  THIP_EIMPL when the two do not fit behind the tokens in the staging buffer (more than half its token area
// used: frames of nearly 32 tokens a fragment) -- _begin takes the frame then; the state is as it was.
int thip_state_token_lists_begin_assigned(thip_state *st, const thip_token_lists *tl, const uint32_t *assign, const uint8_t *last_zzi) {
  if (!st || !tl) return THIP_EFAULT;
  if (tl->ntokens < 0) return THIP_EINVAL;
  int64_t ncoded = 0;
  for (int p = 0; p < 3; p++) ncoded += tl->ncoded[p] > 0 ? tl->ncoded[p] : 0;
  if (ncoded && (!tl->tokens || !assign || !last_zzi)) return THIP_EFAULT;
  if (ncoded > 0x3FFFF) return THIP_EIMPL;   // (eighteen bits of fragment index)
  const int64_t ntok = tl->ntokens;
  const int64_t a_asg = (ntok + 3) & ~(int64_t)3, a_lz = (a_asg + ntok + 3) & ~(int64_t)3, a_end = a_lz + ((ncoded + 15) / 16) * 4;
  if (a_end > tl_token_capacity(st)) return THIP_EIMPL;
  int rc = thip_state_token_lists_open(st, tl);
  if (rc < 0) return rc;
  if (!ncoded) {
    st->tl_z = 64;
    return THIP_OK;
  }
  DeviceGuard dg(st->device);
  hipStream_t s = st->tl_stream;
  uint32_t *h = st->h_tl + st->tl_o_tok;
  if (ntok && tl->tokens != h) memcpy(h, tl->tokens, (size_t)ntok * 4);
  if (ntok) memcpy(h + a_asg, assign, (size_t)ntok * 4);
  memcpy(h + a_lz, last_zzi, (size_t)ncoded);
  TlCopyK C;
  C.src[0] = reinterpret_cast<const int4 *>(st->h_tl);
  C.dst[0] = reinterpret_cast<int4 *>(st->d_tl);
  C.n[0] = THIP_TL_HDR / 4;
  C.src[1] = reinterpret_cast<const int4 *>(h);
  C.dst[1] = reinterpret_cast<int4 *>(st->d_tl + st->tl_o_tok);
  C.n[1] = (size_t)a_end / 4;
  const unsigned cgroups = (unsigned)std::min<size_t>(256, (C.n[1] + 255) / 256 + 1);
  hipLaunchKernelGGL(k_tok_copy, dim3(cgroups), dim3(256), 0, s, C);
  TlK K = st->tl_K;
  K.z0 = 0;
  K.z1 = 64;
  const uint32_t *d_tok = st->d_tl + st->tl_o_tok;
  const int64_t nthreads = std::max<int64_t>(ntok, ncoded);
  hipLaunchKernelGGL(k_tok_scatter, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, s, K, d_tok + a_asg,
                     reinterpret_cast<const uint8_t *>(d_tok + a_lz), (int)ntok);
  if (K.levels) hipLaunchKernelGGL(k_tok_widths, dim3((unsigned)((ncoded + 255) / 256)), dim3(256), 0, s, K);
  if (ncoded <= 4 * kTlSlotChunk) {
    hipLaunchKernelGGL(k_tok_slots, dim3(1), dim3(1024), 0, s, K);
  } else {   // (large frames: many groups, two launches)
    const unsigned ng = (unsigned)((ncoded + kTlSlotChunk - 1) / kTlSlotChunk);
    uint32_t *part = st->d_tl_wide + (((size_t)st->tiles.ntiles + 3) & ~(size_t)3);   // (behind the tiles' words)
    hipLaunchKernelGGL(k_tok_slots_count, dim3(ng), dim3(1024), 0, s, K, part);
    hipLaunchKernelGGL(k_tok_slots_assign, dim3(ng), dim3(1024), 0, s, K, (const uint32_t *)part);
  }
  HIP_TRY(hipGetLastError());
  st->tl_z = 64;
  st->tl_ntok = a_end;
  return THIP_OK;
}

int thip_state_token_lists_finish(thip_state *st, const int16_t *dc) {
  if (!st) return THIP_EFAULT;
  if (!st->tl_pending || st->tl_z != 64) return THIP_EINVAL;   // (lists still to come: _append, or _abort)
  st->tl_pending = 0;
  DeviceGuard dg(st->device);
  hipStream_t s = st->tl_stream;
  thip_frame_desc d = st->tl_desc;
  const int64_t ncoded = st->tl_ncoded;
  if (ncoded) {
    TlK K = st->tl_K;
    if (dc) {   // the caller's un-predicted values: behind k_tok_prepare's copy of the staging buffer, into the same place
      memcpy(st->h_tl + st->tl_o_dcv, dc, (size_t)ncoded * 2);
      if (THIP_OPT("tl_dc_copy") == 1) {
        HIP_TRY(hipMemcpyAsync(st->d_tl + st->tl_o_dcv, st->h_tl + st->tl_o_dcv, (((size_t)ncoded * 2) + 15) & ~(size_t)15, hipMemcpyHostToDevice, s));
      } else {
        // a kernel brings them over, as it brings the tokens: a copy engine's transfer between two kernels of one stream costs 6 us
        // of copy and 18 us of waiting either side of it for 28 KB (profiles/r05_plain_loop_timeline_720p.txt) -- and having the
        // kernel that writes the command words read the pinned buffer itself was slower still (one 2-byte read per eight lanes)
        TlCopyK C;
        C.src[0] = reinterpret_cast<const int4 *>(st->h_tl + st->tl_o_dcv);
        C.dst[0] = reinterpret_cast<int4 *>(st->d_tl + st->tl_o_dcv);
        C.n[0] = ((size_t)ncoded * 2 + 15) / 16;
        C.src[1] = nullptr;
        C.dst[1] = nullptr;
        C.n[1] = 0;
        hipLaunchKernelGGL(k_tok_copy, dim3((unsigned)std::min<size_t>(64, (C.n[0] + 255) / 256)), dim3(256), 0, s, C);
      }
      K.dc_host = reinterpret_cast<const int16_t *>(st->d_tl + st->tl_o_dcv);
    } else {
      K.dc_host = nullptr;
    }
    if (K.levels) hipLaunchKernelGGL(k_tok_write_levels, dim3((unsigned)(((size_t)ncoded * 8 + 255) / 256)), dim3(256), 0, s, K);
    else hipLaunchKernelGGL(k_tok_write, dim3((unsigned)(((size_t)ncoded * 8 + 255) / 256)), dim3(256), 0, s, K);
    HIP_TRY(hipGetLastError());
    d.dc_tokens = dc ? nullptr : st->d_dc_in;
  }
  int32_t res = 0;
  thip_state *sp = st;
  st->redo_owned = 1;
  int rc = thip_decode_frames(&sp, &d, 1, (void *)s, &res);
  st->redo_owned = 0;
  if (rc < 0) return rc;
  if (ncoded) {
    if (!st->ev_staging) HIP_TRY(hipEventCreateWithFlags(&st->ev_staging, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(st->ev_staging, s));
    st->staging_serial = st->frame_serial;
    st->tl_buf_serial[st->tl_buf] = st->frame_serial;
  }
  return res;
}

int thip_state_decode_token_lists(thip_state *st, const thip_token_lists *tl) {
  const int rc = thip_state_token_lists_begin(st, tl);
  if (rc < 0) return rc;
  return thip_state_token_lists_finish(st, tl->dc);
}

}  // extern "C"
