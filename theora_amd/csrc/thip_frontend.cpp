// thip_frontend.cpp -- host front end of the decoder.
//
// Turns Theora packets into calls of the accel-vtable slots implemented by the HIP backend
// (theora_hip.h): header parsing, frame header, coded-block flags, macro-block modes, motion
// vectors, block-level qi, DCT token decode, DC un-prediction, token expansion and
// dequantisation.  Written from the bitstream specification (doc/spec/spec.tex, section
// numbers cited below); the role it plays is the one lib/decode.c plays in the reference
// (th_decode_packetin, decode.c:2740-2986), but none of that code is reproduced.  The one
// structural idea shared with the reference is unpacking the token stream by COUNTS per
// (plane, zig-zag index) list before expanding per fragment (decode.c:993-1201), because
// it is what makes a single pass over the packet possible.
//
// Exports the th_decode_* API declared in include/theoradec_hip.h.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <pthread.h>
#include <sched.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "../../include/theora_hip.h"
#include "../../include/theoradec_hip.h"
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

namespace {

// ---------------------------------------------------------------------------------------
// bit reader: MSB first, reads past the end return zeros (spec 5.2, "Decoding")
// ---------------------------------------------------------------------------------------
struct BitReader {
  const uint8_t *data;
  size_t nbytes, bytepos;   // next byte to pull into the window
  uint64_t win;             // upcoming bits, MSB-aligned
  int have;                 // valid bits in win
  size_t nbits, pos;        // packet size and bits consumed so far (pos may run past nbits)
  BitReader(const uint8_t *d, size_t bytes) : data(d), nbytes(bytes), bytepos(0), win(0), have(0), nbits(bytes * 8), pos(0) {}
  inline void refill() {   // afterwards have >= 57; bytes past the end are zeros
    if (bytepos + 8 <= nbytes) {
      uint64_t v;
      memcpy(&v, data + bytepos, 8);
      v = __builtin_bswap64(v);
      win |= v >> have;
      const int take = (64 - have) >> 3;
      bytepos += (size_t)take;
      have += take * 8;
    } else {
      while (have <= 56) {
        const uint64_t b = bytepos < nbytes ? data[bytepos] : 0;
        bytepos++;
        win |= b << (56 - have);
        have += 8;
      }
    }
  }
  inline uint32_t peek(int n) {   // 1 <= n <= 32, does not consume
    if (have < n) refill();
    return (uint32_t)(win >> (64 - n));
  }
  inline void skip(int n) {       // n <= have
    win <<= n;
    have -= n;
    pos += (size_t)n;
  }
  inline uint32_t read(int n) {   // 0 <= n <= 32
    if (n <= 0) return 0;
    const uint32_t v = peek(n);
    skip(n);
    return v;
  }
  inline uint32_t bit() { return read(1); }
  bool overrun() const { return pos > nbits; }
};

inline int ilog(uint32_t v) {   // number of bits needed to store v (spec 1.4 ilog)
  int n = 0;
  while (v) {
    n++;
    v >>= 1;
  }
  return n;
}

// zig-zag index -> natural position (spec Figure "zig-zag order")
const uint8_t kZigZag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// the same for positions 0..127: everything past 63 goes to entry 64 (a dump slot)
struct ZigZagDump {
  uint8_t t[128];
  ZigZagDump() {
    for (int i = 0; i < 128; i++) t[i] = i < 64 ? kZigZag[i] : 64;
  }
  uint8_t operator[](int i) const { return t[i]; }
};
const ZigZagDump kZigZagDump;

// (row, col) of the k-th block of a super block in coded (Hilbert) order (spec Figure 2.4)
const uint8_t kHilbert[16][2] = {{0, 0}, {0, 1}, {1, 1}, {1, 0}, {2, 0}, {3, 0}, {3, 1}, {2, 1},
                                 {2, 2}, {3, 2}, {3, 3}, {2, 3}, {1, 3}, {1, 2}, {0, 2}, {0, 3}};
// macro blocks of a super block in coded order: (row, col) in units of macro blocks (spec Figure 2.5)
const uint8_t kMbOrder[4][2] = {{0, 0}, {1, 0}, {1, 1}, {0, 1}};

// Table 7.19: mode alphabets of schemes 1..6
const uint8_t kModeAlphabets[6][8] = {{3, 4, 2, 0, 1, 5, 6, 7}, {3, 4, 0, 2, 1, 5, 6, 7}, {3, 2, 4, 0, 1, 5, 6, 7},
                                      {3, 2, 0, 4, 1, 5, 6, 7}, {0, 3, 4, 2, 1, 5, 6, 7}, {0, 5, 3, 4, 2, 1, 6, 7}};
// reference frame of each coding mode (Table 7.46): 0 intra/self, 1 previous, 2 golden -> THIP_FRAME_*
const uint8_t kModeRefi[8] = {THIP_FRAME_PREV, THIP_FRAME_SELF, THIP_FRAME_PREV, THIP_FRAME_PREV,
                              THIP_FRAME_PREV, THIP_FRAME_GOLD, THIP_FRAME_GOLD, THIP_FRAME_PREV};
enum { MODE_INTER_NOMV = 0, MODE_INTRA = 1, MODE_INTER_MV = 2, MODE_INTER_MV_LAST = 3, MODE_INTER_MV_LAST2 = 4,
       MODE_GOLDEN_NOMV = 5, MODE_GOLDEN_MV = 6, MODE_INTER_MV_FOUR = 7 };

constexpr int kHuffLutBits = 9;
struct HuffTree {
  // node i: child[i][b] >= 0 is another node, < 0 is the leaf -(token+1)
  int16_t child[32][2];
  int nnodes;
  int root_leaf;   // a single-leaf tree: token+1, else 0
  // next kHuffLutBits bits -> (code length << 8 | token), or 0x8000 | node to continue from
  uint64_t lut[1 << kHuffLutBits];
};

struct QuantParams {
  uint8_t lflims[64];
  uint16_t acscale[64], dcscale[64];
  int nbms;
  std::vector<uint8_t> bms;   // nbms*64
  int nqrs[2][3];
  int qrsizes[2][3][64];
  int qrbmis[2][3][65];
};

struct Tok {
  int16_t value;   // coefficient value (0 for pure runs / EOB)
  uint8_t skip;    // zeros before the value
  uint8_t adv;     // how far the block advances (0 for EOB tokens)
  uint32_t eob;    // EOB run length, 0 if not an EOB token
};

}  // namespace

struct th_setup_info {
  HuffTree huff[80];
  QuantParams qp;
};

struct MacroBlock {
  int32_t luma[4];     // fragment indices in raster order (A,B,C,D), -1 outside the frame
  int32_t chroma[2][4];
  int nchroma;         // chroma blocks per plane in this macro block
};

// THIP_FE_PROF=1: wall time per section of th_decode_packetin, printed by th_decode_free
enum { FE_FLAGS, FE_MODES, FE_QI, FE_TOKENS, FE_DC, FE_EXPAND, FE_FLUSH, FE_OUT, FE_LPACK, FE_LMETA, FE_LBEGIN, FE_LFINISH, FE_NSEC };
static const char *const kFeNames[FE_NSEC] = {"header+coded flags", "modes+MVs", "block qi", "DCT tokens", "DC unpredict",
                                              "expand+dequant+stage", "flush (H2D, launch, sync)", "ycbcr_out (D2H)",
                                              "lists: tokens packed", "lists: fragment words", "lists: begin (stage, launch)",
                                              "lists: finish (launch)"};
static inline void cpu_relax() {   // a spinning thread's pause (the x86 hint where there is one)
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#elif defined(__aarch64__)
  asm volatile("yield" ::: "memory");
#else
  std::this_thread::yield();
#endif
}
static inline double fe_now() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
// decoder contexts alive in the process (the token-list path is chosen by default while there are few)
static std::atomic<int> g_fe_contexts{0};
constexpr int kFeListsAutoContexts = 4;
struct FeProf {
  bool on, warmed;   // (the first frames carry one-time costs -- streams, allocations -- and are dropped from the sums)
  double acc[FE_NSEC], t;
  long frames, tokens;
  void start() { if (on) t = fe_now(); }
  void lap(int s) {
    if (!on) return;
    const double n = fe_now();
    acc[s] += n - t;
    t = n;
  }
};

// A frame that goes to the device in GROUPS of zig-zag indices (token-list path, option fe_groups > 1): th_decode_packetin's caller
// runs the entropy decoder -- index after index, decode.c:993-1139 -- and writes each finished list, in the device's token format,
// straight into the backend's pinned staging buffer (thip_state_token_lists_staging); whenever a group of indices is complete it is
// handed over (thip_state_token_lists_append) and the device walks it under the rest of the packet instead of behind it.
struct FeStream {
  thip_token_lists tl;             // the frame without its tokens (pointers into the staging buffer)
  thip_token_staging stg;
  const int *ends;                 // where the groups end (kFeGroupEnd*)
  int gi, z0;                      // the group being filled, its first index
  size_t at, gstart;               // tokens written so far; where the group being filled starts (a multiple of 4)
  bool overflow;
  uint32_t list_off[3][64], list_len[3][64];
  int grp_z0[16], grp_z1[16];
  size_t grp_start[16];
  int64_t grp_ntok[16];
};
// The context's second thread (option fe_worker): while the caller decodes the tokens of indices 1..63 it undoes the DC prediction
// (fe_undo_dc: a chain through every plane in raster order that needs the lists of index 0 only) -- 0.15 ms of a 720p frame that
// used to sit between the packet's last bit and the frame's hand-over.  The launches stay on the caller's thread, and so does
// everything that touches the token lists: a first version that packed the tokens and built the fragment words on this thread made
// the DECODER 40 % slower -- every line of the lists went to the other core and had to be fetched back for the next frame.
struct FeWorker {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  bool go = false, quit = false;   // a frame to work on / the context is going away (both under mu)
  std::atomic<int> z0_ready{0};    // the three lists of index 0 are complete
  std::atomic<int> done{0};        // the DC values of the frame are there
  // Where it runs: on the CPUs that share a last-level cache with the caller's (option fe_worker_pin).  What the two threads hand each
  // other -- the coded flags, reference indices and index-0 lists one way, the DC values the other -- crosses between their cores
  // every frame; through a shared L3 that is tens of nanoseconds a line, across sockets several hundred (4K: the stages that
  // write those arrays ran 2.7 x slower with the thread left to the scheduler of a two-socket host).
  cpu_set_t domain;
  bool placed = false;
  bool solo = false;               // the caller's thread may run on ONE CPU only (taskset -c N): nothing to run beside it on
};

struct FeLookahead;
extern "C" {
static void fe_lookahead_free(th_dec_ctx *d);
static int fe_prefetch(th_dec_ctx *d, const ogg_packet *op);
}

struct th_dec_ctx {
  th_info info;
  FeProf prof;
  th_setup_info setup;
  thip_state *hip;
  int nh[3], nv[3], fro[3], nfrags_pl[3];
  int nfrags;
  int hdec, vdec;
  std::vector<int32_t> coded_order;      // all fragments, coded order, planes concatenated
  std::vector<int32_t> sb_start;         // per super block (all planes): first index in coded_order, +1 sentinel
  int nsbs;
  std::vector<MacroBlock> mbs;           // macro blocks in coded order (only those inside the frame)
  std::vector<uint16_t> dequant;         // [qi][pli][qti][zzi]
  // per frame
  std::vector<uint8_t> coded, refi, qii, mbmode_of_frag;
  bool qii_dirty;                // some entry of qii may be non-zero
  std::vector<int8_t> mvx, mvy;
  std::vector<int16_t> dc;
  std::vector<int> clist;        // coded blocks (raster fragment index) in coded order; plane p is [cl_start[p], cl_start[p+1])
  size_t cl_start[4];
  std::vector<ptrdiff_t> ulist;  // the uncoded ones likewise (what frag_copy_list gets)
  size_t ul_start[4];
  std::vector<uint8_t> dc_key;   // scratch of the DC un-prediction (one plane, bordered)
  std::vector<int16_t> dc_val;
  std::vector<uint8_t> sbp, sbf, mbmodes, qi_bits;
  std::vector<Tok> toks[3][64];   // storage; the lists of the current frame are the first ntoks entries
  size_t ntoks[3][64];
  uint32_t eob_carry[3][64];
  int qis[3], nqis, frame_type;
  int64_t keyframe_num, curframe_num, granpos;
  int granpos_bias;
  bool have_frame;
  bool device_dc;   // DC un-prediction left to the backend (THIP_FE_DEVICE_DC=1, or TH_DECCTL_THIP_SET_DEVICE_DC)
  // out-of-loop post-processing (TH_DECCTL_SET_PPLEVEL; decode.c:1203-1325)
  int pp_level;
  bool dc_qis_tracked;
  std::vector<uint8_t> dc_qis, frag_qi;
  int32_t pp_dc_scale[64], pp_sharp_mod[64];
  bool device_tokens;   // token -> coefficient expansion and AC dequantisation left to the backend (THIP_FE_DEVICE_TOKENS=1, ctl)
  // everything behind the entropy decoder left to the backend: the token lists go to the device as they are
  // (option fe_device_lists, TH_DECCTL_THIP_SET_DEVICE_LISTS; thip_state_token_lists_begin / _finish).  1 on, 0 off, -1 (the
  // default) on while few decoder contexts are alive: the device side of a frame is 0.3 ms of small dependent kernels -- a gain
  // of a fifth for one to four streams, a queue for sixteen.
  int device_lists;
  // -1 (the default): MEASURED per context (fe_lists_rule): the time from one plain th_decode_packetin to the next, 16 inter frames
  // with the lists on the device and 16 with the host's own walk, the faster for option fe_assign_settle frames, and again.  The
  // count of contexts alive (the rule of rounds 3 and 4: lists while at most four) is only where a context starts.
  struct {
    int mode = -1;          // 1: the token lists go to the device; 0: the host walks them; -1: not started
    int phase = 0;          // 0: timing `mode`; 1: timing the other one; 2: settled
    int frames = 0, cnt[2] = {0, 0};
    double sum[2] = {0, 0}, last = 0;   // last: when the previous plain inter frame's th_decode_packetin began (0: none)
    int first = 1;          // the mode the measurement began with (ties go to it)
  } lists_rule;
  // option fe_pipeline: the next announced packet's frame handed to the device INSIDE th_decode_ycbcr_out, before that call waits
  // for its own picture (th_decode_ycbcr_out has the story); what its th_decode_packetin will find:
  struct {
    bool valid = false;
    int rc = 0;
    int64_t granpos = 0;
    int64_t key0 = 0, cur0 = 0;        // the frame counters before it (a dropped frame may still come in between)
    std::vector<uint8_t> pkt;          // the packet it was made from
    // what taking it back needs (round 6: another packet than the announced one may come after all -- theoradec.h:279-302 promises
    // nothing about the order of packets --, and the frame decoded ahead is then undone): the backend's reference ring
    // (thip_state_ring_mark), and what of this context outlives a frame
    int64_t mark[8] = {0};
    int frame_type0 = 0, nqis0 = 0, qis0[3] = {0, 0, 0};
    bool qii_dirty0 = false, qii_saved = false;
    std::vector<uint8_t> qii0;         // the blocks' qi indices (they outlive a frame, decode.c:913-917), when the frame ahead writes them
  } early;
  FeWorker *worker;                  // (created with the first frame that takes the token-list path with option fe_worker on)
  FeLookahead *la;                   // packets announced ahead of their th_decode_packetin (TH_DECCTL_THIP_PREFETCH_PACKET), or null
  bool parse_only;                   // a parser context of a look-ahead: no device state, no counters of its own, fe_front only
  FeStream fs;
  uint32_t arrivals[3][64];          // fragments open at every (plane, index) of the current frame
  thip_token_lists tlp;              // the frame in the one-piece form of the token-list path (fe_pack_lists)
  // the pairing of decode.c:1540-1581 done on the host for the device (decode_token_list<true>): for every entry of tl_tokens the fragment
  // it belongs to and the position its value lands at, for every coded fragment its last index (thip_state_token_lists_begin_assigned)
  std::vector<uint32_t> tl_assign;
  std::vector<uint8_t> tl_lastz;
  bool tl_assigned;
  bool pair_on;                      // fe_front pairs tokens and fragments as it decodes the tokens (decode_token_list<true>)
  uint32_t pair_off[3][64];          // ... where every list's tokens and words start in tl_tokens / tl_assign
  std::vector<uint8_t> pair_pos[3];  // ... the index every coded fragment of a plane arrives at next
  std::vector<uint32_t> pair_arr;    // ... the arrivals of the list at hand
  long assign_checked;               // (slot-trace mode: adopted frames whose walk was checked against the host's own)
  bool tl_packed;                    // tlp / tl_tokens / tl_meta / tl_coded hold the frame at hand (an adopted frame's parser did it)
  std::vector<uint32_t> tl_tokens, tl_meta;
  std::vector<int16_t> tl_dc;            // the un-predicted DC values in coded order (token-list path with the DC chain on the host)
  std::vector<int32_t> tl_coded;
  std::vector<uint8_t> mirror[3];
  th_stripe_callback stripe_cb;
  // slot-trace mode (THIP_FE_TRACE_BACKEND=1 at th_decode_alloc): no device state exists; the
  // vtable-slot calls of a frame are recorded instead of made (TH_DECCTL_THIP_GET_SLOT_TRACE)
  bool trace;
  std::vector<int32_t> tr_fragi;
  std::vector<uint8_t> tr_pli, tr_last_zzi, tr_refi;
  std::vector<uint16_t> tr_dcq;
  std::vector<int16_t> tr_mv, tr_coeffs;
  std::vector<int64_t> tr_uncoded;
  int tr_flimit;
};

namespace {

// ---------------------------------------------------------------------------------------
// headers (spec 6.1 - 6.4)
// ---------------------------------------------------------------------------------------
int read_common_header(BitReader &br) {   // packet type octet + codec string (decinfo.c:188, :211-213)
  (void)br.read(8);
  static const char magic[] = "theora";
  for (int i = 0; i < 6; i++)
    if ((char)br.read(8) != magic[i]) return TH_ENOTFORMAT;
  return 0;
}

int parse_info(BitReader &br, th_info *info) {   // spec 6.2
  info->version_major = (uint8_t)br.read(8);
  info->version_minor = (uint8_t)br.read(8);
  info->version_subminor = (uint8_t)br.read(8);
  if (info->version_major != 3 || info->version_minor > 2) return TH_EVERSION;
  info->frame_width = br.read(16) << 4;
  info->frame_height = br.read(16) << 4;
  info->pic_width = br.read(24);
  info->pic_height = br.read(24);
  info->pic_x = br.read(8);
  info->pic_y = br.read(8);
  info->fps_numerator = br.read(32);
  info->fps_denominator = br.read(32);
  if (info->frame_width == 0 || info->frame_height == 0 || info->pic_width + info->pic_x > info->frame_width ||
      info->pic_height + info->pic_y > info->frame_height || info->fps_numerator == 0 ||
      info->fps_denominator == 0)
    return TH_EBADHEADER;
  // the header counts pic_y from the bottom; the API from the top (decinfo.c:99)
  info->pic_y = info->frame_height - info->pic_height - info->pic_y;
  info->aspect_numerator = br.read(24);
  info->aspect_denominator = br.read(24);
  info->colorspace = (th_colorspace)br.read(8);
  info->target_bitrate = (int)br.read(24);
  info->quality = (int)br.read(6);
  info->keyframe_granule_shift = (int)br.read(5);
  info->pixel_fmt = (th_pixel_fmt)br.read(2);
  if (info->pixel_fmt == TH_PF_RSVD) return TH_EBADHEADER;
  if (br.read(3) != 0 || br.overrun()) return TH_EBADHEADER;
  return 0;
}

uint32_t read_le32(BitReader &br) {   // spec 6.3.1: lengths are little-endian
  uint32_t v = 0;
  for (int i = 0; i < 4; i++) v |= br.read(8) << (8 * i);
  return v;
}

int parse_comment(BitReader &br, th_comment *tc) {   // spec 6.3
  uint32_t len = read_le32(br);
  if (len > br.nbits / 8) return TH_EBADHEADER;
  tc->vendor = (char *)malloc((size_t)len + 1);
  if (!tc->vendor) return TH_EFAULT;
  for (uint32_t i = 0; i < len; i++) tc->vendor[i] = (char)br.read(8);
  tc->vendor[len] = 0;
  const uint32_t n = read_le32(br);
  if (n > br.nbits / 32) return TH_EBADHEADER;
  tc->comments = (int)n;
  tc->user_comments = (char **)calloc(n ? n : 1, sizeof(char *));
  tc->comment_lengths = (int *)calloc(n ? n : 1, sizeof(int));
  if (!tc->user_comments || !tc->comment_lengths) {
    tc->comments = 0;
    return TH_EFAULT;
  }
  for (uint32_t k = 0; k < n; k++) {
    len = read_le32(br);
    if (len > br.nbits / 8) return TH_EBADHEADER;
    tc->user_comments[k] = (char *)malloc((size_t)len + 1);
    if (!tc->user_comments[k]) return TH_EFAULT;
    tc->comment_lengths[k] = (int)len;
    for (uint32_t i = 0; i < len; i++) tc->user_comments[k][i] = (char)br.read(8);
    tc->user_comments[k][len] = 0;
  }
  return br.overrun() ? TH_EBADHEADER : 0;
}

constexpr int kHuffErr = 1000;   // neither a node index (0..31) nor a leaf code (-32..-1)
int parse_huff_tree(BitReader &br, HuffTree &t, int depth, int *nleaves) {   // spec 6.4.4
  if (depth > 32) return kHuffErr;
  if (br.bit()) {
    if (++*nleaves > 32) return kHuffErr;
    return -(int)(br.read(5) + 1);   // leaf
  }
  if (t.nnodes >= 32) return kHuffErr;
  const int me = t.nnodes++;
  const int c0 = parse_huff_tree(br, t, depth + 1, nleaves);
  if (c0 == kHuffErr) return kHuffErr;
  const int c1 = parse_huff_tree(br, t, depth + 1, nleaves);
  if (c1 == kHuffErr) return kHuffErr;
  t.child[me][0] = (int16_t)c0;
  t.child[me][1] = (int16_t)c1;
  return me;
}

// extra bits that follow each DCT token (Tables 7.33 / 7.38; kTokDef below has the rest)
const uint8_t kTokExtraBits[32] = {0, 0, 0, 2, 3, 4, 12, 3, 6, 0, 0, 0, 0, 1, 1, 1,
                                   1, 2, 3, 4, 5, 6, 10, 1, 1, 1, 1, 1, 3, 4, 2, 3};
constexpr uint64_t kLutMore = 1ull << 63;
// lut entry: (code length + extra bits) | extra bits << 8 | code length << 16 | token << 24 | first entry of the token in the
// expansion table (TokTable::tab) << 32, so that how far the bit window moves is known one load after the window -- in the entry's
// LOW byte, which is what a variable shift reads its count from --, and the expansion needs nothing but the entry and the window; or
// kLutMore | node when the code is longer than the table covers.
inline uint32_t tok_table_base(int token) {
  uint32_t n = 0;
  for (int i = 0; i < token; i++) n += 1u << kTokExtraBits[i];
  return n;
}
inline uint64_t huff_lut_entry(int token, int len) {
  const int eb = kTokExtraBits[token];
  return (uint64_t)(len + eb) | ((uint64_t)eb << 8) | ((uint64_t)len << 16) | ((uint64_t)token << 24) | ((uint64_t)tok_table_base(token) << 32);
}
void build_huff_lut(HuffTree &t) {
  for (int v = 0; v < (1 << kHuffLutBits); v++) {
    uint64_t e = 0;
    if (t.root_leaf) e = huff_lut_entry(t.root_leaf - 1, 0);   // zero-length code
    else {
      int node = 0, len = 0;
      e = kLutMore;
      while (len < kHuffLutBits) {
        const int c = t.child[node][(v >> (kHuffLutBits - 1 - len)) & 1];
        len++;
        if (c < 0) {
          e = huff_lut_entry(-c - 1, len);
          break;
        }
        node = c;
        e = kLutMore | (uint64_t)node;
      }
    }
    t.lut[v] = e;
  }
}

int parse_setup(BitReader &br, th_setup_info *s) {   // spec 6.4
  QuantParams &q = s->qp;
  int nbits = (int)br.read(3);   // 6.4.1 loop filter limits
  for (int qi = 0; qi < 64; qi++) q.lflims[qi] = (uint8_t)br.read(nbits);
  nbits = (int)br.read(4) + 1;   // 6.4.2 quantization parameters
  for (int qi = 0; qi < 64; qi++) q.acscale[qi] = (uint16_t)br.read(nbits);
  nbits = (int)br.read(4) + 1;
  for (int qi = 0; qi < 64; qi++) q.dcscale[qi] = (uint16_t)br.read(nbits);
  q.nbms = (int)br.read(9) + 1;
  if (q.nbms > 384 || br.overrun()) return TH_EBADHEADER;
  q.bms.resize((size_t)q.nbms * 64);
  for (int i = 0; i < q.nbms * 64; i++) q.bms[i] = (uint8_t)br.read(8);
  for (int qti = 0; qti < 2; qti++)
    for (int pli = 0; pli < 3; pli++) {
      int newqr = 1;
      if (qti > 0 || pli > 0) newqr = (int)br.bit();
      if (!newqr) {
        int rpqr = 0;
        if (qti > 0) rpqr = (int)br.bit();
        int qtj, plj;
        if (rpqr) {
          qtj = qti - 1;
          plj = pli;
        } else {
          qtj = (3 * qti + pli - 1) / 3;
          plj = (pli + 2) % 3;
        }
        q.nqrs[qti][pli] = q.nqrs[qtj][plj];
        memcpy(q.qrsizes[qti][pli], q.qrsizes[qtj][plj], sizeof(q.qrsizes[0][0]));
        memcpy(q.qrbmis[qti][pli], q.qrbmis[qtj][plj], sizeof(q.qrbmis[0][0]));
      } else {
        int qri = 0, qi = 0;
        q.qrbmis[qti][pli][0] = (int)br.read(ilog((uint32_t)q.nbms - 1));
        if (q.qrbmis[qti][pli][0] >= q.nbms) return TH_EBADHEADER;
        do {
          q.qrsizes[qti][pli][qri] = (int)br.read(ilog((uint32_t)(62 - qi))) + 1;
          qi += q.qrsizes[qti][pli][qri];
          qri++;
          q.qrbmis[qti][pli][qri] = (int)br.read(ilog((uint32_t)q.nbms - 1));
          if (q.qrbmis[qti][pli][qri] >= q.nbms) return TH_EBADHEADER;
        } while (qi < 63 && qri < 63);
        if (qi != 63) return TH_EBADHEADER;
        q.nqrs[qti][pli] = qri;
      }
    }
  for (int hti = 0; hti < 80; hti++) {   // 6.4.4
    HuffTree &t = s->huff[hti];
    t.nnodes = 0;
    t.root_leaf = 0;
    int nleaves = 0;
    const int r = parse_huff_tree(br, t, 0, &nleaves);
    if (r == kHuffErr) return TH_EBADHEADER;
    if (t.nnodes == 0) t.root_leaf = -r;   // degenerate one-leaf tree: zero-length code
    else if (r != 0) return TH_EBADHEADER;
    build_huff_lut(t);
  }
  return br.overrun() ? TH_EBADHEADER : 0;
}

// spec 6.4.3 "Computing a Quantization Matrix"; output in ZIG-ZAG order
void compute_qmat(const QuantParams &q, int qti, int pli, int qi, uint16_t out_zz[64]) {
  int qri = 0, qistart = 0;
  while (qri < q.nqrs[qti][pli] - 1 && qi > qistart + q.qrsizes[qti][pli][qri]) {
    qistart += q.qrsizes[qti][pli][qri];
    qri++;
  }
  const int size = q.qrsizes[qti][pli][qri];
  const int qiend = qistart + size;
  const uint8_t *bmi = &q.bms[(size_t)q.qrbmis[qti][pli][qri] * 64];
  const uint8_t *bmj = &q.bms[(size_t)q.qrbmis[qti][pli][qri + 1] * 64];
  for (int zzi = 0; zzi < 64; zzi++) {
    const int ci = kZigZag[zzi];
    const int bm = (2 * (qiend - qi) * bmi[ci] + 2 * (qi - qistart) * bmj[ci] + size) / (2 * size);
    const int qmin = ci == 0 ? (qti == 0 ? 16 : 32) : (qti == 0 ? 8 : 16);
    const int qscale = ci == 0 ? q.dcscale[qi] : q.acscale[qi];
    int v = (qscale * bm / 100) * 4;
    if (v > 4096) v = 4096;
    if (v < qmin) v = qmin;
    out_zz[zzi] = (uint16_t)v;
  }
}

// ---------------------------------------------------------------------------------------
// run-length coded bit strings (spec 7.2)
// ---------------------------------------------------------------------------------------
void read_long_run_bits(BitReader &br, size_t nbits, std::vector<uint8_t> &out) {   // 7.2.1
  out.assign(nbits, 0);
  size_t len = 0;
  if (!nbits) return;
  uint32_t bit = br.bit();
  for (;;) {
    // Table 7.7: the run-length class is the number of leading ones (0..6) of the next six bits
    static const uint8_t kStart[7] = {1, 2, 4, 6, 10, 18, 34}, kBits[7] = {0, 1, 1, 2, 3, 4, 12};
    int ones = __builtin_clz(~(br.peek(6) << 26));
    if (ones > 6) ones = 6;
    br.skip(ones < 6 ? ones + 1 : 6);
    size_t rlen = (size_t)kStart[ones] + br.read(kBits[ones]);
    const bool full = rlen == 4129;
    if (rlen > nbits - len) rlen = nbits - len;   // invalid stream: clip
    memset(&out[len], (int)bit, rlen);
    len += rlen;
    if (len >= nbits || br.overrun()) return;
    bit = full ? br.bit() : 1 - bit;
  }
}

void read_short_run_bits(BitReader &br, size_t nbits, std::vector<uint8_t> &out) {   // 7.2.2
  out.assign(nbits, 0);
  size_t len = 0;
  if (!nbits) return;
  uint32_t bit = br.bit();
  for (;;) {
    // Table 7.11: leading ones (0..5) of the next five bits
    static const uint8_t kStart[6] = {1, 3, 5, 7, 11, 15}, kBits[6] = {1, 1, 1, 2, 2, 4};
    int ones = __builtin_clz(~(br.peek(5) << 27));
    if (ones > 5) ones = 5;
    br.skip(ones < 5 ? ones + 1 : 5);
    size_t rlen = (size_t)kStart[ones] + br.read(kBits[ones]);
    if (rlen > nbits - len) rlen = nbits - len;
    memset(&out[len], (int)bit, rlen);
    len += rlen;
    if (len >= nbits || br.overrun()) return;
    bit = 1 - bit;
  }
}

// ---------------------------------------------------------------------------------------
// motion vectors (spec 7.5.1)
// ---------------------------------------------------------------------------------------
// Table 7.23 is a 3-bit prefix + magnitude bits + sign, eight bits at most: tabulated by the next eight bits (value, length)
struct MvTable {
  int8_t value[256];
  uint8_t len[256];
  MvTable() {
    for (int w = 0; w < 256; w++) {
      const int p = w >> 5;
      int mag = 0, nb = 3, v;
      switch (p) {
        case 0: v = 0; break;
        case 1: v = 1; break;
        case 2: v = -1; break;
        case 3: mag = 2; nb = 3; v = 2; break;
        case 4: mag = 3; nb = 3; v = 2; break;
        case 5: nb = 5; mag = 4 + ((w >> 3) & 3); v = 2; break;
        case 6: nb = 6; mag = 8 + ((w >> 2) & 7); v = 2; break;
        default: nb = 7; mag = 16 + ((w >> 1) & 15); v = 2; break;
      }
      if (p >= 3) {   // the sign follows the magnitude bits
        const int sign = (w >> (7 - nb)) & 1;
        v = sign ? -mag : mag;
        nb++;
      }
      value[w] = (int8_t)v;
      len[w] = (uint8_t)nb;
    }
  }
};
const MvTable kMvTab;
inline int read_mv_component(BitReader &br, int mvmode) {
  if (mvmode) {   // five bits of magnitude, one of sign
    const uint32_t v = br.read(6);
    const int mag = (int)(v >> 1);
    return (v & 1u) ? -mag : mag;
  }
  const uint32_t w = br.peek(8);   // (bits past the packet's end read as zeros)
  br.skip(kMvTab.len[w]);
  return kMvTab.value[w];
}

inline int round_div(int v, int shift) {   // round(v / 2^shift), ties away from zero (spec 7.5.2)
  const int half = 1 << (shift - 1);
  return v >= 0 ? (v + half) >> shift : -((-v + half) >> shift);
}

// ---------------------------------------------------------------------------------------
// DCT tokens (spec 7.7)
// ---------------------------------------------------------------------------------------
inline int read_token_bitwise(BitReader &br, const HuffTree &t, int node) {
  for (;;) {
    const int c = t.child[node][br.bit()];
    if (c < 0) return -c - 1;
    node = c;
    if (br.overrun()) return 0;
  }
}
inline int read_token(BitReader &br, const HuffTree &t) {
  if (t.root_leaf) return t.root_leaf - 1;
  if (br.pos + 32 > br.nbits) return read_token_bitwise(br, t, 0);   // tail of the packet
  const uint32_t w = br.peek(32);   // a code is at most 32 bits long
  const uint64_t e = t.lut[w >> (32 - kHuffLutBits)];
  if (!(e & kLutMore)) {
    br.skip((int)((e >> 16) & 0xFF));
    return (int)((e >> 24) & 0xFF);
  }
  // longer than the table covers: finish the walk on the peeked word, one skip at the end
  int node = (int)(e & 0x7FFFu), len = kHuffLutBits;
  for (;;) {
    const int c = t.child[node][(w >> (31 - len)) & 1u];
    len++;
    if (c < 0) {
      br.skip(len);
      return -c - 1;
    }
    node = c;
  }
}

// Tables 7.33 / 7.38 as data: what token + extra bits expand to.  Extra bits x are read in one
// go; where the token carries a sign it is the first of them.
//   EOB tokens (0-6):   run = vbase + x            (token 6 with x == 0: "all remaining")
//   zero runs (7, 8):   skip = adv = 1 + x, no value
//   values (9-31):      |value| = vbase + ((rest >> vshift) & vmask), skip = sbase + (rest & smask),
//                       adv = skip + 1, rest = the bits after the sign
struct TokDef {
  uint8_t ebits;     // extra bits in total
  uint8_t kind;      // 0 EOB run, 1 zero run, 2 value
  uint8_t sign;      // 0 positive, 1 negative, 2 read from the stream
  uint8_t vshift, sbase, smask;
  uint16_t vmask;
  int16_t vbase;
};
#define TD(eb, kind, sign, vshift, vmask, sbase, smask, vbase) {eb, kind, sign, vshift, sbase, smask, vmask, vbase}
const TokDef kTokDef[32] = {
    TD(0, 0, 0, 0, 0, 0, 0, 1),      TD(0, 0, 0, 0, 0, 0, 0, 2),      TD(0, 0, 0, 0, 0, 0, 0, 3),
    TD(2, 0, 0, 0, 0x3, 0, 0, 4),    TD(3, 0, 0, 0, 0x7, 0, 0, 8),    TD(4, 0, 0, 0, 0xF, 0, 0, 16),
    TD(12, 0, 0, 0, 0xFFF, 0, 0, 0), TD(3, 1, 0, 0, 0, 1, 0x07, 0),   TD(6, 1, 0, 0, 0, 1, 0x3F, 0),
    TD(0, 2, 0, 0, 0, 0, 0, 1),      TD(0, 2, 1, 0, 0, 0, 0, 1),      TD(0, 2, 0, 0, 0, 0, 0, 2),
    TD(0, 2, 1, 0, 0, 0, 0, 2),      TD(1, 2, 2, 0, 0, 0, 0, 3),      TD(1, 2, 2, 0, 0, 0, 0, 4),
    TD(1, 2, 2, 0, 0, 0, 0, 5),      TD(1, 2, 2, 0, 0, 0, 0, 6),      TD(2, 2, 2, 0, 0x01, 0, 0, 7),
    TD(3, 2, 2, 0, 0x03, 0, 0, 9),   TD(4, 2, 2, 0, 0x07, 0, 0, 13),  TD(5, 2, 2, 0, 0x0F, 0, 0, 21),
    TD(6, 2, 2, 0, 0x1F, 0, 0, 37),  TD(10, 2, 2, 0, 0x1FF, 0, 0, 69), TD(1, 2, 2, 0, 0, 1, 0, 1),
    TD(1, 2, 2, 0, 0, 2, 0, 1),      TD(1, 2, 2, 0, 0, 3, 0, 1),      TD(1, 2, 2, 0, 0, 4, 0, 1),
    TD(1, 2, 2, 0, 0, 5, 0, 1),      TD(3, 2, 2, 0, 0, 6, 0x03, 1),   TD(4, 2, 2, 0, 0, 10, 0x07, 1),
    TD(2, 2, 2, 0, 0x01, 1, 0, 2),   TD(3, 2, 2, 1, 0x01, 2, 0x01, 2)};
#undef TD

// kTokDef with every selection turned into a mask.  Straight-line on purpose: which of the 32
// tokens comes next is close to random, so a switch costs a mispredicted branch per token, and the
// token loop of a frame is the one place where a nanosecond per token is a percent of the frame.
// expand_token(kTokFast.t[token], x, k) fills k once the token's extra bits x are known.
struct TokFast {
  uint8_t ebits, sign_shift, sign_and, sign_const;
  uint8_t vshift, sbase, smask, isval;
  uint16_t vmask, rest_mask;
  int32_t vbase;
  int32_t val_mask;    // -1 for value tokens
  uint32_t eob_mask;   // ~0 for EOB tokens
  uint8_t adv_mask;    // 0 for EOB tokens
};
struct TokFastTable {
  TokFast t[32];
  TokFastTable() {
    for (int i = 0; i < 32; i++) {
      const TokDef &d = kTokDef[i];
      TokFast &f = t[i];
      const bool rd = d.sign == 2;
      if (kTokExtraBits[i] != d.ebits) abort();   // the two tables state the same thing
      f.ebits = d.ebits;
      f.sign_shift = (uint8_t)(rd ? d.ebits - 1 : 0);
      f.sign_and = rd ? 1 : 0;
      f.sign_const = (uint8_t)(rd ? 0 : d.sign);
      f.vshift = d.vshift;
      f.sbase = d.sbase;
      f.smask = d.smask;
      f.isval = d.kind == 2;
      f.vmask = d.vmask;
      f.rest_mask = (uint16_t)((1u << (rd ? d.ebits - 1 : d.ebits)) - 1u);
      f.vbase = d.vbase;
      f.val_mask = d.kind == 2 ? -1 : 0;
      f.eob_mask = d.kind == 0 ? 0xFFFFFFFFu : 0u;
      f.adv_mask = d.kind == 0 ? 0 : 0xFF;
    }
  }
};
const TokFastTable kTokFast;

inline void expand_token(const TokFast &t, uint32_t x, Tok &k) {
  const int32_t neg = (int32_t)(((x >> t.sign_shift) & t.sign_and) | t.sign_const);
  const uint32_t rest = x & t.rest_mask;
  const int32_t mag = t.vbase + (int32_t)((rest >> t.vshift) & t.vmask);
  const int32_t skip = t.sbase + (int32_t)(rest & t.smask);
  k.value = (int16_t)(((mag ^ -neg) + neg) & t.val_mask);
  k.skip = (uint8_t)skip;
  k.adv = (uint8_t)((skip + t.isval) & t.adv_mask);
  k.eob = t.eob_mask & (mag ? (uint32_t)mag : 0xFFFFFFFFu);   // token 6 with a zero run field: all remaining
}

// ... and tabulated: every (token, extra bits) pair there is -- 5405 of them, of which the few dozen
// that real streams use stay in L1 -- so the token loop copies eight bytes instead of computing them.
// a token as the device reads it (thip_tokens.h): an EOB run (flag, 24 bits of length) or a value with the zeros before it
inline uint32_t tok_device_word(const Tok &k) {
  const uint32_t e = k.eob, run = e > 0xFFFFFFu ? 0xFFFFFFu : e;
  const uint32_t we = 0x00800000u | (run & 0xFFFFu) | (run >> 16) << 24;
  const uint32_t wv = (uint32_t)(uint16_t)k.value | (uint32_t)k.skip << 16;
  return e ? we : wv;
}
struct TokTable {
  uint16_t base[32];
  Tok tab[5405];
  uint32_t dev[5405];   // tok_device_word(tab[i])
  TokTable() {
    int n = 0;
    for (int i = 0; i < 32; i++) {
      base[i] = (uint16_t)n;
      for (uint32_t x = 0; x < (1u << kTokExtraBits[i]); x++) {
        expand_token(kTokFast.t[i], x, tab[n]);
        dev[n] = tok_device_word(tab[n]);
        n++;
      }
    }
    if (n != 5405) abort();
  }
};
const TokTable kTokTab;

// One (plane, index) token list (7.7.2): n blocks are open at index z of plane p; every token closes
// or advances at least one of them.  Writes the tokens to out (room for n + 1), counts the blocks
// that move on to index z + adv in left[p][z + adv], leaves in *eobs what is left of an EOB run that
// reaches past this list, and returns the end of the written tokens.  A function of its own (not
// inlined: decode_token_list_plain / _bmi2 below) so that the bit window and the counters get registers instead of stack slots.
// PAIR: the pairing of tokens and fragments (decode.c:1540-1581: which fragment a token belongs to) done HERE, as the tokens are
// decoded, for the device (k_tok_scatter; fe_front's token stage has the story).  The lists are decoded index after index, so when
// list (p, z) is read every fragment that arrives at z is known: `arr` holds them in coded order (the ones a carried EOB run ends
// already skipped), token after token takes the next -- an EOB token as many as it ends --, the fragment's next index goes into
// its byte of `pos` (z + adv; for an EOB token that is z itself: nothing changes) and the token's word (fragment, position of the
// value) into `words`.  Two loads and two stores a token in a loop that waits for its bit window most of the time.
// positions of the set bits of a byte, in order (the rest of an entry is zero), and how many there are
struct BitIndexTable {
  alignas(8) uint8_t idx[256][8];
  uint8_t count[256];
  BitIndexTable() {
    for (int m = 0; m < 256; m++) {
      int n = 0;
      for (int b = 0; b < 8; b++) idx[m][b] = 0;
      for (int b = 0; b < 8; b++)
        if (m >> b & 1) idx[m][n++] = (uint8_t)b;
      count[m] = (uint8_t)n;
    }
  }
};
const BitIndexTable kBitIndex;
struct PairArgs {
  const uint32_t *arr;   // arrivals of this list, coded order: index of the fragment in the plane's coded list
  uint32_t *words;       // one per token of the list (room for n + 1)
  uint32_t *tokd;        // the list's tokens in the device's format (thip_tokens.h), beside the words
  uint8_t *pos;          // the plane's fragments: the index each arrives at next
  uint32_t c0;           // the plane's first fragment in the frame's coded order
};
template <bool PAIR>
static inline __attribute__((always_inline)) Tok *decode_token_list_body(BitReader &br, const HuffTree &tree, size_t n, Tok *out,
                                                                         size_t (*left)[128], int p, int z, uint32_t *eobs, const PairArgs *pa) {
  // (ONE counter for the three arrays a token is written to -- the list, the words, the device's tokens: the loop carries the bit
  //  window, its count, the byte position, the open blocks and the arrivals' pointer besides, and what does not fit sixteen
  //  registers goes through memory on the loop's dependency chain)
  Tok *const out0 = out;
  size_t ti = 0;
  const uint32_t *arr = PAIR ? pa->arr : nullptr;
  uint32_t *const words0 = PAIR ? pa->words : nullptr;
  uint32_t *const tokd0 = PAIR ? pa->tokd : nullptr;
  uint8_t *const ppos = PAIR ? pa->pos : nullptr;
  const uint32_t pc0 = PAIR ? pa->c0 : 0u;
  const uint32_t zword = (uint32_t)z << 18;
  // the reader's state as plain locals: br itself is only touched on the slow paths, so nothing
  // here has its address taken.  The bit position is not carried along: pos == 8 * bytepos - have wherever the reader stands
  // (every refill adds the same bits to both, every skip takes them from `have` and adds them to pos), so it is put back
  // from the two when somebody else reads on.
  uint64_t win = br.win;
  int have = br.have;
  size_t bytepos = br.bytepos;
  const uint8_t *const data = br.data;
  size_t *const left_next = &left[p][z];
  const uint64_t *const lut = tree.lut;
  // Fast path: eight bytes can be loaded at the read position -- then the next 64 bits are all inside the packet too (the
  // longest code and the most extra bits together are 44) -- and the tree is not a single leaf: ONE comparison a token,
  // bytepos against fast_last (-1: never).
  const ptrdiff_t fast_last = !tree.root_leaf && br.nbytes >= 8 ? (ptrdiff_t)(br.nbytes - 8) : (ptrdiff_t)-1;
  // what the fast path wants of the window when it starts on a token: fewer than 64 counted bits (a window that is full to the
  // last bit -- possible only right after a refill nobody consumed from -- cannot be shifted by `have`: a byte is un-counted, its
  // bits stay where they are and are OR-ed in again, unchanged) and the code's first kHuffLutBits bits (after a token of the fast
  // path itself there are 12 or more: a topped-up window has 56+ and a token takes at most 44; other readers may leave fewer)
  // (a macro, not a lambda: captured by reference the window, the count and the byte position would live in memory for the whole
  //  loop -- clang keeps them there -- and the loop's dependency chain would run through store-to-load forwarding)
#define THIP_FE_READY()                                                  \
  do {                                                                   \
    if (have >= 64) {                                                    \
      have -= 8;                                                         \
      bytepos -= 1;                                                      \
    }                                                                    \
    if (have < kHuffLutBits && (ptrdiff_t)bytepos <= fast_last) {        \
      uint64_t v0;                                                       \
      memcpy(&v0, data + bytepos, 8);                                    \
      win |= __builtin_bswap64(v0) >> have;                              \
      bytepos += (size_t)((63 - have) >> 3);                             \
      have |= 56;                                                        \
    }                                                                    \
  } while (0)
  THIP_FE_READY();
  uint32_t run_left = *eobs;   // what the last token's EOB run has left for later lists
  while (n > 0) {
    Tok &k = out0[ti];
    uint32_t dw = 0;   // (PAIR) the token as the device reads it
    // Nothing on the fast path depends on the data except through arithmetic: which token comes next and whether the window
    // needs topping up are both close to random, and a mispredicted branch costs more than all the
    // arithmetic of a token.
    if (__builtin_expect((ptrdiff_t)bytepos <= fast_last, 1)) {
      // The table is asked BEFORE the window is topped up: the code's first kHuffLutBits bits are in the window already, so the
      // look-up -- the load the next token's shift waits for -- starts from the shift of the last token alone, and the top-up
      // (a load whose address follows from the last token's length, then a swap, a shift and an OR) runs beside it instead of
      // in front of it.
      uint64_t e = lut[win >> (64 - kHuffLutBits)];
      // top the window up to 56..63 bits: bits that are loaded but not yet counted in `have` are the
      // stream's own and are OR-ed in again, unchanged, next time
      uint64_t v;
      memcpy(&v, data + bytepos, 8);
      win |= __builtin_bswap64(v) >> have;
      bytepos += (size_t)((63u - (unsigned)have) >> 3);
      have |= 56;
      const uint64_t w = win;
      if (__builtin_expect((e & kLutMore) != 0, 0)) {
        // a code longer than the table covers (at most 32 bits): finish the walk on the window
        int node = (int)(e & 0x7FFFu), len = kHuffLutBits;
        for (;;) {
          const int c = tree.child[node][(w >> (63 - len)) & 1u];
          len++;
          if (c < 0) {
            e = huff_lut_entry(-c - 1, len);
            break;
          }
          node = c;
        }
      }
      const int total = (int)(e & 0xFF);   // 1 <= total <= 32 + 12 <= have
      win <<= total;   // the only thing the next token waits for
      have -= total;
      // the extra bits: the last (e >> 8 & 0xFF) of the `total` bits the token takes (none: 0)
      const uint32_t x = (uint32_t)((w >> ((64 - total) & 63)) & ((1ull << ((e >> 8) & 0xFF)) - 1));
      k = kTokTab.tab[(uint32_t)(e >> 32) + x];
      if (PAIR) dw = kTokTab.dev[(uint32_t)(e >> 32) + x];
    } else {   // single-leaf tree or the last bytes of the packet
      br.win = win; br.have = have; br.pos = 8 * bytepos - (size_t)have; br.bytepos = bytepos;
      const TokFast &t = kTokFast.t[read_token(br, tree)];
      expand_token(t, br.read(t.ebits), k);
      win = br.win; have = br.have; bytepos = br.bytepos;
      THIP_FE_READY();
      if (PAIR) dw = tok_device_word(k);
    }
    if (k.eob == 0xFFFFFFFFu) {   // every block still open anywhere ends (7.7.1)
      size_t all = n;
      for (int pp = p + 1; pp < 3; pp++) all += left[pp][z];
      for (int zz = z + 1; zz < 64; zz++)
        for (int pp = 0; pp < 3; pp++) all += left[pp][zz];
      k.eob = (uint32_t)all;
      if (PAIR) dw = tok_device_word(k);
    }
    // an EOB token ends up to k.eob of the open blocks, any other token advances one of them
    // (for an EOB token adv is 0 and the count lands in this list's own, no longer needed, entry)
    const size_t want = k.eob ? k.eob : 1u;
    const size_t take = want < n ? want : n;
    run_left = (uint32_t)(want - take);
    left_next[k.adv]++;   // z + adv <= 127
    if (PAIR) {
      const uint32_t f = *arr;   // (n > 0: there is one)
      arr += take;
      ppos[f] = (uint8_t)(z + k.adv);   // (>= 64: done; an EOB token leaves z, which no later list looks for)
      words0[ti] = (pc0 + f) | (zword + ((uint32_t)k.skip << 18));
      tokd0[ti] = dw;   // (the token as the device reads it: what fe_pack_lists makes of a list afterwards)
    }
    ti++;
    n -= take;
    // (A truncated packet is not special: past the end the reader supplies zero bits, as
    //  oc_pack_read does, and tokens go on being decoded from them -- every token closes or
    //  advances at least one block, so the list still ends -- which is what libtheora outputs.)
  }
  *eobs = run_left;
  br.win = win; br.have = have; br.pos = 8 * bytepos - (size_t)have; br.bytepos = bytepos;
  return out0 + ti;
}
#undef THIP_FE_READY
// The loop is bound by its instruction count (about 53 a token, five or six a cycle), and a fifth of them only move shift counts
// into the one register x86's variable shifts read them from: the same body compiled for BMI2 (three-operand shifts, bzhi for the
// extra bits) is 18 % faster on the CPU harness (tools/fe_tokbench.cpp: 0.428 -> 0.352 ms a 720p frame).  Chosen once, by what the
// CPU says of itself.
template <bool PAIR>
__attribute__((noinline)) Tok *decode_token_list_plain(BitReader &br, const HuffTree &tree, size_t n, Tok *out, size_t (*left)[128], int p,
                                                       int z, uint32_t *eobs, const PairArgs *pa) {
  return decode_token_list_body<PAIR>(br, tree, n, out, left, p, z, eobs, pa);
}
#if defined(__x86_64__)
template <bool PAIR>
__attribute__((noinline, target("bmi,bmi2"))) Tok *decode_token_list_bmi2(BitReader &br, const HuffTree &tree, size_t n, Tok *out,
                                                                        size_t (*left)[128], int p, int z, uint32_t *eobs, const PairArgs *pa) {
  return decode_token_list_body<PAIR>(br, tree, n, out, left, p, z, eobs, pa);
}
static bool cpu_has_bmi2() {
  __builtin_cpu_init();
  return __builtin_cpu_supports("bmi2") && __builtin_cpu_supports("bmi") && !getenv("THIP_FE_NO_BMI2");
}
const bool kHaveBmi2 = cpu_has_bmi2();
#endif
template <bool PAIR>
inline Tok *decode_token_list(BitReader &br, const HuffTree &tree, size_t n, Tok *out, size_t (*left)[128], int p, int z, uint32_t *eobs,
                              const PairArgs *pa) {
#if defined(__x86_64__)
  if (kHaveBmi2) return decode_token_list_bmi2<PAIR>(br, tree, n, out, left, p, z, eobs, pa);
#endif
  return decode_token_list_plain<PAIR>(br, tree, n, out, left, p, z, eobs, pa);
}

// ---------------------------------------------------------------------------------------
// geometry (spec 2.3 - 2.4): coded order, super blocks, macro blocks
// ---------------------------------------------------------------------------------------
void build_geometry(th_dec_ctx *d) {
  const int fmt = (int)d->info.pixel_fmt;
  d->hdec = !(fmt & 1);
  d->vdec = !(fmt & 2);
  const int yh = (int)d->info.frame_width >> 3, yv = (int)d->info.frame_height >> 3;
  int fro = 0;
  for (int p = 0; p < 3; p++) {
    d->nh[p] = p ? (yh + d->hdec) >> d->hdec : yh;
    d->nv[p] = p ? (yv + d->vdec) >> d->vdec : yv;
    d->fro[p] = fro;
    d->nfrags_pl[p] = d->nh[p] * d->nv[p];
    fro += d->nfrags_pl[p];
  }
  d->nfrags = fro;
  d->coded_order.clear();
  d->sb_start.clear();
  for (int p = 0; p < 3; p++)
    for (int sby = 0; sby < d->nv[p]; sby += 4)
      for (int sbx = 0; sbx < d->nh[p]; sbx += 4) {
        d->sb_start.push_back((int32_t)d->coded_order.size());
        for (int k = 0; k < 16; k++) {
          const int by = sby + kHilbert[k][0], bx = sbx + kHilbert[k][1];
          if (by < d->nv[p] && bx < d->nh[p]) d->coded_order.push_back(d->fro[p] + by * d->nh[p] + bx);
        }
      }
  d->nsbs = (int)d->sb_start.size();
  d->sb_start.push_back((int32_t)d->coded_order.size());
  // macro blocks: luma super blocks in raster order, four macro blocks each in coded order
  d->mbs.clear();
  for (int sby = 0; sby < yv; sby += 4)
    for (int sbx = 0; sbx < yh; sbx += 4)
      for (int k = 0; k < 4; k++) {
        const int my = sby + 2 * kMbOrder[k][0], mx = sbx + 2 * kMbOrder[k][1];
        if (my >= yv || mx >= yh) continue;   // frame sizes are multiples of 16: whole MBs only
        MacroBlock mb;
        for (int i = 0; i < 2; i++)
          for (int j = 0; j < 2; j++) mb.luma[i * 2 + j] = (my + i) * yh + mx + j;
        for (int c = 0; c < 2; c++)
          for (int i = 0; i < 4; i++) mb.chroma[c][i] = -1;
        const int cx = mx >> d->hdec, cy = my >> d->vdec;
        const int ncx = d->hdec ? 1 : 2, ncy = d->vdec ? 1 : 2;
        mb.nchroma = ncx * ncy;
        for (int c = 0; c < 2; c++) {
          // raster order inside the macro block; slot = i*2+j so that 4:4:4 lines up with
          // luma A,B,C,D and 4:2:2 uses slots 0 (bottom) and 2 (top)
          for (int i = 0; i < ncy; i++)
            for (int j = 0; j < ncx; j++)
              mb.chroma[c][i * 2 + j] = d->fro[1 + c] + (cy + i) * d->nh[1] + cx + j;
        }
        d->mbs.push_back(mb);
      }
}

}  // namespace

// ---------------------------------------------------------------------------------------
// API
// ---------------------------------------------------------------------------------------
// ---- 7.8 undo DC prediction (the DC token values are the first coefficient of each block) -----------
// Needs the lists of index 0 and the frame's coded flags and reference indices, nothing decoded after them: on the token-list path in
// groups the context's second thread runs it while the caller decodes the other 63 indices (FeWorker).
namespace {
void fe_undo_dc(th_dec_ctx *d) {
  // first pull the DC values out of the zzi == 0 lists: token by token over the coded blocks in order
  {
    for (int p = 0; p < 3; p++) {
      const int *cl = d->clist.data() + d->cl_start[p];
      const size_t nb = d->cl_start[p + 1] - d->cl_start[p];
      size_t i = d->eob_carry[p][0] < nb ? d->eob_carry[p][0] : nb;   // blocks ended by a run from before
      for (size_t j = 0; j < i; j++) d->dc[cl[j]] = 0;
      const Tok *t = d->toks[p][0].data();
      for (const Tok *const tend = t + d->ntoks[p][0]; t < tend && i < nb; t++) {
        if (t->eob) {   // this block and the next eob - 1 have no coefficients at all
          size_t e = nb - i < t->eob ? nb - i : t->eob;
          while (e--) d->dc[cl[i++]] = 0;
        } else d->dc[cl[i++]] = (int16_t)(t->skip == 0 ? t->value : 0);
      }
      while (i < nb) d->dc[cl[i++]] = 0;   // (cannot happen: the list covers every coded block)
    }
    // Which neighbours predict a block is a question of "coded, and from the same reference frame"
    // (7.8.1, Table 7.47): one byte per block (reference index, 0xFF = not coded) in an array with a
    // border of 0xFF all round answers it with four compares and no position tests.
    // (With thip_state_set_device_dc the backend undoes the prediction itself -- k_dc_unpredict, a wavefront
    //  per plane -- and the slots below are handed the token values.)
    for (int p = 0; p < 3 && !d->device_dc; p++) {
      const int nh = d->nh[p], nv = d->nv[p], W = nh + 2;
      std::vector<uint8_t> &key = d->dc_key;
      std::vector<int16_t> &val = d->dc_val;
      key.assign((size_t)W * (nv + 1) + 1, 0xFF);
      val.assign((size_t)W * (nv + 1) + 1, 0);
      for (int y = 0; y < nv; y++) {
        const int f0 = d->fro[p] + y * nh;
        uint8_t *kr = &key[(size_t)(y + 1) * W + 1];
        const uint8_t *__restrict cd = d->coded.data() + f0, *__restrict rf = d->refi.data() + f0;
        for (int x = 0; x < nh; x++) kr[x] = (uint8_t)(rf[x] | (uint8_t)(cd[x] - 1));   // (coded is 0 or 1: 0xFF where it is 0)
      }
      // weights and divisors of Table 7.47, indexed by which neighbours are available (bit 0 left,
      // 1 up-left, 2 up, 3 up-right); the divisors are powers of two, the division truncates
      static const int8_t Wt[16][4] = {{0, 0, 0, 0},  {1, 0, 0, 0},   {0, 1, 0, 0},  {1, 0, 0, 0},
                                       {0, 0, 1, 0},  {1, 0, 1, 0},   {0, 0, 1, 0},  {29, -26, 29, 0},
                                       {0, 0, 0, 1},  {75, 0, 0, 53}, {0, 1, 0, 1},  {75, 0, 0, 53},
                                       {0, 0, 1, 0},  {75, 0, 0, 53}, {0, 3, 10, 3}, {29, -26, 29, 0}};
      static const uint8_t Dsh[16] = {0, 0, 0, 0, 0, 1, 0, 5, 0, 7, 1, 7, 0, 7, 4, 5};
      int last[3] = {0, 0, 0};
      for (int y = 0; y < nv; y++) {
        const int f0 = d->fro[p] + y * nh;
        const uint8_t *kr = &key[(size_t)(y + 1) * W + 1];
        int16_t *vr = &val[(size_t)(y + 1) * W + 1];
        // (the left neighbour's key and value travel in registers: the value is what the next block waits for, and through
        //  memory it would wait for the store to come back as well)
        const uint8_t *const ku = kr - W;
        const int16_t *const vu = vr - W;
        int lk = 0xFF, lv = 0;
        for (int x = 0; x < nh; x++) {
          const int r = kr[x];
          if (r == 0xFF) {
            lk = 0xFF;
            continue;
          }
          const int mask = (lk == r) | (ku[x - 1] == r) << 1 | (ku[x] == r) << 2 | (ku[x + 1] == r) << 3;
          int pred;
          if ((mask & 7) == 7) {
            // L, UL and U all present (every interior block of a key frame): (29 L - 26 UL + 29 U) / 32
            // whether or not UR is, then the outlier clamp (7.8.1 step 5)
            const int l = lv, ul = vu[x - 1], u = vu[x];
            const int num = 29 * (l + u) - 26 * ul;
            pred = (num + ((num >> 31) & 31)) >> 5;   // num / 32, towards zero
            if (abs(pred - u) > 128) pred = u;
            else if (abs(pred - l) > 128) pred = l;
            else if (abs(pred - ul) > 128) pred = ul;
          } else if (mask == 0) pred = last[r];
          else {
            const int l = lv, ul = vu[x - 1], u = vu[x], ur = vu[x + 1];
            const int num = Wt[mask][0] * l + Wt[mask][1] * ul + Wt[mask][2] * u + Wt[mask][3] * ur;
            const int sh = Dsh[mask];
            pred = (num + ((num >> 31) & ((1 << sh) - 1))) >> sh;   // num / 2^sh, towards zero
          }
          const int16_t v = (int16_t)(d->dc[f0 + x] + pred);   // 16-bit wrap
          d->dc[f0 + x] = v;
          vr[x] = v;
          last[r] = v;
          lk = r;
          lv = v;
        }
      }
    }
  }
}
}  // namespace

// ---- a frame in groups of indices (see FeStream, FeWorker) --------------------------------------------------------------------
namespace {
// where a group of indices ends: the lists of low indices are the long ones (a group is worth a pair of launches once the device
// has a few tens of microseconds of work in it), and the last group is what the device still has to walk when the packet's last bit
// has been read
constexpr int kFeGroupEnd9[] = {1, 3, 6, 10, 15, 21, 28, 40, 64};
constexpr int kFeGroupEnd5[] = {1, 6, 15, 28, 64}, kFeGroupEnd4[] = {3, 10, 28, 64}, kFeGroupEnd3[] = {3, 15, 64}, kFeGroupEnd2[] = {6, 64};
// (round 6: the same low groups with a SHORTER last one -- what the device walks behind the packet's last bit is 3 us an index --
//  at the price of one or two more pairs of launches on the caller's thread: option fe_groups = 6 / 7)
constexpr int kFeGroupEnd6[] = {3, 10, 28, 48, 64}, kFeGroupEnd7[] = {3, 10, 28, 44, 56, 64};
inline const int *fe_group_ends(int n) {
  return n >= 9 ? kFeGroupEnd9 : n == 7 ? kFeGroupEnd7 : n == 6 ? kFeGroupEnd6 : n >= 5 ? kFeGroupEnd5 : n == 4 ? kFeGroupEnd4 : n == 3 ? kFeGroupEnd3 : kFeGroupEnd2;
}

// the frame's description and its fragment words, written into the staging buffer (caller's thread)
int fe_stream_open_frame(th_dec_ctx *d, int ngroups) {
  FeStream &fs = d->fs;
  const int rc = thip_state_token_lists_staging(d->hip, &fs.stg);
  if (rc < 0) return rc;
  thip_token_lists &tl = fs.tl;
  memset(&tl, 0, sizeof(tl));
  tl.frame_type = d->frame_type;
  tl.flimit = d->setup.qp.lflims[d->qis[0]];
  for (int p = 0; p < 3; p++) {
    tl.ncoded[p] = (int32_t)(d->cl_start[p + 1] - d->cl_start[p]);
    for (size_t ci = d->cl_start[p]; ci < d->cl_start[p + 1]; ci++) {
      const int f = d->clist[ci];
      const uint32_t qti = d->mbmode_of_frag[f] != MODE_INTRA;
      fs.stg.coded[ci] = f;
      fs.stg.frag_meta[ci] = (uint32_t)d->refi[f] | ((uint32_t)(p * 3 + d->qii[f]) * 2u + qti) << 2 |
                             ((uint32_t)d->mvx[f] & 0xFFu) << 8 | ((uint32_t)d->mvy[f] & 0xFFu) << 16 | (uint32_t)p << 24;
    }
    for (int qti = 0; qti < 2; qti++) tl.dc_quant[p][qti] = d->dequant[(((size_t)d->qis[0] * 3 + p) * 2 + qti) * 64];
  }
  memset(fs.stg.dequant, 0, 18 * 64 * 2);
  for (int p = 0; p < 3; p++)
    for (int qii = 0; qii < d->nqis; qii++)
      for (int qti = 0; qti < 2; qti++)
        memcpy(fs.stg.dequant + ((p * 3 + qii) * 2 + qti) * 64, &d->dequant[(((size_t)d->qis[qii] * 3 + p) * 2 + qti) * 64], 128);
  tl.coded = fs.stg.coded;
  tl.frag_meta = fs.stg.frag_meta;
  tl.dequant = fs.stg.dequant;
  fs.ends = fe_group_ends(ngroups);
  fs.gi = 0;
  fs.z0 = 0;
  fs.at = fs.gstart = 0;
  fs.overflow = false;
  return 0;
}

// The three lists of index z are complete: into the staging buffer in the device's token format (thip_tokens.h).  Returns true when
// that completes a group (its description is in grp_*[gi - 1] then).
bool fe_stream_pack_index(th_dec_ctx *d, int z) {
  FeStream &fs = d->fs;
  for (int p = 0; p < 3; p++) {
    const size_t nk = d->ntoks[p][z];
    fs.list_off[p][z] = (uint32_t)(fs.at - fs.gstart);
    fs.list_len[p][z] = (uint32_t)nk;
    if (fs.overflow || (int64_t)(fs.at + nk) > fs.stg.token_capacity) {   // (cannot happen: a token closes an open block of its list)
      fs.overflow = true;
      fs.list_len[p][z] = 0;
      continue;
    }
    const Tok *__restrict t = d->toks[p][z].data();
    uint32_t *__restrict w = fs.stg.tokens + fs.at;
    for (size_t k = 0; k < nk; k++) {   // (branch-free, one list at a time: the compiler vectorises it)
      const uint32_t e = t[k].eob;
      const uint32_t run = e > 0xFFFFFFu ? 0xFFFFFFu : e;   // (more than any plane the backend takes has)
      const uint32_t we = 0x00800000u | (run & 0xFFFFu) | (run >> 16) << 24;
      const uint32_t wv = (uint32_t)(uint16_t)t[k].value | (uint32_t)t[k].skip << 16;
      w[k] = e ? we : wv;
    }
    fs.at += nk;
  }
  if (z + 1 != fs.ends[fs.gi]) return false;
  fs.grp_z0[fs.gi] = fs.z0;
  fs.grp_z1[fs.gi] = z + 1;
  fs.grp_start[fs.gi] = fs.gstart;
  fs.grp_ntok[fs.gi] = (int64_t)(fs.at - fs.gstart);
  fs.gi++;
  fs.z0 = z + 1;
  fs.at = (fs.at + 3) & ~(size_t)3;   // the next group starts on a 16-byte unit
  fs.gstart = fs.at;
  return true;
}
int fe_stream_append(th_dec_ctx *d, int g) {
  FeStream &fs = d->fs;
  return thip_state_token_lists_append(d->hip, fs.grp_z0[g], fs.grp_z1[g], fs.stg.tokens + fs.grp_start[g], fs.grp_ntok[g], fs.list_off,
                                       fs.list_len, d->eob_carry, d->arrivals);
}

// the CPUs that share the last-level cache of `cpu` (sysfs list format: "0-7,128-135"); false when the host does not say
bool fe_llc_cpus(int cpu, cpu_set_t *set) {
  char path[128], buf[512];
  snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", cpu);
  FILE *f = fopen(path, "r");
  if (!f) return false;
  const bool got = fgets(buf, sizeof(buf), f) != nullptr;
  fclose(f);
  if (!got) return false;
  CPU_ZERO(set);
  int n = 0;
  for (const char *c = buf; *c && *c != '\n';) {
    char *e;
    const long a = strtol(c, &e, 10);
    if (e == c) return false;
    long b = a;
    c = e;
    if (*c == '-') {
      b = strtol(c + 1, &e, 10);
      if (e == c + 1) return false;
      c = e;
    }
    for (long k = a; k <= b && k < CPU_SETSIZE; k++) {
      CPU_SET((int)k, set);
      n++;
    }
    if (*c == ',') c++;
  }
  return n > 1;
}
// called by th_decode_packetin's thread before it wakes the worker: keep the worker next to it
void fe_worker_place(FeWorker &w) {
  const int cpu = sched_getcpu();
  if (cpu < 0 || cpu >= CPU_SETSIZE) return;
  if (w.placed && CPU_ISSET(cpu, &w.domain)) return;   // (the caller has not left the cache domain the worker is in)
  cpu_set_t set, allowed;
  if (!fe_llc_cpus(cpu, &set)) return;
  // inside what the caller's own thread is allowed (a process pinned with taskset keeps its threads where it was put)
  if (sched_getaffinity(0, sizeof(allowed), &allowed) == 0) {
    w.solo = CPU_COUNT(&allowed) < 2;
    CPU_AND(&set, &set, &allowed);
    if (CPU_COUNT(&set) < 2) {   // (nothing to choose: asked again only when the caller turns up somewhere else)
      w.domain = allowed;
      w.placed = true;
      return;
    }
  }
  if (pthread_setaffinity_np(w.th.native_handle(), sizeof(set), &set) != 0) return;   // (CPUs outside the process's set: left to the scheduler)
  w.domain = set;
  w.placed = true;
}

// the worker's side of a frame
void fe_worker_frame(th_dec_ctx *d) {
  FeWorker &w = *d->worker;
  // (the lists of index 0 are a few tens of microseconds of the decoder's work away)
  for (unsigned spins = 0; !w.z0_ready.load(std::memory_order_acquire); spins++) {
    if (spins < 8192) cpu_relax();
    else std::this_thread::yield();
  }
  fe_undo_dc(d);
  const size_t nc = d->cl_start[3];
  d->tl_dc.resize(nc + 1);
  for (size_t ci = 0; ci < nc; ci++) d->tl_dc[ci] = d->dc[d->clist[ci]];
}

void fe_worker_main(th_dec_ctx *d) {
  FeWorker &w = *d->worker;
  for (;;) {
    {
      std::unique_lock<std::mutex> lk(w.mu);
      w.cv.wait(lk, [&] { return w.go || w.quit; });
      if (w.quit) return;
      w.go = false;
    }
    fe_worker_frame(d);
    w.done.store(1, std::memory_order_release);
  }
}
}  // namespace

extern "C" {

void th_info_init(th_info *info) {
  if (!info) return;
  memset(info, 0, sizeof(*info));
  info->version_major = 3;
  info->version_minor = 2;
  info->version_subminor = 1;
  info->keyframe_granule_shift = 6;
}
void th_info_clear(th_info *info) {
  if (info) memset(info, 0, sizeof(*info));
}
void th_comment_init(th_comment *tc) {
  if (tc) memset(tc, 0, sizeof(*tc));
}
void th_comment_clear(th_comment *tc) {
  if (!tc) return;
  for (int i = 0; i < tc->comments; i++) free(tc->user_comments[i]);
  free(tc->user_comments);
  free(tc->comment_lengths);
  free(tc->vendor);
  memset(tc, 0, sizeof(*tc));
}

// codec.h:545-589: tags are compared without regard to case, up to the '='
static int tag_matches(const char *comment, const char *tag, size_t n) {
  for (size_t i = 0; i < n; i++) {
    char a = comment[i], b = tag[i];
    if (a >= 'a' && a <= 'z') a = (char)(a - 32);
    if (b >= 'a' && b <= 'z') b = (char)(b - 32);
    if (a != b || !a) return 0;
  }
  return comment[n] == '=';
}
void th_comment_add(th_comment *tc, const char *comment) {
  if (!tc || !comment) return;
  char **uc = (char **)realloc(tc->user_comments, sizeof(char *) * (size_t)(tc->comments + 2));
  if (!uc) return;
  tc->user_comments = uc;
  int *cl = (int *)realloc(tc->comment_lengths, sizeof(int) * (size_t)(tc->comments + 2));
  if (!cl) return;
  tc->comment_lengths = cl;
  const size_t len = strlen(comment);
  char *c = (char *)malloc(len + 1);
  if (!c) return;
  memcpy(c, comment, len + 1);
  tc->user_comments[tc->comments] = c;
  tc->comment_lengths[tc->comments] = (int)len;
  tc->comments++;
  tc->user_comments[tc->comments] = nullptr;
}
void th_comment_add_tag(th_comment *tc, const char *tag, const char *val) {
  if (!tc || !tag || !val) return;
  const size_t tl = strlen(tag), vl = strlen(val);
  char *c = (char *)malloc(tl + vl + 2);
  if (!c) return;
  memcpy(c, tag, tl);
  c[tl] = '=';
  memcpy(c + tl + 1, val, vl + 1);
  th_comment_add(tc, c);
  free(c);
}
char *th_comment_query(th_comment *tc, const char *tag, int count) {
  if (!tc || !tag) return nullptr;
  const size_t n = strlen(tag);
  int found = 0;
  for (int i = 0; i < tc->comments; i++)
    if ((size_t)tc->comment_lengths[i] > n && tag_matches(tc->user_comments[i], tag, n)) {
      if (found == count) return tc->user_comments[i] + n + 1;
      found++;
    }
  return nullptr;
}
int th_comment_query_count(th_comment *tc, const char *tag) {
  if (!tc || !tag) return 0;
  const size_t n = strlen(tag);
  int found = 0;
  for (int i = 0; i < tc->comments; i++)
    if ((size_t)tc->comment_lengths[i] > n && tag_matches(tc->user_comments[i], tag, n)) found++;
  return found;
}

// codec.h "Basic shared functions" (internal.c:189-210, state.c:1259-1267)
const char *th_version_string(void) { return "theora-hip 0.2 (MI355X backend; bitstream 3.2.1)"; }
uint32_t th_version_number(void) { return (3u << 16) + (2u << 8) + 1u; }
int th_packet_isheader(ogg_packet *op) { return op && op->bytes > 0 ? op->packet[0] >> 7 : 0; }
int th_packet_iskeyframe(ogg_packet *op) {
  return !op || op->bytes <= 0 ? 0 : (op->packet[0] & 0x80) ? -1 : !(op->packet[0] & 0x40);
}

int th_decode_headerin(th_info *info, th_comment *tc, th_setup_info **setup, ogg_packet *op) {
  if (!op) return TH_EBADHEADER;
  if (!info) return TH_EFAULT;
  if (op->bytes <= 0 || !op->packet) return TH_EBADHEADER;
  // a data packet after all three headers ends header decode (theoradec.h:218-233); the order of
  // the checks and their codes are decinfo.c:191-210
  if (!(op->packet[0] & 0x80)) {
    if (info->frame_width <= 0) return TH_ENOTFORMAT;
    if (!tc) return TH_EFAULT;
    if (!tc->vendor) return TH_EBADHEADER;
    if (!setup) return TH_EFAULT;
    if (!*setup) return TH_EBADHEADER;
    return 0;
  }
  BitReader br(op->packet, (size_t)op->bytes);
  const int type = op->packet[0];
  int rc;
  // the codec string comes before anything that depends on the packet type (decinfo.c:211-213)
  if ((rc = read_common_header(br)) < 0) return rc;
  if (type == 0x80) {
    if (!op->b_o_s || info->frame_width > 0) return TH_EBADHEADER;
    if ((rc = parse_info(br, info)) < 0) {
      th_info_clear(info);
      return rc;
    }
    return 3;
  }
  if (type == 0x81) {
    if (!tc) return TH_EFAULT;
    if (!info->frame_width || tc->vendor) return TH_EBADHEADER;
    if ((rc = parse_comment(br, tc)) < 0) {
      th_comment_clear(tc);
      return rc;
    }
    return 2;
  }
  if (type == 0x82) {
    if (!tc || !setup) return TH_EFAULT;
    if (!info->frame_width || !tc->vendor || *setup) return TH_EBADHEADER;
    th_setup_info *s = new (std::nothrow) th_setup_info();
    if (!s) return TH_EFAULT;
    if ((rc = parse_setup(br, s)) < 0) {
      delete s;
      return rc;
    }
    *setup = s;
    return 1;
  }
  return TH_EBADHEADER;   // an unknown header type (decinfo.c:251-254)
}

void th_setup_free(th_setup_info *setup) { delete setup; }

th_dec_ctx *th_decode_alloc(const th_info *info, const th_setup_info *setup) { return th_decode_alloc_on(info, setup, -1); }

th_dec_ctx *th_decode_alloc_on(const th_info *info, const th_setup_info *setup, int device) {
  if (!info || !setup) return nullptr;
  if (device < 0) {   // option "device" (THIP_DEVICE=<n>|rr): -1 the current device, -2 round robin over the node's devices
    const int e = thip_option("device");
    if (e == -2) {
      static std::atomic<unsigned> next{0};
      const int n = thip_device_count();
      device = n > 0 ? (int)(next.fetch_add(1) % (unsigned)n) : -1;
    } else if (e >= 0) {
      device = e;
    }
  }
  if ((info->frame_width & 15) || (info->frame_height & 15) || !info->frame_width || !info->frame_height ||
      info->pixel_fmt == TH_PF_RSVD || (int)info->pixel_fmt < 0 || (int)info->pixel_fmt > 3)
    return nullptr;
  th_dec_ctx *d = new th_dec_ctx();
  d->info = *info;
  d->setup = *setup;
  d->hip = nullptr;
  d->worker = nullptr;
  d->la = nullptr;
  d->parse_only = false;
  d->pair_on = false;
  d->tl_packed = d->tl_assigned = false;
  d->trace = thip_option("fe_trace_backend") != 0;
  d->tr_flimit = 0;
  if (!d->trace &&
      thip_state_create_on(&d->hip, device, (int)info->frame_width, (int)info->frame_height, (int)info->pixel_fmt) < 0) {
    delete d;
    return nullptr;
  }
  if (d->hip) thip_state_set_eager_output(d->hip, 1);   // every frame is wanted on the host (th_decode_ycbcr_out)
  d->device_dc = false;
  if (d->hip && thip_option("fe_device_dc") != 0)
    d->device_dc = thip_state_set_device_dc(d->hip, 1) == 0;   // (refused for planes of more than 1024 fragment rows)
  d->device_tokens = d->hip && thip_option("fe_device_tokens") != 0;
  d->device_lists = d->hip ? thip_option("fe_device_lists") : 0;
  g_fe_contexts.fetch_add(1, std::memory_order_relaxed);
  build_geometry(d);
  d->dequant.resize((size_t)64 * 3 * 2 * 64);
  for (int qi = 0; qi < 64; qi++)
    for (int p = 0; p < 3; p++)
      for (int qti = 0; qti < 2; qti++)
        compute_qmat(d->setup.qp, qti, p, qi, &d->dequant[(((size_t)qi * 3 + p) * 2 + qti) * 64]);
  // post-processing tables: pp_dc_scale as oc_dequant_tables_init leaves it (quant.c:88 -- every (qti, pli) pass
  // overwrites it, the inter / Cr pass is the last), pp_sharp_mod from decode.c:398-409
  for (int qi = 0; qi < 64; qi++) {
    const QuantParams &q = d->setup.qp;
    const int qti = 1, p = 2;
    int qri = 0, qistart = 0;
    while (qri < q.nqrs[qti][p] - 1 && qi > qistart + q.qrsizes[qti][p][qri]) {
      qistart += q.qrsizes[qti][p][qri];
      qri++;
    }
    const int size = q.qrsizes[qti][p][qri], qiend = qistart + size;
    const int bmi = q.bms[(size_t)q.qrbmis[qti][p][qri] * 64], bmj = q.bms[(size_t)q.qrbmis[qti][p][qri + 1] * 64];
    const uint32_t base0 = (uint32_t)((2 * (qiend - qi) * bmi + 2 * (qi - qistart) * bmj + size) / (2 * size)) & 0xFFu;
    d->pp_dc_scale[qi] = (int32_t)(((uint32_t)q.dcscale[qi] * base0) / 160u);
    int qsum = 0;
    for (int t = 0; t < 2; t++)
      for (int pl = 0; pl < 3; pl++) {
        const uint16_t *dq = &d->dequant[(((size_t)qi * 3 + pl) * 2 + t) * 64];
        qsum += (dq[12] + dq[17] + dq[18] + dq[24]) << (pl == 0);
      }
    d->pp_sharp_mod[qi] = -(qsum >> 11);
  }
  d->pp_level = 0;
  d->dc_qis_tracked = false;
  d->coded.assign(d->nfrags, 0);
  d->refi.assign(d->nfrags, 0);
  d->qii.assign(d->nfrags, 0);
  d->qii_dirty = false;
  d->mvx.assign(d->nfrags, 0);
  d->mvy.assign(d->nfrags, 0);
  d->dc.assign(d->nfrags, 0);
  d->mbmode_of_frag.assign(d->nfrags, 0);
  d->mbmodes.assign(d->mbs.size(), 0);
  d->keyframe_num = d->curframe_num = 0;
  d->granpos = 0;
  // streams of bitstream version 3.2.1 and later count frames from 1 (state.c:740-745)
  d->granpos_bias = (info->version_major > 3 ||
                     (info->version_major == 3 && (info->version_minor > 2 ||
                                                   (info->version_minor == 2 && info->version_subminor >= 1))))
                        ? 1 : 0;
  d->have_frame = false;
  d->stripe_cb.ctx = nullptr;
  d->stripe_cb.stripe_decoded = nullptr;
  memset(&d->prof, 0, sizeof(d->prof));
  d->prof.on = thip_option("fe_prof") != 0;
  for (int p = 0; p < 3; p++) d->mirror[p].assign((size_t)d->nh[p] * 8 * d->nv[p] * 8, 0);
  return d;
}

void th_decode_free(th_dec_ctx *d) {
  if (!d) return;
  if (d->prof.on && d->prof.frames) {
    double tot = 0;
    for (int s = 0; s < FE_NSEC; s++) tot += d->prof.acc[s];
    fprintf(stderr, "[thip front end] %ld frames, %.3f ms/frame, %ld tokens/frame\n", d->prof.frames, 1e3 * tot / (double)d->prof.frames, d->prof.tokens / d->prof.frames);
    for (int s = 0; s < FE_NSEC; s++)
      fprintf(stderr, "  %-28s %8.3f ms/frame %5.1f %%\n", kFeNames[s], 1e3 * d->prof.acc[s] / (double)d->prof.frames,
              100.0 * d->prof.acc[s] / tot);
  }
  fe_lookahead_free(d);
  if (d->worker) {
    {
      std::lock_guard<std::mutex> lk(d->worker->mu);
      d->worker->quit = true;
    }
    d->worker->cv.notify_one();
    if (d->worker->th.joinable()) d->worker->th.join();
    delete d->worker;
  }
  if (d->hip) thip_state_free(d->hip);
  g_fe_contexts.fetch_sub(1, std::memory_order_relaxed);
  delete d;
}

int th_decode_ctl(th_dec_ctx *d, int req, void *buf, size_t buf_sz) {
  switch (req) {
    case TH_DECCTL_GET_PPLEVEL_MAX:
      if (!d || !buf) return TH_EFAULT;
      if (buf_sz != sizeof(int)) return TH_EINVAL;
      *(int *)buf = 7;   // OC_PP_LEVEL_MAX, decode.c:48
      return 0;
    case TH_DECCTL_SET_PPLEVEL: {
      if (!d || !buf) return TH_EFAULT;
      if (buf_sz != sizeof(int)) return TH_EINVAL;
      const int lvl = *(int *)buf;
      if (lvl < 0 || lvl > 7) return TH_EINVAL;   // decode.c:1994
      d->pp_level = lvl;
      // every frame is sent to its host image by the decoding launch itself -- unless a post-processed one
      // is going to replace it: then the image is made when th_decode_ycbcr_out asks
      if (d->hip) thip_state_set_eager_output(d->hip, lvl < 2);
      return 0;
    }
    case TH_DECCTL_SET_GRANPOS: {
      if (!d || !buf) return TH_EFAULT;
      if (buf_sz != sizeof(int64_t)) return TH_EINVAL;
      const int64_t g = *(int64_t *)buf;
      if (g < 0) return TH_EINVAL;
      d->granpos = g;
      d->keyframe_num = (g >> d->info.keyframe_granule_shift) - d->granpos_bias;
      d->curframe_num = d->keyframe_num + (g & (((int64_t)1 << d->info.keyframe_granule_shift) - 1));
      return 0;
    }
    case TH_DECCTL_SET_STRIPE_CB:
      if (!d || !buf) return TH_EFAULT;
      if (buf_sz != sizeof(th_stripe_callback)) return TH_EINVAL;
      d->stripe_cb = *(const th_stripe_callback *)buf;
      return 0;
    case TH_DECCTL_THIP_SET_DEVICE_DC: {
      if (!d || !buf) return TH_EFAULT;
      if (buf_sz != sizeof(int)) return TH_EINVAL;
      if (d->trace || !d->hip) return TH_EINVAL;
      const int on = *(int *)buf != 0;
      if (thip_state_set_device_dc(d->hip, on) < 0) return TH_EIMPL;
      d->device_dc = on != 0;
      return 0;
    }
    case TH_DECCTL_THIP_SET_DEVICE_TOKENS: {
      if (!d || !buf) return TH_EFAULT;
      if (buf_sz != sizeof(int)) return TH_EINVAL;
      if (d->trace || !d->hip) return TH_EINVAL;
      d->device_tokens = *(int *)buf != 0;
      return 0;
    }
    case TH_DECCTL_THIP_SET_DEVICE_LISTS: {
      if (!d || !buf) return TH_EFAULT;
      if (buf_sz != sizeof(int)) return TH_EINVAL;
      if (d->trace || !d->hip) return TH_EINVAL;
      d->device_lists = *(int *)buf != 0 ? 1 : 0;
      return 0;
    }
    case TH_DECCTL_THIP_GET_DEVICE: {
      if (!d || !buf) return TH_EFAULT;
      if (buf_sz != sizeof(int)) return TH_EINVAL;
      if (!d->hip) return TH_EINVAL;
      *(int *)buf = thip_state_device(d->hip);
      return 0;
    }
    case TH_DECCTL_THIP_PREFETCH_PACKET:
      if (!d || !buf) return TH_EFAULT;
      if (buf_sz != sizeof(ogg_packet)) return TH_EINVAL;
      return fe_prefetch(d, (const ogg_packet *)buf);
    case TH_DECCTL_THIP_GET_SLOT_TRACE: {
      if (!d || !buf) return TH_EFAULT;
      if (buf_sz != sizeof(thip_slot_trace)) return TH_EINVAL;
      if (!d->trace) return TH_EINVAL;
      thip_slot_trace *t = (thip_slot_trace *)buf;
      t->ncoded = (int64_t)d->tr_fragi.size();
      t->fragi = d->tr_fragi.data();
      t->pli = d->tr_pli.data();
      t->last_zzi = d->tr_last_zzi.data();
      t->refi = d->tr_refi.data();
      t->dc_quant = d->tr_dcq.data();
      t->mv = d->tr_mv.data();
      t->coeffs = d->tr_coeffs.data();
      t->nuncoded = (int64_t)d->tr_uncoded.size();
      t->uncoded = d->tr_uncoded.data();
      t->flimit = d->tr_flimit;
      t->frame_type = d->frame_type;
      return 0;
    }
    default: return TH_EIMPL;
  }
}

int64_t th_granule_frame(void *encdec, int64_t granpos) {
  th_dec_ctx *d = (th_dec_ctx *)encdec;
  if (!d || granpos < 0) return -1;
  const int shift = d->info.keyframe_granule_shift;
  const int64_t iframe = granpos >> shift;
  const int64_t pframe = granpos - (iframe << shift);
  return iframe + pframe - d->granpos_bias;
}

double th_granule_time(void *encdec, int64_t granpos) {
  th_dec_ctx *d = (th_dec_ctx *)encdec;
  if (!d || granpos < 0 || !d->info.fps_numerator) return -1;
  return (double)(th_granule_frame(encdec, granpos) + 1) * ((double)d->info.fps_denominator / (double)d->info.fps_numerator);
}

// One data packet: spec 7.1 - 7.11, ending in the vtable slots of the HIP backend.  In two halves: fe_front is the entropy decoder
// (7.1 - 7.7: everything that reads the packet; on the token-list path in groups it hands the lists over as it goes), fe_back
// what follows the packet's last bit (7.8 and the hand-over of the frame).  A frame's front half depends on nothing an earlier
// frame left behind -- which is what the look-ahead of th_decode_ctl(TH_DECCTL_THIP_PREFETCH_PACKET) uses: the front halves of the
// packets a caller has announced run on parser contexts of their own (FeLookahead), th_decode_packetin adopts the result.
struct FeRun {
  bool lists_now = false, streaming = false, with_worker = false, dc_done = false;
  int stream_rc = 0;   // (inline hand-over: the first failure)
};
constexpr int kFeContinue = 0x7F00;   // fe_front: the frame goes on to fe_back (anything else is th_decode_packetin's return value)

// is everything behind the entropy decoder the device's (the token-list path) for the frame at hand?
static bool fe_lists_now(const th_dec_ctx *d) {
  if (d->trace || !d->hip) return false;
  if (d->device_lists >= 0) return d->device_lists > 0;
  if (d->device_dc || d->device_tokens) return false;
  if (thip_option("fe_lists_rule") == 0 || d->lists_rule.mode < 0)   // (the count of contexts: option fe_lists_rule = 0, and where a context starts)
    return g_fe_contexts.load(std::memory_order_relaxed) <= kFeListsAutoContexts;
  return d->lists_rule.mode != 0;
}
constexpr int kFeListsSample = 16, kFeListsWarm = 8;
// option fe_device_lists = -1, fe_lists_rule = 1: which of the two is faster for THIS context (see th_dec_ctx::lists_rule); called
// when a th_decode_packetin begins, with whether the packet takes the plain path (not adopted from the look-ahead, not empty)
static void fe_lists_rule(th_dec_ctx *d, double now, bool plain) {
  auto &R = d->lists_rule;
  if (d->trace || !d->hip || d->device_lists >= 0 || d->device_dc || d->device_tokens || thip_option("fe_lists_rule") == 0) return;
  if (R.mode < 0) {
    R.mode = R.first = g_fe_contexts.load(std::memory_order_relaxed) <= kFeListsAutoContexts ? 1 : 0;
    R.frames = -kFeListsWarm;   // (a stream's first frames: buffers, streams and threads come into being)
  }
  // the interval that ends now is the frame before this one: counted if it was a plain inter frame (a key frame costs several
  // inter frames, and whether one falls into a sample is chance)
  const bool counted = R.last > 0 && plain;
  const double dt = now - R.last;
  R.last = plain ? now : 0;
  if (!counted || d->frame_type == THIP_INTRA_FRAME) return;
  if (R.phase < 2) {
    if (++R.frames <= 2) return;   // (the first two frames behind a change of sides still carry the other side's work)
    const int which = R.phase;
    R.sum[which] += dt;
    if (++R.cnt[which] >= kFeListsSample) {
      const int before = R.mode;
      if (R.phase == 0) {
        R.mode ^= 1;
      } else {
        const double first = R.sum[0] / R.cnt[0], other = R.sum[1] / R.cnt[1];
        R.mode = first <= 1.03 * other ? R.first : R.first ^ 1;
      }
      if (R.mode != before) thip_option_add(R.mode ? "fe_lists_to_device" : "fe_lists_to_host", 1);
      R.phase++;
      R.frames = 0;
    }
  } else if (++R.frames >= std::max(1, thip_option("fe_assign_settle"))) {
    R.phase = 0;
    R.frames = 0;
    R.first = R.mode;
    R.sum[0] = R.sum[1] = 0;
    R.cnt[0] = R.cnt[1] = 0;
  }
}

static int fe_front(th_dec_ctx *d, const ogg_packet *op, int64_t *granpos, FeRun &r) {
  const int N = d->nfrags;
  d->tl_packed = d->tl_assigned = false;
  int ncoded_total = 0;
  // (a negative length reads as all-zero bits here and in the reference alike -- oc_pack_readinit with
  //  a stop pointer before the start -- i.e. as an intra frame header; only bytes == 0 is a drop)
  BitReader br(op->packet, op->bytes > 0 ? (size_t)op->bytes : 0);
  d->prof.start();
  if (op->bytes == 0) {
    // an empty packet is a dropped frame: an inter frame with no coded blocks (decode.c:2746)
    d->frame_type = THIP_INTER_FRAME;
    memset(d->coded.data(), 0, (size_t)N);
  } else {
    // ---- 7.1 frame header -------------------------------------------------------------------
    if (br.bit()) return TH_EBADPACKET;
    d->frame_type = (int)br.bit();
    d->nqis = 0;
    do {
      d->qis[d->nqis++] = (int)br.read(6);
    } while (d->nqis < 3 && br.bit());
    if (d->frame_type == THIP_INTRA_FRAME) {
      if (br.read(3) != 0) return TH_EIMPL;
      memset(d->coded.data(), 1, (size_t)N);   // 7.3 step 1
      ncoded_total = N;
    } else {
      // ---- 7.3 coded block flags ---------------------------------------------------------------
      read_long_run_bits(br, (size_t)d->nsbs, d->sbp);
      size_t nfull = 0;
      for (int s = 0; s < d->nsbs; s++) nfull += !d->sbp[s];
      std::vector<uint8_t> fbits;
      read_long_run_bits(br, nfull, fbits);
      d->sbf.assign((size_t)d->nsbs, 0);
      size_t fi = 0, nblk = 0;
      for (int s = 0; s < d->nsbs; s++) {
        if (!d->sbp[s]) d->sbf[s] = fbits[fi++];
        else nblk += (size_t)(d->sb_start[s + 1] - d->sb_start[s]);
      }
      std::vector<uint8_t> bbits;
      read_short_run_bits(br, nblk, bbits);
      size_t bi = 0;
      for (int s = 0; s < d->nsbs; s++)
        for (int k = d->sb_start[s]; k < d->sb_start[s + 1]; k++) {
          const uint8_t c = d->sbp[s] ? bbits[bi++] : d->sbf[s];
          d->coded[d->coded_order[k]] = c;
          ncoded_total += c;
        }
    }
  }
  // no reference yet on an inter frame: the backend substitutes mid-grey (decode.c:2757-2762)
  d->granpos = ((d->keyframe_num + d->granpos_bias) << d->info.keyframe_granule_shift) +
               (d->curframe_num - d->keyframe_num);
  if (ncoded_total == 0) {   // decode.c:2764-2772
    // Nothing decoded yet: the reference has just made its mid-grey dummy frame (decode.c:2757-2762,
    // oc_dec_init_dummy_frame) and th_decode_ycbcr_out shows that; the backend substitutes the same
    // grey when the first inter frame with coded blocks arrives (thip_decode_frames).
    if (!d->have_frame && !d->parse_only)   // (a look-ahead's parser has no pictures: its owner says TH_DUPFRAME itself)
      for (int p = 0; p < 3; p++) memset(d->mirror[p].data(), 0x80, d->mirror[p].size());
    d->curframe_num++;
    if (granpos) *granpos = d->granpos;
    return TH_DUPFRAME;
  }
  // the coded blocks in coded order, per plane: what every later stage walks
  {
    d->clist.resize((size_t)ncoded_total + 1);   // + 1: the slot a block is written to and not kept
    d->ulist.resize((size_t)(N - ncoded_total) + 1);
    int cstart[4] = {0, d->nfrags_pl[0], d->nfrags_pl[0] + d->nfrags_pl[1], N};
    size_t n = 0, u = 0;
    for (int p = 0; p < 3; p++) {
      d->cl_start[p] = n;
      d->ul_start[p] = u;
      for (int k = cstart[p]; k < cstart[p + 1]; k++) {
        const int f = d->coded_order[k];
        d->clist[n] = f;   // both unconditionally; the right one keeps it (no branch on a random flag)
        d->ulist[u] = f;
        n += d->coded[f];
        u += !d->coded[f];
      }
    }
    d->cl_start[3] = n;
    d->ul_start[3] = u;
  }
  d->prof.lap(FE_FLAGS);
  if (d->frame_type == THIP_INTRA_FRAME) {
    d->keyframe_num = d->curframe_num;
    d->granpos = ((d->keyframe_num + d->granpos_bias) << d->info.keyframe_granule_shift);
    memset(d->refi.data(), THIP_FRAME_SELF, (size_t)N);
    memset(d->mbmode_of_frag.data(), MODE_INTRA, (size_t)N);
    memset(d->mvx.data(), 0, (size_t)N);
    memset(d->mvy.data(), 0, (size_t)N);
  } else {
    // ---- 7.4 macro block coding modes -----------------------------------------------------------
    const int mscheme = (int)br.read(3);
    uint8_t alphabet[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // a malformed scheme-0 list may leave entries unset
    if (mscheme == 0) {
      for (int mode = 0; mode < 8; mode++) alphabet[br.read(3)] = (uint8_t)mode;
    } else if (mscheme != 7) {
      memcpy(alphabet, kModeAlphabets[mscheme - 1], 8);
    }
    for (size_t m = 0; m < d->mbs.size(); m++) {
      const MacroBlock &mb = d->mbs[m];
      const bool any = d->coded[mb.luma[0]] | d->coded[mb.luma[1]] | d->coded[mb.luma[2]] | d->coded[mb.luma[3]];
      int mode = MODE_INTER_NOMV;
      if (any) {
        if (mscheme != 7) {
          // Table 7.19's codes are unary: up to seven ones, closed by a zero unless there are seven.  One look at seven bits instead
          // of a read per bit (bits past the packet's end read as zeros either way).
          int mi = __builtin_clz(~(br.peek(7) << 25));
          if (mi > 7) mi = 7;
          br.skip(mi < 7 ? mi + 1 : 7);
          mode = alphabet[mi];
        } else {
          mode = (int)br.read(3);
        }
      }
      d->mbmodes[m] = (uint8_t)mode;
    }
    // ---- 7.5 motion vectors ----------------------------------------------------------------------
    const int mvmode = (int)br.bit();
    int last1x = 0, last1y = 0, last2x = 0, last2y = 0;
    for (size_t m = 0; m < d->mbs.size(); m++) {
      const MacroBlock &mb = d->mbs[m];
      const int mode = d->mbmodes[m];
      int mx = 0, my = 0;
      int lx[4] = {0, 0, 0, 0}, ly[4] = {0, 0, 0, 0};
      if (mode == MODE_INTER_MV_FOUR) {
        for (int k = 0; k < 4; k++)
          if (d->coded[mb.luma[k]]) {
            lx[k] = read_mv_component(br, mvmode);
            ly[k] = read_mv_component(br, mvmode);
            mx = lx[k];
            my = ly[k];
          }
        last2x = last1x; last2y = last1y;
        last1x = mx; last1y = my;
      } else if (mode == MODE_GOLDEN_MV) {
        mx = read_mv_component(br, mvmode);
        my = read_mv_component(br, mvmode);
      } else if (mode == MODE_INTER_MV_LAST2) {
        mx = last2x; my = last2y;
        last2x = last1x; last2y = last1y;
        last1x = mx; last1y = my;
      } else if (mode == MODE_INTER_MV_LAST) {
        mx = last1x; my = last1y;
      } else if (mode == MODE_INTER_MV) {
        mx = read_mv_component(br, mvmode);
        my = read_mv_component(br, mvmode);
        last2x = last1x; last2y = last1y;
        last1x = mx; last1y = my;
      }
      const uint8_t refi = kModeRefi[mode];
      for (int k = 0; k < 4; k++) {
        const int f = mb.luma[k];
        d->refi[f] = refi;
        d->mbmode_of_frag[f] = (uint8_t)mode;
        d->mvx[f] = (int8_t)(mode == MODE_INTER_MV_FOUR ? lx[k] : mx);
        d->mvy[f] = (int8_t)(mode == MODE_INTER_MV_FOUR ? ly[k] : my);
      }
      for (int c = 0; c < 2; c++)
        for (int k = 0; k < 4; k++) {
          const int f = mb.chroma[c][k];
          if (f < 0) continue;
          d->refi[f] = refi;
          d->mbmode_of_frag[f] = (uint8_t)mode;
          int cxv = mx, cyv = my;
          if (mode == MODE_INTER_MV_FOUR) {
            if (d->hdec && d->vdec) {          // 4:2:0: one block, average of four
              cxv = round_div(lx[0] + lx[1] + lx[2] + lx[3], 2);
              cyv = round_div(ly[0] + ly[1] + ly[2] + ly[3], 2);
            } else if (d->hdec) {              // 4:2:2: bottom = A,B ; top = C,D
              const int a = k == 0 ? 0 : 2;
              cxv = round_div(lx[a] + lx[a + 1], 1);
              cyv = round_div(ly[a] + ly[a + 1], 1);
            } else if (d->vdec) {              // (decimated vertically only) left = A,C ; right = B,D
              cxv = round_div(lx[k] + lx[k + 2], 1);
              cyv = round_div(ly[k] + ly[k + 2], 1);
            } else {                           // 4:4:4
              cxv = lx[k];
              cyv = ly[k];
            }
          }
          d->mvx[f] = (int8_t)cxv;
          d->mvy[f] = (int8_t)cyv;
        }
    }
  }
  d->prof.lap(FE_MODES);
  // ---- 7.6 block-level qi ---------------------------------------------------------------------------
  {
    const size_t nc = d->cl_start[3];
    const int *cl = d->clist.data();
    // only the coded blocks get a new qii (decode.c:913-917): an uncoded block keeps the one it was last coded
    // with, which the de-ringing filter reads (decode.c:1926)
    // (with one qi in every frame since the last key frame nothing is anything but zero: the pass over the coded blocks is skipped)
    if (d->nqis > 1 || d->qii_dirty) {
      for (size_t i = 0; i < nc; i++) d->qii[cl[i]] = 0;
      if (d->frame_type == THIP_INTRA_FRAME) d->qii_dirty = false;   // every block was coded: every entry is zero now
    }
    if (d->nqis > 1) d->qii_dirty = true;
    std::vector<uint8_t> &bits = d->qi_bits;
    for (int q = 0; q + 1 < d->nqis; q++) {
      size_t nb = 0;
      for (size_t i = 0; i < nc; i++) nb += d->qii[cl[i]] == q;
      read_long_run_bits(br, nb, bits);
      bits.push_back(0);   // the entry a block that takes no bit looks at
      size_t bi = 0;
      for (size_t i = 0; i < nc; i++) {
        const int f = cl[i];
        const unsigned m = d->qii[f] == q;
        d->qii[f] = (uint8_t)(d->qii[f] + (m & bits[bi]));
        bi += m;
      }
    }
  }
  d->prof.lap(FE_QI);
  // ---- everything behind the entropy decoder on the device, when asked for and possible (decided here: the worker starts now) ----
  const bool lists_now = r.lists_now = fe_lists_now(d);
  const int fe_groups = lists_now ? thip_option("fe_groups") : 0;
  bool &streaming = r.streaming;                 // in groups of indices, as they are decoded (FeStream)
  streaming = lists_now && fe_groups > 1;
  int &stream_rc = r.stream_rc;
  if (streaming && fe_stream_open_frame(d, fe_groups) < 0) streaming = false;   // (the one-piece path below says why)
  if (streaming) {
    stream_rc = thip_state_token_lists_open(d->hip, &d->fs.tl);
    if (stream_rc < 0) streaming = false;   // (THIP_EIMPL: the slots; anything else: reported by the one-piece path below)
  }
  bool &with_worker = r.with_worker;
  // the DC chain on the second thread -- where it is worth a thread: the device needs the values LAST (thip_state_token_lists_finish),
  // behind its walk of the last group of indices (3 us an index: 110-160 us), so up to 720p the caller undoes the prediction itself
  // while that walk runs (0.16 ms at 720p) and a second thread only adds its hand-overs (measured: + 5 % without it at 720p, - 10 %
  // at 1080p, where the chain is 0.36 ms); fe_worker = 2 (default) draws the line at 32 768 fragments, 1 / 0: always / never
  const int fe_worker = thip_option("fe_worker");
  with_worker = lists_now && !d->device_dc && (fe_worker == 1 || (fe_worker == 2 && d->nfrags > 32768));
  if (with_worker && !d->worker) {
    d->worker = new (std::nothrow) FeWorker();
    if (d->worker) {
      try {
        d->worker->th = std::thread(fe_worker_main, d);
      } catch (...) {   // (no thread to be had: the caller does it behind the tokens)
        delete d->worker;
        d->worker = nullptr;
      }
    }
    if (!d->worker) with_worker = false;
  }
  if (with_worker) {
    FeWorker &w = *d->worker;
    if (thip_option("fe_worker_pin") != 0) fe_worker_place(w);
    if (w.solo) with_worker = false;
  }
  if (with_worker) {
    FeWorker &w = *d->worker;
    w.z0_ready.store(0, std::memory_order_relaxed);
    w.done.store(0, std::memory_order_relaxed);
    {
      std::lock_guard<std::mutex> lk(w.mu);
      w.go = true;
    }
    w.cv.notify_one();
  }
  d->prof.lap(FE_LMETA);
  // ---- 7.7 DCT tokens, unpacked by counts per (plane, index) list -------------------------------------
  {
    size_t left[3][128];   // [64..127]: where advances past the last index land
    memset(left, 0, sizeof(left));
    for (int p = 0; p < 3; p++) left[p][0] = d->cl_start[p + 1] - d->cl_start[p];
    memset(d->ntoks, 0, sizeof(d->ntoks));
    memset(d->eob_carry, 0, sizeof(d->eob_carry));
    uint32_t eobs = 0;
    int htil = 0, htic = 0;
    // Pairing tokens and fragments while the tokens are decoded (a look-ahead's parser, option fe_assign): see decode_token_list.
    const bool pair = d->pair_on;
    size_t pair_at = 0;   // tokens written so far
    if (pair) {
      size_t nmax = 0;
      for (int p = 0; p < 3; p++) {
        const size_t np = d->cl_start[p + 1] - d->cl_start[p], np16 = (np + 15) & ~(size_t)15;
        d->pair_pos[p].assign(np16 + 16, 0xFF);   // 0xFF: not a fragment
        if (np) memset(d->pair_pos[p].data(), 0, np);   // every coded fragment arrives at index 0
        nmax = np16 > nmax ? np16 : nmax;
      }
      d->pair_arr.resize(nmax + 32);                   // (+ the eight entries a block's last store may reach past its arrivals)
      d->tl_lastz.assign(d->cl_start[3] + 1 + 16, 0);   // (+ 16: the last block of a plane is written whole)
    }
    for (int z = 0; z < 64; z++) {
      if (z < 2) {
        htil = (int)br.read(4);
        htic = (int)br.read(4);
      }
      const int hg = z == 0 ? 0 : z <= 5 ? 1 : z <= 14 ? 2 : z <= 27 ? 3 : 4;
      for (int p = 0; p < 3; p++) {
        size_t n = left[p][z];
        d->arrivals[p][z] = (uint32_t)n;
        // an EOB run still open from an earlier list ends the first blocks of this one
        if (eobs) {
          const uint32_t take = eobs < n ? eobs : (uint32_t)n;
          d->eob_carry[p][z] = take;
          eobs -= take;
          n -= take;
        }
        const HuffTree &tree = d->setup.huff[16 * hg + (p == 0 ? htil : htic)];
        // every token closes at least one open block of this list: at most n tokens (+1 for the
        // closing entry of a truncated packet, +1 for the sentinel).  The lists only ever grow; ntoks
        // is their length.
        std::vector<Tok> &list = d->toks[p][z];
        if (list.size() < n + 2) list.resize(n + 2);
        Tok *out;
        if (pair) {
          // the fragments that arrive at index z of this plane, in coded order: the bytes of `pos` equal to z.  Every one of them
          // has its last index set to z here (decode.c:1545; whoever arrives again later overwrites it).
          // tokens (device format) and words go straight to where the device's copy is made from, list after list in the order
          // they are decoded (the lists' places are a table of the hand-over: any order will do)
          if (d->tl_tokens.size() < pair_at + n + 2) d->tl_tokens.resize((pair_at + n + 2) * 2);
          if (d->tl_assign.size() < pair_at + n + 2) d->tl_assign.resize((pair_at + n + 2) * 2);
          d->pair_off[p][z] = (uint32_t)pair_at;
          const size_t c0 = d->cl_start[p], np = d->cl_start[p + 1] - c0, np16 = (np + 15) & ~(size_t)15;
          uint8_t *const pos = d->pair_pos[p].data();
          uint32_t *const arr = d->pair_arr.data();
          uint8_t *const lastz = d->tl_lastz.data() + c0;
          size_t na = 0;
          if (d->arrivals[p][z]) {
#if defined(__SSE2__)
            // sixteen fragments at a time, nothing per arrival: the last indices are blended in as a vector, the arrivals' numbers
            // come out of a table indexed by eight bits of the comparison's mask (how many bits a block has set is close to
            // random for the middle indices: a loop over them costs a mispredicted branch per block, more than everything else)
            const __m128i zz = _mm_set1_epi8((char)z), zero = _mm_setzero_si128();
            for (size_t i = 0; i < np16; i += 16) {
              const __m128i eq = _mm_cmpeq_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i *>(pos + i)), zz);
              const unsigned m = (unsigned)_mm_movemask_epi8(eq);
              if (!m) continue;   // (nobody of these sixteen: the rule at high indices)
              __m128i *const lzp = reinterpret_cast<__m128i *>(lastz + i);   // (tl_lastz has room for whole blocks)
              _mm_storeu_si128(lzp, _mm_or_si128(_mm_andnot_si128(eq, _mm_loadu_si128(lzp)), _mm_and_si128(eq, zz)));
              for (int half = 0; half < 2; half++) {
                const unsigned m8 = (m >> (8 * half)) & 0xFFu;
                const __m128i b = _mm_unpacklo_epi8(_mm_loadl_epi64(reinterpret_cast<const __m128i *>(kBitIndex.idx[m8])), zero);
                const __m128i base = _mm_set1_epi32((int)(i + 8 * (size_t)half));
                _mm_storeu_si128(reinterpret_cast<__m128i *>(arr + na), _mm_add_epi32(_mm_unpacklo_epi16(b, zero), base));
                _mm_storeu_si128(reinterpret_cast<__m128i *>(arr + na + 4), _mm_add_epi32(_mm_unpackhi_epi16(b, zero), base));
                na += kBitIndex.count[m8];
              }
            }
#else
            for (size_t k = 0; k < np; k++)
              if (pos[k] == z) {
                arr[na++] = (uint32_t)k;
                lastz[k] = (uint8_t)z;
              }
#endif
          }
          // (na == the list's arrivals by construction; a carried run has ended the first eob_carry of them)
          PairArgs pa = {arr + d->eob_carry[p][z], d->tl_assign.data() + pair_at, d->tl_tokens.data() + pair_at, pos, (uint32_t)c0};
          out = decode_token_list<true>(br, tree, n, list.data(), left, p, z, &eobs, &pa);
          pair_at += (size_t)(out - list.data());
        } else {
          out = decode_token_list<false>(br, tree, n, list.data(), left, p, z, &eobs, nullptr);
        }
        d->ntoks[p][z] = (size_t)(out - list.data());
        // behind the list: an EOB run without end, so that the expansion needs no end-of-list test
        // (a malformed stream that asks for more tokens than the list has finds its blocks ended)
        out->value = 0; out->skip = 0; out->adv = 0; out->eob = 0xFFFFFFFFu;
        d->prof.tokens += (long)d->ntoks[p][z];
      }
      if (z == 0 && with_worker) d->worker->z0_ready.store(1, std::memory_order_release);   // the DC chain can start
      // the three lists of index z are complete; with them maybe a group
      if (streaming && fe_stream_pack_index(d, z) && stream_rc >= 0) stream_rc = fe_stream_append(d, d->fs.gi - 1);
    }
  }
  d->prof.lap(FE_TOKENS);
  return kFeContinue;
}

// The frame's token lists in the device's format (thip_tokens.h), its coded-fragment list and fragment words: d->tlp, tl_tokens,
// tl_coded, tl_meta (the one-piece form of the token-list path; a look-ahead's parser does it for its owner).  The tables that
// need the owner's dequantisation matrices are filled in by fe_back.
static void fe_pack_lists(th_dec_ctx *d) {
  thip_token_lists &tl = d->tlp;
  memset(&tl, 0, sizeof(tl));
  tl.frame_type = d->frame_type;
  tl.flimit = d->setup.qp.lflims[d->qis[0]];
  size_t nt = 0;
  for (int p = 0; p < 3; p++)
    for (int z = 0; z < 64; z++) nt += d->ntoks[p][z];
  if (d->pair_on) {   // fe_front has written the tokens and their words where they belong (decode_token_list<true>): the tables only
    for (int p = 0; p < 3; p++)
      for (int z = 0; z < 64; z++) {
        tl.list_off[p][z] = d->pair_off[p][z];
        tl.list_len[p][z] = (uint32_t)d->ntoks[p][z];
        tl.eob_carry[p][z] = d->eob_carry[p][z];
        tl.arrivals[p][z] = d->arrivals[p][z];
      }
    if (d->tl_assign.size() < nt + 1) d->tl_assign.resize(nt + 1);
    d->tl_assign[nt] = 0xFFFFFFFFu;
  } else {
    d->tl_tokens.resize(nt + 1);
    uint32_t *o = d->tl_tokens.data();
    size_t at = 0;
    for (int p = 0; p < 3; p++)
      for (int z = 0; z < 64; z++) {
        tl.list_off[p][z] = (uint32_t)at;
        tl.list_len[p][z] = (uint32_t)d->ntoks[p][z];
        tl.eob_carry[p][z] = d->eob_carry[p][z];
        tl.arrivals[p][z] = d->arrivals[p][z];
        // (branch-free, one list at a time: the compiler vectorises it)
        const Tok *__restrict t = d->toks[p][z].data();
        uint32_t *__restrict w = o + at;
        const size_t nk = d->ntoks[p][z];
        for (size_t k = 0; k < nk; k++) {
          const uint32_t e = t[k].eob;
          const uint32_t run = e > 0xFFFFFFu ? 0xFFFFFFu : e;   // (more than any plane the backend takes has)
          const uint32_t we = 0x00800000u | (run & 0xFFFFu) | (run >> 16) << 24;
          const uint32_t wv = (uint32_t)(uint16_t)t[k].value | (uint32_t)t[k].skip << 16;
          w[k] = e ? we : wv;
        }
        at += nk;
      }
  }
  tl.ntokens = (int64_t)nt;
  d->tl_assigned = d->pair_on;   // (tl_assign and tl_lastz are fe_front's too)
  d->prof.lap(FE_LPACK);
  const size_t nc = d->cl_start[3];
  d->tl_meta.resize(nc + 1);
  d->tl_coded.resize(nc + 1);
  for (int p = 0; p < 3; p++) {
    tl.ncoded[p] = (int32_t)(d->cl_start[p + 1] - d->cl_start[p]);
    for (size_t ci = d->cl_start[p]; ci < d->cl_start[p + 1]; ci++) {
      const int f = d->clist[ci];
      const uint32_t qti = d->mbmode_of_frag[f] != MODE_INTRA;
      d->tl_coded[ci] = f;
      d->tl_meta[ci] = (uint32_t)d->refi[f] | ((uint32_t)(p * 3 + d->qii[f]) * 2u + qti) << 2 |
                       ((uint32_t)d->mvx[f] & 0xFFu) << 8 | ((uint32_t)d->mvy[f] & 0xFFu) << 16 | (uint32_t)p << 24;
    }
  }
}

// Slot-trace mode: what k_tok_scatter would make of an adopted frame's assignment -- every token on its own, as the device has
// them -- against the coefficients the host's own walk has just recorded (fe_back).  False: they differ.
static bool fe_check_assignment(const th_dec_ctx *d) {
  const thip_token_lists &tl = d->tlp;
  const size_t nc = d->cl_start[3];
  if (d->tr_fragi.size() != nc || d->tl_assign.size() < (size_t)tl.ntokens || d->tl_lastz.size() < nc) return false;
  std::vector<int16_t> c(nc * 64, 0);
  for (int64_t j = 0; j < tl.ntokens; j++) {
    const uint32_t tk = d->tl_tokens[(size_t)j];
    if (tk & 0x00800000u) continue;
    const uint32_t w = d->tl_assign[(size_t)j];
    const uint32_t ci = w & 0x3FFFFu;
    const int at = (int)((w >> 18) & 127u);
    if (w == 0xFFFFFFFFu || ci >= nc) continue;
    const int value = (int16_t)(tk & 0xFFFFu);
    if (!value || at == 0 || at > 63) continue;
    const int f = d->clist[ci];
    int p = 0;
    while (p < 2 && ci >= d->cl_start[p + 1]) p++;
    const int qti = d->mbmode_of_frag[f] != MODE_INTRA;
    const uint16_t *acq = &d->dequant[(((size_t)d->qis[d->qii[f]] * 3 + p) * 2 + qti) * 64];
    c[ci * 64 + kZigZag[at]] = (int16_t)(value * (int)acq[at]);
  }
  for (size_t ci = 0; ci < nc; ci++) {
    if (d->tl_lastz[ci] != d->tr_last_zzi[ci]) return false;
    for (int k = 1; k < 64; k++)
      if (c[ci * 64 + k] != d->tr_coeffs[ci * 64 + k]) return false;
  }
  return true;
}

static int fe_back(th_dec_ctx *d, int64_t *granpos, FeRun &r) {
  const int N = d->nfrags;
  const bool lists_now = r.lists_now, streaming = r.streaming, with_worker = r.with_worker;
  const int stream_rc = r.stream_rc;
  // ---- 7.8 undo DC prediction: fe_undo_dc -------------------------------------------------------------
  bool &dc_done = r.dc_done;
  auto undo_dc = [&]() {
    dc_done = true;
    fe_undo_dc(d);
  };
  // (with the second thread: it has been at it since the lists of index 0 were complete, and leaves the values in coded order in tl_dc too)
  auto join_worker = [&]() {
    if (!with_worker || dc_done) return;
    FeWorker &w = *d->worker;
    for (unsigned spins = 0; !w.done.load(std::memory_order_acquire); spins++) {
      if (spins < 8192) cpu_relax();
      else std::this_thread::yield();
    }
    dc_done = true;
  };
  // ---- everything from here to the pictures on the device, when asked for and possible ------------------
  bool lists_done = false;
  if (streaming) {
    // The lists have been going over group by group; the device is a group or two behind.  The DC prediction is undone here
    // meanwhile (the device needs the values last), then the frame is finished.
    const size_t nc = d->cl_start[3];
    const int16_t *dcv = nullptr;
    if (with_worker) {
      join_worker();
      dcv = d->tl_dc.data();
      d->prof.lap(FE_DC);
    } else if (!d->device_dc) {
      undo_dc();
      d->tl_dc.resize(nc + 1);
      for (size_t ci = 0; ci < nc; ci++) d->tl_dc[ci] = d->dc[d->clist[ci]];
      dcv = d->tl_dc.data();
      d->prof.lap(FE_DC);
    }
    int lrc = stream_rc;
    if (lrc < 0) (void)thip_state_token_lists_abort(d->hip);
    if (lrc >= 0 && d->fs.overflow) {
      (void)thip_state_token_lists_abort(d->hip);
      lrc = THIP_EINVAL;
    }
    d->prof.lap(FE_LBEGIN);
    if (lrc >= 0) {
      lrc = thip_state_token_lists_finish(d->hip, dcv);
      d->prof.lap(FE_LFINISH);
    }
    if (lrc >= 0) lists_done = true;
    else if (lrc != THIP_EIMPL) return TH_EFAULT;   // (THIP_EIMPL: a plane too large for that path -- the slots below)
  } else if (lists_now) {
    if (!d->tl_packed) {   // (an adopted frame may bring them packed, and its tokens paired with their fragments)
      fe_pack_lists(d);
      d->tl_assigned = false;
    }
    d->tl_packed = false;
    thip_token_lists &tl = d->tlp;
    const size_t nc = d->cl_start[3];
    for (int p = 0; p < 3; p++)
      for (int qti = 0; qti < 2; qti++) tl.dc_quant[p][qti] = d->dequant[(((size_t)d->qis[0] * 3 + p) * 2 + qti) * 64];
    uint16_t dq[18 * 64];
    memset(dq, 0, sizeof(dq));
    for (int p = 0; p < 3; p++)
      for (int qii = 0; qii < d->nqis; qii++)
        for (int qti = 0; qti < 2; qti++)
          memcpy(dq + ((p * 3 + qii) * 2 + qti) * 64, &d->dequant[(((size_t)d->qis[qii] * 3 + p) * 2 + qti) * 64], 128);
    tl.tokens = d->tl_tokens.data();
    tl.coded = d->tl_coded.data();
    tl.frag_meta = d->tl_meta.data();
    tl.dequant = dq;
    // The device starts on the lists; the DC prediction is undone on this side meanwhile unless the device is asked for that
    // too (option fe_device_dc) -- a chain through the plane in raster order, a few nanoseconds a fragment here, a dependent
    // step of a wave there -- and the values follow with the second call: the device needs them last.
    d->prof.lap(FE_LMETA);
    int lrc = THIP_EIMPL;
    if (d->tl_assigned) lrc = thip_state_token_lists_begin_assigned(d->hip, &tl, d->tl_assign.data(), d->tl_lastz.data());
    d->tl_assigned = false;
    if (lrc == THIP_EIMPL) lrc = thip_state_token_lists_begin(d->hip, &tl);   // (not paired, or no room for the pairing's arrays)
    d->prof.lap(FE_LBEGIN);
    if (lrc >= 0) {
      const int16_t *dcv = nullptr;
      if (with_worker) {
        join_worker();
        dcv = d->tl_dc.data();
        d->prof.lap(FE_DC);
      } else if (!d->device_dc) {
        if (!dc_done) {   // (an adopted frame brings the values along, in both orders)
          undo_dc();
          d->tl_dc.resize(nc + 1);
          for (size_t ci = 0; ci < nc; ci++) d->tl_dc[ci] = d->dc[d->clist[ci]];
        }
        dcv = d->tl_dc.data();
        d->prof.lap(FE_DC);
      }
      lrc = thip_state_token_lists_finish(d->hip, dcv);
      d->prof.lap(FE_LFINISH);
    }
    join_worker();   // (whatever happened: the second thread is done with the context's arrays)
    if (lrc >= 0) lists_done = true;
    else if (lrc != THIP_EIMPL) return TH_EFAULT;   // (THIP_EIMPL: a plane too large for that path -- the slots below)
    d->prof.lap(FE_EXPAND);
  }
  if (!lists_done) {
  if (!dc_done) undo_dc();
  d->prof.lap(FE_DC);
  // ---- 7.9 reconstruction through the backend's vtable slots -------------------------------------------
  int rc = 0;
  if (d->trace) {
    d->tr_fragi.clear(); d->tr_pli.clear(); d->tr_last_zzi.clear(); d->tr_refi.clear();
    d->tr_dcq.clear(); d->tr_mv.clear(); d->tr_coeffs.clear(); d->tr_uncoded.clear();
  } else {
    rc = thip_frame_begin(d->hip, d->frame_type);
    if (rc < 0) return TH_EFAULT;
  }
  const int flimit = d->setup.qp.lflims[d->qis[0]];
  d->tr_flimit = flimit;
  const bool use_tokens = d->device_tokens && !d->trace;
  // Option fe_levels = 1: the host's own walk hands the slot quantised LEVELS (thip_state_frag_recon_levels): the multiplication of
  // decode.c:1573 is the reconstruction kernel's and the staging a block needs halves.  Off by default: end to end it measured
  // the same within 2 % (720p and 1080p, 1 and 16 threads, tools/exp_fe_levels.sh) -- the walk is bound by the tokens.  The
  // slot-trace mode of the tests always records the dequantised coefficients.
  const bool use_levels = !use_tokens && !d->trace && thip_option("fe_levels") != 0;
  if (use_tokens || use_levels) {   // the frame's AC dequantisation tables (decode.c:1358-1366): number = (plane * 3 + qii) * 2 + qti
    for (int p = 0; p < 3; p++)
      for (int qii = 0; qii < d->nqis; qii++)
        for (int qti = 0; qti < 2; qti++)
          if (thip_frame_dequant_table(d->hip, (p * 3 + qii) * 2 + qti,
                                       &d->dequant[(((size_t)d->qis[qii] * 3 + p) * 2 + qti) * 64]) < 0)
            return TH_EFAULT;
  }
  {
    alignas(16) int16_t block[128];
    uint32_t toks[64];
    memset(block, 0, sizeof(block));
    for (int p = 0; p < 3; p++) {
      const Tok *tp[64];   // next token of every index list (each list ends in an endless EOB run)
      uint32_t run[64];
      for (int z = 0; z < 64; z++) {
        tp[z] = d->toks[p][z].data();
        run[z] = d->eob_carry[p][z];
      }
      for (size_t ci = d->cl_start[p]; ci < d->cl_start[p + 1]; ci++) {
        const int f = d->clist[ci];
        const int qti = d->mbmode_of_frag[f] != MODE_INTRA;
        const uint16_t *acq = &d->dequant[(((size_t)d->qis[d->qii[f]] * 3 + p) * 2 + qti) * 64];
        const uint16_t dcq = d->dequant[(((size_t)d->qis[0] * 3 + p) * 2 + qti) * 64];
        int z = 0, last_zzi = 0, ntok = 0;
        if (use_tokens) {
          // The tokens only get delimited here: which of them belong to this block.  Zero fill, zig-zag
          // scatter and `value * ac_quant` (decode.c:1573) happen on the device (k_expand_tokens).
          while (z < 64) {
            last_zzi = z;
            if (run[z]) {
              run[z]--;
              break;
            }
            const Tok &t = *tp[z]++;
            if (t.eob) {
              run[z] = t.eob - 1;
              break;
            }
            const int at = z + t.skip;
            if (t.value != 0 && at >= 1 && at <= 63) toks[ntok++] = (uint32_t)at << 16 | (uint16_t)t.value;
            z += t.adv;
          }
          const int16_t mvt = (int16_t)(((int)d->mvx[f] & 0xFF) | ((int)d->mvy[f] * 256));
          rc = thip_state_frag_recon_tokens(d->hip, f, p, toks, ntok, d->dc[f], last_zzi, dcq, (p * 3 + d->qii[f]) * 2 + qti,
                                            d->refi[f], mvt);
          if (rc < 0) return TH_EFAULT;
          continue;
        }
        while (z < 64) {
          last_zzi = z;
          if (run[z]) {   // inside an EOB run at this index
            run[z]--;
            break;
          }
          const Tok &t = *tp[z]++;
          if (t.eob) {
            run[z] = t.eob - 1;
            break;
          }
          // spec 7.9.2.  Unconditional: a pure zero run stores a zero where nothing has been stored
          // yet (positions only grow), and positions past 63 (a malformed run) land in the second
          // half of block[], which nobody reads (the reference keeps such a dump slot too,
          // decint.h:96).
          const int at = z + t.skip;   // <= 63 + 63
          block[kZigZagDump[at]] = use_levels ? (int16_t)t.value : (int16_t)(t.value * (int)acq[at & 63]);
          z += t.adv;
        }
        block[0] = d->dc[f];   // raw un-predicted DC; the slot dequantises it (state.c:967-979)
        const int16_t mv = (int16_t)(((int)d->mvx[f] & 0xFF) | ((int)d->mvy[f] * 256));
        if (d->trace) {
          d->tr_fragi.push_back(f);
          d->tr_pli.push_back((uint8_t)p);
          d->tr_last_zzi.push_back((uint8_t)last_zzi);
          d->tr_refi.push_back(d->refi[f]);
          d->tr_dcq.push_back(dcq);
          d->tr_mv.push_back(mv);
          d->tr_coeffs.insert(d->tr_coeffs.end(), block, block + 64);
          memset(block, 0, 64 * sizeof(block[0]));   // what the slot does (idct.c:245,276,295)
          continue;
        }
        rc = use_levels ? thip_state_frag_recon_levels(d->hip, f, p, block, last_zzi, dcq, d->qii[f], d->refi[f], mv)
                        : thip_state_frag_recon(d->hip, f, p, block, last_zzi, dcq, d->refi[f], mv);
        if (rc < 0) return TH_EFAULT;
      }
      const ptrdiff_t *const uncoded = d->ulist.data() + d->ul_start[p];
      const size_t nuncoded = d->ul_start[p + 1] - d->ul_start[p];
      if (d->trace) {
        for (size_t u = 0; u < nuncoded; u++) d->tr_uncoded.push_back((int64_t)uncoded[u]);
        continue;
      }
      if (nuncoded && thip_frag_copy_list(d->hip, uncoded, (ptrdiff_t)nuncoded) < 0) return TH_EFAULT;
      if (flimit && thip_state_loop_filter_frag_rows(d->hip, flimit, THIP_FRAME_SELF, p, 0, d->nv[p]) < 0)
        return TH_EFAULT;
    }
  }
  d->prof.lap(FE_EXPAND);
  if (!d->trace) {
    rc = thip_frame_flush(d->hip);
    if (rc < 0) return TH_EFAULT;
  } else if (d->tl_assigned) {
    // an adopted frame whose parser paired tokens and fragments for the device: every token applied on its own, as k_tok_scatter applies
    // them, must give the coefficients recorded above
    d->tl_assigned = false;
    if (!fe_check_assignment(d)) return TH_EFAULT;
    d->assign_checked++;
  }
  }   // (!lists_done)
  // ---- out-of-loop post-processing (decode.c:1203-1325, :2893-2911), on the backend -----------------------
  if (!d->trace) {
    if (d->pp_level <= 0) {
      d->dc_qis_tracked = false;                                  // decode.c:1209-1219
    } else if (!d->dc_qis_tracked) {
      if (d->frame_type == THIP_INTRA_FRAME) {                     // "no point in starting now" otherwise, decode.c:1221-1227
        d->dc_qis.assign((size_t)N, (uint8_t)d->qis[0]);
        d->dc_qis_tracked = true;
      }
    } else {
      const size_t nc = d->cl_start[3];
      const int *cl = d->clist.data();
      for (size_t i = 0; i < nc; i++) d->dc_qis[cl[i]] = (uint8_t)d->qis[0];   // decode.c:1236-1243
    }
    int lvl = d->dc_qis_tracked ? d->pp_level : 0;
    if (lvl >= 2) {
      d->frag_qi.resize((size_t)N);
      // decode.c:1926, as it is: an uncoded block's stale qii indexes THIS frame's qis[] (entries beyond nqis are older still)
      for (int f = 0; f < N; f++) d->frag_qi[f] = (uint8_t)d->qis[d->qii[f]];
      if (thip_state_postprocess(d->hip, lvl, d->dc_qis.data(), d->frag_qi.data(), d->pp_dc_scale, d->pp_sharp_mod) < 0)
        return TH_EFAULT;
    } else {
      static const uint8_t none = 0;
      static const int32_t zeros[64] = {0};
      (void)thip_state_postprocess(d->hip, 0, &none, &none, zeros, zeros);   // th_decode_ycbcr_out shows the decoded frame
    }
  }
  d->prof.lap(FE_FLUSH);
  d->prof.frames++;
  if (d->prof.on && !d->prof.warmed && d->prof.frames == 4) {
    memset(d->prof.acc, 0, sizeof(d->prof.acc));
    d->prof.frames = 0;
    d->prof.tokens = 0;
    d->prof.warmed = true;
  }
  d->have_frame = true;
  d->curframe_num++;
  if (granpos) *granpos = d->granpos;
  if (d->stripe_cb.stripe_decoded && !d->trace) {   // decode.c:2929-2941, once for the whole frame
    th_ycbcr_buffer yb;
    if (th_decode_ycbcr_out(d, yb) < 0) return TH_EFAULT;
    d->stripe_cb.stripe_decoded(d->stripe_cb.ctx, yb, 0, d->nv[0]);
  }
  return 0;
}

// ---- look-ahead: the front halves of announced packets on parser contexts of their own ---------------------------------------
// What th_decode_packetin reads out of a packet depends on the headers and on nothing an earlier frame left behind (the coded
// flags, modes, vectors, qi indices and tokens of a frame are self-contained; DC prediction runs inside a frame; only the
// PICTURES form a chain).  A caller that has packets in hand before their turn -- a demultiplexer's queue, a file -- announces them
// in decode order with th_decode_ctl(TH_DECCTL_THIP_PREFETCH_PACKET): each is copied and parsed (fe_front, then the DC chain) by
// a thread of its own on a parser context; the th_decode_packetin that later gets the same bytes adopts the result (vectors
// swapped, not copied) and does only what the chain of pictures needs: the hand-over to the device.  One stream is no longer one
// host thread.  A packet that was not announced, or does not match what was, is parsed the ordinary way (whatever was announced is
// dropped first); nothing changes for a caller that never asks.
constexpr int kFeLookaheadMax = 16;
struct FeSlot {
  th_dec_ctx *ctx = nullptr;        // the parser context (created with the slot's first packet)
  std::thread th;
  std::mutex mu;
  std::condition_variable cv, cv_done;
  bool go = false, quit = false;    // (under mu)
  std::atomic<int> done{0};         // the packet is parsed, rc says how it went (set under mu, read without)
  int rc = 0;
  bool want_lists = false;          // the owner takes the token-list path: the parser packs the lists too (fe_pack_lists)
  bool want_assign = false;         // ... and pairs tokens and fragments as it decodes them (decode_token_list<true>, option fe_assign)
  bool timed = false;               // option fe_prof: the parser's stages (its own thread's clock)
  double acc[4] = {0, 0, 0, 0};     // entropy decoder, DC chain, lists packed, (unused)
  long jobs = 0;
  std::vector<uint8_t> pkt;
  long bytes = 0;
};
struct FeLookahead {
  int nslots = 0, head = 0, count = 0;   // a ring in announcement order: the oldest packet is in slot `head`
  FeSlot slots[kFeLookaheadMax];
  cpu_set_t domain;
  bool placed = false;
  long adopted = 0, missed = 0;
  // Who pairs tokens and fragments -- the parser, as it decodes (k_tok_scatter on the device: nothing to walk), or the device
  // (k_tok_assign: a frame's 64 dependent rounds on one compute unit per plane) -- moves work between the two sides of the
  // pipeline: pairing costs a parser 3.5-4.3 ns a token (720p dense 1.1 ms a frame instead of 0.75), the device's walk 0.2 / 0.25 /
  // 1.0 ms of a 720p / 1080p / 4K frame behind the packet.  Which side has room depends on the frame size and on how many packets
  // the caller announces (4K, four ahead: the parsers are the bound and the device's walk is free, 184 against 136 frames/s;
  // eight ahead: 334 against 295), so option fe_assign = 2 (the default) MEASURES: the time from one adopted frame's
  // th_decode_packetin to the next, 24 frames with the parsers pairing, 24 without (the frames parsed under the other rule
  // skipped), then the better of the two for 1024 frames, and again.  A caller that paces itself sees no difference and keeps
  // the parsers pairing (less device time).
  int pair_mode = 1;       // what the next announced packet's parser is told
  int pair_phase = 0;      // 0: timing pair_mode 1; 1: timing pair_mode 0; 2: settled
  int pair_frames = 0;     // adopted frames in this phase
  double pair_sum[2] = {0, 0};
  int pair_cnt[2] = {0, 0};
  double pair_last = 0;    // when the previous adopted frame's th_decode_packetin began (0: the previous packet was not adopted)
  long pair_settled[2] = {0, 0};   // (statistics: how often each rule won)
};
constexpr int kFePairSample = 24;   // (how long a measured rule is kept: option fe_assign_settle, 1024 frames)

static void fe_init_frame_arrays(th_dec_ctx *d) {
  d->coded.assign(d->nfrags, 0);
  d->refi.assign(d->nfrags, 0);
  d->qii.assign(d->nfrags, 0);
  d->qii_dirty = false;
  d->mvx.assign(d->nfrags, 0);
  d->mvy.assign(d->nfrags, 0);
  d->dc.assign(d->nfrags, 0);
  d->mbmode_of_frag.assign(d->nfrags, 0);
  d->mbmodes.assign(d->mbs.size(), 0);
}

static th_dec_ctx *fe_new_parser(const th_dec_ctx *m) {
  th_dec_ctx *s = new (std::nothrow) th_dec_ctx();
  if (!s) return nullptr;
  s->info = m->info;
  s->setup = m->setup;   // (its own copy of the code tables: they are what the parser's core keeps in its cache)
  s->hip = nullptr;
  s->worker = nullptr;
  s->la = nullptr;
  s->parse_only = true;
  s->pair_on = false;
  s->tl_packed = s->tl_assigned = false;
  s->trace = false;
  s->device_dc = s->device_tokens = false;
  s->device_lists = 0;
  s->pp_level = 0;
  s->dc_qis_tracked = false;
  s->granpos_bias = m->granpos_bias;
  s->keyframe_num = s->curframe_num = 0;
  s->granpos = 0;
  s->have_frame = false;
  s->stripe_cb.ctx = nullptr;
  s->stripe_cb.stripe_decoded = nullptr;
  memset(&s->prof, 0, sizeof(s->prof));
  build_geometry(s);
  fe_init_frame_arrays(s);
  return s;
}

// one announced packet on a parser context: everything that reads the packet, then the DC chain (values in both orders)
static void fe_parse_job(FeSlot &sl) {
  th_dec_ctx *s = sl.ctx;
  ogg_packet op;
  memset(&op, 0, sizeof(op));
  op.packet = sl.pkt.data();
  op.bytes = sl.bytes;
  FeRun r;
  s->qii_dirty = true;   // (the coded blocks' entries are always written: the owner takes exactly those)
  double t[5] = {0, 0, 0, 0, 0};
  if (sl.timed) t[0] = fe_now();
  s->pair_on = sl.want_assign;
  int rc = fe_front(s, &op, nullptr, r);
  if (sl.timed) t[1] = fe_now();
  if (rc == kFeContinue) {
    fe_undo_dc(s);
    const size_t nc = s->cl_start[3];
    s->tl_dc.resize(nc + 1);
    for (size_t ci = 0; ci < nc; ci++) s->tl_dc[ci] = s->dc[s->clist[ci]];
    if (sl.timed) t[2] = fe_now();
    s->tl_packed = false;
    if (sl.want_lists) {
      fe_pack_lists(s);
      s->tl_packed = true;
      if (sl.timed) t[3] = t[4] = fe_now();
    }
    if (sl.timed && t[4] > 0) {
      for (int k = 0; k < 4; k++) sl.acc[k] += t[k + 1] - t[k];
      sl.jobs++;
    }
  }
  sl.rc = rc;
}

static void fe_slot_main(FeSlot *sl) {
  for (;;) {
    {
      std::unique_lock<std::mutex> lk(sl->mu);
      sl->cv.wait(lk, [&] { return sl->go || sl->quit; });
      if (sl->quit) return;
      sl->go = false;
    }
    fe_parse_job(*sl);
    {
      std::lock_guard<std::mutex> lk(sl->mu);   // (a waiter that has given up spinning sleeps on cv_done under mu)
      sl->done.store(1, std::memory_order_release);
    }
    sl->cv_done.notify_one();
  }
}

// the packet in this slot is parsed: a short spin (the usual case: the parsers are ahead, or nearly), then asleep -- a caller whose
// parsers are the bound would otherwise burn a core, and with it a share of a CPU quota, doing nothing
static void fe_slot_wait(FeSlot &sl) {
  for (unsigned spins = 0; spins < 4096; spins++) {
    if (sl.done.load(std::memory_order_acquire)) return;
    cpu_relax();
  }
  std::unique_lock<std::mutex> lk(sl.mu);
  sl.cv_done.wait(lk, [&] { return sl.done.load(std::memory_order_acquire) != 0; });
}

// every announced packet dropped (their parsers are waited for: they write into the slots' contexts)
static void fe_lookahead_drop(FeLookahead *la) {
  for (int i = 0; i < la->count; i++) {
    FeSlot &sl = la->slots[(la->head + i) % la->nslots];
    fe_slot_wait(sl);
  }
  la->missed += la->count;
  thip_option_add("fe_lookahead_missed", la->count);
  la->head = la->count = 0;
}

static void fe_lookahead_free(th_dec_ctx *d) {
  FeLookahead *la = d->la;
  if (!la) return;
  for (int i = 0; i < kFeLookaheadMax; i++) {
    FeSlot &sl = la->slots[i];
    if (sl.th.joinable()) {
      {
        std::lock_guard<std::mutex> lk(sl.mu);
        sl.quit = true;
      }
      sl.cv.notify_one();
      sl.th.join();
    }
    delete sl.ctx;   // (a parser context owns no device state, no threads and no count in g_fe_contexts)
  }
  if (d->prof.on) {
    double a[4] = {0, 0, 0, 0};
    long jobs = 0;
    for (int i = 0; i < kFeLookaheadMax; i++) {
      for (int k = 0; k < 4; k++) a[k] += la->slots[i].acc[k];
      jobs += la->slots[i].jobs;
    }
    if (jobs)
      fprintf(stderr, "[thip front end] look-ahead: a parser's frame: entropy decoder%s %.3f, DC chain %.3f, lists packed %.3f ms (%ld frames, %d parsers)\n",
              thip_option("fe_assign") ? " (pairing tokens and fragments)" : "", 1e3 * a[0] / jobs, 1e3 * a[1] / jobs, 1e3 * (a[2] + a[3]) / jobs, jobs, la->nslots);
  }
  if (d->prof.on)
    fprintf(stderr, "[thip front end] look-ahead: %ld packets adopted, %ld announced and not used, %ld walks checked (slot-trace mode); fe_assign = %d"
            " (measured: the parsers pairing won %ld times, the device's walk %ld)\n", la->adopted, la->missed, d->assign_checked, thip_option("fe_assign"),
            la->pair_settled[1], la->pair_settled[0]);
  delete la;
  d->la = nullptr;
}

// The parser threads next to the caller and never ON its CPU: on the CPUs that share its last-level cache, inside what the caller
// itself may use (see fe_worker_place), without the one the caller is running on -- a thread woken on its waker's CPU runs FIRST
// there, and the th_decode_ctl that announced a packet would return when the packet is parsed (measured: pthread_cond_signal
// 0.46 ms with a 0.5 ms job on the same CPU, 0.012 ms next to it).  Asked again when the caller turns up on one of their CPUs.
// false: the caller is confined to one CPU -- nothing would run beside it.
static bool fe_lookahead_place(FeLookahead *la) {
  cpu_set_t allowed;
  CPU_ZERO(&allowed);
  const bool know = sched_getaffinity(0, sizeof(allowed), &allowed) == 0;
  if (know && CPU_COUNT(&allowed) < 2) return false;
  if (thip_option("fe_worker_pin") == 0 || !know) return true;
  const int cpu = sched_getcpu();
  if (cpu < 0 || cpu >= CPU_SETSIZE) return true;
  if (la->placed && !CPU_ISSET(cpu, &la->domain)) return true;
  cpu_set_t set;
  if (fe_llc_cpus(cpu, &set)) CPU_AND(&set, &set, &allowed);
  else set = allowed;
  // (fewer CPUs in the cache domain than threads that want to run: the whole of what the caller may use)
  if (CPU_COUNT(&set) < la->nslots + 1) set = allowed;
  CPU_CLR(cpu, &set);
  for (int i = 0; i < la->nslots; i++)
    if (la->slots[i].th.joinable()) (void)pthread_setaffinity_np(la->slots[i].th.native_handle(), sizeof(set), &set);
  la->domain = set;
  la->placed = true;
  return true;
}

// TH_DECCTL_THIP_PREFETCH_PACKET.  0: announced; 1: not taken (no slot free, an empty packet, a context that leaves the DC chain or
// the token expansion to the device, a caller confined to one CPU) -- th_decode_packetin parses it itself then, as ever.
static int fe_prefetch(th_dec_ctx *d, const ogg_packet *op) {
  if (d->parse_only) return TH_EINVAL;
  if (op->bytes <= 0 || !op->packet) return 1;   // (a dropped frame: nothing to parse)
  if (d->device_dc || d->device_tokens) return 1;
  int want = thip_option("fe_lookahead");
  if (want <= 0) return 1;
  if (want > kFeLookaheadMax) want = kFeLookaheadMax;
  FeLookahead *la = d->la;
  if (!la) {
    la = d->la = new (std::nothrow) FeLookahead();
    if (!la) return 1;
    la->nslots = want;
  }
  if (la->count >= la->nslots) return 1;
  FeSlot &sl = la->slots[(la->head + la->count) % la->nslots];
  if (!sl.ctx) {
    sl.ctx = fe_new_parser(d);
    if (!sl.ctx) return 1;
  }
  if (!sl.th.joinable()) {
    try {
      sl.th = std::thread(fe_slot_main, &sl);
    } catch (...) {
      return 1;
    }
    la->placed = false;   // (a new thread: placed below with the others)
  }
  if (!fe_lookahead_place(la)) return 1;
  sl.bytes = op->bytes;
  sl.pkt.resize((size_t)op->bytes);
  memcpy(sl.pkt.data(), op->packet, (size_t)op->bytes);
  sl.want_lists = fe_lists_now(d) || d->trace;   // (slot-trace mode: packed and paired too, and checked against the host's own walk)
  const int asg = thip_option("fe_assign");
  sl.want_assign = sl.want_lists && (asg == 1 || (asg >= 2 && la->pair_mode));
  sl.timed = d->prof.on;
  sl.done.store(0, std::memory_order_relaxed);
  la->count++;
  {
    std::lock_guard<std::mutex> lk(sl.mu);
    sl.go = true;
  }
  sl.cv.notify_one();
  return 0;
}

// the oldest announced packet if it is `op`, parsed; null (and everything announced dropped) otherwise
static FeSlot *fe_lookahead_take(th_dec_ctx *d, const ogg_packet *op) {
  FeLookahead *la = d->la;
  if (!la || !la->count) return nullptr;
  // A dropped frame (a zero-byte packet, decode.c:2746) is never announced -- there is nothing to parse --, so it says nothing
  // about the announcements around it: they stay where they are.
  if (op->bytes <= 0) return nullptr;
  FeSlot &sl = la->slots[la->head];
  const bool same = op->bytes > 0 && op->bytes == sl.bytes && op->packet && !memcmp(op->packet, sl.pkt.data(), (size_t)sl.bytes) &&
                    !d->device_dc && !d->device_tokens;
  if (!same) {
    fe_lookahead_drop(la);
    return nullptr;
  }
  fe_slot_wait(sl);
  la->head = (la->head + 1) % la->nslots;
  la->count--;
  if (sl.rc != kFeContinue) {   // a frame without coded blocks, a packet the parser refused: the owner says so itself (cheap)
    la->missed++;
    thip_option_add("fe_lookahead_missed", 1);
    return nullptr;
  }
  la->adopted++;
  thip_option_add("fe_lookahead_adopted", 1);
  return &sl;
}

// the parsed frame of a parser context becomes the owner's: what fe_front would have left behind
static void fe_adopt(th_dec_ctx *d, th_dec_ctx *s) {
  d->coded.swap(s->coded);
  d->refi.swap(s->refi);
  d->mbmode_of_frag.swap(s->mbmode_of_frag);
  d->mvx.swap(s->mvx);
  d->mvy.swap(s->mvy);
  d->dc.swap(s->dc);
  d->clist.swap(s->clist);
  d->ulist.swap(s->ulist);
  d->tl_dc.swap(s->tl_dc);
  d->tl_assigned = false;
  d->tl_packed = s->tl_packed;
  if (s->tl_packed) {
    d->tl_tokens.swap(s->tl_tokens);
    d->tl_meta.swap(s->tl_meta);
    d->tl_coded.swap(s->tl_coded);
    d->tlp = s->tlp;
    s->tl_packed = false;
    d->tl_assigned = s->tl_assigned;
    if (s->tl_assigned) {
      d->tl_assign.swap(s->tl_assign);
      d->tl_lastz.swap(s->tl_lastz);
      s->tl_assigned = false;
    }
  }
  for (int p = 0; p < 3; p++)
    for (int z = 0; z < 64; z++) d->toks[p][z].swap(s->toks[p][z]);
  memcpy(d->cl_start, s->cl_start, sizeof(d->cl_start));
  memcpy(d->ul_start, s->ul_start, sizeof(d->ul_start));
  memcpy(d->ntoks, s->ntoks, sizeof(d->ntoks));
  memcpy(d->eob_carry, s->eob_carry, sizeof(d->eob_carry));
  memcpy(d->arrivals, s->arrivals, sizeof(d->arrivals));
  memcpy(d->qis, s->qis, sizeof(d->qis));
  d->nqis = s->nqis;
  d->frame_type = s->frame_type;
  // the qi index of a block outlives the frame (an uncoded block keeps the one it was last coded with, decode.c:913-917): the
  // owner's array takes the coded blocks' entries, as the pass of 7.6 in fe_front would have written them
  const size_t nc = d->cl_start[3];
  const int *cl = d->clist.data();
  if (d->nqis > 1 || d->qii_dirty) {
    const uint8_t *sq = s->qii.data();
    uint8_t *dq = d->qii.data();
    for (size_t i = 0; i < nc; i++) dq[cl[i]] = sq[cl[i]];
    if (d->frame_type == THIP_INTRA_FRAME) d->qii_dirty = false;
  }
  if (d->nqis > 1) d->qii_dirty = true;
  // frame counters (fe_front, 7.1 and 7.4)
  d->granpos = ((d->keyframe_num + d->granpos_bias) << d->info.keyframe_granule_shift) + (d->curframe_num - d->keyframe_num);
  if (d->frame_type == THIP_INTRA_FRAME) {
    d->keyframe_num = d->curframe_num;
    d->granpos = ((d->keyframe_num + d->granpos_bias) << d->info.keyframe_granule_shift);
  }
  if (d->prof.on)
    for (int p = 0; p < 3; p++)
      for (int z = 0; z < 64; z++) d->prof.tokens += (long)d->ntoks[p][z];
}

// option fe_assign = 2: which of the two rules is faster here (see FeLookahead); called with the time a th_decode_packetin began and
// whether it found its packet parsed
static void fe_pair_rule(FeLookahead *la, double now, bool adopted) {
  if (!adopted) {
    la->pair_last = 0;   // (the next interval would not be one between two adopted frames)
    return;
  }
  if (la->adopted < 4 * la->nslots + 8) {   // (a stream's first frames: parser contexts and threads come into being, their arrays grow)
    la->pair_last = now;
    return;
  }
  if (la->pair_last > 0 && la->pair_phase < 2) {
    la->pair_frames++;
    if (la->pair_frames > la->nslots + 2) {   // (the packets announced before the rule changed have gone through)
      la->pair_sum[la->pair_phase] += now - la->pair_last;
      if (++la->pair_cnt[la->pair_phase] >= kFePairSample) {
        if (la->pair_phase == 0) {
          la->pair_mode = 0;
          thip_option_add("fe_assign_to_device", 1);
        } else {
          const double with = la->pair_sum[0] / la->pair_cnt[0], without = la->pair_sum[1] / la->pair_cnt[1];
          la->pair_mode = with <= 1.03 * without ? 1 : 0;
          la->pair_settled[la->pair_mode]++;
          if (la->pair_mode) thip_option_add("fe_assign_to_parsers", 1);
        }
        la->pair_phase++;
        la->pair_frames = 0;
      }
    }
  } else if (la->pair_phase == 2 && ++la->pair_frames >= std::max(1, thip_option("fe_assign_settle"))) {
    la->pair_phase = 0;
    la->pair_frames = 0;
    if (!la->pair_mode) thip_option_add("fe_assign_to_parsers", 1);
    la->pair_mode = 1;
    la->pair_sum[0] = la->pair_sum[1] = 0;
    la->pair_cnt[0] = la->pair_cnt[1] = 0;
  }
  la->pair_last = now;
}

int th_decode_packetin(th_dec_ctx *d, const ogg_packet *op, int64_t *granpos) {
  if (!d || !op) return TH_EFAULT;
  if (d->parse_only) return TH_EINVAL;
  if (d->early.valid) {
    // th_decode_ycbcr_out has decoded the next announced packet ahead (option fe_pipeline)
    auto G = [&](int64_t key, int64_t cur) { return ((key + d->granpos_bias) << d->info.keyframe_granule_shift) + (cur - key); };
    if (op->bytes == 0) {
      // a dropped frame in between (decode.c:2746; such a packet cannot be announced): it changes no picture, so the frame decoded
      // ahead is still right; only the counters move -- this packet takes the number the other one had
      const int64_t k0 = d->early.key0, c0 = d->early.cur0;
      if (granpos) *granpos = G(k0, c0);
      if (d->frame_type == THIP_INTRA_FRAME) {
        d->keyframe_num = c0 + 1;
        d->early.granpos = G(c0 + 1, c0 + 1);
      } else {
        d->early.granpos = G(k0, c0 + 1);
      }
      d->granpos = d->early.granpos;
      d->early.cur0 = c0 + 1;
      d->curframe_num = c0 + 2;
      return TH_DUPFRAME;
    }
    d->early.valid = false;
    if ((size_t)op->bytes == d->early.pkt.size() && op->packet && !memcmp(op->packet, d->early.pkt.data(), d->early.pkt.size())) {
      if (granpos) *granpos = d->early.granpos;
      return d->early.rc;
    }
    // Another packet than the one th_decode_ycbcr_out decoded ahead (a seek, a caller that changed its mind): the frame is TAKEN
    // BACK.  The backend's reference ring goes back to where it stood (thip_state_ring_rewind: the discarded frame's kernels still
    // run, ahead of whatever comes next on the state's stream; what they wrote counts as unknown), the context's counters and the
    // blocks' qi indices likewise, and this packet is decoded as if nothing had happened (it may itself be the next
    // announced one).  (Round 5 returned TH_EINVAL here -- an announcement was a promise.)
    if (thip_state_ring_rewind(d->hip, d->early.mark) < 0) return TH_EFAULT;
    d->keyframe_num = d->early.key0;
    d->curframe_num = d->early.cur0;
    d->granpos = G(d->early.key0, d->early.cur0 - 1);
    d->frame_type = d->early.frame_type0;
    d->nqis = d->early.nqis0;
    memcpy(d->qis, d->early.qis0, sizeof(d->qis));
    d->qii_dirty = d->early.qii_dirty0;
    if (d->early.qii_saved && d->early.qii0.size() == d->qii.size()) d->qii.swap(d->early.qii0);
    d->early.qii_saved = false;
    thip_option_add("fe_pipeline_taken_back", 1);
    // (what else was announced is looked at below like any announcement: adopted if this packet is the oldest of them, dropped if not)
  }
  FeRun r;
  if (d->la && !d->la->count) d->la->pair_last = 0;
  if (d->la && d->la->count) {
    d->prof.start();
    const double now = fe_now();
    FeSlot *const sl = fe_lookahead_take(d, op);
    fe_pair_rule(d->la, now, sl != nullptr);
    if (sl) {
      fe_lists_rule(d, now, false);
      fe_adopt(d, sl->ctx);
      r.lists_now = fe_lists_now(d);
      r.dc_done = true;
      d->prof.lap(FE_TOKENS);   // (the wait for the parser, if any, and the adoption)
      return fe_back(d, granpos, r);
    }
  }
  if (d->device_lists < 0) fe_lists_rule(d, fe_now(), op->bytes > 0);
  const int rc = fe_front(d, op, granpos, r);
  if (rc != kFeContinue) return rc;
  return fe_back(d, granpos, r);
}

int th_decode_ycbcr_out(th_dec_ctx *d, th_ycbcr_buffer ycbcr) {
  if (!d || !ycbcr) return TH_EFAULT;
  // Before the first frame (and in slot-trace mode) the picture is the context's own blank image;
  // afterwards it is the backend's pinned image of the last decoded frame, which the decoding
  // launch itself has been filling (thip_state_set_eager_output): nothing is copied here.
  const uint8_t *src[3] = {d->mirror[0].data(), d->mirror[1].data(), d->mirror[2].data()};
  int32_t strides[3] = {d->nh[0] * 8, d->nh[1] * 8, d->nh[2] * 8};
  d->prof.start();
  if (d->have_frame && !d->trace) {
    // Option fe_pipeline (on by default since round 6: an announcement is no promise any more).  The caller's loop is th_decode_packetin(N),
    // th_decode_ycbcr_out(N), th_decode_packetin(N + 1), ...: the device works on frame N while this thread waits here, and sits idle
    // while this thread hands frame N + 1 over -- adoption, copies and a dozen launches, as long as the device's own share at 720p.
    // With the packets announced ahead the next frame is usually parsed by now, so it is handed over HERE, before the wait: the
    // picture of frame N is named first (thip_state_ycbcr_map_begin), frame N + 1 goes to the state's other host image behind
    // frame N's kernels, and its th_decode_packetin finds the work done.  What it costs: if ANOTHER packet comes next the frame is
    // taken back (th_decode_packetin: reference ring and counters put back, the work thrown away; a dropped frame in between is
    // fine), and a failed tile hand-over of frame N can no longer be repaired by decoding it again (THIP_EFAULT instead; never
    // observed outside its test).  On by default since round 6.
    bool held = d->early.valid;   // (a second th_decode_ycbcr_out for the same frame: the picture named the first time)
    FeLookahead *const la = d->la;
    if (!held && la && la->count && d->pp_level <= 0 && !d->stripe_cb.stripe_decoded && !d->device_dc && !d->device_tokens &&
        thip_option("fe_pipeline") != 0) {
      FeSlot &sl = la->slots[la->head];
      if (sl.done.load(std::memory_order_acquire) && sl.rc == kFeContinue && sl.bytes > 0) {   // (parsed already: nobody is waited for here)
        if (thip_state_ycbcr_map_begin(d->hip) < 0) return TH_EFAULT;
        held = true;
        la->head = (la->head + 1) % la->nslots;
        la->count--;
        la->adopted++;
        thip_option_add("fe_lookahead_adopted", 1);
        thip_option_add("fe_pipelined", 1);
        const double now = fe_now();
        fe_pair_rule(la, now, true);
        fe_lists_rule(d, now, false);
        d->early.pkt.assign(sl.pkt.begin(), sl.pkt.begin() + sl.bytes);
        d->early.key0 = d->keyframe_num;
        d->early.cur0 = d->curframe_num;
        // (what th_decode_packetin needs to take this frame back, should another packet come)
        if (thip_state_ring_mark(d->hip, d->early.mark) < 0) return TH_EFAULT;
        d->early.frame_type0 = d->frame_type;
        d->early.nqis0 = d->nqis;
        memcpy(d->early.qis0, d->qis, sizeof(d->qis));
        d->early.qii_dirty0 = d->qii_dirty;
        d->early.qii_saved = sl.ctx->nqis > 1 || d->qii_dirty;   // (fe_adopt writes the coded blocks' entries then)
        if (d->early.qii_saved) d->early.qii0.assign(d->qii.begin(), d->qii.end());
        fe_adopt(d, sl.ctx);
        FeRun r;
        r.lists_now = fe_lists_now(d);
        r.dc_done = true;
        d->early.granpos = 0;
        d->early.rc = fe_back(d, &d->early.granpos, r);
        d->early.valid = true;
      }
    }
    if ((held ? thip_state_ycbcr_map_end(d->hip, src, strides) : thip_state_ycbcr_map(d->hip, src, strides)) < 0) return TH_EFAULT;
  }
  d->prof.lap(FE_OUT);
  for (int p = 0; p < 3; p++) {
    ycbcr[p].width = d->nh[p] * 8;
    ycbcr[p].height = d->nv[p] * 8;
    ycbcr[p].stride = strides[p];
    ycbcr[p].data = const_cast<uint8_t *>(src[p]);
  }
  return 0;
}

}  // extern "C"
