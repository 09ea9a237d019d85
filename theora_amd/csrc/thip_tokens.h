// thip_tokens.h -- SURVEY section 8(f) rank 1 in its parallel form: the DCT tokens of a frame, as the entropy
// decoder leaves them -- one list per (plane, zig-zag index), decode.c:993-1139 -- go to the device, and the
// device works out which token belongs to which fragment, expands and dequantises (decode.c:1511-1587),
// builds the command words, the coefficient slots and the DC array that k_dc_unpredict / k_recon read.
// Included by thip_decode.hip only.
//
// What the reference does per fragment, in coded order (decode.c:1540-1581): zzi = 0; while zzi < 64: if an
// EOB run is pending at this index of this plane, take one of it and stop; else take the NEXT token of the
// list (plane, zzi): an EOB token starts a run and stops the fragment, any other token puts a value at
// zzi + skip and moves the fragment on to index zzi + adv.  So at every index z the fragments that ARRIVE at
// z, in coded order, meet the list's entries in list order: first `carry` fragments are ended by the run left
// over from earlier lists, then every EOB token ends as many fragments as its run says and every other token
// serves one.  With r = rank of a fragment among the arrivals at z and S_j = carry + (sum of what the tokens
// before token j consume), token j serves the fragment of rank S_j: two prefix sums per index, 64 indices one
// after the other (a fragment's arrival at z depends on what it was given before), one work group per plane
// with the fragments' next index (a byte each) and the rank -> fragment map in LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// device token: bits 0-15 value (or, for an EOB token, the low 16 bits of its run), 16-22 zeros before the
// value (a zero run can be 64 long), bit 23 EOB token, 24-31 the high bits of an EOB run.  A token with value 0
// and bit 23 clear is a pure zero run (the fragment advances by skip), a value token advances by skip + 1.
#define THIP_TOK_EOB 0x00800000u
// per coded fragment: bits 0-1 refi, 2-6 dequantisation table (plane * 3 + qii) * 2 + qti, 8-15 mvx, 16-23 mvy,
// 24-25 plane
// header tables, 3 x 64 dwords each, index plane * 64 + zzi
#define THIP_TL_OFF 0       // first token of the list
#define THIP_TL_LEN 192     // tokens in the list
#define THIP_TL_CARRY 384   // fragments ended at this index by an EOB run from an earlier list
#define THIP_TL_ARRIVE 576  // fragments that reach this index at all
#define THIP_TL_DCQ 768     // dc_quant[plane][qti], 6 dwords
#define THIP_TL_HDR 776     // dwords

struct TlPlaneK {
  int n;    // coded fragments of the plane
  int c0;   // index of its first one in the frame's coded order
};
struct TlK {
  const uint32_t *hdr;
  const uint32_t *tok;
  const int32_t *clist;     // coded fragments (fragment numbers), coded order, plane after plane
  const uint32_t *meta;     // per coded fragment, same order
  const uint16_t *dq;       // [18][64] AC dequantisation tables, zig-zag order
  const int32_t *frag_pos;  // fragment number -> tile * 64 + lane
  int16_t *tmp;             // [ncoded][64] coefficients, natural order (zeroed before k_tok_assign)
  uint8_t *last_zzi;        // [ncoded]
  uint32_t *slot;           // [ncoded] coefficient slot of a fragment that has one
  int16_t *dc_in;           // [nfrags] DC token values, fragment order (zeroed)
  uint32_t *info;           // command words (zeroed)
  uint32_t *slot0;          // first slot of every tile (zeroed)
  uint32_t *arr;            // [ncoded] scratch: rank -> fragment map of planes too large to keep it in LDS
  int4 *coeffs;             // coefficient slots, tile layout
  int ncoded;
  TlPlaneK p[3];
};

// exclusive prefix sum over the work group (blockDim.x a multiple of 64, at most 1024); scr: 16 dwords of LDS
__device__ __forceinline__ uint32_t tl_exscan(uint32_t v, uint32_t *scr, uint32_t &total) {
  const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6, nw = (int)blockDim.x >> 6;
  uint32_t incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(incl, d);
    if (lane >= d) incl += o;
  }
  if (lane == 63) scr[wave] = incl;
  __syncthreads();
  uint32_t before = 0, all = 0;
  for (int w = 0; w < nw; w++) {
    const uint32_t s = scr[w];
    before += w < wave ? s : 0u;
    all += s;
  }
  __syncthreads();   // scr may be written again
  total = all;
  return before + incl - v;
}

__device__ __forceinline__ int tl_nat(int zzi) {   // natural-order position of zig-zag index zzi
  constexpr uint8_t T[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
  return T[zzi];
}

// One work group per plane.  LDS: pos[n] bytes (the index a fragment arrives at next; 64 + z = finished, the
// last index it arrived at was z), arr[n] 16-bit (rank among this index's arrivals -> fragment), 16 dwords.
// BIG: planes of more than kTlLdsFrags coded fragments (4K luma: 129 600) keep the rank -> fragment map in memory instead
// (K.arr, 32-bit entries, read back past the L1 after the work-group barrier): one byte of LDS per fragment.
constexpr int kTlLdsFrags = 49152;   // per plane with the map in LDS: 3 bytes of LDS per fragment
constexpr int kTlMaxFrags = 147456;  // per plane at all: 1 byte of LDS per fragment
template <bool BIG>
__global__ __launch_bounds__(1024) void k_tok_assign(const TlK K) {
  extern __shared__ __attribute__((aligned(16))) uint8_t s_tl[];
  __shared__ uint16_t s_dq[18 * 64];
  __shared__ uint32_t s_scr[16];
  const int p = (int)blockIdx.x, n = K.p[p].n, c0 = K.p[p].c0;
  if (n == 0) return;
  uint8_t *pos = s_tl;
  uint16_t *arr = reinterpret_cast<uint16_t *>(s_tl + ((n + 15) & ~15));
  uint32_t *garr = K.arr + c0;
  const int t = (int)threadIdx.x, T = (int)blockDim.x;
  for (int i = t; i < 18 * 64; i += T) s_dq[i] = K.dq[i];
  const int Kf = (n + T - 1) / T;
  const int f0 = min(t * Kf, n), f1 = min(f0 + Kf, n);   // this thread's fragments
  for (int i = f0; i < f1; i++) pos[i] = 0;
  __syncthreads();
  const uint32_t *hdr = K.hdr + p * 64;
  for (int z = 0; z < 64; z++) {
    if (hdr[THIP_TL_ARRIVE + z] == 0) continue;          // (uniform) nobody gets this far
    // ---- the arrivals at z, ranked in coded order --------------------------------------------------
    uint32_t cnt = 0;
    for (int i = f0; i < f1; i++) cnt += pos[i] == z ? 1u : 0u;
    uint32_t narr;
    uint32_t r = tl_exscan(cnt, s_scr, narr);
    for (int i = f0; i < f1; i++)
      if (pos[i] == z) {
        if (BIG) garr[r++] = (uint32_t)i;
        else arr[r++] = (uint16_t)i;
      }
    // ---- what the tokens of the list consume ----------------------------------------------------------
    const uint32_t off = hdr[THIP_TL_OFF + z], m = hdr[THIP_TL_LEN + z], carry = hdr[THIP_TL_CARRY + z];
    const uint32_t Kt = (m + (uint32_t)T - 1u) / (uint32_t)T;
    const uint32_t j0 = min((uint32_t)t * Kt, m), j1 = min(j0 + Kt, m);
    uint32_t use = 0;
    for (uint32_t j = j0; j < j1; j++) {
      const uint32_t tk = K.tok[off + j];
      use += (tk & THIP_TOK_EOB) ? ((tk & 0xFFFFu) | (tk >> 24) << 16) : 1u;
    }
    uint32_t dummy;
    if (BIG) __threadfence_block();                      // the map's stores are done before the barriers below let anyone read it
    uint32_t S = carry + tl_exscan(use, s_scr, dummy);   // (its barriers also publish arr)
    // ---- every token that is not an EOB token serves the arrival of rank S ------------------------------
    for (uint32_t j = j0; j < j1; j++) {
      const uint32_t tk = K.tok[off + j];
      if (tk & THIP_TOK_EOB) {
        S += (tk & 0xFFFFu) | (tk >> 24) << 16;
        continue;
      }
      if (S < narr) {   // (a list longer than its arrivals: a malformed stream; the surplus is ignored)
        const int i = BIG ? (int)__hip_atomic_load(garr + S, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (int)arr[S];
        const int skip = (int)((tk >> 16) & 127u);
        const int value = (int)(int16_t)(tk & 0xFFFFu);
        const int at = z + skip;
        if (value != 0) {
          if (at == 0) K.dc_in[K.clist[c0 + i]] = (int16_t)value;   // the DC token value (un-predicted later)
          else if (at <= 63) {
            const uint32_t qs = (K.meta[c0 + i] >> 2) & 31u;
            K.tmp[(size_t)(c0 + i) * 64 + tl_nat(at)] = (int16_t)(value * (int)s_dq[qs * 64 + at]);   // decode.c:1573
          }
        }
        const int np = at + (value != 0 ? 1 : 0);
        pos[i] = (uint8_t)(np < 64 ? np : 64 + z);   // (np <= 63 + 64)
      }
      S += 1;
    }
    __syncthreads();   // pos as the next index finds it
  }
  // last_zzi (decode.c:1545: the index the fragment's last token -- or the run that ended it -- was met at)
  for (int i = f0; i < f1; i++) K.last_zzi[c0 + i] = (uint8_t)(pos[i] < 64 ? pos[i] : pos[i] - 64);
}

// Slots are handed out in coded order to the fragments that need one (last_zzi >= 2, state.c:967).  One group.
__global__ __launch_bounds__(1024) void k_tok_slots(const TlK K) {
  __shared__ uint32_t s_scr[16];
  const int t = (int)threadIdx.x, T = (int)blockDim.x, n = K.ncoded;
  const int Kf = (n + T - 1) / T;
  const int f0 = min(t * Kf, n), f1 = min(f0 + Kf, n);
  uint32_t cnt = 0;
  for (int i = f0; i < f1; i++) cnt += K.last_zzi[i] >= 2 ? 1u : 0u;
  uint32_t total;
  uint32_t s = tl_exscan(cnt, s_scr, total);
  for (int i = f0; i < f1; i++) {
    K.slot[i] = s;
    s += K.last_zzi[i] >= 2 ? 1u : 0u;
  }
}

// Eight threads per coded fragment: its command words, its tile's first slot, its coefficients into the tile layout
// (piece q = 2j + h: columns 4h..4h+3 of rows 2j and 2j+1 as pairs, see thip_state_frag_recon).
__global__ __launch_bounds__(256) void k_tok_write(const TlK K) {
  const int g = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  const int i = g >> 3, q = g & 7;
  if (i >= K.ncoded) return;
  const int lz = K.last_zzi[i];
  const uint32_t m = K.meta[i];
  const int pos = K.frag_pos[K.clist[i]];
  const uint32_t slot = K.slot[i];
  if (q == 0) {
    const uint32_t pli = (m >> 24) & 3u, qti = (m >> 2) & 1u;
    const uint32_t dcq = K.hdr[THIP_TL_DCQ + pli * 2 + qti];
    uint32_t flags = THIP_INFO_CODED | (m & 3u) << THIP_INFO_REFI_SHIFT | (uint32_t)lz << THIP_INFO_LAST_ZZI_SHIFT |
                     ((m >> 8) & 0xFFu) << THIP_INFO_MVX_SHIFT | ((m >> 16) & 0xFFu) << THIP_INFO_MVY_SHIFT;
    if (lz < 2) flags |= THIP_INFO_DC_ONLY;
    K.info[2 * (size_t)pos] = flags;
    K.info[2 * (size_t)pos + 1] = dcq << 16;   // (the DC itself comes from the DC array: StreamK::dc)
    // the first coded fragment of a tile: the slots handed out before it are the tile's first slot number
    if (i == 0 || (K.frag_pos[K.clist[i - 1]] >> 6) != (pos >> 6)) K.slot0[pos >> 6] = slot;
  }
  if (lz >= 2) {
    const int j = q >> 1, h = q & 1;
    const uint2 a = *reinterpret_cast<const uint2 *>(K.tmp + (size_t)i * 64 + (2 * j) * 8 + 4 * h);
    const uint2 b = *reinterpret_cast<const uint2 *>(K.tmp + (size_t)i * 64 + (2 * j + 1) * 8 + 4 * h);
    int4 o;
    o.x = (int)__builtin_amdgcn_perm(b.x, a.x, 0x05040100u);   // {a0, b0}
    o.y = (int)__builtin_amdgcn_perm(b.x, a.x, 0x07060302u);   // {a1, b1}
    o.z = (int)__builtin_amdgcn_perm(b.y, a.y, 0x05040100u);
    o.w = (int)__builtin_amdgcn_perm(b.y, a.y, 0x07060302u);
    K.coeffs[(size_t)(slot >> 6) * 512 + (size_t)q * 64 + (slot & 63)] = o;
  }
}

// Staging -> device and the zero fills, one launch: dst[0..ncopy) = src (16-byte units; src is pinned host memory the
// kernel reads across PCIe), then the four areas that must be zero before the frame's kernels run.
struct TlPrepK {
  const int4 *src;
  int4 *dst;
  size_t ncopy;
  int4 *z[4];
  size_t nz[4];   // 16-byte units
};
__global__ __launch_bounds__(256) void k_tok_prepare(const TlPrepK P) {
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x, G = (size_t)gridDim.x * blockDim.x;
  for (size_t i = g; i < P.ncopy; i += G) P.dst[i] = P.src[i];
  const int4 zero = make_int4(0, 0, 0, 0);
#pragma unroll
  for (int a = 0; a < 4; a++)
    for (size_t i = g; i < P.nz[a]; i += G) P.z[a][i] = zero;
}
