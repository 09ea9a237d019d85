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
#define THIP_TL_ROFF 776    // where the list's entries start in the rank-ordered token array (k_tok_rank / k_tok_walk)
#define THIP_TL_HDR 968     // dwords

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
  const int16_t *dc_host;   // [ncoded] un-predicted DC values from the caller, coded order, or null (then dc_in is un-predicted on the device)
  uint32_t *info;           // command words (zeroed)
  uint32_t *slot0;          // first slot of every tile (zeroed)
  uint32_t *arr;            // [ncoded] scratch: rank -> fragment map of planes too large to keep it in LDS
  int4 *coeffs;             // coefficient slots, tile layout
  int ncoded;
  TlPlaneK p[3];
  // the indices this launch of k_tok_assign walks, [z0, z1): a frame's lists may arrive in groups of indices (the entropy decoder
  // finishes index after index, thip_state_token_lists_append), and between two launches the fragments' next index waits in pos_save
  int z0, z1;
  uint8_t *pos_save;        // [3][pos_pitch]
  int pos_pitch;            // bytes, a multiple of 16, >= the largest plane's coded fragments rounded up to 32
  // levels != 0: the coefficient slots are written in the LEVELS form (include/theora_hip.h: 64-byte units of int8, a tile whose
  // levels do not all fit eight bits as int16 over two units) and the multiplication of decode.c:1573 is the reconstruction kernel's:
  // tmp holds the levels as the tokens carry them, `wide` one word per tile, set by whoever meets a level beyond eight bits
  int levels;
  uint32_t *wide;           // [ntiles] (zeroed)
  uint32_t *rank;           // k_tok_rank -> k_tok_walk: for every list, its arrivals' tokens in the order of the arrivals
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

// One work group per plane.  LDS: pos[] bytes (the index a fragment arrives at next; 64 + z = finished, the
// last index it arrived at was z; 0xFF padding up to whole dwords), qs[n] bytes (its dequantisation table), arr[n] 16-bit
// (rank among this index's arrivals -> fragment), the header tables, 2 x 2 x 16 dwords for the scans.
// BIG: planes of more than kTlLdsFrags coded fragments (4K luma: 129 600) keep the rank -> fragment map in memory instead
// (K.arr, 32-bit entries, read back past the L1 after the work-group barrier) and read the table number from the fragment
// words: one byte of LDS per fragment.
// The kernel lives on ONE compute unit and is bound by the instructions that unit can issue, so a round (one index) is
// kept short: a thread looks at its fragments four at a time (`pos` as dwords, the bytes equal to z found with three
// integer operations and counted with one), both prefix sums of the index share one scan, done in registers (DPP) inside a
// wave and by sixteen lanes across the waves, the list's header words are in LDS, a thread's first tokens of the NEXT list are
// requested before it serves this list's and summed after -- three work-group barriers, no exposed memory round trip -- and
// the work group is no larger than the plane needs (the per-round cost of a wave that has nothing to do is the same as that
// of one that has): 256 threads up to 16 K coded fragments, 512 up to 48 K, 1024 beyond.
constexpr int kTlLdsFrags = 36864;   // per plane with the map in LDS: 4 bytes of LDS per fragment (1080p luma: 32 640)
constexpr int kTlMaxFrags = 147456;  // per plane at all: 1 byte of LDS per fragment
constexpr int kTlPrefetch = 4;       // tokens of the next list a thread asks for a round ahead
__host__ __device__ inline int tl_threads(int nmax) { return nmax <= 16384 ? 256 : (nmax <= 49152 ? 512 : 1024); }   // (a thread: up to kTlGroups x 32 fragments)

// Work-group barrier that waits for this wave's LDS traffic only.  __syncthreads() also waits for every global store to be
// acknowledged -- a round trip per round for stores (the coefficients, the DC values) nobody in this kernel reads back.
__device__ __forceinline__ void tl_barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// inclusive prefix sum over the 64 lanes of a wave, in registers (row_shr 1, 2, 4, 8 inside the rows of 16, then the last lane
// of a row to the rows behind it)
__device__ __forceinline__ uint32_t tl_wave_scan(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1, 3
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2, 3
  return v;
}
// exclusive prefix sums of two values at once over the work group (at most 16 waves); scr: 2 x 16 dwords nobody else writes
// before the next barrier
__device__ __forceinline__ void tl_exscan2(uint32_t a, uint32_t b, uint32_t (*scr)[16], uint32_t &ea, uint32_t &ta, uint32_t &eb) {
  const int lane = (int)threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), nw = (int)blockDim.x >> 6;
  const uint32_t ia = tl_wave_scan(a), ib = tl_wave_scan(b);
  if (lane == 63) {
    scr[0][wave] = ia;
    scr[1][wave] = ib;
  }
  tl_barrier_lds();
  uint32_t wa = lane < nw ? scr[0][lane & 15] : 0u, wb = lane < nw ? scr[1][lane & 15] : 0u;   // the waves' totals, lanes 0..15
  wa += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)wa, 0x111, 0xF, 0xF, true);
  wb += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)wb, 0x111, 0xF, 0xF, true);
  wa += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)wa, 0x112, 0xF, 0xF, true);
  wb += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)wb, 0x112, 0xF, 0xF, true);
  wa += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)wa, 0x114, 0xF, 0xF, true);
  wb += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)wb, 0x114, 0xF, 0xF, true);
  wa += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)wa, 0x118, 0xF, 0xF, true);
  wb += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)wb, 0x118, 0xF, 0xF, true);
  const uint32_t upto_a = (uint32_t)__builtin_amdgcn_readlane((int)wa, wave), upto_b = (uint32_t)__builtin_amdgcn_readlane((int)wb, wave);
  const uint32_t mine_a = (uint32_t)__builtin_amdgcn_readlane((int)ia, 63), mine_b = (uint32_t)__builtin_amdgcn_readlane((int)ib, 63);
  ta = (uint32_t)__builtin_amdgcn_readlane((int)wa, 15);
  ea = upto_a - mine_a + ia - a;
  eb = upto_b - mine_b + ib - b;
}
__device__ __forceinline__ uint32_t tl_cost(uint32_t tk) { return (tk & THIP_TOK_EOB) ? ((tk & 0xFFFFu) | (tk >> 24) << 16) : 1u; }
// bit 7 of every byte of w that equals the byte in zz (zz = z * 0x01010101)
__device__ __forceinline__ uint32_t tl_eq_bytes(uint32_t w, uint32_t zz) {
  const uint32_t x = w ^ zz;
  return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;
}

constexpr int kTlGroups = 5;         // groups of 32 fragments a thread looks after, at most (147 456 fragments / 32 / 1024 threads)
// one bit per fragment of a group of 32: set where the byte of `pos` equals z (zz = z * 0x01010101)
__device__ __forceinline__ uint32_t tl_group_mask(const uint32_t *posw, int g, uint32_t zz) {
  const uint4 a = *reinterpret_cast<const uint4 *>(posw + 8 * g), b = *reinterpret_cast<const uint4 *>(posw + 8 * g + 4);
  const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  uint32_t m = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) m |= ((((tl_eq_bytes(w[j], zz) >> 7) * 0x00204081u) >> 21) & 0xFu) << (4 * j);   // bits 0, 8, 16, 24 -> a nibble
  return m;
}

template <bool BIG>
__global__ __launch_bounds__(1024) void k_tok_assign(const TlK K) {
  extern __shared__ __attribute__((aligned(16))) uint8_t s_tl[];
  __shared__ uint16_t s_dq[18 * 64];
  __shared__ uint32_t s_scr[2][2][16];
  __shared__ uint32_t s_hdr[4][64];                      // first token, tokens, carry, arrivals of every list of the plane
  __shared__ uint8_t s_nat[64];                          // zig-zag index -> natural position
  const int p = (int)blockIdx.x, n = K.p[p].n, c0 = K.p[p].c0;
  if (n == 0) return;
  const int n32 = (n + 31) & ~31;
  uint8_t *pos = s_tl;
  uint32_t *posw = reinterpret_cast<uint32_t *>(s_tl);
  uint8_t *qsl = s_tl + n32;                             // (!BIG)
  uint16_t *arr = reinterpret_cast<uint16_t *>(s_tl + 2 * n32);
  uint32_t *garr = K.arr + c0;
  const int t = (int)threadIdx.x, T = (int)blockDim.x;
  const int lgT = 31 - __clz(T);                         // (the work group is 256, 512 or 1024 threads: shifts, not divisions)
  if (!K.levels)
    for (int i = t; i < 18 * 64; i += T) s_dq[i] = K.dq[i];
  if (t < 256) s_hdr[t >> 6][t & 63] = K.hdr[(t >> 6) * 192 + p * 64 + (t & 63)];   // THIP_TL_OFF / _LEN / _CARRY / _ARRIVE
  if (t < 64) s_nat[t] = (uint8_t)tl_nat(t);
  const int G = n32 >> 5;                                // groups of 32 fragments
  const int Kg = (G + T - 1) >> lgT;                     // <= kTlGroups
  const int g0 = min(t * Kg, G), g1 = min(g0 + Kg, G);   // this thread's fragments: 32 g0 .. 32 g1 - 1
  const int zend = K.z1;
  const bool lv = K.levels != 0;
  uint32_t *const saved = reinterpret_cast<uint32_t *>(K.pos_save + (size_t)p * K.pos_pitch);
  if (K.z0 == 0) {
    for (int i = t; i < n32 / 4; i += T) {
      const int left = n - 4 * i;                        // fragments in this dword (the rest is padding that never matches)
      posw[i] = left >= 4 ? 0u : (left <= 0 ? 0xFFFFFFFFu : (0xFFFFFFFFu << (8 * left)));
    }
  } else {
    for (int i = t; i < n32 / 4; i += T) posw[i] = saved[i];   // where the launch for the indices before z0 left the fragments
  }
  if (!BIG) {   // products: the fragment's table; levels: "has a level beyond eight bits" (this launch)
    if (!K.levels)
      for (int i = t; i < n; i += T) qsl[i] = (uint8_t)((K.meta[c0 + i] >> 2) & 31u);
    else
      for (int i = t; i < n32 / 4; i += T) reinterpret_cast<uint32_t *>(qsl)[i] = 0u;
  }
  __syncthreads();
  // this thread's share [j0, j1) of a list's tokens
  auto share = [&](int z, uint32_t &off, uint32_t &j0, uint32_t &j1) {
    off = s_hdr[0][z];
    const uint32_t m = s_hdr[1][z];
    const uint32_t Kt = (m + (uint32_t)T - 1u) >> lgT;
    j0 = min((uint32_t)t * Kt, m);
    j1 = min(j0 + Kt, m);
  };
  // (-DTHIP_TL_PROF: thread 0 of plane 0 adds up, phase by phase over the rounds, where the time goes and prints it)
#ifdef THIP_TL_PROF
  unsigned long long tp[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tq = 0;
  int rounds = 0;
#define TLP(k) { unsigned long long now_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)); tp[k] += now_ - tq; tq = now_; }
#else
#define TLP(k)
#endif
  // Four tokens served: the arrival of rank S gets a token that is not an EOB token.  Stage by stage for the four at once,
  // so that the dependent LDS reads of one token (rank -> fragment -> table -> factor) overlap with the others'.
  auto serve4 = [&](const uint32_t tk[kTlPrefetch], uint32_t jb, uint32_t jend, int z, uint32_t &S, uint32_t narr) {
    bool live[kTlPrefetch];
    uint32_t Sq[kTlPrefetch];
#pragma unroll
    for (int q = 0; q < kTlPrefetch; q++) {
      const bool valid = jb + (uint32_t)q < jend;
      const bool eob = (tk[q] & THIP_TOK_EOB) != 0;
      Sq[q] = S;
      live[q] = valid && !eob && S < narr;               // (a list longer than its arrivals: a malformed stream; the surplus is ignored)
      S += valid ? tl_cost(tk[q]) : 0u;
    }
    int fi[kTlPrefetch];
#pragma unroll
    for (int q = 0; q < kTlPrefetch; q++)
      fi[q] = !live[q] ? 0 : (BIG ? (int)__hip_atomic_load(garr + Sq[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (int)arr[Sq[q]]);
    asm volatile("" ::"v"(fi[0]), "v"(fi[1]), "v"(fi[2]), "v"(fi[3]));
    TLP(6)
    uint32_t qs[kTlPrefetch], nat[kTlPrefetch];
    int at[kTlPrefetch], cf[kTlPrefetch];
#pragma unroll
    for (int q = 0; q < kTlPrefetch; q++) {
      at[q] = z + (int)((tk[q] >> 16) & 127u);
      qs[q] = (!live[q] || lv) ? 0u : (BIG ? (K.meta[c0 + fi[q]] >> 2) & 31u : (uint32_t)qsl[fi[q]]);
      nat[q] = s_nat[at[q] & 63];
      cf[q] = (live[q] && at[q] == 0 && (tk[q] & 0xFFFFu)) ? K.clist[c0 + fi[q]] : 0;   // (DC tokens: the fragment's number)
    }
    int fac[kTlPrefetch];
#pragma unroll
    for (int q = 0; q < kTlPrefetch; q++) fac[q] = lv ? 1 : (int)s_dq[qs[q] * 64 + (uint32_t)(at[q] & 63)];
    asm volatile("" ::"v"(fac[0]), "v"(fac[1]), "v"(fac[2]), "v"(fac[3]), "v"(cf[0]), "v"(cf[1]), "v"(cf[2]), "v"(cf[3]));
    TLP(7)
#pragma unroll
    for (int q = 0; q < kTlPrefetch; q++) {
      if (!live[q]) continue;
      const int value = (int)(int16_t)(tk[q] & 0xFFFFu);
      if (value != 0) {
        if (at[q] == 0) K.dc_in[cf[q]] = (int16_t)value;   // the DC token value (un-predicted later, or the caller's is used)
        else if (at[q] <= 63) {
          K.tmp[(size_t)(c0 + fi[q]) * 64 + nat[q]] = (int16_t)(value * fac[q]);   // decode.c:1573 (levels form: the level itself)
          if (lv && (value > 127 || value < -128)) {   // the fragment's tile turns wide
            if (BIG) K.wide[K.frag_pos[K.clist[c0 + fi[q]]] >> 6] = 1u;
            else qsl[fi[q]] = 1;   // (a byte of LDS now; the two look-ups that find the tile when the launch ends, for the marked fragments only)
          }
        }
      }
      const int np = at[q] + (value != 0 ? 1 : 0);
      pos[fi[q]] = (uint8_t)(np < 64 ? np : 64 + z);   // (np <= 63 + 64)
    }
    TLP(8)
  };
  int z = K.z0;
  while (z < zend && s_hdr[3][z] == 0) z++;              // (uniform) the first index anybody arrives at
  uint32_t off = 0, j0 = 0, j1 = 0, use = 0, cur[kTlPrefetch];
#pragma unroll
  for (int q = 0; q < kTlPrefetch; q++) cur[q] = 0u;
  if (z < zend) {
    share(z, off, j0, j1);
#pragma unroll
    for (int q = 0; q < kTlPrefetch; q++) cur[q] = j0 + (uint32_t)q < j1 ? K.tok[off + j0 + (uint32_t)q] : 0u;
#pragma unroll
    for (int q = 0; q < kTlPrefetch; q++) use += j0 + (uint32_t)q < j1 ? tl_cost(cur[q]) : 0u;
    for (uint32_t j = j0 + kTlPrefetch; j < j1; j++) use += tl_cost(K.tok[off + j]);
  }
  int buf = 0;
#ifdef THIP_TL_PROF
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tq));
#endif
  while (z < zend) {
    int zn = z + 1;
    while (zn < zend && s_hdr[3][zn] == 0) zn++;
    // ---- the next list's first tokens are asked for now; they are looked at before this round's first store goes out (this
    //      chip counts loads and stores in one in-order counter: a load behind a store waits for the store's acknowledgement) ----
    uint32_t offn = 0, j0n = 0, j1n = 0, nx[kTlPrefetch];
#pragma unroll
    for (int q = 0; q < kTlPrefetch; q++) nx[q] = 0u;
    if (zn < zend) {
      share(zn, offn, j0n, j1n);
#pragma unroll
      for (int q = 0; q < kTlPrefetch; q++)
        if (j0n + (uint32_t)q < j1n) nx[q] = K.tok[offn + j0n + (uint32_t)q];
    }
    // ---- the arrivals at z, ranked in coded order; what the tokens before this thread's consume ----------------
    const uint32_t zz = (uint32_t)z * 0x01010101u;
    uint32_t cnt = 0, gm[kTlGroups];
#pragma unroll
    for (int k = 0; k < kTlGroups; k++) {
      gm[k] = g0 + k < g1 ? tl_group_mask(posw, g0 + k, zz) : 0u;
      cnt += (uint32_t)__popc(gm[k]);
    }
    uint32_t r, narr, S;
    TLP(0)
    tl_exscan2(cnt, use, s_scr[buf], r, narr, S);
    TLP(1)
    buf ^= 1;
    S += s_hdr[2][z];
#pragma unroll
    for (int k = 0; k < kTlGroups; k++) {
      uint32_t m = gm[k];
      const uint32_t base = (uint32_t)(g0 + k) * 32u;
      while (m) {
        const uint32_t i = base + (uint32_t)(__ffs((int)m) - 1);
        m &= m - 1u;
        if (BIG) garr[r++] = i;
        else arr[r++] = (uint16_t)i;
      }
    }
    if (BIG) {
      __threadfence_block();                             // the map's stores are done before the barrier lets anyone read it
      __syncthreads();
    } else {
      tl_barrier_lds();                                  // the map is there
    }
    TLP(2)
    // ---- the next list's share of the scan ------------------------------------------------------------------
    uint32_t usen = 0;
    if (zn < zend) {
#pragma unroll
      for (int q = 0; q < kTlPrefetch; q++) usen += j0n + (uint32_t)q < j1n ? tl_cost(nx[q]) : 0u;
      for (uint32_t j = j0n + kTlPrefetch; j < j1n; j++) usen += tl_cost(K.tok[offn + j]);
    }
    TLP(3)
    // ---- the tokens of this list, four at a time (the first four have been in registers since the round before) ---------
    serve4(cur, j0, j1, z, S, narr);
    for (uint32_t jb = j0 + kTlPrefetch; jb < j1; jb += kTlPrefetch) {   // (long lists: four loads, then their stores)
      uint32_t tk[kTlPrefetch];
#pragma unroll
      for (int q = 0; q < kTlPrefetch; q++) tk[q] = jb + (uint32_t)q < j1 ? K.tok[off + jb + (uint32_t)q] : 0u;
      serve4(tk, jb, j1, z, S, narr);
    }
    TLP(4)
    tl_barrier_lds();  // pos as the next index finds it
    TLP(5)
#ifdef THIP_TL_PROF
    rounds++;
#endif
    z = zn;
    off = offn;
    j0 = j0n;
    j1 = j1n;
    use = usen;
#pragma unroll
    for (int q = 0; q < kTlPrefetch; q++) cur[q] = nx[q];
  }
#ifdef THIP_TL_PROF
  if (t == 0 && p == 0)
    printf("k_tok_assign plane 0: n %d T %d rounds %d | count %llu scan %llu rank+barrier %llu nextuse %llu serve-rest %llu endbarrier %llu | serve: ranks %llu tables %llu stores %llu (10 ns ticks)\n", n, T, rounds,
           tp[0], tp[1], tp[2], tp[3], tp[4], tp[5], tp[6], tp[7], tp[8]);
#endif
  if (!BIG && lv)
    for (int i = t; i < n; i += T)
      if (qsl[i]) K.wide[K.frag_pos[K.clist[c0 + i]] >> 6] = 1u;
  if (zend < 64) {   // more indices to come: the next launch starts from here
    for (int i = t; i < n32 / 4; i += T) saved[i] = posw[i];
    return;
  }
  // last_zzi (decode.c:1545: the index the fragment's last token -- or the run that ended it -- was met at)
  for (int i = t; i < n; i += T) K.last_zzi[c0 + i] = (uint8_t)(pos[i] < 64 ? pos[i] : pos[i] - 64);
}
#undef TLP

// ---- the same walk with every fragment looked after by ONE thread (option tl_algo = 2) -----------------------------------------
// k_tok_assign pairs tokens and fragments through a rank -> fragment map that all threads write and all threads read: two prefix sums, three
// work-group barriers and two dependent LDS look-ups per token and round.  Which token an arrival gets depends on the arrival's RANK only --
// the rank-th unit of what the list's entries consume --, and that half of the question has nothing to do with the fragments: k_tok_rank
// answers it for a whole group of lists at once, one work group per list, into an array indexed by rank (EOB_MARK where the rank is
// swallowed by the carry or by an EOB run, the token word where a token serves it).  What is left for the 64 dependent rounds of
// k_tok_walk is: count my fragments' arrivals, ONE prefix sum over the work group, fetch my arrivals' entries (consecutive ranks:
// consecutive addresses), move my fragments on.  `pos` belongs to its owner: one barrier a round (inside the prefix sum), one byte of
// LDS per fragment for every plane size (no separate kernel for 4K), bit 7 of the byte = "has a level beyond eight bits".
#define THIP_TOK_RANK_EOB 0xFFFFFFFFu   // (not a token: bit 23 with a zero run of 127 and value 0xFFFF never leaves the front end)
constexpr int kTlRankThreads = 1024;
constexpr int kTlWalkBatch = 8;       // arrivals of a thread whose entries are requested together
__global__ __launch_bounds__(kTlRankThreads) void k_tok_rank(const TlK K) {
  __shared__ uint32_t s_scr[16];
  const int p = (int)blockIdx.x % 3, z = K.z0 + (int)blockIdx.x / 3;
  const uint32_t narr = K.hdr[THIP_TL_ARRIVE + p * 64 + z];
  if (narr == 0) return;
  const uint32_t off = K.hdr[THIP_TL_OFF + p * 64 + z], m = K.hdr[THIP_TL_LEN + p * 64 + z];
  uint32_t *const R = K.rank + K.hdr[THIP_TL_ROFF + p * 64 + z];
  const int t = (int)threadIdx.x;
  for (uint32_t i = (uint32_t)t; i < narr; i += kTlRankThreads) R[i] = THIP_TOK_RANK_EOB;
  __syncthreads();
  uint32_t S0 = K.hdr[THIP_TL_CARRY + p * 64 + z];   // what the entries before this chunk consume
  for (uint32_t j0 = 0; j0 < m && S0 < narr; j0 += 4u * kTlRankThreads) {
    const uint32_t j = j0 + 4u * (uint32_t)t;
    uint32_t tk[4], c[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      tk[q] = j + (uint32_t)q < m ? K.tok[off + j + (uint32_t)q] : 0u;
      // (a run counts for no more than the list has arrivals -- beyond that it means "everything" anyway --, so that the sums of a
      //  chunk stay far inside 32 bits whatever a malformed packet says: 4096 x 147 456 < 2^30)
      c[q] = j + (uint32_t)q < m ? min(tl_cost(tk[q]), narr) : 0u;
    }
    uint32_t total;
    uint32_t S = S0 + tl_exscan(c[0] + c[1] + c[2] + c[3], s_scr, total);
#pragma unroll
    for (int q = 0; q < 4; q++) {
      if (j + (uint32_t)q < m && !(tk[q] & THIP_TOK_EOB) && S < narr) R[S] = tk[q];
      S += c[q];
    }
    S0 += total;
  }
}

template <int T>
__global__ __launch_bounds__(T) void k_tok_walk(const TlK K) {
  extern __shared__ __attribute__((aligned(16))) uint8_t s_tl[];
  __shared__ uint16_t s_dq[18 * 64];
  __shared__ uint32_t s_scr[2][16];
  __shared__ uint32_t s_hdr[2][64];                      // arrivals, rank offset of every list of the plane
  __shared__ uint8_t s_nat[64];
  const int p = (int)blockIdx.x, n = K.p[p].n, c0 = K.p[p].c0;
  if (n == 0) return;
  const int n32 = (n + 31) & ~31;
  uint8_t *pos = s_tl;
  uint32_t *posw = reinterpret_cast<uint32_t *>(s_tl);
  const int t = (int)threadIdx.x;
  const bool lv = K.levels != 0;
  if (!lv)
    for (int i = t; i < 18 * 64; i += T) s_dq[i] = K.dq[i];
  if (t < 64) {
    s_hdr[0][t] = K.hdr[THIP_TL_ARRIVE + p * 64 + t];
    s_hdr[1][t] = K.hdr[THIP_TL_ROFF + p * 64 + t];
    s_nat[t] = (uint8_t)tl_nat(t);
  }
  constexpr int lgT = T == 256 ? 8 : T == 512 ? 9 : 10;
  const int G = n32 >> 5;
  const int Kg = (G + T - 1) >> lgT;                     // <= kTlGroups
  const int g0 = min(t * Kg, G), g1 = min(g0 + Kg, G);   // this thread's fragments: 32 g0 .. 32 g1 - 1, nobody else's
  uint32_t *const saved = reinterpret_cast<uint32_t *>(K.pos_save + (size_t)p * K.pos_pitch);
  // (a thread initialises / restores what it owns: no barrier between that and its first look at it)
  for (int i = 8 * g0; i < 8 * g1; i++) {
    const int left = n - 4 * i;
    posw[i] = K.z0 == 0 ? (left >= 4 ? 0u : (left <= 0 ? 0xFFFFFFFFu : (0xFFFFFFFFu << (8 * left)))) : saved[i];
  }
  __syncthreads();
  const int zend = K.z1;
  int buf = 0;
  for (int z = K.z0; z < zend; z++) {
    if (s_hdr[0][z] == 0) continue;                      // (uniform) nobody arrives at this index
    const uint32_t zz = (uint32_t)z * 0x01010101u;
    uint32_t cnt = 0, gm[kTlGroups];
#pragma unroll
    for (int k = 0; k < kTlGroups; k++) {
      gm[k] = 0u;
      if (g0 + k < g1) {
        const uint4 a = *reinterpret_cast<const uint4 *>(posw + 8 * (g0 + k)), b = *reinterpret_cast<const uint4 *>(posw + 8 * (g0 + k) + 4);
        const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
        for (int j = 0; j < 8; j++) gm[k] |= ((((tl_eq_bytes(w[j] & 0x7F7F7F7Fu, zz) >> 7) * 0x00204081u) >> 21) & 0xFu) << (4 * j);
      }
      cnt += (uint32_t)__popc(gm[k]);
    }
    // exclusive prefix sum of the arrival counts over the work group: the rank of this thread's first arrival
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const uint32_t incl = tl_wave_scan(cnt);
    if (lane == 63) s_scr[buf][wave] = incl;
    tl_barrier_lds();
    uint32_t wsum = lane < (T >> 6) ? s_scr[buf][lane & 15] : 0u;
    wsum += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)wsum, 0x111, 0xF, 0xF, true);
    wsum += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)wsum, 0x112, 0xF, 0xF, true);
    wsum += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)wsum, 0x114, 0xF, 0xF, true);
    wsum += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)wsum, 0x118, 0xF, 0xF, true);
    const uint32_t upto = (uint32_t)__builtin_amdgcn_readlane((int)wsum, wave), mine = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    uint32_t r = upto - mine + incl - cnt;
    buf ^= 1;
    const uint32_t *const R = K.rank + s_hdr[1][z];
    const uint32_t narr = s_hdr[0][z];
    // this thread's arrivals, kTlWalkBatch at a time: their entries (consecutive ranks: one round trip for the batch) first, then
    // what they say
#pragma unroll 1
    for (int k = 0; k < kTlGroups; k++) {
      uint32_t m = gm[k];
      const uint32_t base = (uint32_t)(g0 + k) * 32u;
      while (m) {
        int fi[kTlWalkBatch];
        uint32_t tk[kTlWalkBatch];
        bool live[kTlWalkBatch];
#pragma unroll
        for (int q = 0; q < kTlWalkBatch; q++) {
          live[q] = m != 0u;
          fi[q] = (int)base + (live[q] ? __ffs((int)m) - 1 : 0);
          m &= m - 1u;
        }
#pragma unroll
        for (int q = 0; q < kTlWalkBatch; q++) tk[q] = live[q] && r + (uint32_t)q < narr ? R[r + (uint32_t)q] : THIP_TOK_RANK_EOB;
#pragma unroll
        for (int q = 0; q < kTlWalkBatch; q++) {
          if (!live[q]) continue;
          r++;
          const uint8_t old = pos[fi[q]];
          if (tk[q] == THIP_TOK_RANK_EOB) {             // ended at this index (the carry, an EOB run, or a list that ran out)
            pos[fi[q]] = (uint8_t)((old & 0x80u) | (uint32_t)(64 + z));
            continue;
          }
          const int at = z + (int)((tk[q] >> 16) & 127u);
          const int value = (int)(int16_t)(tk[q] & 0xFFFFu);
          uint32_t big = old & 0x80u;
          if (value != 0) {
            if (at == 0) K.dc_in[K.clist[c0 + fi[q]]] = (int16_t)value;   // the DC token value (un-predicted later, or the caller's is used)
            else if (at <= 63) {
              const int fac = lv ? 1 : (int)s_dq[((K.meta[c0 + fi[q]] >> 2) & 31u) * 64 + (uint32_t)at];
              K.tmp[(size_t)(c0 + fi[q]) * 64 + s_nat[at]] = (int16_t)(value * fac);   // decode.c:1573 (levels form: the level itself)
              if (lv && (value > 127 || value < -128)) big = 0x80u;
            }
          }
          const int np = at + (value != 0 ? 1 : 0);
          pos[fi[q]] = (uint8_t)(big | (uint32_t)(np < 64 ? np : 64 + z));
        }
      }
    }
  }
  // (pos is its owner's: what follows needs no barrier either)
  if (zend < 64) {
    for (int i = 8 * g0; i < 8 * g1; i++) saved[i] = posw[i];
    return;
  }
  for (int i = 32 * g0; i < min(32 * g1, n); i++) {
    const uint32_t v = pos[i], q = v & 0x7Fu;
    K.last_zzi[c0 + i] = (uint8_t)(q < 64u ? q : q - 64u);   // decode.c:1545
    if (v & 0x80u) K.wide[K.frag_pos[K.clist[c0 + i]] >> 6] = 1u;
  }
}

// Levels form: which fragments lie in a wide tile -- bit 7 of their last_zzi byte (a thread per coded fragment: k_tok_slots, a single
// work group that walks the fragments in a loop, would pay the two dependent look-ups fragment after fragment).
__global__ __launch_bounds__(256) void k_tok_widths(const TlK K) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= K.ncoded) return;
  if (K.wide[K.frag_pos[K.clist[i]] >> 6]) K.last_zzi[i] |= 0x80u;
}

// Slots are handed out in coded order to the fragments that need one (last_zzi >= 2, state.c:967).  One group.
__global__ __launch_bounds__(1024) void k_tok_slots(const TlK K) {
  __shared__ uint32_t s_scr[16];
  const int t = (int)threadIdx.x, T = (int)blockDim.x, n = K.ncoded;
  const int Kf = (n + T - 1) / T;
  const int f0 = min(t * Kf, n), f1 = min(f0 + Kf, n);
  // (levels form: a block of a wide tile takes two units -- bit 7 of its last_zzi byte, k_tok_widths)
  auto need = [&](int i) -> uint32_t {
    const uint32_t lz = K.last_zzi[i];
    return (lz & 0x7Fu) < 2u ? 0u : 1u + (lz >> 7);
  };
  uint32_t cnt = 0;
  for (int i = f0; i < f1; i++) cnt += need(i);
  uint32_t total;
  uint32_t s = tl_exscan(cnt, s_scr, total);
  for (int i = f0; i < f1; i++) {
    K.slot[i] = s;
    s += need(i);
  }
}

// The same for large frames, in two launches of many groups: a group of 1024 threads looks after kTlSlotChunk fragments -- counts what
// they need (k_tok_slots_count -> part[group]), then adds up the groups before it and hands its own fragments their slots
// (k_tok_slots_assign).  (One group walking 194 400 fragments of a 4K frame took 181 us at the end of every frame.)
constexpr int kTlSlotChunk = 8192;
__device__ __forceinline__ uint32_t tl_slot_need(const TlK &K, int i) {
  const uint32_t lz = K.last_zzi[i];
  return (lz & 0x7Fu) < 2u ? 0u : 1u + (lz >> 7);
}
__global__ __launch_bounds__(1024) void k_tok_slots_count(const TlK K, uint32_t *part) {
  __shared__ uint32_t s_scr[16];
  const int i0 = (int)blockIdx.x * kTlSlotChunk + (int)threadIdx.x * (kTlSlotChunk / 1024);
  uint32_t cnt = 0;
  for (int i = i0; i < min(i0 + kTlSlotChunk / 1024, K.ncoded); i++) cnt += tl_slot_need(K, i);
  uint32_t total;
  (void)tl_exscan(cnt, s_scr, total);
  if (threadIdx.x == 0) part[blockIdx.x] = total;
}
__global__ __launch_bounds__(1024) void k_tok_slots_assign(const TlK K, const uint32_t *part) {
  __shared__ uint32_t s_scr[16];
  __shared__ uint32_t s_before;
  if (threadIdx.x < 64) {   // the groups before this one (a 4K frame has 24)
    uint32_t v = 0;
    for (int g = (int)threadIdx.x; g < (int)blockIdx.x; g += 64) v += part[g];
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    if (threadIdx.x == 0) s_before = v;
  }
  const int i0 = (int)blockIdx.x * kTlSlotChunk + (int)threadIdx.x * (kTlSlotChunk / 1024);
  const int i1 = min(i0 + kTlSlotChunk / 1024, K.ncoded);
  uint32_t cnt = 0;
  for (int i = i0; i < i1; i++) cnt += tl_slot_need(K, i);
  uint32_t total;
  uint32_t sl = tl_exscan(cnt, s_scr, total);   // (its barriers publish s_before too)
  sl += s_before;
  for (int i = i0; i < i1; i++) {
    K.slot[i] = sl;
    sl += tl_slot_need(K, i);
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
    // (the DC itself comes from the DC array, StreamK::dc, unless the caller has un-predicted it: then it travels as it does
    //  from the slots -- in the command word of a block without coefficients, as coefficient 0 of one with)
    K.info[2 * (size_t)pos + 1] = dcq << 16 | (K.dc_host ? (uint32_t)(uint16_t)K.dc_host[i] : 0u);
    // the first coded fragment of a tile: the slots handed out before it are the tile's first slot number
    if (i == 0 || (K.frag_pos[K.clist[i - 1]] >> 6) != (pos >> 6)) K.slot0[pos >> 6] = slot;
  }
  if (lz >= 2) {
    const int j = q >> 1, h = q & 1;
    uint2 a = *reinterpret_cast<const uint2 *>(K.tmp + (size_t)i * 64 + (2 * j) * 8 + 4 * h);
    if (q == 0 && K.dc_host) a.x = (a.x & 0xFFFF0000u) | (uint32_t)(uint16_t)K.dc_host[i];
    const uint2 b = *reinterpret_cast<const uint2 *>(K.tmp + (size_t)i * 64 + (2 * j + 1) * 8 + 4 * h);
    int4 o;
    o.x = (int)__builtin_amdgcn_perm(b.x, a.x, 0x05040100u);   // {a0, b0}
    o.y = (int)__builtin_amdgcn_perm(b.x, a.x, 0x07060302u);   // {a1, b1}
    o.z = (int)__builtin_amdgcn_perm(b.y, a.y, 0x05040100u);
    o.w = (int)__builtin_amdgcn_perm(b.y, a.y, 0x07060302u);
    K.coeffs[(size_t)(slot >> 6) * 512 + (size_t)q * 64 + (slot & 63)] = o;
  }
}

// The same in the levels form.  Eight threads per coded fragment again: thread q = 2j + h holds columns 4h..4h+3 of rows 2j, 2j+1 -- half
// of piece j of a narrow unit (eight int8: dwords 2h and 2h+1, bytes {x[2j][2d], x[2j][2d+1], x[2j+1][2d], x[2j+1][2d+1]}), or piece q of a
// wide block (int16 pairs, piece q of the block that starts at unit u lies in unit u + (q >> 2), piece q & 3).  The command word carries
// qii (the reconstruction kernel picks the table), word 1 the raw DC of EVERY block (coefficient 0 of a unit stays zero), the tile's
// first unit bit 31 when the tile is wide.
__global__ __launch_bounds__(256) void k_tok_write_levels(const TlK K) {
  const int g = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  const int i = g >> 3, q = g & 7;
  if (i >= K.ncoded) return;
  const int lzw = K.last_zzi[i], lz = lzw & 0x7F;
  const uint32_t m = K.meta[i];
  const int pos = K.frag_pos[K.clist[i]];
  const uint32_t unit = K.slot[i];
  const bool wide = (lzw & 0x80) != 0;
  if (q == 0) {
    const uint32_t pli = (m >> 24) & 3u, tab = (m >> 2) & 31u, qti = tab & 1u, qii = (tab >> 1) % 3u;
    const uint32_t dcq = K.hdr[THIP_TL_DCQ + pli * 2 + qti];
    uint32_t flags = THIP_INFO_CODED | (m & 3u) << THIP_INFO_REFI_SHIFT | qii << THIP_INFO_QII_SHIFT | (uint32_t)lz << THIP_INFO_LAST_ZZI_SHIFT |
                     ((m >> 8) & 0xFFu) << THIP_INFO_MVX_SHIFT | ((m >> 16) & 0xFFu) << THIP_INFO_MVY_SHIFT;
    if (lz < 2) flags |= THIP_INFO_DC_ONLY;
    K.info[2 * (size_t)pos] = flags;
    K.info[2 * (size_t)pos + 1] = dcq << 16 | (K.dc_host ? (uint32_t)(uint16_t)K.dc_host[i] : 0u);
    // the first coded fragment of a tile: the units handed out before it are the tile's first unit number
    if (i == 0 || (K.frag_pos[K.clist[i - 1]] >> 6) != (pos >> 6)) K.slot0[pos >> 6] = unit | (wide ? THIP_SLOT_WIDE : 0u);
  }
  if (lz >= 2) {
    const int j = q >> 1, h = q & 1;
    const uint2 a = *reinterpret_cast<const uint2 *>(K.tmp + (size_t)i * 64 + (2 * j) * 8 + 4 * h);       // x[2j][4h..4h+3]
    const uint2 b = *reinterpret_cast<const uint2 *>(K.tmp + (size_t)i * 64 + (2 * j + 1) * 8 + 4 * h);   // x[2j+1][4h..4h+3]
    uint8_t *const base = reinterpret_cast<uint8_t *>(K.coeffs);
    if (wide) {
      int4 o;
      o.x = (int)__builtin_amdgcn_perm(b.x, a.x, 0x05040100u);   // {a0, b0}
      o.y = (int)__builtin_amdgcn_perm(b.x, a.x, 0x07060302u);   // {a1, b1}
      o.z = (int)__builtin_amdgcn_perm(b.y, a.y, 0x05040100u);
      o.w = (int)__builtin_amdgcn_perm(b.y, a.y, 0x07060302u);
      const uint32_t u = unit + (uint32_t)(q >> 2);
      *reinterpret_cast<int4 *>(base + (size_t)(u >> 6) * THIP_UNIT_GROUP_BYTES + (size_t)(q & 3) * 1024 + (size_t)(u & 63) * 16) = o;
    } else {
      uint2 o;   // the low bytes of {a0, a1, b0, b1} and of {a2, a3, b2, b3}
      o.x = __builtin_amdgcn_perm(b.x, a.x, 0x06040200u);
      o.y = __builtin_amdgcn_perm(b.y, a.y, 0x06040200u);
      *reinterpret_cast<uint2 *>(base + (size_t)(unit >> 6) * THIP_UNIT_GROUP_BYTES + (size_t)j * 1024 + (size_t)(unit & 63) * 16 + 8 * h) = o;
    }
  }
}

// Staging -> device and the zero fills, one launch: dst[0..ncopy) = src (16-byte units; src is pinned host memory the
// kernel reads across PCIe), then the four areas that must be zero before the frame's kernels run.
struct TlPrepK {
  const int4 *src;
  int4 *dst;
  size_t ncopy;
  int4 *z[5];
  size_t nz[5];   // 16-byte units
};
// A group of lists goes to the device: the header tables (they grow index by index) and the group's tokens.
// ---- the walk done by the caller: every token comes with the fragment it belongs to ---------------------------------------------
// thip_state_token_lists_begin_assigned: a caller that has walked the lists itself (th_decode_*'s look-ahead does it on its parser
// threads, where a frame's 64 dependent rounds cost nothing on anybody's critical path) hands over, beside the tokens, one word
// per token -- bits 0-17 the fragment's index in coded order, bits 18-24 the zig-zag position the value lands at (list index +
// zeros before it), all ones for a token nobody consumes -- and the last index of every fragment.  What is left for the device is
// what k_tok_assign does AFTER it knows the pairing, and that is independent per token: one thread a token, the whole chip.
__global__ __launch_bounds__(256) void k_tok_scatter(const TlK K, const uint32_t *__restrict__ asg, const uint8_t *__restrict__ lz, int ntok) {
  const int j = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (j < K.ncoded) K.last_zzi[j] = (uint8_t)(lz[j] & 63u);   // decode.c:1545
  if (j >= ntok) return;
  const uint32_t tk = K.tok[j];
  if (tk & THIP_TOK_EOB) return;
  const uint32_t w = asg[j];
  const uint32_t ci = w & 0x3FFFFu;
  const int at = (int)((w >> 18) & 127u);
  if (w == 0xFFFFFFFFu || ci >= (uint32_t)K.ncoded) return;   // (a token no fragment consumed: the surplus of a malformed list)
  const int value = (int)(int16_t)(tk & 0xFFFFu);
  if (value == 0) return;                                      // a pure zero run
  if (at == 0) {
    K.dc_in[K.clist[ci]] = (int16_t)value;                     // the DC token value (un-predicted later, or the caller's is used)
  } else if (at <= 63) {
    const uint32_t qs = (K.meta[ci] >> 2) & 31u;
    const int fac = K.levels ? 1 : (int)K.dq[qs * 64 + (uint32_t)at];
    K.tmp[(size_t)ci * 64 + tl_nat(at)] = (int16_t)(value * fac);   // decode.c:1573 (levels form: the level itself)
    if (K.levels && (value > 127 || value < -128)) K.wide[K.frag_pos[K.clist[ci]] >> 6] = 1u;   // the fragment's tile turns wide
  }
}

struct TlCopyK {
  const int4 *src[2];
  int4 *dst[2];
  size_t n[2];   // 16-byte units
};
__global__ __launch_bounds__(256) void k_tok_copy(const TlCopyK C) {
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x, G = (size_t)gridDim.x * blockDim.x;
#pragma unroll
  for (int a = 0; a < 2; a++)
    for (size_t i = g; i < C.n[a]; i += G) C.dst[a][i] = C.src[a][i];
}
__global__ __launch_bounds__(256) void k_tok_prepare(const TlPrepK P) {
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x, G = (size_t)gridDim.x * blockDim.x;
  for (size_t i = g; i < P.ncopy; i += G) P.dst[i] = P.src[i];
  const int4 zero = make_int4(0, 0, 0, 0);
#pragma unroll
  for (int a = 0; a < 5; a++)
    for (size_t i = g; i < P.nz[a]; i += G) P.z[a][i] = zero;
}
