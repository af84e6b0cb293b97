// k_tile_core.h -- what the tile kernel (k_tiles.hip) and the word-mode kernels (k_words.hip) share: the wavefront's tile in LDS, the
// workgroup's aggregator of count deltas, the site search in registers, the single-site rewrite, and process_tile -- K3's pair count and
// K4's merge-apply on a staged tile (bpe.cpp:436-478 and :491-812).  Device code, all of it inline or templates: a kernel of either file
// instantiates what it needs.  (Until round 4 the head of k_merge.hip.)
#pragma once
#include <stdlib.h>

#include <algorithm>

#include <type_traits>

#include "yttm_device.h"
#include "yttm_kernels.h"
#include "k_merge_shared.h"

namespace yttm {


// Staged (LDS) token word: bit31 = first token of a word, bit30 = id is the y of some batch rule, bit29 = id is the x
// of some batch rule, bits 0..28 = id.  HBM tokens carry only bit31 + id.
constexpr uint32_t L_ISX = 1u << 29, L_ISY = 1u << 30, L_ID = (1u << 29) - 1;

// per-wavefront tile state in LDS
template <int SLOT>
struct WaveLds {
  uint32_t tk[SLOT + 4];                   // staged tokens (+ sentinels)
  uint16_t ridx[SLOT];                     // merge site at p: index of its rule in the batch (z = z_base + ridx)
  unsigned long long wsmask[SLOT / 64];    // K3: bit p of chunk c: token 64 c + p starts a word.  K4: the same bits as the registers hold
                                           // them, mask 4 j + i = ballot over lanes l of "token 256 j + 4 l + i starts a word" (stage_ws_masks)
  unsigned long long sitemask[SLOT / 64];  // bit p: a merge (tk[p],tk[p+1]) starts at p
  uint32_t wsbase[SLOT / 64];              // number of word starts before the chunk (K4: before row j)
  uint16_t sitepos[64];                    // positions of the (up to) 64 merge sites a pass of phase 2 works on
  unsigned int sctl[2];                    // K4: number of merge sites found in the tile, position of the first one
};
struct AggLds {
  unsigned long long key[AGG_SLOTS];
  unsigned long long val[AGG_SLOTS];
  uint32_t flagbits[FLAG_LDS_IDS / 16];  // 2 bits per token id: bit0 = x of a batch rule, bit1 = y of a batch rule
  unsigned int new_keys;                 // slots claimed by this workgroup (added to pt.n_keys once, at the end)
  unsigned long long st[6];              // workgroup-local stats (one global atomic each at the end); [4],[5]: measurement pass only
#ifdef YTTM_K4_PROF
  unsigned long long miss_n, miss_cyc;
#endif
};


// Count deltas of hot pairs are summed in a small LDS hash shared by the workgroup before they become HBM atomics
// (cdna guide, Guideline 12): early in training there are few distinct pairs with huge counts, and without the hash
// every tile would hammer the same few HBM addresses.  What misses the hash goes straight to the HBM pair table.
template <int SLOT>
__device__ inline void emit(AggLds &A, WaveLds<SLOT> &W, const PairTable &pt, const DeltaBuf &db, unsigned long long key, long long delta) {
  (void)W;
  unsigned int h = (pair_hash32(key) >> 7) & (AGG_SLOTS - 1);
  for (int probe = 0; probe < 8; probe++) {
    unsigned long long k = __hip_atomic_load(&A.key[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // ds_read, not a flat load
    if (k == PT_EMPTY) {
      k = atomicCAS(&A.key[h], PT_EMPTY, key);
      if (k == PT_EMPTY) k = key;
    }
    if (k == key) {
      atomicAdd(&A.val[h], (unsigned long long)delta);
      return;
    }
    h = (h + 1) & (AGG_SLOTS - 1);
  }
#ifdef YTTM_K4_PROF
  const unsigned long long t0_ = (unsigned long long)clock64();
#endif
  global_emit(pt, db, key, delta, &A.new_keys);
#ifdef YTTM_K4_PROF
  atomicAdd(&A.miss_n, 1ull);
  atomicAdd(&A.miss_cyc, (unsigned long long)clock64() - t0_);
#endif
}


// The LDS half of an emit alone: false if the workgroup's aggregator had no room for the key (word mode batches what is left, below).
__device__ inline bool agg_try(AggLds &A, unsigned long long key, long long delta) {
  unsigned int h = (pair_hash32(key) >> 7) & (AGG_SLOTS - 1);
  for (int probe = 0; probe < 8; probe++) {
    unsigned long long k = __hip_atomic_load(&A.key[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (k == PT_EMPTY) {
      k = atomicCAS(&A.key[h], PT_EMPTY, key);
      if (k == PT_EMPTY) k = key;
    }
    if (k == key) {
      atomicAdd(&A.val[h], (unsigned long long)delta);
      return true;
    }
    h = (h + 1) & (AGG_SLOTS - 1);
  }
  return false;
}

// ---- word mode: count updates leave k_words as RECORDS.  k_words gathers words from all over the table, so its deltas are mostly of
// pairs the workgroup's aggregator has never seen, and pt_add's way -- probe, then an add whose old value tells whether the count crossed
// a list threshold -- made the wave wait two dependent trips per emit, five emits per pass, behind every other add to the same pair
// (measured: 9 us per emit call with a miss; adds that return nothing only move the wait to the wave's next load: one in-order counter).
// So what the aggregator does not take is written to the workgroup's region of a record buffer -- plain 16-byte stores, positions from an
// LDS counter -- and k_delta_apply, one thread per record, puts the records into the pair table after the words are done.
struct DeltaOut {
  DeltaRec *recs;          // this workgroup's region
  unsigned int *n;         // (LDS) records written / asked for
  unsigned int cap;
};
template <int N>
__device__ inline void rec_emit_batch(const DeltaOut &D, const PairTable &pt, const DeltaBuf &db, const unsigned long long (&key)[N],
                                      const long long (&delta)[N], const bool (&miss)[N], unsigned int *new_keys) {
#pragma unroll
  for (int j = 0; j < N; j++) {
    const unsigned long long m = __ballot(miss[j]);
    if (!m) continue;
    unsigned int b0 = 0;
    const int fl = __ffsll((long long)m) - 1;
    if (lane_id() == fl) b0 = atomicAdd(D.n, (unsigned int)__popcll(m));
    b0 = (unsigned int)__shfl((int)b0, fl);
    if (miss[j]) {
      const unsigned int at = b0 + (unsigned int)__popcll(m & lanemask_lt());
      if (at < D.cap) {
        DeltaRec r;
        r.key = key[j];
        r.delta = delta[j];
        D.recs[at] = r;
      } else {
        global_emit(pt, db, key[j], delta[j], new_keys);  // (the region is full: the slow way)
      }
    }
  }
}

template <int NT>
__device__ inline void agg_init(AggLds &A, const uint32_t *__restrict__ flagbits_g) {
  for (int s = (int)threadIdx.x; s < AGG_SLOTS; s += NT) {
    A.key[s] = PT_EMPTY;
    A.val[s] = 0;
  }
  if (flagbits_g)
    for (int s = (int)threadIdx.x; s < (int)(FLAG_LDS_IDS / 16); s += NT) A.flagbits[s] = flagbits_g[s];
  if (threadIdx.x == 0) {
    A.new_keys = 0;
    A.st[0] = A.st[1] = A.st[2] = A.st[3] = A.st[4] = A.st[5] = 0;
#ifdef YTTM_K4_PROF
    A.miss_n = A.miss_cyc = 0;
#endif
  }
}
template <int NT>
__device__ inline void agg_flush(AggLds &A, const PairTable &pt, const DeltaBuf &db) {
  __syncthreads();
  for (int s = (int)threadIdx.x; s < AGG_SLOTS; s += NT) {
    unsigned long long k = A.key[s];
    if (k != PT_EMPTY) {
      long long v = (long long)A.val[s];
      if (v != 0) global_emit(pt, db, k, v, &A.new_keys);
    }
  }
}

// 16 B/lane coalesced loads of a tile into registers (issued one tile ahead of use: the HBM latency of the next tile
// hides behind the processing of the current one)
template <int SLOT>
__device__ inline void tile_fetch(uint4 (&r)[SLOT / 256], const TileSet &ts, uint32_t t, int n) {
  const uint4 *src = reinterpret_cast<const uint4 *>(ts.tok + (size_t)t * SLOT);
#pragma unroll
  for (int j = 0; j < SLOT / 256; j++) {
    const int i = lane_id() + 64 * j;
    r[j] = (4 * i < n) ? src[i] : make_uint4(0, 0, 0, 0);
  }
}


// The batch's rule hash as the apply kernel sees it: in LDS when it fits (the usual case: <= APPLY_LDS_RULES/2 rules), so
// that processing a tile issues NO global load -- any such load would also wait (vmcnt is in-order) for the prefetch of
// the wave's next tile, a random HBM access that costs several microseconds late in training.
constexpr unsigned int APPLY_LDS_RULES = 512;
template <bool IN_LDS>
struct RuleTab {
  const unsigned long long *lds_keys;  // [mask+1] (IN_LDS)
  const uint16_t *lds_ridx;            // z - z_base
  const RuleSlot *g;                   // the hash in HBM (!IN_LDS: batches of more than APPLY_LDS_RULES/2 rules)
  unsigned int mask;
  uint32_t z_base;
  // index of the rule in the batch, or 0xffffffff.  Two instantiations, not a run-time choice: with both paths in one
  // function the compiler waits for vmcnt(0) where they join, prefetch included.
  __device__ uint32_t find(uint32_t a, uint32_t b) const {
    const unsigned long long key = pair_key(a, b);
    unsigned int h = pair_hash32(key) & mask;
    for (;;) {
      unsigned long long k;
      if (IN_LDS) k = __hip_atomic_load(&lds_keys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else k = g[h].key;
      if (k == key) return IN_LDS ? (uint32_t)lds_ridx[h] : g[h].z - z_base;
      if (k == PT_EMPTY) return 0xffffffffu;
      h = (h + 1) & mask;
    }
  }
};

// a merge site at position p joins the tile's list (any order; the first 64 are listed, the count goes on) and the minimum
template <int SLOT>
__device__ inline void site_listed(WaveLds<SLOT> &W, int p) {
  const unsigned int idx = atomicAdd(&W.sctl[0], 1u) & 0xffffu;
  if (idx < 64u) W.sitepos[idx] = (uint16_t)p;
  atomicMin(&W.sctl[1], (unsigned int)p);
}
// K4: word-start bits of a dirty tile go to LDS the way the registers hold them (no pass over the staged tokens); the word
// that contains position p is then tile_word_index_rl(p)
template <int SLOT>
__device__ inline void stage_ws_masks(WaveLds<SLOT> &W, const uint4 (&r)[SLOT / 256], int n) {
  uint32_t before = 0;
#pragma unroll
  for (int j = 0; j < SLOT / 256; j++) {
    if (256 * j < n) {
      const unsigned long long m0 = __ballot(r[j].x >> 31), m1 = __ballot(r[j].y >> 31), m2 = __ballot(r[j].z >> 31), m3 = __ballot(r[j].w >> 31);
      if (lane_id() == 0) {
        W.wsmask[4 * j] = m0; W.wsmask[4 * j + 1] = m1; W.wsmask[4 * j + 2] = m2; W.wsmask[4 * j + 3] = m3;
        W.wsbase[j] = before;
      }
      before += (uint32_t)(__popcll(m0) + __popcll(m1) + __popcll(m2) + __popcll(m3));
    }
  }
}
template <int SLOT>
__device__ inline uint32_t tile_word_index_rl(const WaveLds<SLOT> &W, int p) {
  const int j = p >> 8, l = (p >> 2) & 63, c = p & 3;
  const unsigned long long lt = (1ull << l) - 1ull;
  uint32_t k = W.wsbase[j];
#pragma unroll
  for (int cc = 0; cc < 4; cc++) {
    const unsigned long long m = W.wsmask[4 * j + cc];
    k += (uint32_t)__popcll(m & lt);
    if (cc <= c) k += (uint32_t)((m >> l) & 1ull);
  }
  return k - 1u;
}

// The same decision with the batch's PAIR filter (k_merge_shared.h: Bloom filter of the batch's pairs, two bits per rule in one word;
// BatchArgs::bloom): bit 4 j + i of `hb` = the adjacency that starts at my token i of row j passes the filter.  The x / y flags are per
// token -- with k rules up to k * k flagged adjacencies, of which k are rules: from the middle of a training on, two thirds of the
// flag-dirty tiles hold no merge site, and every flagged adjacency costs an exact look-up.  The pair filter's hits are nearly all sites.
template <int SLOT, class bits_t>
__device__ inline bool reg_bloom_test(const uint4 (&r)[SLOT / 256], int n, const uint32_t *bloom, uint32_t self_x, bits_t &hb) {
  const int lane = lane_id();
  hb = 0;
  bool selfc = false;
  const bool has_self = self_x != 0xffffffffu;
#define BLOOM_BIT(HA, HB, T1, S)                                                                  \
  {                                                                                               \
    const uint32_t h_ = (HA) ^ (HB);                                                              \
    const uint32_t bits_ = pm_bits(h_);                                                           \
    if ((bloom[pm_word(h_)] & bits_) == bits_ && !((T1) >> 31)) hb |= (bits_t)1 << (S);           \
  }
#pragma unroll
  for (int j = 0; j < SLOT / 256; j++) {
    if (256 * j < n) {
      uint32_t nx = from_lane_right(r[j].x);
      uint32_t nx0 = TOK_WS;
      if (j + 1 < SLOT / 256) nx0 = from_lane0(r[j + 1 < SLOT / 256 ? j + 1 : j].x);
      if (lane == 63) nx = nx0;
      const uint32_t a0 = r[j].x & TOK_MASK, a1 = r[j].y & TOK_MASK, a2 = r[j].z & TOK_MASK, a3 = r[j].w & TOK_MASK, a4 = nx & TOK_MASK;
      BLOOM_BIT(pm_mul24(a0, PM_K1), pm_mul24(a1, PM_K2), r[j].y, 4 * j)
      BLOOM_BIT(pm_mul24(a1, PM_K1), pm_mul24(a2, PM_K2), r[j].z, 4 * j + 1)
      BLOOM_BIT(pm_mul24(a2, PM_K1), pm_mul24(a3, PM_K2), r[j].w, 4 * j + 2)
      BLOOM_BIT(pm_mul24(a3, PM_K1), pm_mul24(a4, PM_K2), nx, 4 * j + 3)
      if (has_self)
        selfc = selfc || (a0 == self_x && a1 == self_x && !(r[j].y >> 31)) || (a1 == self_x && a2 == self_x && !(r[j].z >> 31)) ||
                (a2 == self_x && a3 == self_x && !(r[j].w >> 31)) || (a3 == self_x && a4 == self_x && !(nx >> 31));
    }
  }
#undef BLOOM_BIT
  return hb != 0 || selfc;
}

// K4, the tile still in registers: find the merge sites of the batch's x != y rules -- rule index to W.ridx[p], bit p of
// W.sitemask -- with one hash lookup per flagged adjacency, before anything is staged.  Returns 0 for a tile with neither
// such a site nor an x x of the self rule (nothing to do: the x/y flags are per token, and late in training two thirds of
// the tiles with a flagged adjacency hold no merge site), else 1, plus 2 if the self rule may have sites (those need the
// run they sit in and are found from LDS once the tile is staged).
template <int SLOT, bool LDSR, bool DIRECT = false>
__device__ inline int reg_find_sites(WaveLds<SLOT> &W, const uint4 (&r)[SLOT / 256], int n, const uint32_t *flagbits_lds /* the batch's pair filter; DIRECT: the pair -> rule table */,
                                     uint32_t self_x, const RuleTab<LDSR> &rtab,
                                     uint32_t &my_cnt /* sites found by this lane */, uint32_t &my_site /* the last one: position << 16 | rule index */,
                                     uint32_t direct_v = 0) {
  const int lane = lane_id();
  my_cnt = 0;
  my_site = 0;
  typedef typename std::conditional<(SLOT / 64 > 32), unsigned long long, uint32_t>::type bits_t;
  bits_t hb = 0;  // bit 4 j + i: the adjacency that starts at my token i of row j may be a rule of the batch
  if constexpr (DIRECT) {
    // Small alphabets' first rounds -- every tile holds dozens of sites, every one of the lane's eight adjacencies is some lane's candidate:
    // the pair filter only adds its cost to the hash probes.  While all ids are below direct_v the pair itself indexes a byte table in LDS
    // (rule number, 0xff: none): one ds_read_u8 per adjacency.  (Slots behind the tile's end hold id 0, a special token: never in a rule.)
    const uint8_t *tab = reinterpret_cast<const uint8_t *>(flagbits_lds);
    static_assert(SLOT / 64 <= 8, "rule numbers of a lane's adjacencies: two words");
    uint32_t ri_lo = 0xffffffffu, ri_hi = 0xffffffffu;  // byte 4 j + i: rule of the adjacency that starts at my token i of row j
    bool selfp = false;
    const bool has_self = self_x != 0xffffffffu;
#pragma unroll
    for (int j = 0; j < SLOT / 256; j++) {
      if (256 * j < n) {
        uint32_t nx = from_lane_right(r[j].x);
        uint32_t nx0 = TOK_WS;
        if (j + 1 < SLOT / 256) nx0 = from_lane0(r[j + 1 < SLOT / 256 ? j + 1 : j].x);
        if (lane == 63) nx = nx0;
        const uint32_t a0 = r[j].x & L_ID, a1 = r[j].y & L_ID, a2 = r[j].z & L_ID, a3 = r[j].w & L_ID, a4 = nx & L_ID;
        const uint32_t q0 = (r[j].y >> 31) ? 0xffu : (uint32_t)tab[a0 * direct_v + a1], q1 = (r[j].z >> 31) ? 0xffu : (uint32_t)tab[a1 * direct_v + a2];
        const uint32_t q2 = (r[j].w >> 31) ? 0xffu : (uint32_t)tab[a2 * direct_v + a3], q3 = (nx >> 31) ? 0xffu : (uint32_t)tab[a3 * direct_v + a4];
        const uint32_t packed = q0 | (q1 << 8) | (q2 << 16) | (q3 << 24);
        if (j == 0) ri_lo = packed; else ri_hi = packed;
        if (has_self)
          selfp = selfp || (a0 == self_x && a1 == self_x && !(r[j].y >> 31)) || (a1 == self_x && a2 == self_x && !(r[j].z >> 31)) ||
                  (a2 == self_x && a3 == self_x && !(r[j].w >> 31)) || (a3 == self_x && a4 == self_x && !(nx >> 31));
      }
    }
    // bit s of my_bits: byte s is a rule number (its top bit is clear: rule numbers are below 128)
    const uint32_t nl = ~ri_lo & 0x80808080u, nh = ~ri_hi & 0x80808080u;
    const bool found = (nl | nh) != 0u;
    if (__ballot(found || selfp) == 0) return 0;
    if (lane < SLOT / 64) W.sitemask[lane] = 0ull;
    if (lane == 0) {
      W.sctl[0] = 0u;
      W.sctl[1] = 0xffffffffu;
    }
    wave_sync();
    uint32_t *sm32 = reinterpret_cast<uint32_t *>(W.sitemask);
    if (found) {
      uint32_t my_bits = 0, my_ri = 0;
#pragma unroll
      for (int s = 0; s < 4 * (SLOT / 256); s++) {
        const uint32_t ri = ((s < 4 ? ri_lo : ri_hi) >> (8 * (s & 3))) & 0xffu;
        if (ri != 0xffu) {
          W.ridx[256 * (s >> 2) + 4 * lane + (s & 3)] = (uint16_t)ri;
          my_bits |= 1u << s;
          my_ri = ri;
        }
      }
#pragma unroll
      for (int j = 0; j < SLOT / 256; j++) {
        const uint32_t nib = (my_bits >> (4 * j)) & 15u;
        if (nib) atomicOr(&sm32[(256 * j + 4 * lane) >> 5], nib << ((4 * lane) & 31));
      }
      const int s_first = __ffs((int)my_bits) - 1, s_last = 31 - __clz((int)my_bits);
      const uint32_t p_first = (uint32_t)(256 * (s_first >> 2) + 4 * lane + (s_first & 3));
      my_cnt = (uint32_t)__popc(my_bits);
      my_site = ((uint32_t)(256 * (s_last >> 2) + 4 * lane + (s_last & 3)) << 16) | my_ri;
      const unsigned int idx = atomicAdd(&W.sctl[0], my_cnt | (my_cnt > 1 ? 0x10000u : 0u)) & 0xffffu;
      if (my_cnt == 1 && idx < 64u) W.sitepos[idx] = (uint16_t)p_first;
      atomicMin(&W.sctl[1], p_first);
    }
    wave_sync();
    return (__ballot(found) ? 1 : 0) | (__ballot(selfp) ? 3 : 0);
  }
  const bool cand = reg_bloom_test<SLOT, bits_t>(r, n, flagbits_lds, self_x, hb);
  if (__ballot(cand) == 0) return 0;
  if (lane < SLOT / 64) W.sitemask[lane] = 0ull;
  if (lane == 0) {
    W.sctl[0] = 0u;
    W.sctl[1] = 0xffffffffu;
  }
  wave_sync();
  uint32_t *sm32 = reinterpret_cast<uint32_t *>(W.sitemask);
  bool selfp = false;
  // what this lane finds: bit 4 j + i = a site starts at my token i of row j (position 256 j + 4 lane + i); the site bits,
  // the list and the count go to LDS once, after the look-ups (a lane's positions are looked at in ascending order)
  bits_t my_bits = 0;
  uint32_t my_ri = 0;  // rule of my last site
  const bool has_self = self_x != 0xffffffffu;  // (uniform: most batches have no x x rule)
#define PAIR_SITE(T0, T1, P, S)                                                              \
  if (!((T1)&TOK_WS)) {                                                                  \
    const uint32_t a_ = (T0)&L_ID, b_ = (T1)&L_ID;                                       \
    if (has_self && a_ == self_x && b_ == self_x) {                                      \
      selfp = true;                                                                      \
    } else if ((hb >> (S)) & 1u) {                                                        \
      const uint32_t ri = rtab.find(a_, b_);                                             \
      if (ri != 0xffffffffu) {                                                           \
        W.ridx[(P)] = (uint16_t)ri;                                                      \
        my_bits |= (bits_t)1 << (S);                                                     \
        my_ri = ri;                                                                      \
      }                                                                                  \
    }                                                                                    \
  }
#pragma unroll
  for (int j = 0; j < SLOT / 256; j++) {
    if (256 * j < n) {
      // first token of the lane to my right (lane 63: of the next row), and its flags; slots behind the tile's end hold zeros
      uint32_t nx = from_lane_right(r[j].x);
      uint32_t nx0 = TOK_WS;
      if (j + 1 < SLOT / 256) nx0 = from_lane0(r[j + 1 < SLOT / 256 ? j + 1 : j].x);
      if (lane == 63) nx = nx0;
      int p = 256 * j + 4 * lane;
      YTTM_OPAQUE_V(p);  // (recomputed per tile: hoisted out of the tile loop, the LDS addresses derived from it are spilled to scratch)
      PAIR_SITE(r[j].x, r[j].y, p, 4 * j)
      PAIR_SITE(r[j].y, r[j].z, p + 1, 4 * j + 1)
      PAIR_SITE(r[j].z, r[j].w, p + 2, 4 * j + 2)
      PAIR_SITE(r[j].w, nx, p + 3, 4 * j + 3)
    }
  }
#undef PAIR_SITE
  const bool found = my_bits != 0;
  if (found) {
#pragma unroll
    for (int j = 0; j < SLOT / 256; j++) {
      const uint32_t nib = (uint32_t)(my_bits >> (4 * j)) & 15u;
      if (nib) atomicOr(&sm32[(256 * j + 4 * lane) >> 5], nib << ((4 * lane) & 31));
    }
    const int s_first = sizeof(bits_t) == 8 ? __ffsll((long long)my_bits) - 1 : __ffs((int)my_bits) - 1;
    const int s_last = sizeof(bits_t) == 8 ? 63 - __clzll((long long)my_bits) : 31 - __clz((int)my_bits);
    const uint32_t p_first = (uint32_t)(256 * (s_first >> 2) + 4 * lane + (s_first & 3));
    my_cnt = sizeof(bits_t) == 8 ? (uint32_t)__popcll((unsigned long long)my_bits) : (uint32_t)__popc((unsigned int)my_bits);
    my_site = ((uint32_t)(256 * (s_last >> 2) + 4 * lane + (s_last & 3)) << 16) | my_ri;
    // the tile's site list (any order).  A lane that found more than one site only counts them and asks for the list in
    // position order (bit 16 of the counter), which phase 2 then builds from the site masks.
    const unsigned int idx = atomicAdd(&W.sctl[0], my_cnt | (my_cnt > 1 ? 0x10000u : 0u)) & 0xffffu;
    if (my_cnt == 1 && idx < 64u) W.sitepos[idx] = (uint16_t)p_first;
    atomicMin(&W.sctl[1], p_first);
  }
  wave_sync();
  return (__ballot(found) ? 1 : 0) | (__ballot(selfp) ? 3 : 0);
}

// K4, a tile with exactly ONE merge site (nine dirty tiles in ten late in training), handled where it is -- in registers:
// the four tokens around the site and the word's frequency are fetched as wave-uniform scalars, lanes 0..3 emit the (at
// most) four count deltas together, and the tokens behind the site move up by one with a lane-to-lane shift before the
// rows are written back with the same 16-byte stores they were loaded with.  Nothing is staged.  Returns false (nothing
// done) if a run of equal tokens touches the site: those cases need the run's length and go the general way.
// site = position << 16 | rule index (uniform).
template <int SLOT>
__device__ inline bool single_site_tile(const uint4 (&r)[SLOT / 256], AggLds &A, WaveLds<SLOT> &W, const TileSet &ts, const PairTable &pt, const DeltaBuf &db,
                                        uint32_t t, int n, uint32_t word0, uint32_t site, uint32_t z_base) {
  static_assert(SLOT >= 512, "rows 0 and 1");
  const int lane = lane_id();
  const int p = (int)(site >> 16);
  const uint32_t z = z_base + (site & 0xffffu);
  // token at tile position q (uniform): component q & 3 of row q >> 8 in lane (q >> 2) & 63
#define TOK_AT(OUT, Q)                                                                     \
  {                                                                                        \
    const int q_ = (Q), c_ = q_ & 3;                                                       \
    /* (masks, not selects: a select chain over the components becomes an indexed access and puts r[] into scratch) */ \
    const uint32_t m0_ = 0u - (uint32_t)(c_ == 0), m1_ = 0u - (uint32_t)(c_ == 1), m2_ = 0u - (uint32_t)(c_ == 2), \
                   m3_ = 0u - (uint32_t)(c_ == 3), hi_ = 0u - (uint32_t)(q_ >> 8); /* (class A: two rows) */            \
    const uint32_t a0_ = (r[0].x & m0_) | (r[0].y & m1_) | (r[0].z & m2_) | (r[0].w & m3_);                              \
    const uint32_t a1_ = (r[1].x & m0_) | (r[1].y & m1_) | (r[1].z & m2_) | (r[1].w & m3_);                              \
    const uint32_t v_ = (a1_ & hi_) | (a0_ & ~hi_);                                                                       \
    OUT = (uint32_t)__builtin_amdgcn_readlane((int)v_, (q_ >> 2) & 63);                    \
  }
  uint32_t t0, t1;
  TOK_AT(t0, p)
  TOK_AT(t1, p + 1)
  const uint32_t x = t0 & L_ID, y = t1 & L_ID;
  const bool hasL = p > 0 && !(t0 & TOK_WS);
  uint32_t L = 0, R = 0;
  if (hasL) {
    TOK_AT(L, p - 1)
    L &= L_ID;
  }
  bool hasR = p + 2 < n;
  if (hasR) {
    uint32_t t2;
    TOK_AT(t2, p + 2)
    hasR = !(t2 & TOK_WS);
    R = t2 & L_ID;
  }
#undef TOK_AT
  if ((hasL && L == x) || (hasR && R == y)) return false;
  // the word that contains p: number of word starts at positions <= p (lane l holds positions 256 j + 4 l + {0..3})
  const int jp = p >> 8, lp = (p >> 2) & 63, cp = p & 3;
  uint32_t widx = 0;
#pragma unroll
  for (int j = 0; j < SLOT / 256; j++) {
    if (j <= jp) {
      const unsigned long long m0 = __ballot(r[j].x >> 31), m1 = __ballot(r[j].y >> 31), m2 = __ballot(r[j].z >> 31), m3 = __ballot(r[j].w >> 31);
      if (j < jp) {
        widx += (uint32_t)(__popcll(m0) + __popcll(m1) + __popcll(m2) + __popcll(m3));
      } else {
        const unsigned long long lt = (1ull << lp) - 1ull;
        widx += (uint32_t)(__popcll(m0 & lt) + __popcll(m1 & lt) + __popcll(m2 & lt) + __popcll(m3 & lt));
        widx += (uint32_t)((m0 >> lp) & 1ull);
        if (cp >= 1) widx += (uint32_t)((m1 >> lp) & 1ull);
        if (cp >= 2) widx += (uint32_t)((m2 >> lp) & 1ull);
        if (cp >= 3) widx += (uint32_t)((m3 >> lp) & 1ull);
      }
    }
  }
  const long long f = (long long)ts.wcnt[word0 + widx - 1u];
  // ---- tokens behind the site move up by one; the site becomes z
  const uint32_t zw = z | (t0 & TOK_WS);
  uint4 *dst = reinterpret_cast<uint4 *>(ts.tok + (size_t)t * SLOT);
#pragma unroll
  for (int j = 0; j < SLOT / 256; j++) {
    if (j >= jp && 256 * j < n) {
      uint32_t nxt = from_lane_right(r[j].x);  // first token of the lane to my right (lane 63: of the next row; zeros behind the end)
      uint32_t nxt0 = 0;
      if (j + 1 < SLOT / 256) nxt0 = from_lane0(r[j + 1 < SLOT / 256 ? j + 1 : j].x);
      if (lane == 63) nxt = nxt0;
      const int q0 = 256 * j + 4 * lane;
      const uint4 o = r[j];
      uint4 v;
      v.x = q0 < p ? o.x : (q0 == p ? zw : o.y);
      v.y = q0 + 1 < p ? o.y : (q0 + 1 == p ? zw : o.z);
      v.z = q0 + 2 < p ? o.z : (q0 + 2 == p ? zw : o.w);
      v.w = q0 + 3 < p ? o.w : (q0 + 3 == p ? zw : nxt);
      if (q0 + 3 >= p && q0 < n) dst[lane + 64 * j] = v;
    }
  }
  if (lane == 0) ts.tile_len[t] = (uint32_t)(n - 1);
  // ---- count deltas: (L,x) -> (L,z) and (y,R) -> (z,R); the merged pair itself is zeroed after the round
  const bool v = lane < 2 ? hasL : (lane < 4 && hasR);
  if (__ballot(v)) {
    if (v) {
      const unsigned long long key = lane == 0 ? pair_key(L, x) : lane == 1 ? pair_key(L, z) : lane == 2 ? pair_key(y, R) : pair_key(z, R);
      emit<SLOT>(A, W, pt, db, key, (lane & 1) ? f : -f);
    }
  }
  return true;
}

// registers -> LDS, sentinels (wave-local)
template <int SLOT>
__device__ inline void tile_stage(WaveLds<SLOT> &W, const uint4 (&r)[SLOT / 256], int n) {
  const int lane = lane_id();
#pragma unroll
  for (int j = 0; j < SLOT / 256; j++)
    if (256 * j < n) reinterpret_cast<uint4 *>(W.tk)[lane + 64 * j] = r[j];
  wave_sync();
  if (lane == 0) {
    W.tk[n] = TOK_WS;  // sentinel: "next token starts a word" => no adjacency past the end
    W.tk[n + 1] = TOK_WS;
    W.tk[n + 2] = TOK_WS;
  }
  wave_sync();
}

// frequency of the word that contains tile position p
// Index (within the tile) of the word that contains tile position p.
template <int SLOT>
__device__ inline uint32_t tile_word_index(const WaveLds<SLOT> &W, int p) {
  const int c = p >> 6;
  const unsigned long long le = (2ull << (p & 63)) - 1ull;  // bits 0..(p&63)
  return W.wsbase[c] + (uint32_t)__popcll(W.wsmask[c] & le) - 1u;
}
// the same for lane l asking about position 64 c + l (c uniform): the chunk's mask and base go through scalar registers
template <int SLOT>
__device__ inline uint32_t chunk_word_index(const WaveLds<SLOT> &W, int c) {
  const unsigned long long wm = uni64(W.wsmask[c]);
  return uni(W.wsbase[c]) + lanes_below(wm) + (lane_bit(wm) ? 1u : 0u) - 1u;
}
// Frequencies of the first 64 N words of a tile travel with it in registers: lane j holds words j, j+64, ... -- ALL words of a
// freshly built tile (a class-A tile then has at most SLOT/2 words, a class-B tile -- words of more than TILE_NOM_A tokens -- at
// most 16); once words have been merged down and a repack has re-dealt them a tile can hold more, and phase 2 reads the
// frequencies of those behind the window from HBM (process_tile).  They are loaded together with
// the tokens, one tile ahead, and read by cross-lane shuffles: the gather from HBM that this replaces cost ~6 us per
// active chunk late in training (random 4-byte reads into a 64 MB array: a TLB miss almost every time).
template <int SLOT>
struct WReg {
  static constexpr int N = SLOT == TILE_SLOT_A ? SLOT / 128 : 1;
  uint32_t v[N];
};
template <int SLOT>
__device__ inline void wreg_load(WReg<SLOT> &w, const uint32_t *__restrict__ wcnt, uint32_t word0) {  // wcnt is padded by 64*N
  const int lane = lane_id();
#pragma unroll
  for (int i = 0; i < WReg<SLOT>::N; i++) w.v[i] = wcnt[word0 + (uint32_t)(lane + 64 * i)];
}
// Frequency of word k of the tile.  MUST be called by all lanes of the wave (ds_bpermute).
template <int SLOT>
__device__ inline long long word_weight_all(const WReg<SLOT> &wreg, uint32_t k) {
  uint32_t f = 0;
#pragma unroll
  for (int i = 0; i < WReg<SLOT>::N; i++) {
    const uint32_t fi = __shfl(wreg.v[i], (int)(k & 63u));
    if ((k >> 6) == (uint32_t)i) f = fi;
  }
  return (long long)f;
}
// Frequency of the word that contains tile position 64 c + lane.  MUST be called by all lanes of the wave (ds_bpermute);
// lanes behind the end of the tile get some word's frequency (never used).
template <int SLOT>
__device__ inline long long tile_weight_all(const WaveLds<SLOT> &W, const WReg<SLOT> &wreg, int c) {
  const uint32_t k = chunk_word_index<SLOT>(W, c);
  uint32_t f = 0;
#pragma unroll
  for (int i = 0; i < WReg<SLOT>::N; i++) {
    const uint32_t fi = __shfl(wreg.v[i], (int)(k & 63u));
    if ((k >> 6) == (uint32_t)i) f = fi;
  }
  return (long long)f;
}

// ------------------------------------------------------------------------------------------------- K3 / K4
// One wavefront per tile.  MERGE=false: K3, weighted bigram histogram of the whole table (every
// adjacency counts the word frequency; a run of L equal tokens counts floor(L/2) for its self pair).
// MERGE=true: K4, apply the batch rules (z ids are consecutive: rule j of the batch creates z_base + j) and emit the
// exact count deltas around the merge sites.
#if defined(YTTM_K4_PROF) && YTTM_K4_PROF >= 2  // PROF=2: phase marks (they cost ~100 cycles each); PROF=1: workgroup timeline only
#define K4_MARK(k) do { const unsigned long long t_ = (unsigned long long)clock64(); S.pt[k] += t_ - S.t_last; S.t_last = t_; } while (0)
#define K4_COUNT(k) (S.pt[k]++)
#else
#define K4_MARK(k) ((void)0)
#define K4_COUNT(k) ((void)0)
#endif
struct TileStats {
#ifdef YTTM_K4_PROF
  unsigned long long pt[16] = {0}, t_last = 0;
#endif
  unsigned long long sites = 0, touched = 0, scanned = 0, touched_tok = 0;
  unsigned long long words_hit = 0, words_hit_tok = 0;  // measurement pass (BatchArgs::instr): words with a merge site, their tokens
};

// K3's dense pair table: how many copies of an n x n table fit in the 1024 counters (a power of two, at most one per lane)
__device__ inline uint32_t dense_copies(uint32_t n) {
  if (n == 0) return 1u;
  uint32_t c = 1024u / (n * n);
  if (c > 64u) c = 64u;
  return c ? 1u << (31 - __clz(c)) : 1u;
}

// everything that happens to one staged tile (K3 count or K4 merge)
template <int SLOT, bool MERGE, bool LDSR, bool WORDS = false>
__device__ inline void process_tile(WaveLds<SLOT> &W, AggLds &A, const TileSet &ts, const PairTable &pt, const DeltaBuf &db,
                                    const RuleTab<LDSR> &rtab, uint32_t self_x, uint32_t self_z,
                                    uint32_t z_base, uint32_t t, int n, uint32_t word0, const WReg<SLOT> &wreg, TileStats &S,
                                    bool self_pass /* MERGE: the tile may hold sites of the x x rule */, bool instr = false,
                                    const DeltaOut *dout = nullptr /* WORDS: where the count updates go (rec_emit_batch) */) {
  const int lane = lane_id();
  unsigned long long &my_sites = S.sites, &st_touched = S.touched, &st_scanned = S.scanned, &st_touched_tok = S.touched_tok;
    const int nchunks = (n + 63) >> 6;
    st_scanned += (unsigned long long)n;

    // ---- phase 1a.  K3: word-start masks per 64-token chunk.  K4: word-start masks and the sites of the x != y rules came
    // from the registers (stage_ws_masks, reg_find_sites); sites of an x x rule are found here: left-to-right greedy inside a
    // run of x's = the positions at an even offset from the run's start.
    bool any = false;
    int nsites = 0, first_site_chunk = nchunks;  // (MERGE)
    bool list_in_order = false;                  // (MERGE) phase 2 builds its site lists from the site masks
    if (!MERGE) {
      uint32_t wbase = 0;
      for (int c = 0; c < nchunks; c++) {
        const int p = c * 64 + lane;
        const bool ws = p < n && (W.tk[p] & TOK_WS);
        const unsigned long long m = __ballot(ws);
        if (lane == 0) {
          W.wsmask[c] = m;
          W.wsbase[c] = wbase;
        }
        wbase += (uint32_t)__popcll(m);
      }
    } else {
      if (self_pass) {
        for (int c = 0; c < nchunks; c++) {
          const int p = c * 64 + lane;
          bool self_site = false;
          if (p < n) {
            const uint32_t t0 = W.tk[p], t1 = W.tk[p + 1];
            if (!(t1 & TOK_WS) && (t0 & L_ID) == self_x && (t1 & L_ID) == self_x) {
              int q = p;
              while (q > 0 && !(W.tk[q] & TOK_WS) && (W.tk[q - 1] & L_ID) == self_x) q--;
              if (((p - q) & 1) == 0) {
                self_site = true;
                W.ridx[p] = (uint16_t)(self_z - z_base);
                site_listed<SLOT>(W, p);
              }
            }
          }
          const unsigned long long ssm = __ballot(self_site);
          if (ssm != 0ull && lane == 0) W.sitemask[c] |= ssm;
        }
        wave_sync();
      }
      const uint32_t sc = uni(W.sctl[0]);
      nsites = (int)(sc & 0xffffu);
      list_in_order = nsites > 64 || (sc >> 16) != 0u;
      any = nsites != 0;
      if (any) first_site_chunk = (int)(uni(W.sctl[1]) >> 6);
    }
    wave_sync();
    if (MERGE) K4_MARK(3);

    if (!MERGE) {
      for (int c = 0; c < nchunks; c++) {
        const int p = c * 64 + lane;
        const long long f = tile_weight_all<SLOT>(W, wreg, c);
        if (p >= n) continue;
        const uint32_t t0 = W.tk[p], t1 = W.tk[p + 1];
        if (t1 & TOK_WS) continue;
        const uint32_t a = t0 & TOK_MASK, b = t1 & TOK_MASK;
        // K3 on a small alphabet (self_z = smallest id, z_base = number of ids, <= 32): the pair IS the index of a dense table of
        // counts in LDS -- one ds_add_u64 per adjacency instead of a hash probe (compare, CAS, add)
        // -- and a lane adds into its own copy of the table when the alphabet leaves room for copies (dense_copies), so the lanes
        // of one instruction (on 'abcd ': 64 lanes, 25 pairs) do not queue up on one address
        unsigned long long *dense = reinterpret_cast<unsigned long long *>(A.flagbits) + (size_t)(lane & (dense_copies(z_base) - 1)) * (z_base * z_base);
        const bool use_dense = z_base != 0;
        if (a != b) {
          if (use_dense) atomicAdd(&dense[(a - self_z) * z_base + (b - self_z)], (unsigned long long)f);
          else emit<SLOT>(A, W, pt, db, pair_key(a, b), f);
        } else {
          const bool run_start = (t0 & TOK_WS) || p == 0 || (W.tk[p - 1] & TOK_MASK) != a;
          if (run_start) {
            int q = p + 1;
            while (!(W.tk[q + 1] & TOK_WS) && (W.tk[q + 1] & TOK_MASK) == a) q++;
            const long long len = q - p + 1;
            if (use_dense) atomicAdd(&dense[(a - self_z) * z_base + (a - self_z)], (unsigned long long)((len / 2) * f));
            else emit<SLOT>(A, W, pt, db, pair_key(a, a), (len / 2) * f);
          }
        }
      }
    } else {
      K4_MARK(4);
      if (any) {
#define SITE(q) ((q) >= 0 && (((W.sitemask[(q) >> 6] >> ((q)&63)) & 1ull) != 0))
#define NEWTOK(q) (z_base + (uint32_t)W.ridx[(q)])
        // ---- phase 2: count deltas, ONE LANE PER MERGE SITE -------------------------------------------------------------
        // The sites of the tile, 64 at a time: lane i takes site number base + i, reads the few tokens around it from LDS and
        // works out every delta the merge causes -- what worker_doing_merge does per list node (bpe.cpp:491-812): the left
        // neighbour's (L,x) -> (L,z), the right neighbour's (y,R) -> (z,R), runs of equal tokens losing a member, and runs of
        // the new token.  (A pass over the tile chunk by chunk with one token per lane did the same with ~4 lanes of 64
        // busy: sites are sparse even in the first rounds.)  The merged pair itself is not retracted site by site: every
        // occurrence goes, its count is zeroed after the round.
        if (lane == 0) my_sites += (unsigned long long)nsites;
        for (int base = 0; base < nsites; base += 64) {
          int before = 0;  // sites in the chunks already looked at
          // (up to 64 sites: the list made while they were found, in any order; more: the sites in position order, 64 per pass)
          for (int c = first_site_chunk; list_in_order && c < nchunks && before < base + 64; c++) {
            const unsigned long long smc = uni64(W.sitemask[c]);
            const int cnt = __popcll(smc);
            if (cnt != 0 && before + cnt > base && lane_bit(smc)) {
              const int rk = before + (int)lanes_below(smc) - base;
              if (rk >= 0 && rk < 64) W.sitepos[rk] = (uint16_t)(c * 64 + lane);
            }
            before += cnt;
          }
          wave_sync();
          const bool have = base + lane < nsites;
          const int p = have ? (int)W.sitepos[lane] : 0;
          const uint32_t widx = tile_word_index_rl<SLOT>(W, p);
          long long f = word_weight_all<SLOT>(wreg, widx);  // (all lanes: shuffles)
          // The registers hold the frequencies of the tile's first 64 N words -- every word of a fresh tile (a word has at least two
          // tokens then).  Merged down to one or two tokens and re-dealt by a repack, more words than that can share a tile: theirs come
          // from HBM.  (Found by tools/soak_sim.py: 276 words in one tile, the sites of words 256.. applied with frequency 0.)
          if (have && widx >= 64u * (uint32_t)WReg<SLOT>::N) f = (long long)ts.wcnt[word0 + widx];
          bool v0 = false, v1 = false, v2 = false, v3 = false, v4 = false;
          unsigned long long k0 = 0, k1 = 0, k2 = 0, k3 = 0, k4 = 0;
          long long d0 = 0, d2 = 0, d4 = 0;
          if (have) {
            const uint32_t t0 = W.tk[p], t1 = W.tk[p + 1], t2 = W.tk[p + 2];
            const uint32_t x = t0 & L_ID, y = t1 & L_ID, z = NEWTOK(p);
            const bool hasL = p > 0 && !(t0 & TOK_WS);       // a token of the same word before the site
            const bool hasR = p + 2 < n && !(t2 & TOK_WS);   // ... and behind it
            const bool s_m2 = hasL && p >= 2 && SITE(p - 2);  // that token is the y of another site
            const bool s_p2 = hasR && SITE(p + 2);            // ... the x of another site
            if (hasL) {
              const uint32_t L = W.tk[p - 1] & L_ID;
              if (!s_m2) {  // L stays: (L,x) -> (L,z)
                if (L != x) { v0 = true; k0 = pair_key(L, x); d0 = -f; }
                v1 = true; k1 = pair_key(L, z);
              }
              if (L == x && x != self_x) {  // x != y rule whose x is the last token of a run of x's: the run shrinks by one
                int rr = p;
                while (rr > 0 && !(W.tk[rr] & TOK_WS) && (W.tk[rr - 1] & L_ID) == x) rr--;
                if (((p - rr + 1) & 1) == 0) { v0 = true; k0 = pair_key(x, x); d0 = -f; }
              }
            }
            // run of new z tokens (x y x y ... or the halves of an x-run): counted floor(Lz/2) by its first site
            if (s_p2 && NEWTOK(p + 2) == z && !(s_m2 && NEWTOK(p - 2) == z)) {
              int q = p + 2, lz = 2;
              while (!(W.tk[q + 2] & TOK_WS) && q + 2 < n && SITE(q + 2) && NEWTOK(q + 2) == z) { q += 2; lz++; }
              v4 = true; k4 = pair_key(z, z); d4 = (long long)(lz / 2) * f;
            }
            if (hasR) {
              const uint32_t R = t2 & L_ID;
              const uint32_t B = s_p2 ? NEWTOK(p + 2) : R;  // new adjacency (z, right neighbour)
              if (B != z) { v3 = true; k3 = pair_key(z, B); }
              // the old adjacency (y, R) disappears
              if (y != R) {
                v2 = true; k2 = pair_key(y, R); d2 = -f;
              } else if (y != self_x) {  // x != y rule whose y is the first token of a run of y's: the run shrinks by one
                int q = p + 1;
                while (!(W.tk[q + 1] & TOK_WS) && (W.tk[q + 1] & L_ID) == y) q++;
                if (((q - p) & 1) == 0) { v2 = true; k2 = pair_key(y, y); d2 = -f; }
              }
            }
          }
          K4_MARK(13);  // (PROF=2: phase 2 up to here = the sites' context and deltas; from here to mark 5 = the emits)
          if (WORDS) {
            const unsigned long long ks[5] = {k0, k1, k2, k3, k4};
            const long long ds[5] = {d0, f, d2, f, d4};
            const bool ms[5] = {v0 && !agg_try(A, k0, d0), v1 && !agg_try(A, k1, f), v2 && !agg_try(A, k2, d2), v3 && !agg_try(A, k3, f),
                                v4 && !agg_try(A, k4, d4)};
            if (__ballot(ms[0] || ms[1] || ms[2] || ms[3] || ms[4])) {
#ifdef YTTM_K4_PROF
              const unsigned long long t0_ = (unsigned long long)clock64();
#endif
              rec_emit_batch<5>(*dout, pt, db, ks, ds, ms, &A.new_keys);
#ifdef YTTM_K4_PROF
              if (ms[0] || ms[1] || ms[2] || ms[3] || ms[4]) {
                atomicAdd(&A.miss_n, (unsigned long long)((int)ms[0] + (int)ms[1] + (int)ms[2] + (int)ms[3] + (int)ms[4]));
                atomicAdd(&A.miss_cyc, (unsigned long long)clock64() - t0_);
              }
#endif
            }
          } else {
          if (__ballot(v0)) { if (v0) emit<SLOT>(A, W, pt, db, k0, d0); }
          if (__ballot(v1)) { if (v1) emit<SLOT>(A, W, pt, db, k1, f); }
          if (__ballot(v2)) { if (v2) emit<SLOT>(A, W, pt, db, k2, d2); }
          if (__ballot(v3)) { if (v3) emit<SLOT>(A, W, pt, db, k3, f); }
          if (__ballot(v4)) { if (v4) emit<SLOT>(A, W, pt, db, k4, d4); }
          }
          wave_sync();  // (the list is rebuilt by the next pass)
          K4_MARK(5);
        }
        K4_MARK(5);
        if (instr) {
          // measurement pass: the words that hold a site, and how many tokens they have (what the contract's roofline formula
          // calls W_touched and T_touched).  One bit per word of the tile in the site list's space: 1024 bits, enough for every class-A
          // tile (<= SLOT words); a re-dealt class-B tile with more words than that would only blur this statistic.
          uint32_t *bm = reinterpret_cast<uint32_t *>(W.sitepos);
          if (lane < 32) bm[lane] = 0u;
          wave_sync();
          for (int c = first_site_chunk; c < nchunks; c++) {
            const int p = c * 64 + lane;
            if (p < n && SITE(p)) {
              const uint32_t w = tile_word_index_rl<SLOT>(W, p) & 1023u;
              atomicOr(&bm[w >> 5], 1u << (w & 31u));
            }
          }
          wave_sync();
          for (int c = 0; c < nchunks; c++) {
            const int p = c * 64 + lane;
            bool hit = false, start = false;
            if (p < n) {
              const uint32_t w = tile_word_index_rl<SLOT>(W, p) & 1023u;
              hit = (bm[w >> 5] >> (w & 31u)) & 1u;
              start = hit && (W.tk[p] & TOK_WS);
            }
            S.words_hit_tok += (unsigned long long)__popcll(__ballot(hit));
            S.words_hit += (unsigned long long)__popcll(__ballot(start));
          }
          wave_sync();
        }
        // (word mode, k_words: the gathered words go back to their own slots -- words_out() -- not to a tile)
        if (WORDS) return;
        // ---- phase 3: compact in place (all reads come from LDS, so overwriting the slot in HBM is safe) ----------------
        // survivors of a chunk = its positions that are not the y of a site; tokens before the first site neither move nor change
        uint32_t *dst = ts.tok + (size_t)t * SLOT;
        uint32_t abase = (uint32_t)first_site_chunk * 64u;
        unsigned long long sm_prev = 0ull;
        for (int c = first_site_chunk; c < nchunks; c++) {
          const int p = c * 64 + lane;
          const unsigned long long smc = uni64(W.sitemask[c]);
          const int left = n - c * 64;
          const unsigned long long am = (left >= 64 ? ~0ull : (1ull << left) - 1ull) & ~((smc << 1) | (sm_prev >> 63));
          if (lane_bit(am)) {
            const uint32_t np = abase + lanes_below(am);
            const uint32_t t0 = W.tk[p];
            dst[np] = lane_bit(smc) ? (NEWTOK(p) | (t0 & TOK_WS)) : (t0 & ~(L_ISX | L_ISY));
          }
          abase += (uint32_t)__popcll(am);
          sm_prev = smc;
        }
        // invariant: slots behind the live prefix hold zeros (id 0 is a special token: never flagged, never part of a rule),
        // so the register-level dismissal needs no bounds checks
        for (int p = (int)abase + lane; p < n; p += 64) dst[p] = 0;
        if (lane == 0) ts.tile_len[t] = abase;
        st_touched++;
        st_touched_tok += (unsigned long long)n;
        K4_MARK(6);
#undef SITE
#undef NEWTOK
      }
    }
}


// class-A apply kernels (k_tiles<MERGE>, k_words): waves per workgroup x workgroups per CU (8 x 3 = 6 waves per SIMD: <= 80 VGPRs, <= 48.5 KB LDS per workgroup)
constexpr int APPLY_WPB = 8, APPLY_BPC = 3;
constexpr int APPLY_WPB_B = 4;  // class-B tiles (4096-token slots): waves per workgroup of the merge-apply launch
// workgroups of a tile launch: blocks_per_cu per CU, never more than there are tiles to hand out
static inline unsigned int tile_grid(unsigned int n_tiles, unsigned int wpb, unsigned int blocks_per_cu) {
  unsigned int need = (n_tiles + wpb - 1) / wpb;
  unsigned int g = 256u * blocks_per_cu;
  if (g > need) g = need;
  return g ? g : 1u;
}

}  // namespace yttm
