// k_merge.hip -- K3 (pair-frequency count) and K4 (batched merge-apply + count deltas) over LDS-staged token tiles,
// plus the HBM pair table kernels (candidate filter, rehash, query, remote-delta apply) for gfx950.
//
// Replaces, in the reference trainer:
//   K3  build_linked_list (pair2cnt part)   bpe.cpp:436-478, summed over threads :1076-1088
//   K4  worker_doing_merge                  bpe.cpp:491-812  (list splice, +-pair2cnt, run handling :625-691/:719-785,
//                                           new-pair reports :789-804)
//   pair table + candidate filter           pair2cnt_g :891, check_cnt :1099-1108, PriorityQueue :271-314 (the final
//                                           ordered pick stays on the host: host_trainer.cpp)
// Design: no linked lists and no per-pair position lists.  Each round the host picks a batch of mutually
// non-intersecting rules; one streaming pass over the token tiles applies all of them at once.  A
// WAVEFRONT owns a tile (no workgroup barriers in the loop) and prefetches the next tile into registers while it works on
// the current one.  K4 per tile: (1) in registers, one flag lookup per token (LDS bitmap: is the id the x / the y of a
// batch rule) -- a tile without a flagged adjacency is dismissed here, at HBM speed; flagged adjacencies are looked up
// in the LDS rule hash: merge sites.  (2) A tile with a single site is rewritten in registers (single_site_tile).
// (3) Otherwise the tile is staged into LDS, x==y sites are resolved by parity from the run start, ONE LANE PER SITE
// works out the exact count deltas around it (summed in an LDS hash shared by the workgroup, then 64-bit atomics into
// the HBM pair table), and the tile is compacted in place from its first site on.
// Wave-uniform values go through scalar registers (uni / lane_bit / lanes_below, yttm_device.h).
// HBM-bound integer work: no MFMA.
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


template <int SLOT, int WPB, bool MERGE, bool LDSR, bool DIRECT = false>
__global__ __launch_bounds__(WPB * 64, MERGE ? (WPB == 4 ? 5 : WPB == 8 ? 6 : WPB) : WPB) void k_tiles(TileSet ts, PairTable pt, DeltaBuf db, const RuleSlot *__restrict__ rules,
                                                    unsigned int rule_mask, const uint8_t *__restrict__ tokflag,
                                                    const uint32_t *__restrict__ flagbits, uint32_t self_x, uint32_t self_z, uint32_t z_base,
                                                    const uint32_t *__restrict__ worklist, const unsigned int *__restrict__ work_n,
                                                    unsigned long long *__restrict__ stats /* [0]=sites [1]=tiles touched [2]=tokens scanned [3]=tokens in touched tiles */,
                                                    BatchArgs ba, ScanArgs sa) {
  __shared__ WaveLds<SLOT> WL[WPB];
  __shared__ AggLds A;
  __shared__ unsigned long long rkeys[LDSR ? APPLY_LDS_RULES : 1];
  __shared__ uint16_t rridx[LDSR ? APPLY_LDS_RULES : 1];
#ifdef YTTM_K4_PROF
  const unsigned long long wall0_ = wall_clock64();
#endif
  if (MERGE && ba.mark && blockIdx.x == 0 && threadIdx.x == 0)
    __hip_atomic_store(&stats[STAT_T0], (unsigned long long)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const bool from_args = MERGE && LDSR && ba.k != 0;  // tables built from the kernel argument, nothing read from HBM
  agg_init<WPB * 64>(A, (MERGE && !from_args) ? flagbits : nullptr);
  if (from_args) {
    // (A.flagbits: the batch's pair filter, or -- DIRECT -- the pair -> rule table: direct_v * direct_v bytes, 0xff = no rule)
    for (int s = (int)threadIdx.x; s < (int)(FLAG_LDS_IDS / 16); s += WPB * 64) A.flagbits[s] = DIRECT ? 0xffffffffu : 0u;
    for (unsigned int i = threadIdx.x; i <= rule_mask; i += WPB * 64) rkeys[i] = PT_EMPTY;
    __syncthreads();
    for (unsigned int j = threadIdx.x; j < ba.k; j += WPB * 64) {  // (class B: one wave per workgroup)
      const uint32_t x = ba.xy[2 * j], y = ba.xy[2 * j + 1];
      if (x != y) {
        if (DIRECT) {
          reinterpret_cast<uint8_t *>(A.flagbits)[x * ba.direct_v + y] = (uint8_t)j;
        } else {
          const uint32_t bh = pm_hash(x, y);  // (A.flagbits holds the batch's pair filter)
          atomicOr(&A.flagbits[pm_word(bh)], pm_bits(bh));
        }
        const unsigned long long key = pair_key(x, y);
        unsigned int h = pair_hash32(key) & rule_mask;
        for (;;) {
          if (atomicCAS(&rkeys[h], PT_EMPTY, key) == PT_EMPTY) {
            rridx[h] = (uint16_t)j;
            break;
          }
          h = (h + 1) & rule_mask;
        }
      }
    }
  } else if (LDSR) {
    for (unsigned int i = threadIdx.x; i <= rule_mask; i += WPB * 64) {
      rkeys[i] = rules[i].key;
      rridx[i] = (uint16_t)(rules[i].z - z_base);
    }
  }
  if (!MERGE && z_base) {  // K3, small alphabet: the dense pair table (see process_tile) lives where K4 keeps its flag bitmap
    static_assert(FLAG_LDS_IDS / 16 * sizeof(uint32_t) >= 32 * 32 * sizeof(unsigned long long), "32 x 32 counts");
    unsigned long long *dense = reinterpret_cast<unsigned long long *>(A.flagbits);
    for (unsigned int i = threadIdx.x; i < z_base * z_base * dense_copies(z_base); i += WPB * 64) dense[i] = 0;
  }
  const RuleTab<LDSR> rtab{rkeys, rridx, rules, rule_mask, z_base};
  __syncthreads();
  const int wave = uni((int)(threadIdx.x >> 6)), lane = lane_id();
  WaveLds<SLOT> &W = WL[wave];
  const uint32_t stride = gridDim.x * WPB;
  // K4 runs over the worklist of dirty tiles written by k_filter; K3 over all tiles
  uint32_t wn[WL_PARTS];  // lengths of the sub-lists
  uint32_t NT = ts.n_tiles;
  // (a worklist gathered from the pair index -- k_gather -- that could not find one of the batch's pairs there is not used:
  // the launch takes every tile instead; work_n[WL_PARTS + 1] is that verdict)
  if (worklist && work_n[WL_PARTS + 1]) worklist = nullptr;
  if (worklist) {
    uint32_t mx = 0;
#pragma unroll
    for (uint32_t s = 0; s < WL_PARTS; s++) {
      wn[s] = work_n[s];
      mx = wn[s] > mx ? wn[s] : mx;
    }
    NT = mx * WL_PARTS;  // item i = entry i / WL_PARTS of sub-list i % WL_PARTS (or nothing, past that list's end)
  }
  const size_t wl_seg = WL_SEG(ts.n_tiles);
  // Tile loop of this wave.  Headers (live length, first word) of the next 64 tiles are loaded with ONE vector load
  // each (lane j holds tile i+j) and handed out by shuffles, so a tile costs no header round trip.  Tokens of tile i+1
  // are fetched right after tile i has been staged into LDS and arrive while tile i is processed.  (All waits the
  // compiler emits are vmcnt(0), so a deeper prefetch buys nothing; measured.)
  uint32_t t = blockIdx.x * WPB + wave;  // work item i (tile index, or index into the worklist)
  int hn = 0;                            // lane j: live length of work item t_batch + j*stride
  uint32_t hw = 0, ht = 0;               // lane j: first word / tile id of that work item
  uint32_t t_batch = t;
  auto load_headers = [&](uint32_t tb) {
    const unsigned long long tj = (unsigned long long)tb + (unsigned long long)lane * stride;
    hn = 0; hw = 0; ht = 0;
    bool have = tj < NT;
    if (have && worklist) {
      const uint32_t part = (uint32_t)tj % WL_PARTS, idx = (uint32_t)(tj / WL_PARTS);
      uint32_t len = 0;
#pragma unroll
      for (uint32_t s = 0; s < WL_PARTS; s++) len = part == s ? wn[s] : len;
      have = idx < len;
      if (have) ht = worklist[part * wl_seg + idx];
    } else if (have) {
      ht = (uint32_t)tj;
    }
    if (have) {
      hn = (int)ts.tile_len[ht];
      hw = ts.tile_word0[ht];  // (independent of the length, so that the two loads share a round trip; an empty tile's is never used)
    }
  };
  uint4 r[SLOT / 256];
  WReg<SLOT> wq{};  // word frequencies of the tile held in r
  TileStats S;
#ifdef YTTM_K4_PROF
  S.t_last = (unsigned long long)clock64();
#endif
  // tile i is in registers: flag/stage it; then (prefetch of tile i+1 by the caller); then process it from LDS
  // tile i is in registers: look for merge sites / stage it; then (prefetch of tile i+1 by the caller); then process it from LDS
  int site_state = 0;  // reg_find_sites() of the tile just looked at
  auto stage_part = [&](int n0, uint32_t tile, uint32_t w0) {
    // K4: a tile without a merge site is dismissed in registers and never touches LDS
    uint32_t my_cnt = 0, my_site = 0;
    site_state = MERGE ? reg_find_sites<SLOT, LDSR, DIRECT>(W, r, n0, A.flagbits, self_x, rtab, my_cnt, my_site, ba.direct_v) : 1;
    if (MERGE) K4_MARK(0);
    bool dirty = site_state != 0;
    if (MERGE && SLOT == TILE_SLOT_A && site_state == 1 && !ba.instr) {  // sites of x != y rules only: is it a single one?
      const unsigned long long fm = __ballot(my_cnt != 0);
      if (__popcll(fm) == 1) {
        const int src = __ffsll((long long)fm) - 1;
        if (__builtin_amdgcn_readlane((int)my_cnt, src) == 1) {
          const uint32_t site = (uint32_t)__builtin_amdgcn_readlane((int)my_site, src);
          if (single_site_tile<SLOT>(r, A, W, ts, pt, db, tile, n0, w0, site, z_base)) {
            dirty = false;
            uint32_t one = 1;
            YTTM_OPAQUE_V(one);  // (a 64-bit constant 1 kept in registers across the tile loop gets spilled)
            if (lane == 0) S.sites += one;
            S.touched += one;
            S.touched_tok += (unsigned long long)n0;
          }
        }
      }
    }
    if (dirty) {
      if (MERGE) stage_ws_masks<SLOT>(W, r, n0);
      tile_stage<SLOT>(W, r, n0);
    }
    if (MERGE) K4_MARK(1);
    return dirty;
  };
  auto process_part = [&](bool dirty, uint32_t tile, int n0, uint32_t w0, const WReg<SLOT> &wcur) {
    if (MERGE) K4_MARK(2);
    if (dirty) {
      K4_COUNT(8);
      process_tile<SLOT, MERGE, LDSR>(W, A, ts, pt, db, rtab, self_x, self_z, z_base, tile, n0, w0, wcur, S, (site_state & 2) != 0, MERGE && ba.instr != 0);
      wave_sync();  // everyone is done with this tile's LDS state before it is restaged
    } else {
      S.scanned += (unsigned long long)n0;
    }
  };
  // (A dynamic hand-out of worklist items through a global counter was tried for short worklists and was slower: the
  // counter's latency lands in every tile because all waits are vmcnt(0).  Static striding it is.)
  // word frequencies travel with the tile's prefetch when (nearly) every tile will need them: K3, and K4 over a worklist
  const bool eager_w = !MERGE || worklist != nullptr;
  int j = 0;
  if (t < NT) {
    load_headers(t_batch);
    tile_fetch<SLOT>(r, ts, uni(from_lane0(ht)), uni(from_lane0(hn)));
    if (eager_w) wreg_load<SLOT>(wq, ts.wcnt, uni(from_lane0(hw)));
  }
  while (t < NT) {
    const int n0 = uni(__shfl(hn, j));  // (uniform, and now the compiler knows: tile loops and branches run on the scalar unit)
    const uint32_t w0 = uni(__shfl(hw, j));
    const uint32_t tile = uni(__shfl(ht, j));
    const bool dirty = stage_part(n0, tile, w0);
    // K4: most tiles are dismissed in registers late in training -- their word frequencies are never needed, so they are
    // loaded only now, for a dirty tile, ahead of the next tile's prefetch (first use is in phase 2)
    if (MERGE && dirty && !eager_w) wreg_load<SLOT>(wq, ts.wcnt, w0);
    // next tile of this wave: header from the batch (reload the batch every 64 tiles), tokens prefetched now
    const uint32_t t_next = t + stride;
    j++;
    if (j == 64 && t_next < NT) {
      j = 0;
      t_batch = t_next;
      load_headers(t_batch);
    }
    const WReg<SLOT> wcur = wq;
    if (t_next < NT) {
      tile_fetch<SLOT>(r, ts, uni(__shfl(ht, j)), uni(__shfl(hn, j)));
      if (eager_w) wreg_load<SLOT>(wq, ts.wcnt, uni(__shfl(hw, j)));
    }
    process_part(dirty, tile, n0, w0, wcur);
    t = t_next;
  }
  if (MERGE) {
    S.sites = wave_sum_u64(S.sites);
    if (lane == 0) {
      if (S.sites) atomicAdd(&A.st[0], S.sites);
      if (S.touched) atomicAdd(&A.st[1], S.touched);
      if (S.scanned && !worklist) atomicAdd(&A.st[2], S.scanned);
      if (S.touched_tok) atomicAdd(&A.st[3], S.touched_tok);
      if (S.words_hit) atomicAdd(&A.st[4], S.words_hit);
      if (S.words_hit_tok) atomicAdd(&A.st[5], S.words_hit_tok);
    }
  }
#ifdef YTTM_K4_PROF
  if (MERGE) K4_MARK(7);   // end of own tile loop
  __syncthreads();
  if (MERGE) K4_MARK(11);  // waiting for the other waves of the workgroup
#endif
  agg_flush<WPB * 64>(A, pt, db);
  if (!MERGE && z_base) {  // (after agg_flush's barrier: every wave is done counting)
    const unsigned long long *dense = reinterpret_cast<const unsigned long long *>(A.flagbits);
    for (unsigned int i = threadIdx.x; i < z_base * z_base; i += WPB * 64) {
      unsigned long long v = 0;
      for (uint32_t c = 0; c < dense_copies(z_base); c++) v += dense[c * z_base * z_base + i];
      if (v) global_emit(pt, db, pair_key(self_z + i / z_base, self_z + i % z_base), (long long)v, &A.new_keys);
    }
  }
  __syncthreads();
#ifdef YTTM_K4_PROF
  if (MERGE) {
    K4_MARK(12);  // flush
    if (lane == 0)
      for (int i = 0; i < 16; i++)
        if (S.pt[i]) atomicAdd(&stats[8 + i], S.pt[i]);
    if (threadIdx.x == 0 && A.miss_n) {  // emits that found no room in the workgroup's LDS hash (they went to the HBM table one by one)
      atomicAdd(&stats[8 + 14], A.miss_n);
      atomicAdd(&stats[8 + 15], A.miss_cyc);
    }
  }
#endif
#ifdef YTTM_K4_PROF
  if (MERGE && threadIdx.x == 0) {  // per-workgroup timeline (100 MHz wall clock) for YTTM_TRACE_ROUNDS
    unsigned long long *row = stats + BLK_BASE + 8 * (blockIdx.x % BLK_ROWS);
    row[5] = wall0_;
    row[6] = wall_clock64();
    row[7] = A.st[1];
  }
#endif
  if (threadIdx.x == 0) {
    if (MERGE) {
      blk_add(stats, 4, A.new_keys);
      for (int i = 0; i < 4; i++) blk_add(stats, i, A.st[i]);
      if (ba.instr) {  // (measurement pass: plain global atomics)
        if (A.st[4]) atomicAdd(&stats[4], A.st[4]);
        if (A.st[5]) atomicAdd(&stats[5], A.st[5]);
      }
    } else if (A.new_keys) {
      atomicAdd(pt.n_keys, A.new_keys);  // K3: one launch
    }
  }
  if (MERGE && sa.on) {  // the round's candidate scan, by the last workgroup to get here (scan_top)
    // Everything this workgroup leaves for the tail went out as device-scope atomics or write-through stores (pair table, hot
    // list, statistics row), so the ticket only has to wait until those have completed -- a workgroup-scope release: an
    // agent-scope one would also write the XCD's L2 back, once per workgroup (measured: +150 us per round at 768 workgroups).
    // (Publishing needs those operations COMPLETE: every wave drains its memory operations -- s_waitcnt vmcnt(0), written out because
    // the compiler may drop the wait of a fence it thinks has nothing to wait for -- before one lane takes the ticket with an agent-scope
    // atomic.  MI355X_MICROARCH.md lists "sc1 payload -> vmcnt(0) -> flag" among the valid cross-CU hand-offs; the reader side is the
    // agent-scope acquire below plus agent-scope loads.  tools/dbg/fuse_check.py diffs the candidate traces of fused and unfused runs.)
    __shared__ unsigned int is_last;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0) is_last = atomicAdd(sa.done_ctr, 1u) == gridDim.x - 1;
    __syncthreads();
    if (is_last) {
      __threadfence();  // (acquire: nothing stale in this CU's caches)
      static_assert(sizeof(WL) >= (CAND_BINS + 160) * sizeof(unsigned int), "tile buffers double as the tail's scratch");
      const RuleProbe zprobe{LDSR ? rkeys : nullptr, LDSR ? nullptr : rules, rule_mask};
      scan_top<WPB * 64>(pt, sa, stats, zprobe, self_x != 0xffffffffu ? pair_key(self_x, self_x) : PT_EMPTY, reinterpret_cast<unsigned int *>(&WL[0]), nullptr);
    }
  }
}

// ------------------------------------------------------------------------------------------------- pair table kernels
// Candidate filter: appends every pair with (count > tau_cnt) or (count == tau_cnt and max(x,y) <= tau_mx) and
// histograms all live counts (CAND_BINS log-ish bins) so the host can choose the next threshold.
__global__ __launch_bounds__(BLOCK) void k_cand_scan(PairTable pt, unsigned long long tau_cnt, uint32_t tau_mx,
                                                     CandRec *__restrict__ out, unsigned int cap, unsigned int *__restrict__ n_out,
                                                     unsigned long long *__restrict__ hist) {
  __shared__ unsigned int lh[CAND_BINS];
  for (int b = (int)threadIdx.x; b < CAND_BINS; b += BLOCK) lh[b] = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) n_out[1] = *pt.n_keys;  // rides along in the host's per-round read-back
  __syncthreads();
  const unsigned long long n_slots = pt.mask + 1;
  const unsigned long long n_iter = (n_slots + BLOCK - 1) / BLOCK;
  for (unsigned long long it = blockIdx.x; it < n_iter; it += gridDim.x) {
    const unsigned long long i = it * BLOCK + threadIdx.x;
    bool pass = false;
    unsigned long long k = PT_EMPTY, c = 0;
    if (i < n_slots) {
      c = (*pt.cnt_p(i)) & PT_CNT;  // empty and dead slots have count 0: their keys are never read
      if (c > 0) {
        k = (*pt.key_p(i));
        if (hist) atomicAdd(&lh[cand_bin(c)], 1u);
        const uint32_t x = (uint32_t)(k >> 32), y = (uint32_t)k;
        const uint32_t mx = x > y ? x : y;
        pass = c > tau_cnt || (c == tau_cnt && mx <= tau_mx);
      }
    }
    const unsigned long long m = __ballot(pass);
    if (m) {
      unsigned int base = 0;
      if (lane_id() == 0) base = atomicAdd(n_out, (unsigned int)__popcll(m));
      base = from_lane0(base);
      if (pass) {
        unsigned int o = base + (unsigned int)__popcll(m & lanemask_lt());
        if (o < cap) {
          out[o].key = k;
          out[o].cnt = c;
        }
      }
    }
  }
  __syncthreads();
  if (hist) {
    for (int b = (int)threadIdx.x; b < CAND_BINS; b += BLOCK) {
      unsigned int v = lh[b];
      if (v) atomicAdd(&hist[b], (unsigned long long)v);
    }
  }
}

// One workgroup copies header, histogram and the first `fast` candidates of a finished candidate scan into the host's pinned
// mailbox, publishes `round_id` there (system-scope release) and clears the device-side counters for the next call: the host
// polls the mailbox instead of paying a copy + stream synchronisation every round.  Mailbox: [0..15] n_out[0..3], [32] round
// id, [40] tokens streamed so far, [48] tiles touched so far, [56..79] xstat (multi-GPU: ranks whose delta block overflowed,
// largest record count of a rank this round, number of ranks whose hot list overflowed, "a rank's send buffer overflowed"),
// histogram at byte MB_HIST = 128, candidates at byte 8192.
__device__ inline void publish_round(const PairTable &pt, CandRec *__restrict__ out, unsigned int cap, unsigned int *__restrict__ n_out,
                                     unsigned long long *__restrict__ hist, unsigned int *__restrict__ done_ctr, unsigned char *__restrict__ mailbox,
                                     unsigned int fast, uint32_t round_id, unsigned long long *__restrict__ stats,
                                     unsigned long long *__restrict__ xstat) {
  if (threadIdx.x == 0) n_out[1] = __hip_atomic_load(pt.n_keys, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const unsigned int n = __hip_atomic_load(&n_out[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  unsigned int *mb_hdr = reinterpret_cast<unsigned int *>(mailbox);
  unsigned long long *mb_hist = reinterpret_cast<unsigned long long *>(mailbox + MB_HIST);
  uint4 *mb_out = reinterpret_cast<uint4 *>(mailbox + 8192);
  if (threadIdx.x < 4) mb_hdr[threadIdx.x] = __hip_atomic_load(&n_out[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (threadIdx.x == 4)  // tokens streamed by the K4 filters so far: the host derives the tiles' fill from it (repack trigger)
    *reinterpret_cast<unsigned long long *>(mailbox + 40) = __hip_atomic_load(&stats[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (threadIdx.x == 5)  // tiles that held a merge site so far: a dense round skips the filter's exact rule test
    *reinterpret_cast<unsigned long long *>(mailbox + 48) = __hip_atomic_load(&stats[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  xstat_forward(mailbox, xstat, (int)threadIdx.x - 6);
  for (int b = (int)threadIdx.x; b < CAND_BINS; b += BLOCK) {
    mb_hist[b] = __hip_atomic_load(&hist[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    hist[b] = 0;
  }
  unsigned int take = n < cap ? n : cap;
  if (take > fast) take = fast;
  const unsigned long long *src = reinterpret_cast<const unsigned long long *>(out);
  for (unsigned int i = threadIdx.x; i < take; i += BLOCK) {
    uint4 v;
    const unsigned long long a = __hip_atomic_load(&src[2 * i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long b = __hip_atomic_load(&src[2 * i + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v.x = (uint32_t)a; v.y = (uint32_t)(a >> 32); v.z = (uint32_t)b; v.w = (uint32_t)(b >> 32);
    mb_out[i] = v;
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    n_out[0] = n_out[1] = n_out[2] = n_out[3] = 0;
    *done_ctr = 0;
    __hip_atomic_store(&mb_hdr[8], round_id, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// Candidate filter over the hot list: same outputs as k_cand_scan, but only the listed slots are inspected and the
// histogram covers the counts >= hot_tau.  n_out: [0] candidates, [1] n_keys, [2] list length, [3] listed slots that
// are still >= hot_tau.
// The last workgroup to finish copies header, histogram and the first `fast` candidates into the host's pinned mailbox,
// publishes `round_id` there (system-scope release) and clears the device-side counters for the next call: the host
// polls the mailbox instead of paying a copy + stream synchronisation every round.
__global__ __launch_bounds__(BLOCK) void k_hot_scan(PairTable pt, unsigned long long tau_cnt, uint32_t tau_mx, CandRec *__restrict__ out,
                                                    unsigned int cap, unsigned int *__restrict__ n_out, unsigned long long *__restrict__ hist,
                                                    unsigned int *__restrict__ done_ctr, unsigned char *__restrict__ mailbox, unsigned int fast,
                                                    uint32_t round_id, unsigned long long *__restrict__ stats, const RuleSlot *__restrict__ zrules,
                                                    unsigned int zmask, unsigned long long zself, BatchArgs zba,
                                                    unsigned long long *__restrict__ xstat /* multi-GPU: the exchange's report, forwarded */) {
  // zrules != nullptr: the batch that was just applied -- every occurrence of its pairs was merged, so their counts are
  // exactly zero now; they are all on the list (that is where they were picked from), so they are zeroed here instead of
  // by a kernel of their own
  __shared__ unsigned int lh[CAND_BINS];
  __shared__ unsigned int live_blk;
  __shared__ unsigned long long zkeys[FILTER_LDS_KEYS];
  const bool zero_any = zrules != nullptr || zba.k != 0;
  bool zkeys_in_lds = zrules && zmask < FILTER_LDS_KEYS;
  if (zba.k) {  // the batch came as a kernel argument: build the key table here (4 slots per possible rule)
    zmask = 4 * BATCH_ARGS_MAX - 1;
    zkeys_in_lds = true;
    for (unsigned int s = threadIdx.x; s <= zmask; s += BLOCK) zkeys[s] = PT_EMPTY;
    __syncthreads();
    if (threadIdx.x < zba.k && zba.xy[2 * threadIdx.x] != zba.xy[2 * threadIdx.x + 1]) {  // (BATCH_ARGS_MAX <= the block size)
      const unsigned long long key = pair_key(zba.xy[2 * threadIdx.x], zba.xy[2 * threadIdx.x + 1]);
      unsigned int h = pair_hash32(key) & zmask;
      while (atomicCAS(&zkeys[h], PT_EMPTY, key) != PT_EMPTY) h = (h + 1) & zmask;
    }
  } else if (zkeys_in_lds) {
    for (unsigned int s = threadIdx.x; s <= zmask; s += BLOCK) zkeys[s] = zrules[s].key;
  }
  const RuleProbe zprobe{zkeys_in_lds ? zkeys : nullptr, zrules, zmask};
  for (int b = (int)threadIdx.x; b < CAND_BINS; b += BLOCK) lh[b] = 0;
  if (threadIdx.x == 0) live_blk = 0;
  const unsigned int hn_raw = *pt.hot_n;
  const unsigned int hn = hn_raw < pt.hot_cap ? hn_raw : pt.hot_cap;
  if (blockIdx.x == 0 && threadIdx.x == 0) n_out[2] = hn_raw;
  {  // every workgroup folds its share of the per-workgroup statistics rows (see fold_blk_stats) into the totals: its rows are
     // summed across the first lanes of wave 0 first, so a total receives one atomic per workgroup
    const int per = (BLK_ROWS + (int)gridDim.x - 1) / (int)gridDim.x;  // <= 64
    if (threadIdx.x < 64) {
      unsigned long long v[5] = {0, 0, 0, 0, 0};
      const int b = (int)blockIdx.x * per + (int)threadIdx.x;
      if ((int)threadIdx.x < per && b < BLK_ROWS) {
        unsigned long long *row = stats + BLK_BASE + 8 * b;
        for (int j = 0; j < 5; j++) {
          v[j] = row[j];
          if (v[j]) row[j] = 0;
        }
      }
      for (int j = 0; j < 5; j++) {
        const unsigned long long t = wave_sum_u64(v[j]);
        if (threadIdx.x == 0 && t) {
          if (j < 4) atomicAdd(&stats[j], t);
          else atomicAdd(pt.n_keys, (unsigned int)t);
        }
      }
    }
  }
  __syncthreads();
  unsigned int live = 0;
  for (unsigned int i0 = blockIdx.x * BLOCK; i0 < hn; i0 += gridDim.x * BLOCK) {
    const unsigned int i = i0 + threadIdx.x;
    bool pass = false;
    unsigned long long k = PT_EMPTY, c = 0;
    if (i < hn) {
      const uint32_t sl = pt.hot_slots[i];
      const uint4 rec = *reinterpret_cast<const uint4 *>(pt.key_p(sl));  // key and count in one 16-byte load
      c = (((unsigned long long)rec.w << 32) | rec.z) & PT_CNT;
      k = ((unsigned long long)rec.y << 32) | rec.x;
      if (c && zero_any && (k == zself || zprobe.has((uint32_t)(k >> 32), (uint32_t)k))) {
        *pt.cnt_p(sl) = (((unsigned long long)rec.w << 32) | rec.z) & PT_FLAGS;  // count 0, still on the lists it was on
        c = 0;
      }
      if (c >= pt.hot_tau) {
        live++;
        atomicAdd(&lh[cand_bin(c)], 1u);
        const uint32_t x = (uint32_t)(k >> 32), y = (uint32_t)k;
        const uint32_t mx = x > y ? x : y;
        pass = c > tau_cnt || (c == tau_cnt && mx <= tau_mx);
      }
    }
    const unsigned long long m = __ballot(pass);
    if (m) {
      unsigned int base = 0;
      if (lane_id() == 0) base = atomicAdd(n_out, (unsigned int)__popcll(m));
      base = from_lane0(base);
      if (pass) {
        const unsigned int o = base + (unsigned int)__popcll(m & lanemask_lt());
        if (o < cap) {
          out[o].key = k;
          out[o].cnt = c;
        }
      }
    }
  }
  if (live) atomicAdd(&live_blk, live);
  __syncthreads();
  for (int b = (int)threadIdx.x; b < CAND_BINS; b += BLOCK) {
    const unsigned int v = lh[b];
    if (v) atomicAdd(&hist[b], (unsigned long long)v);
  }
  if (threadIdx.x == 0 && live_blk) atomicAdd(&n_out[3], live_blk);
  // ---- last workgroup: publish
  __shared__ unsigned int is_last;
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) is_last = atomicAdd(done_ctr, 1u) == gridDim.x - 1;
  __syncthreads();
  if (!is_last) return;
  __threadfence();
  publish_round(pt, out, cap, n_out, hist, done_ctr, mailbox, fast, round_id, stats, xstat);
}

// The scan of the top list as a kernel of its own (ONE workgroup): rounds that are more than one launch (multi-GPU exchange,
// class-B / class-C tiles), rescans with another threshold, and the first scan after a refill.  zrules / zba: the batch whose
// pairs are still to be zeroed (as for k_hot_scan).
constexpr int TOP_SCAN_NT = 512;
__global__ __launch_bounds__(TOP_SCAN_NT) void k_top_scan(PairTable pt, ScanArgs sa, unsigned long long *__restrict__ stats, const RuleSlot *__restrict__ zrules,
                                                          unsigned int zmask, unsigned long long zself, BatchArgs zba, unsigned long long *__restrict__ xstat) {
  __shared__ unsigned long long zkeys[FILTER_LDS_KEYS];
  __shared__ unsigned int scratch[CAND_BINS + 160];
  bool zkeys_in_lds = zrules && zmask < FILTER_LDS_KEYS;
  if (zba.k) {
    zmask = 4 * BATCH_ARGS_MAX - 1;
    zkeys_in_lds = true;
    for (unsigned int s = threadIdx.x; s <= zmask; s += TOP_SCAN_NT) zkeys[s] = PT_EMPTY;
    __syncthreads();
    if (threadIdx.x < zba.k && zba.xy[2 * threadIdx.x] != zba.xy[2 * threadIdx.x + 1]) {  // (BATCH_ARGS_MAX <= the block size)
      const unsigned long long key = pair_key(zba.xy[2 * threadIdx.x], zba.xy[2 * threadIdx.x + 1]);
      unsigned int h = pair_hash32(key) & zmask;
      while (atomicCAS(&zkeys[h], PT_EMPTY, key) != PT_EMPTY) h = (h + 1) & zmask;
    }
  } else if (zkeys_in_lds) {
    for (unsigned int s = threadIdx.x; s <= zmask; s += TOP_SCAN_NT) zkeys[s] = zrules[s].key;
  } else if (!zrules) {  // nothing to zero: an empty table
    zmask = 0;
    zkeys_in_lds = true;
    if (threadIdx.x == 0) zkeys[0] = PT_EMPTY;
  }
  __syncthreads();
  const RuleProbe zprobe{zkeys_in_lds ? zkeys : nullptr, zrules, zmask};
  scan_top<TOP_SCAN_NT>(pt, sa, stats, zprobe, zself, scratch, xstat);
}

// Refill of the top list from the hot list: PT_TOP is set exactly on the listed slots with count >= pt.top_tau, and those are
// appended (pt.top_n was reset by the host).
__global__ __launch_bounds__(BLOCK) void k_top_rebuild(PairTable pt) {
  const unsigned int hn_raw = *pt.hot_n;
  const unsigned int hn = hn_raw < pt.hot_cap ? hn_raw : pt.hot_cap;
  for (unsigned int i0 = blockIdx.x * BLOCK; i0 < hn; i0 += gridDim.x * BLOCK) {
    const unsigned int i = i0 + threadIdx.x;
    bool top = false;
    uint32_t sl = 0;
    if (i < hn) {
      sl = pt.hot_slots[i];
      const unsigned long long raw = *pt.cnt_p(sl), c = raw & PT_CNT;
      top = c >= pt.top_tau && c > 0;
      const unsigned long long want = (raw & ~PT_TOP) | (top ? PT_TOP : 0ull);
      if (want != raw) *pt.cnt_p(sl) = want;
    }
    const unsigned long long m = __ballot(top);
    if (m) {
      unsigned int base = 0;
      if (lane_id() == 0) base = atomicAdd(pt.top_n, (unsigned int)__popcll(m));
      base = from_lane0(base);
      if (top) {
        const unsigned int o = base + (unsigned int)__popcll(m & lanemask_lt());
        if (o < pt.top_cap) pt.top_slots[o] = sl;
      }
    }
  }
}

// (Re)build the hot list: every slot with count >= pt.hot_tau, in one streaming pass; PT_HOT is set exactly on those.
__global__ __launch_bounds__(BLOCK) void k_hot_rebuild(PairTable pt) {
  const unsigned long long n_slots = pt.mask + 1;
  const unsigned long long n_iter = (n_slots + BLOCK - 1) / BLOCK;
  for (unsigned long long it = blockIdx.x; it < n_iter; it += gridDim.x) {
    const unsigned long long i = it * BLOCK + threadIdx.x;
    bool hot = false;
    if (i < n_slots) {
      const unsigned long long raw = (*pt.cnt_p(i)), c = raw & PT_CNT;
      hot = c >= pt.hot_tau && c > 0;
      const unsigned long long want = hot ? (c | PT_HOT) : c;
      if (want != raw) (*pt.cnt_p(i)) = want;
    }
    const unsigned long long m = __ballot(hot);
    if (m) {
      unsigned int base = 0;
      if (lane_id() == 0) base = atomicAdd(pt.hot_n, (unsigned int)__popcll(m));
      base = from_lane0(base);
      if (hot) {
        const unsigned int o = base + (unsigned int)__popcll(m & lanemask_lt());
        if (o < pt.hot_cap) pt.hot_slots[o] = (uint32_t)i;
      }
    }
  }
}

__global__ __launch_bounds__(BLOCK) void k_fold_stats(unsigned long long *stats, unsigned int *n_keys) { fold_blk_stats(stats, n_keys); }

__global__ __launch_bounds__(BLOCK) void k_pt_rehash(PairTable src, PairTable dst) {
  const unsigned long long n_slots = src.mask + 1;
  unsigned long long i = (unsigned long long)blockIdx.x * BLOCK + threadIdx.x;
  const unsigned long long stride = (unsigned long long)gridDim.x * BLOCK;
  for (; i < n_slots; i += stride) {
    unsigned long long k = (*src.key_p(i));
    if (k == PT_EMPTY) continue;
    unsigned long long c = (*src.cnt_p(i)) & PT_CNT;
    if (c) pt_add(dst, k, (long long)c);  // dead pairs (count 0) can never come back: drop them
  }
}

__global__ __launch_bounds__(BLOCK) void k_pt_query(PairTable pt, const unsigned long long *__restrict__ keys, unsigned int n,
                                                    unsigned long long *__restrict__ out) {
  unsigned int i = blockIdx.x * BLOCK + threadIdx.x;
  if (i < n) out[i] = pt_get(pt, keys[i]);
}

// the rules of a finished batch: all their occurrences were merged, their counts are exactly zero now
__global__ __launch_bounds__(BLOCK) void k_pt_zero(PairTable pt, const RuleSlot *__restrict__ rules, unsigned int n_slots,
                                                   unsigned long long self_key) {
  unsigned int i = blockIdx.x * BLOCK + threadIdx.x;
  unsigned long long key = PT_EMPTY;
  if (i < n_slots) key = rules[i].key;
  else if (i == n_slots) key = self_key;
  if (key == PT_EMPTY) return;
  unsigned long long j = mix64(key) & pt.mask;
  for (;;) {
    const unsigned long long k = (*pt.key_p(j));
    if (k == PT_EMPTY) return;
    if (k == key) { (*pt.cnt_p(j)) &= PT_FLAGS; return; }  // a listed slot stays listed (once)
    j = (j + 1) & pt.mask;
  }
}

// multi-GPU: fold the count deltas received from the other ranks into the local replica of the global pair table
__global__ __launch_bounds__(BLOCK) void k_pt_apply(PairTable pt, const DeltaRec *__restrict__ recs, unsigned long long n) {
  unsigned long long i = (unsigned long long)blockIdx.x * BLOCK + threadIdx.x;
  const unsigned long long stride = (unsigned long long)gridDim.x * BLOCK;
  for (; i < n; i += stride) pt_add(pt, recs[i].key, recs[i].delta);
}

// multi-GPU, behind a round's exchange and off its critical path (it runs during the host's turn): the delta table's slots of the round
// just exchanged are freed (db.send = that round's block, which itself stays as it is: a repeat of the exchange may want it again), and
// the OTHER block -- the round before's, long settled -- is made ready for the round to come: its records' sums zeroed, its header
// written (no records yet; this rank's statistics for the ranks' common decisions, yttm_device.h: XHDR).
__global__ __launch_bounds__(BLOCK) void k_dt_clean(DeltaBuf db, DeltaRec *__restrict__ other, unsigned long long *__restrict__ stats, uint32_t tiles_a,
                                                    unsigned int *__restrict__ done_ctr) {
  __shared__ unsigned int is_last;
  const unsigned long long n_cur = db.send[0].key < db.send_cap ? db.send[0].key : db.send_cap;
  const unsigned long long n_oth = other[0].key < db.send_cap ? other[0].key : db.send_cap;
  const unsigned long long stride = (unsigned long long)gridDim.x * BLOCK;
  for (unsigned long long j = (unsigned long long)blockIdx.x * BLOCK + threadIdx.x; j < n_cur; j += stride) {
    const uint32_t sl = db.touched[j];
    db.keys[sl].key = PT_EMPTY;
    db.keys[sl].idx = DT_NOIDX;
  }
  for (unsigned long long j = (unsigned long long)blockIdx.x * BLOCK + threadIdx.x; j < n_oth; j += stride) other[XHDR + j].delta = 0;
  // the other block's count goes to zero when every workgroup has read it: the last one to get here.  (No fence: the ticket orders READS of
  // that count -- each workgroup's are long done -- and what is written here only has to be there at the kernel's end.  An agent-scope fence
  // per workgroup writes the XCD's L2 back: this kernel took 30 us with one.)
  __syncthreads();
  if (threadIdx.x == 0) is_last = atomicAdd(done_ctr, 1u) == gridDim.x - 1;
  __syncthreads();
  if (!is_last || threadIdx.x != 0) return;
  other[0].key = 0;
  other[0].delta = (long long)db.send_cap;
  other[1].key = stats ? ld_agent(&stats[0]) : 0ull;                 // merge sites so far (folded by the scans: a round or two old)
  other[1].delta = (long long)(stats ? ld_agent(&stats[2]) : 0ull);  // tokens streamed so far
  other[2].key = tiles_a;
  other[2].delta = 0;
  other[3].key = 0;
  other[3].delta = 0;
  *done_ctr = 0;
}
__global__ __launch_bounds__(BLOCK) void k_dt_init(DtSlot *__restrict__ slots, unsigned long long n) {
  const DtSlot e{PT_EMPTY, DT_NOIDX, 0u};
  for (unsigned long long i = (unsigned long long)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * BLOCK) slots[i] = e;
}

// multi-GPU, per round, phase 1: the ranks' delta blocks as ncclAllGather left them -- block r = { header, records... } of `blk` 16-byte
// units -- and the OTHER ranks' deltas folded into the local replica (pt comes with its list thresholds off: nothing is listed here, see
// k_fold_list).  A rank whose count does not fit its block is skipped as a whole and reported in xstat[0] (bit r); the host then repeats
// the exchange with larger blocks for exactly those ranks (only_mask).  xstat[1] = largest count seen (sizes the next round's blocks),
// xstat[4..7] = sums over the headers.  No host round trip: counts are read on the device.
// the ranks' block headers -> xstat (one thread of the fold): sums, largest count, blocks that did not fit, "a rank lost records"
__device__ inline void fold_headers(const DeltaRec *__restrict__ blocks, unsigned long long blk, int world, unsigned long long only_mask,
                                    unsigned long long *__restrict__ xstat) {
  unsigned long long sites = 0, toks = 0, tiles = 0, xmask = 0, xmax = 0, lost = 0;
  for (int r = 0; r < world; r++) {
    const DeltaRec *b = blocks + (size_t)r * blk;
    const unsigned long long n = b[0].key;  // header: record count of rank r, capacity of its send buffer
    sites += b[1].key;
    toks += (unsigned long long)b[1].delta;
    tiles += b[2].key;
    xmax = n > xmax ? n : xmax;
    if (n > (unsigned long long)b[0].delta) lost = 1ull;  // rank r lost records: every rank reads this verdict and stops
    if (only_mask && !((only_mask >> r) & 1ull)) continue;
    if (n > blk - XHDR) xmask |= 1ull << r;  // (reported for the own block too: every rank must reach the same verdict)
  }
  if (!only_mask) {  // (a repeat gathers the same headers again)
    xstat[4] = sites;
    xstat[5] = toks;
    xstat[6] = tiles;
    xstat[7] = (unsigned long long)world;
  }
  atomicMax(&xstat[1], xmax);
  if (lost) xstat[3] = 1ull;
  if (xmask) atomicOr(&xstat[0], xmask);
}

__global__ __launch_bounds__(BLOCK) void k_pt_apply_blocks(PairTable pt, const DeltaRec *__restrict__ blocks, unsigned long long blk, int world,
                                                           int rank, unsigned long long only_mask, unsigned long long *__restrict__ xstat,
                                                           unsigned long long *__restrict__ stats) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (stats && !only_mask) __hip_atomic_store(&stats[STAT_T1], (unsigned long long)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (apply kernels and all-gather are done)
    fold_headers(blocks, blk, world, only_mask, xstat);
  }
  for (int r = 0; r < world; r++) {
    const DeltaRec *b = blocks + (size_t)r * blk;
    const unsigned long long n = b[0].key;
    if (only_mask && !((only_mask >> r) & 1ull)) continue;
    if (n > blk - XHDR) continue;  // (reported by fold_headers: the repeat brings it)
    if (r == rank) continue;  // own deltas went into the table when they were made
    unsigned long long i = (unsigned long long)blockIdx.x * BLOCK + threadIdx.x;
    const unsigned long long stride = (unsigned long long)gridDim.x * BLOCK;
    for (; i < n; i += stride)
      if (b[XHDR + i].delta) pt_add(pt, b[XHDR + i].key, b[XHDR + i].delta);  // (updates of a round often cancel)
  }
}

// multi-GPU, per round, phase 2 (behind phase 1's kernel boundary: every rank's deltas are in the table), ONE workgroup.  A count that
// reached a list threshold during this round puts its slot on that list HERE, judged by the FINAL count -- the same on every rank -- and
// not by whichever adder happened to see a crossing (the apply kernels and phase 1 run with the thresholds off): a transient crossing
// -- this rank's +5 before another's -3 -- would list the slot on one rank and not on the other, and the lists' lengths (so: whether one
// overflowed) would have to be agreed on by a collective of their own every round.  Which slots to look at: the adds' notes
// (PairTable::maybe, a superset of the slots that can have crossed; a few dozen per round); should they have overflowed, every record
// with a positive delta of every block, this rank's included.  Several notes of one slot meet at the flag (atomicOr: whoever sets it
// appends).  Then the round's candidate scan (scan_top, straight into the host's mailbox), with the fold's report on the exchange (xstat).
constexpr int FOLD_NT = 512;
__device__ inline void fold_list_slot(const PairTable &pt, unsigned long long j) {
  const unsigned long long raw = ld_agent(pt.cnt_p(j)), c = raw & PT_CNT;
  unsigned long long want = 0;
  if (!(raw & PT_HOT) && c >= pt.hot_tau) want |= PT_HOT;
  if (!(raw & PT_TOP) && c >= pt.top_tau) want |= PT_TOP;
  if (!want) return;
  const unsigned long long fresh = want & ~atomicOr(pt.cnt_p(j), want);
  if (fresh & PT_HOT) {
    const unsigned int o = atomicAdd(pt.hot_n, 1u);
    if (o < pt.hot_cap) __hip_atomic_store(&pt.hot_slots[o], (uint32_t)j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (fresh & PT_TOP) {
    const unsigned int o = atomicAdd(pt.top_n, 1u);
    if (o < pt.top_cap) __hip_atomic_store(&pt.top_slots[o], (uint32_t)j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
__global__ __launch_bounds__(FOLD_NT) void k_fold_list(PairTable pt, const DeltaRec *__restrict__ blocks, unsigned long long blk, int world,
                                                        unsigned long long only_mask, ScanArgs sa, unsigned long long *__restrict__ stats,
                                                        const RuleSlot *__restrict__ zrules, unsigned int zmask, unsigned long long zself, BatchArgs zba,
                                                        unsigned long long *__restrict__ xstat, int read_headers) {
  __shared__ unsigned long long zkeys[FILTER_LDS_KEYS];
  __shared__ unsigned int scratch[CAND_BINS + 160];
  // (a communicator of one rank: phase 1 was not launched -- there is no other rank's block -- and the header is read here)
  if (read_headers && threadIdx.x == 0) {
    if (stats && !only_mask) __hip_atomic_store(&stats[STAT_T1], (unsigned long long)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    fold_headers(blocks, blk, world, only_mask, xstat);
  }
  const unsigned int n_maybe = pt.maybe_n ? *pt.maybe_n : 0u;
  if (pt.hot_tau != ~0ull) {
    if (n_maybe <= pt.maybe_cap) {
      for (unsigned int e = threadIdx.x; e < n_maybe; e += FOLD_NT) fold_list_slot(pt, pt.maybe[e]);
    } else {  // the notes overflowed (a round with tens of thousands of new candidates): every record that raised a count
      for (int r = 0; r < world; r++) {
        const DeltaRec *b = blocks + (size_t)r * blk;
        const unsigned long long n = b[0].key;
        if (n > blk - XHDR) continue;  // (skipped by phase 1 as well; the repeat brings it -- with the notes still overflowed)
        for (unsigned long long i = threadIdx.x; i < n; i += FOLD_NT) {
          if (b[XHDR + i].delta <= 0) continue;
          const unsigned long long key = b[XHDR + i].key;
          unsigned long long j = mix64(key) & pt.mask;
          for (;;) {
            const unsigned long long k = ld_agent(pt.key_p(j));
            if (k == PT_EMPTY) break;  // (cannot happen: a positive delta was added, so the key is there)
            if (k == key) { fold_list_slot(pt, j); break; }
            j = (j + 1) & pt.mask;
          }
        }
      }
    }
  }
  __syncthreads();
  // (the notes are consumed -- unless blocks were skipped and a repeat is to come: its pass must see the overflow verdict again)
  if (threadIdx.x == 0 && pt.maybe_n && !(n_maybe > pt.maybe_cap && ld_agent(&xstat[0]))) *pt.maybe_n = 0u;
  if (!sa.on) return;
  // ---- the round's candidate scan
  bool zkeys_in_lds = zrules && zmask < FILTER_LDS_KEYS;  // (the finished batch's pairs, to be zeroed: as in k_top_scan)
  if (zba.k) {
    zmask = 4 * BATCH_ARGS_MAX - 1;
    zkeys_in_lds = true;
    for (unsigned int sl = threadIdx.x; sl <= zmask; sl += FOLD_NT) zkeys[sl] = PT_EMPTY;
    __syncthreads();
    if (threadIdx.x < zba.k && zba.xy[2 * threadIdx.x] != zba.xy[2 * threadIdx.x + 1]) {  // (BATCH_ARGS_MAX <= the block size)
      const unsigned long long key = pair_key(zba.xy[2 * threadIdx.x], zba.xy[2 * threadIdx.x + 1]);
      unsigned int h = pair_hash32(key) & zmask;
      while (atomicCAS(&zkeys[h], PT_EMPTY, key) != PT_EMPTY) h = (h + 1) & zmask;
    }
  } else if (zkeys_in_lds) {
    for (unsigned int sl = threadIdx.x; sl <= zmask; sl += FOLD_NT) zkeys[sl] = zrules[sl].key;
  } else if (!zrules) {  // nothing to zero: an empty table
    zmask = 0;
    zkeys_in_lds = true;
    if (threadIdx.x == 0) zkeys[0] = PT_EMPTY;
  }
  __syncthreads();
  const RuleProbe zprobe{zkeys_in_lds ? zkeys : nullptr, zrules, zmask};
  ScanArgs sb = sa;
  sb.done_ctr = nullptr;  // (a single-workgroup launch)
  scan_top<FOLD_NT>(pt, sb, stats, zprobe, zself, scratch, xstat);
}

// Start of a merge round whose batch does not fit the kernel arguments, one launch instead of copies and memsets: the batch's rule hash
// and its pair filter are read straight from the host's pinned staging area (a few KB over PCIe), the worklist counters are reset.
__global__ __launch_bounds__(BLOCK) void k_round_begin(const RuleSlot *__restrict__ src_rules, unsigned int n_slots, RuleSlot *__restrict__ dst_rules,
                                                       unsigned int *__restrict__ work_n_a, unsigned int *__restrict__ work_n_b,
                                                       const uint32_t *__restrict__ src_bloom, uint32_t *__restrict__ dst_bloom) {
  const unsigned int tid = blockIdx.x * BLOCK + threadIdx.x, nt = gridDim.x * BLOCK;
  if (src_bloom)
    for (unsigned int i = tid; i < (unsigned int)PM_BLOOM_WORDS_H; i += nt) dst_bloom[i] = src_bloom[i];
  for (unsigned int i = tid; i < n_slots; i += nt)
    reinterpret_cast<uint4 *>(dst_rules)[i] = reinterpret_cast<const uint4 *>(src_rules)[i];
  if (tid == 0) {
    for (uint32_t i = 0; i <= WL_PARTS + 1; i++) {  // sub-list lengths, hand-out counter, "worklist incomplete" verdict
      if (work_n_a) work_n_a[i] = 0;
      if (work_n_b) work_n_b[i] = 0;
    }
  }
}

__global__ __launch_bounds__(BLOCK) void k_fill_u64(unsigned long long *__restrict__ p, unsigned long long v, unsigned long long n) {
  unsigned long long i = (unsigned long long)blockIdx.x * BLOCK + threadIdx.x;
  const unsigned long long stride = (unsigned long long)gridDim.x * BLOCK;
  for (; i < n; i += stride) p[i] = v;
}

// ------------------------------------------------------------------------------------------------- pair index (K4 worklists)
// Late in training a batch touches a few percent of the tiles, and which ones cannot be told from the tokens a tile holds (random
// text: every tile holds both halves of nearly every late rule).  So the pairs that can still be merged -- the hot list -- get an
// inverted index, pair -> tiles that hold it (the reference's pair2pos, bpe.cpp:438/:626/:694, at tile granularity): built with two
// streaming passes (count, fill) when the hot list is rebuilt, exact for pairs of tokens that existed then (a merge only creates
// adjacencies of its NEW token, and tokens never change tiles until a repack, which invalidates the index).  A round whose rules
// are all in the index gathers their posting lists into the worklist of the apply kernel instead of streaming every tile.
struct PairIndex {
  unsigned long long *key;  // [mask + 1] open addressing, PT_EMPTY = free
  uint32_t *cnt;            // [(mask + 1) * IDX_SHARDS] postings per key and shard (count pass), then the fill cursors
  unsigned long long *off;  // [(mask + 1) * IDX_SHARDS + 2] start of a (key, shard)'s postings (launch_exclusive_scan of the counts; the last = their total)
  uint32_t *bloom;          // [ENC_BLOOM_WORDS] blocked Bloom filter of the keys (staged into LDS by the streaming passes)
  uint32_t *post;           // tile ids
  unsigned int mask;
};
__device__ inline uint32_t idx_find(const PairIndex &ix, unsigned long long key, uint32_t h) {
  uint32_t s = h & ix.mask;
  for (;;) {
    const unsigned long long k = ix.key[s];
    if (k == key) return s;
    if (k == PT_EMPTY) return 0xffffffffu;
    s = (s + 1) & ix.mask;
  }
}
// every hot-list slot that is still at or above hot_tau becomes a key of the index
__global__ __launch_bounds__(BLOCK) void k_idx_seed(PairTable pt, PairIndex ix) {
  const unsigned int hn_raw = *pt.hot_n;
  const unsigned int hn = hn_raw < pt.hot_cap ? hn_raw : pt.hot_cap;
  for (unsigned int i = blockIdx.x * BLOCK + threadIdx.x; i < hn; i += gridDim.x * BLOCK) {
    const uint32_t sl = pt.hot_slots[i];
    const unsigned long long c = *pt.cnt_p(sl) & PT_CNT;
    if (c < pt.hot_tau || c == 0) continue;
    const unsigned long long key = *pt.key_p(sl);
    const uint32_t h = enc_hash((uint32_t)(key >> 32), (uint32_t)key);
    uint32_t s = h & ix.mask;
    for (;;) {
      const unsigned long long k = atomicCAS(&ix.key[s], PT_EMPTY, key);
      if (k == PT_EMPTY || k == key) break;
      s = (s + 1) & ix.mask;
    }
    atomicOr(&ix.bloom[enc_bloom_word(h)], enc_bloom_bits(h));
  }
}
// One wavefront per tile, tokens in registers: every adjacency whose pair is a key of the index is counted (FILL = false) or has
// its tile / its word appended to the key's postings (FILL = true; one posting per adjacency: duplicates are harmless, the gather
// claims a tile / a word once).  At the word-mode switch nearly every adjacency is a posting (1 GB corpus: 78 M of 94 M tokens, 10 000
// keys): one global atomic per posting was 6 ms per pass.  So a workgroup sums its postings per key in an LDS table first -- count pass:
// one global add per key and workgroup; fill pass: count, reserve the workgroup's run of each key with ONE add, then go over the tiles
// again and hand the run out from LDS cursors.  Keys that find no room in the table take the global atomic per posting as before.
constexpr int IDXA_NT = 512, IDXA_SLOTS = 2048, IDXA_BITS = 11, IDXA_PROBES = 8;
struct IdxAgg {                          // the workgroup's table: pair -> its slot of the index and this workgroup's postings of it.  The
  unsigned long long key[IDXA_SLOTS];    // frequent pairs get in first (PT_EMPTY = free) and are then resolved without leaving the CU:
  uint32_t slot[IDXA_SLOTS];             // no Bloom test, no probe of the index in L2
  uint32_t cnt[IDXA_SLOTS];              // postings of this workgroup; fill pass, second sweep: the cursor
  uint32_t base[IDXA_SLOTS];             // fill pass: start of this workgroup's run of the key's postings
};
// entry of `key` in its probe window, or -1
__device__ inline int idxa_lookup(const IdxAgg &T, unsigned long long key) {
  uint32_t h = pair_hash32(key) >> (32 - IDXA_BITS);
  static_assert(IDXA_SLOTS == 1 << IDXA_BITS, "hash bits");
  for (int p = 0; p < IDXA_PROBES; p++) {
    const unsigned long long k = __hip_atomic_load(&T.key[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (k == key) return (int)h;
    if (k == PT_EMPTY) return -1;
    h = (h + 1) & (IDXA_SLOTS - 1);
  }
  return -1;
}
// SWEEP 0: count into the table (count pass: + global adds for what finds no room); SWEEP 1 (fill pass): write the postings
template <int SLOT, bool FILL, bool WORDS, int SWEEP>
__device__ inline void idx_sweep(const TileSet &ts, const PairIndex &ix, const uint32_t *bloom, IdxAgg &T, uint32_t shard) {
  const int lane = lane_id();
  const uint32_t stride = gridDim.x * (IDXA_NT / 64);
  for (uint32_t t = uni(blockIdx.x * (IDXA_NT / 64) + (uint32_t)(threadIdx.x >> 6)); t < ts.n_tiles; t += stride) {  // (uni: a wave's tile is the same in its lanes)
    const int n = (int)ts.tile_len[t];
    uint4 r[SLOT / 256];
    tile_fetch<SLOT>(r, ts, t, n);
    // WORDS: a posting is the WORD that holds the adjacency (word mode: the tile may hold TOK_HOLEs -- word-start bit set, so never the
    // second token of an adjacency; never counted as a word start).
    uint32_t wrow = WORDS ? ts.tile_word0[t] : 0u;
    (void)wrow;
#define IDX_PAIR(T0, T1, WIDX)                                                        \
  if (!((T1)&TOK_WS)) {                                                                \
    const uint32_t a_ = (T0)&TOK_MASK, b_ = (T1)&TOK_MASK;                             \
    const unsigned long long key_ = pair_key(a_, b_);                                  \
    const uint32_t h_ = enc_hash(a_, b_);                                              \
    const uint32_t bits_ = enc_bloom_bits(h_);                                         \
    const bool maybe_ = (bloom[enc_bloom_word(h_)] & bits_) == bits_;  /* (late builds: few adjacencies are keys of the index) */ \
    int e_ = maybe_ ? idxa_lookup(T, key_) : -1;                                       \
    uint32_t s_ = 0xffffffffu;                                                         \
    if (maybe_ && e_ < 0) {  /* not in the table: is it a key of the index at all? */   \
      s_ = idx_find(ix, key_, h_);                                                     \
      if (SWEEP == 0 && s_ != 0xffffffffu) {  /* a place in the table, if its probe window has one */ \
        uint32_t w_ = pair_hash32(key_) >> (32 - IDXA_BITS);                           \
        for (int p_ = 0; p_ < IDXA_PROBES; p_++) {                                     \
          unsigned long long k_ = __hip_atomic_load(&T.key[w_], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
          if (k_ == PT_EMPTY) {                                                        \
            k_ = atomicCAS(&T.key[w_], PT_EMPTY, key_);                                \
            if (k_ == PT_EMPTY) k_ = key_;                                             \
          }                                                                            \
          if (k_ == key_) {                                                            \
            T.slot[w_] = s_;                                                           \
            e_ = (int)w_;                                                              \
            break;                                                                     \
          }                                                                            \
          w_ = (w_ + 1) & (IDXA_SLOTS - 1);                                            \
        }                                                                              \
      }                                                                                \
    }                                                                                  \
    if (SWEEP == 0) {                                                                  \
      if (e_ >= 0) atomicAdd(&T.cnt[e_], 1u);                                          \
      else if (!FILL && s_ != 0xffffffffu) atomicAdd(&ix.cnt[(size_t)s_ * IDX_SHARDS + shard], 1u); \
    } else if (e_ >= 0 || s_ != 0xffffffffu) {                                         \
      unsigned long long at_;                                                          \
      if (e_ >= 0) {                                                                   \
        at_ = (unsigned long long)T.base[e_] + atomicAdd(&T.cnt[e_], 1u);              \
      } else {                                                                         \
        const size_t cs_ = (size_t)s_ * IDX_SHARDS + shard;                            \
        at_ = ix.off[cs_] + atomicAdd(&ix.cnt[cs_], 1u);                               \
      }                                                                                \
      ix.post[at_] = WORDS ? (WIDX) : t;                                               \
    }                                                                                  \
  }
#pragma unroll
    for (int j = 0; j < SLOT / 256; j++) {
      if (256 * j < n) {
        uint32_t nx = from_lane_right(r[j].x);
        uint32_t nx0 = TOK_WS;
        if (j + 1 < SLOT / 256) nx0 = from_lane0(r[j + 1 < SLOT / 256 ? j + 1 : j].x);
        if (lane == 63) nx = nx0;
        uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;  // word of my token i
        if (WORDS && SWEEP == 1) {
          const bool s0 = tok_is_ws(r[j].x), s1 = tok_is_ws(r[j].y), s2 = tok_is_ws(r[j].z), s3 = tok_is_ws(r[j].w);
          const unsigned long long m0 = __ballot(s0), m1 = __ballot(s1), m2 = __ballot(s2), m3 = __ballot(s3);
          const unsigned long long lt = lanemask_lt();
          const uint32_t wb = wrow + (uint32_t)(__popcll(m0 & lt) + __popcll(m1 & lt) + __popcll(m2 & lt) + __popcll(m3 & lt));
          w0 = wb + (s0 ? 1u : 0u) - 1u;
          w1 = w0 + (s1 ? 1u : 0u);
          w2 = w1 + (s2 ? 1u : 0u);
          w3 = w2 + (s3 ? 1u : 0u);
          wrow += (uint32_t)(__popcll(m0) + __popcll(m1) + __popcll(m2) + __popcll(m3));
        }
        // (slots behind the live prefix hold zeros: id 0 is a special token, never part of a pair of the index)
        IDX_PAIR(r[j].x, r[j].y, w0)
        IDX_PAIR(r[j].y, r[j].z, w1)
        IDX_PAIR(r[j].z, r[j].w, w2)
        IDX_PAIR(r[j].w, nx, w3)
      }
    }
#undef IDX_PAIR
  }
}
// `save` [gridDim.x]: the count pass leaves every workgroup's table there, and the fill pass of a build with many postings (agg) starts from
// it instead of counting again (same grid, same tiles per workgroup: the table is exactly what that sweep would rebuild -- a third of the
// build's work).
template <int SLOT, bool FILL, bool WORDS>
__global__ __launch_bounds__(IDXA_NT) void k_idx_stream(TileSet ts, PairIndex ix, int agg /* fill pass: 0 = few postings, one sweep with an atomic each */,
                                                        IdxAgg *__restrict__ save) {
  __shared__ IdxAgg T;
  __shared__ uint32_t bloom[ENC_BLOOM_WORDS];
  for (int i = (int)threadIdx.x; i < ENC_BLOOM_WORDS; i += IDXA_NT) bloom[i] = ix.bloom[i];
  const bool reload = FILL && agg && save;
  for (int i = (int)threadIdx.x; i < IDXA_SLOTS; i += IDXA_NT) {
    T.key[i] = reload ? save[blockIdx.x].key[i] : PT_EMPTY;
    T.cnt[i] = reload ? save[blockIdx.x].cnt[i] : 0u;
    if (reload) T.slot[i] = save[blockIdx.x].slot[i];
  }
  __syncthreads();
  const uint32_t shard = blockIdx.x % IDX_SHARDS;
  if (!FILL || (agg && !reload)) idx_sweep<SLOT, FILL, WORDS, 0>(ts, ix, bloom, T, shard);
  __syncthreads();
  for (int i = (int)threadIdx.x; i < IDXA_SLOTS; i += IDXA_NT) {
    const uint32_t c = T.cnt[i];
    if (!FILL && save) {
      save[blockIdx.x].key[i] = T.key[i];
      save[blockIdx.x].slot[i] = T.slot[i];
      save[blockIdx.x].cnt[i] = c;
    }
    if (T.key[i] == PT_EMPTY || !c) continue;
    const size_t cs = (size_t)T.slot[i] * IDX_SHARDS + shard;
    const uint32_t b0 = atomicAdd(&ix.cnt[cs], c);
    if (FILL) {
      T.base[i] = (uint32_t)ix.off[cs] + b0;  // (the postings number fewer than 2^32: build_index checks)
      T.cnt[i] = 0;
    }
  }
  if (!FILL) return;
  __syncthreads();
  idx_sweep<SLOT, FILL, WORDS, 1>(ts, ix, bloom, T, shard);
}
// ------------------------------------------------------------------------------------------------- word mode
// Once a round's merge sites are few against the tokens a pass over the tiles streams, class-A words leave the tiles (yttm_device.h:
// WordSet): a round then (1) k_wgather looks the batch's rules up -- postings of the pair index, or the instance list of the pair's
// younger token -- and claims each word that may hold a site once; (2) k_words gathers those words, 64 at a time, into a wave's LDS
// tile, runs the same site search and count-delta code as a tile does (reg_find_sites, process_tile), writes the shrunk words back
// into their slots and records every new token instance in its token's list.  Work follows the merge sites (the reference's
// pair2pos, bpe.cpp:438/:626/:694), not the table.

// tile -> wmeta of its words (the switch; one wave per tile)
template <int SLOT>
__global__ __launch_bounds__(BLOCK) void k_words_init(TileSet ts, unsigned long long *__restrict__ wmeta) {
  const int lane = lane_id();
  constexpr int NC = SLOT / 64;
  const uint32_t stride = gridDim.x * NWAVES;
  for (uint32_t t = uni(blockIdx.x * NWAVES + (uint32_t)(threadIdx.x >> 6)); t < ts.n_tiles; t += stride) {
    const int n = (int)ts.tile_len[t];
    const uint32_t *src = ts.tok + (size_t)t * SLOT;
    unsigned long long m[NC];
    uint32_t wbase[NC];
    uint32_t acc = ts.tile_word0[t];
#pragma unroll
    for (int c = 0; c < NC; c++) {
      const int p = c * 64 + lane;
      m[c] = __ballot(p < n && (src[p] & TOK_WS));
      wbase[c] = acc;
      acc += (uint32_t)__popcll(m[c]);
    }
    int next_start = n;  // first word start behind the chunk
#pragma unroll
    for (int c = NC - 1; c >= 0; c--) {
      const unsigned long long mc = m[c];
      if ((mc >> lane) & 1ull) {
        const int p = c * 64 + lane;
        const unsigned long long after = mc & ~((2ull << lane) - 1ull);
        const int end = after ? c * 64 + (__ffsll((long long)after) - 1) : next_start;
        const uint32_t w = wbase[c] + (uint32_t)__popcll(mc & lanemask_lt());
        wmeta[w] = (((unsigned long long)t * SLOT + (unsigned long long)p) << 16) | (unsigned long long)(end - p);
      }
      if (mc) next_start = c * 64 + (__ffsll((long long)mc) - 1);
    }
  }
}

// The rules of the batch -> the round's worklist of words.  Every workgroup works out where each rule's candidates are (a posting
// run of the index, or the instance list of the younger token and the neighbour to look for), lays the runs end to end and takes its
// share of the whole -- a rule with a million records and one with ten cost the same per record.  The last workgroup to finish allots
// the instance lists of the batch's new tokens (at most one record per record matched: every site was one of them).
constexpr int WG_NT = 512, WG_BUF = 4096;
__global__ __launch_bounds__(WG_NT) void k_wgather(WGatherArgs g, BatchArgs ba) {
  __shared__ unsigned long long s_base[WGATHER_MAXK];
  __shared__ unsigned long long s_pref[WGATHER_MAXK + 1];
  __shared__ uint32_t s_filt[WGATHER_MAXK];
  __shared__ uint32_t s_cnt[WGATHER_MAXK];
  __shared__ uint8_t s_mode[WGATHER_MAXK];  // 0: postings (word ids), 1: records whose left neighbour is s_filt, 2: ... right neighbour
  __shared__ uint32_t s_buf[WG_BUF];
  __shared__ unsigned long long s_wsum[WG_NT / 64];
  __shared__ unsigned int s_n, s_gbase, s_last;
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (ba.mark && blockIdx.x == 0 && tid == 0)
    __hip_atomic_store(&g.stats[STAT_T0], (unsigned long long)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const uint32_t k = g.k;
  const PairIndex ix{g.ix.key, g.ix.cnt, g.ix.off, g.ix.bloom, g.ix.post, g.ix.mask};
  if (tid == 0) s_n = 0;
  // ---- the rules' runs; thread t owns rules [t * per, (t + 1) * per)
  const uint32_t per = (k + WG_NT - 1) / WG_NT;
  unsigned long long mine = 0;
  for (uint32_t j = (uint32_t)tid * per; j < ((uint32_t)tid + 1) * per && j < k; j++) {
    const uint32_t x = g.xyz ? g.xyz[3 * j] : ba.xy[2 * j], y = g.xyz ? g.xyz[3 * j + 1] : ba.xy[2 * j + 1];
    const uint32_t m = x > y ? x : y;
    unsigned long long base = 0, len = 0;
    uint32_t filt = 0, mode = 0;
    if (m < g.z_static) {
      uint32_t s = 0xffffffffu;
      if (g.ix_valid) s = idx_find(ix, pair_key(x, y), enc_hash(x, y));
      if (s == 0xffffffffu) {
        g.work_n[WL_PARTS + 1] = 1u;  // not in the index: this round takes every word
      } else {
        base = ix.off[(size_t)s * IDX_SHARDS];
        len = ix.off[((size_t)s + 1) * IDX_SHARDS] - base;
      }
    } else {
      base = g.tl.base[m];
      const uint32_t f = g.tl.fill[m], c = g.tl.cap[m];
      len = f < c ? f : c;
      if (x > y) { mode = 2; filt = y; } else { mode = 1; filt = x; }
    }
    s_base[j] = base;
    s_pref[j] = len;
    s_filt[j] = filt;
    s_mode[j] = (uint8_t)mode;
    s_cnt[j] = 0;
    mine += len;
  }
  {  // exclusive scan of the run lengths over the workgroup
    unsigned long long inc = mine;
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned long long t = __shfl_up(inc, o);
      if (lane >= o) inc += t;
    }
    if (lane == 63) s_wsum[wave] = inc;
    __syncthreads();
    unsigned long long before = inc - mine;
    for (int w = 0; w < wave; w++) before += s_wsum[w];
    for (uint32_t j = (uint32_t)tid * per; j < ((uint32_t)tid + 1) * per && j < k; j++) {
      const unsigned long long l = s_pref[j];
      s_pref[j] = before;
      before += l;
    }
    if (tid == WG_NT - 1) s_pref[k] = before;  // (the last thread's runs end the sequence, whether it owns rules or not)
    __syncthreads();
  }
  const unsigned long long total = s_pref[k];
  const unsigned long long chunk = (total + gridDim.x - 1) / gridDim.x;
  const unsigned long long lo = (unsigned long long)blockIdx.x * chunk, hi = lo + chunk < total ? lo + chunk : total;
  const uint32_t part = 0;  // (word mode keeps ONE list: a workgroup appends once, at its end)
  for (unsigned long long i0 = lo; i0 < hi; i0 += WG_NT) {
    const unsigned long long i = i0 + (unsigned long long)tid;
    bool hit = false;
    uint32_t w = 0;
    if (i < hi) {
      uint32_t a = 0, b = k;  // the rule whose run holds record i: the last j with s_pref[j] <= i
      while (b - a > 1) {
        const uint32_t mid = (a + b) >> 1;
        if (s_pref[mid] <= i) a = mid; else b = mid;
      }
      const unsigned long long at = s_base[a] + (i - s_pref[a]);
      const uint32_t mode = s_mode[a];
      bool match = true;
      if (mode == 0) {
        w = ix.post[at];
      } else {
        match = (mode == 1 ? g.tl.rec_l[at] : g.tl.rec_r[at]) == s_filt[a];
        if (match) w = g.tl.rec_word[at];
      }
      if (match) {
        atomicAdd(&s_cnt[a], 1u);
        hit = atomicExch(&g.stamp[w], g.round_id) != g.round_id;  // each word once per round
      }
    }
    const unsigned long long hm = __ballot(hit);
    if (hm) {
      unsigned int b0 = 0;
      const int first = __ffsll((long long)hm) - 1;
      if (lane == first) b0 = atomicAdd(&s_n, (unsigned int)__popcll(hm));
      b0 = (unsigned int)__shfl((int)b0, first);
      if (hit) {
        const unsigned int pos = b0 + (unsigned int)__popcll(hm & lanemask_lt());
        if (pos < (unsigned int)WG_BUF) s_buf[pos] = w;
        else g.worklist[part * g.wl_seg + atomicAdd(&g.work_n[part], 1u)] = w;  // (the buffer is full: one by one)
      }
    }
  }
  __syncthreads();
  const unsigned int nbuf = s_n < (unsigned int)WG_BUF ? s_n : (unsigned int)WG_BUF;
  if (tid == 0 && nbuf) s_gbase = atomicAdd(&g.work_n[part], nbuf);
  __syncthreads();
  for (unsigned int i = (unsigned int)tid; i < nbuf; i += WG_NT) g.worklist[part * g.wl_seg + s_gbase + i] = s_buf[i];
  for (uint32_t j = (uint32_t)tid; j < k; j += WG_NT)
    if (s_cnt[j]) atomicAdd(&g.gm[j], s_cnt[j]);
  // ---- the last workgroup allots the new tokens' lists (the counts went out as device-scope atomics: see k_tiles on the ticket)
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  if (tid == 0) s_last = atomicAdd(g.done_ctr, 1u) == gridDim.x - 1 ? 1u : 0u;
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  unsigned long long mine2 = 0;
  for (uint32_t j = (uint32_t)tid * per; j < ((uint32_t)tid + 1) * per && j < k; j++) {
    const unsigned int c = __hip_atomic_load(&g.gm[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_cnt[j] = c;
    mine2 += c;
  }
  unsigned long long inc = mine2;
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned long long t = __shfl_up(inc, o);
    if (lane >= o) inc += t;
  }
  __syncthreads();
  if (lane == 63) s_wsum[wave] = inc;
  __syncthreads();
  unsigned long long before = inc - mine2, all = 0;
  for (int w = 0; w < WG_NT / 64; w++) {
    if (w < wave) before += s_wsum[w];
    all += s_wsum[w];
  }
  const unsigned long long cur = g.tl.cursor[g.round_id & 1u];  // (a round reads the cursor of its parity and leaves the other one: k_words<FUSED>)
  const bool fits = cur + all <= g.tl.log_cap && !g.work_n[WL_PARTS + 1];
  for (uint32_t j = (uint32_t)tid * per; j < ((uint32_t)tid + 1) * per && j < k; j++) {
    const uint32_t z = g.z_base + j;
    g.tl.base[z] = cur + before;
    g.tl.cap[z] = fits ? s_cnt[j] : 0u;
    g.tl.fill[z] = 0u;
    before += s_cnt[j];
    g.gm[j] = 0u;
  }
  __syncthreads();
  if (tid == 0) {
    g.tl.cursor[(g.round_id + 1u) & 1u] = fits ? cur + all : cur;
    if (!fits) __hip_atomic_store(g.tl.broken, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // (log full, or a round that took every word: its matches say nothing about its sites)
    *g.done_ctr = 0u;
  }
}

// per-wave state of k_words on top of the tile state
struct WordsLds {
  unsigned long long lin[TILE_SLOT_A / 64];      // bit p: a word starts at position p of the gathered tile
  unsigned long long newsite[TILE_SLOT_A / 64];  // bit q: the token at position q of the compacted tile is a new one
  uint32_t wnew[66];                             // start of word i in the compacted tile; [nw] = its length
};
// FUSED (a small round, a batch that travels in the kernel arguments): no k_wgather before this kernel and no worklist in HBM -- every
// workgroup looks the rules up itself, takes its share of their runs, claims the words (stamps) into a list in LDS and works through that
// list.  The new tokens' lists are allotted without a count of the matches: rule j gets min(its run's length, the pair's count the host
// picked it by) records -- no fewer than its sites (a site is a candidate record, and every site adds at least one to the count) --
// from the cursor of the round's parity; every workgroup works that out alike and workgroup 0 writes it down for the rounds to come.
// (Moving the records AFTER the candidates were published, by the last workgroup, was tried: that tail -- one workgroup reading every
// region across XCDs -- took 40 us a round, far longer than the host's turn it was meant to hide in.)
constexpr int FUSE_LIST = 4096, FUSE_PASS = 2048, FUSE_WORD_COST = 32;  // (words claimed go to a list in LDS, FUSE_PASS candidate records at a time; the words are worked on when another pass might not fit)
template <int WPB, bool LDSR, bool FUSED>
__global__ __launch_bounds__(WPB * 64) void k_words(WordSet ws, PairTable pt, DeltaBuf db, const RuleSlot *__restrict__ rules, unsigned int rule_mask,
                                                    const uint32_t *__restrict__ bloom_g, uint32_t self_x, uint32_t self_z, uint32_t z_base, uint32_t k_rules,
                                                    const uint32_t *__restrict__ worklist, unsigned long long wl_seg,
                                                    const unsigned int *__restrict__ work_n, unsigned long long *__restrict__ stats, TokLists tl,
                                                    DeltaRec *__restrict__ drec, unsigned int drec_cap /* per workgroup */, unsigned int *__restrict__ drec_n,
                                                    uint4 *__restrict__ irec /* new-instance records, a region of drec_cap per workgroup too */, unsigned int wpi,
                                                    unsigned int inline_apply /* needs sa.on */, BatchArgs ba, ScanArgs sa, WGatherArgs g /* FUSED */) {
  constexpr int SLOT = TILE_SLOT_A;
  static_assert(!FUSED || (LDSR && BATCH_ARGS_MAX <= WPB * 64 && FUSE_PASS % (WPB * 64) == 0 && FUSE_PASS <= FUSE_LIST), "the fused round: rules from the arguments, one thread per rule");
  __shared__ unsigned long long f_base[FUSED ? BATCH_ARGS_MAX : 1], f_pref[FUSED ? BATCH_ARGS_MAX + 1 : 1];
  __shared__ uint32_t f_filt[FUSED ? BATCH_ARGS_MAX : 1], f_list[FUSED ? FUSE_LIST : 1];
  __shared__ uint8_t f_mode[FUSED ? BATCH_ARGS_MAX : 1];
  __shared__ unsigned long long f_lbase[FUSED ? BATCH_ARGS_MAX : 1];  // the new tokens' lists: allotted by every workgroup alike, from the same numbers
  __shared__ uint32_t f_lcap[FUSED ? BATCH_ARGS_MAX : 1];
  __shared__ unsigned long long f_wpref[FUSED ? BATCH_ARGS_MAX + 1 : 1];  // the runs' cost (records to read + words to work on), summed like f_pref
  __shared__ unsigned long long f_tmp[4];
  __shared__ unsigned int f_n, f_every;
  __shared__ WaveLds<SLOT> WL[WPB];
  __shared__ WordsLds XL[WPB];
  __shared__ unsigned int dn;  // records of this workgroup
  __shared__ AggLds A;
  __shared__ unsigned int rn;  // new-instance records of this workgroup (its region of irec; put into the tokens' lists at the end)
  __shared__ unsigned long long rkeys[LDSR ? APPLY_LDS_RULES : 1];
  __shared__ uint16_t rridx[LDSR ? APPLY_LDS_RULES : 1];
  const bool from_args = LDSR && ba.k != 0;
  agg_init<WPB * 64>(A, from_args ? nullptr : bloom_g);  // (A.flagbits holds the batch's pair filter)
  if (threadIdx.x == 0) rn = 0;
  if (threadIdx.x == 0) dn = 0;
  if (from_args) {
    for (int s = (int)threadIdx.x; s < (int)(FLAG_LDS_IDS / 16); s += WPB * 64) A.flagbits[s] = 0;
    for (unsigned int i = threadIdx.x; i <= rule_mask; i += WPB * 64) rkeys[i] = PT_EMPTY;
    __syncthreads();
    for (unsigned int j = threadIdx.x; j < ba.k; j += WPB * 64) {
      const uint32_t x = ba.xy[2 * j], y = ba.xy[2 * j + 1];
      if (x != y) {
        const uint32_t bh = pm_hash(x, y);
        atomicOr(&A.flagbits[pm_word(bh)], pm_bits(bh));
        const unsigned long long key = pair_key(x, y);
        unsigned int h = pair_hash32(key) & rule_mask;
        for (;;) {
          if (atomicCAS(&rkeys[h], PT_EMPTY, key) == PT_EMPTY) {
            rridx[h] = (uint16_t)j;
            break;
          }
          h = (h + 1) & rule_mask;
        }
      }
    }
  } else if (LDSR) {
    for (unsigned int i = threadIdx.x; i <= rule_mask; i += WPB * 64) {
      rkeys[i] = rules[i].key;
      rridx[i] = (uint16_t)(rules[i].z - z_base);
    }
  }
  const RuleTab<LDSR> rtab{rkeys, rridx, rules, rule_mask, z_base};
  if (FUSED && threadIdx.x == 0) {
    f_n = 0;
    f_every = 0;
    if (ba.mark && blockIdx.x == 0) __hip_atomic_store(&stats[STAT_T0], (unsigned long long)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  const int wave = uni((int)(threadIdx.x >> 6)), lane = lane_id();
  if (FUSED) {  // where each rule's candidates are (as k_wgather: a posting run of the index, or the younger token's instance list)
    const PairIndex ix{g.ix.key, g.ix.cnt, g.ix.off, g.ix.bloom, g.ix.post, g.ix.mask};
    const uint32_t j = threadIdx.x;
    unsigned long long len = 0, lcap = 0;
    bool f_mode_of_mine = false;
    if (j < ba.k) {
      const uint32_t x = ba.xy[2 * j], y = ba.xy[2 * j + 1];
      const uint32_t m = x > y ? x : y;
      unsigned long long base = 0;
      uint32_t filt = 0, mode = 0;
      bool found = true;
      if (m < g.z_static) {
        uint32_t s = 0xffffffffu;
        if (g.ix_valid) s = idx_find(ix, pair_key(x, y), enc_hash(x, y));
        if (s == 0xffffffffu) {
          f_every = 1u;  // not in the index: this round takes every word
          found = false;
        } else {
          base = ix.off[(size_t)s * IDX_SHARDS];
          len = ix.off[((size_t)s + 1) * IDX_SHARDS] - base;
        }
      } else {
        base = g.tl.base[m];
        const uint32_t f = g.tl.fill[m], c = g.tl.cap[m];
        len = f < c ? f : c;
        if (x > y) { mode = 2; filt = y; } else { mode = 1; filt = x; }
      }
      f_base[j] = base;
      f_filt[j] = filt;
      f_mode[j] = (uint8_t)mode;
      f_mode_of_mine = mode != 0;
      const unsigned long long cj = g.cnt[j];
      lcap = found && len < cj ? len : cj;
    }
    static_assert(BATCH_ARGS_MAX <= 128, "two waves scan the runs");
    // what a run costs: a record to read each, and FUSE_WORD_COST of those per word to work on -- every posting's word, but only the
    // records of an instance list that have the right neighbour (about as many as the pair's count says)
    const unsigned long long wgt = len + (unsigned long long)FUSE_WORD_COST * (f_mode_of_mine ? lcap : len);
    unsigned long long inc = len, cinc = lcap, winc = wgt;
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned long long t = __shfl_up(inc, o), tc = __shfl_up(cinc, o), tw = __shfl_up(winc, o);
      if (lane >= o) { inc += t; cinc += tc; winc += tw; }
    }
    if (threadIdx.x == 63) { f_tmp[0] = inc; f_tmp[1] = cinc; f_tmp[2] = winc; }  // (wave 0's sums)
    if (threadIdx.x == 127) f_tmp[3] = cinc;                                      // (wave 1's sum of the allotments)
    __syncthreads();
    const unsigned long long w0 = f_tmp[0], wc0 = f_tmp[1], ww0 = f_tmp[2];
    const unsigned long long call = wc0 + f_tmp[3];
    if (wave == 1) { inc += w0; cinc += wc0; winc += ww0; }
    const unsigned long long cur = g.tl.cursor[g.round_id & 1u];
    const bool fits = cur + call <= g.tl.log_cap;
    if (j < ba.k) {
      f_pref[j] = inc - len;
      f_wpref[j] = winc - wgt;
      f_lbase[j] = cur + (cinc - lcap);
      f_lcap[j] = fits ? (uint32_t)lcap : 0u;
      if (blockIdx.x == 0) {
        g.tl.base[z_base + j] = cur + (cinc - lcap);
        g.tl.cap[z_base + j] = fits ? (uint32_t)lcap : 0u;  // (their fill counts are at zero: enter_word_mode -- a token is new once)
      }
    }
    if (j + 1 == ba.k) { f_pref[ba.k] = inc; f_wpref[ba.k] = winc; }
    if (ba.k == 0 && j == 0) { f_pref[0] = 0; f_wpref[0] = 0; }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      g.tl.cursor[(g.round_id + 1u) & 1u] = fits ? cur + call : cur;
      if (!fits) __hip_atomic_store(g.tl.broken, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // (log full: the index is rebuilt)
    }
    __syncthreads();
  }
  WaveLds<SLOT> &W = WL[wave];
  WordsLds &X = XL[wave];
  const TileSet ts{ws.tok, nullptr, nullptr, ws.wcnt, 0u};
  const DeltaOut dout{drec + (size_t)blockIdx.x * drec_cap, &dn, drec_cap};
  uint4 *my_irec = irec + (size_t)blockIdx.x * drec_cap;
  // work items: runs of 64 worklist entries, or of 64 words
  const bool listed = FUSED && !f_every;  // (uniform; a fused round that must take every word walks them like an unfused one)
  if (FUSED) worklist = nullptr;
  else if (worklist && work_n[WL_PARTS + 1]) worklist = nullptr;
  const uint32_t wl_n = worklist ? work_n[0] : 0u;
  // (wpi words per work item: 64 when there are words for every wave; fewer in the small late rounds -- a wave's time goes with the tokens
  // of its tile, and the chip has thousands of idle wave slots then)
  const unsigned long long n_glob = worklist ? (unsigned long long)wl_n : (unsigned long long)ws.n_words;
  (void)wl_seg;
  // fused: my share of the rules' runs laid end to end -- one stretch of records (the words of neighbouring postings are neighbours in
  // HBM; shares dealt out in small blocks cost the big rounds of random text 180 -> 270 us), cut by COST, not by records: the words
  // are what takes the time, and a posting is a word where a record of an instance list mostly is not (equal record counts left some
  // workgroups with all the words: CJK-shaped text, rounds 200 .. 700, 260 -> 440 us)
  const unsigned long long f_total = listed ? f_pref[ba.k] : 0ull;
  unsigned long long f_pos = 0, f_hi = 0;
  if (listed) {
    const unsigned long long wtotal = f_wpref[ba.k], wchunk = (wtotal + gridDim.x - 1) / gridDim.x;
    auto rec_of = [&](unsigned long long w) -> unsigned long long {  // the record at cost w from the start (monotonic)
      if (w >= wtotal) return f_total;
      uint32_t a = 0, b = ba.k;
      while (b - a > 1) {
        const uint32_t mid = (a + b) >> 1;
        if (f_wpref[mid] <= w) a = mid; else b = mid;
      }
      const unsigned long long wa = f_wpref[a + 1] - f_wpref[a], la = f_pref[a + 1] - f_pref[a];
      if (!wa) return f_pref[a];
      unsigned long long off = (unsigned long long)((double)(w - f_wpref[a]) / (double)wa * (double)la);
      if (off > la) off = la;
      return f_pref[a] + off;
    };
    const unsigned long long lo_w = (unsigned long long)blockIdx.x * wchunk < wtotal ? (unsigned long long)blockIdx.x * wchunk : wtotal;
    f_pos = rec_of(lo_w);
    f_hi = rec_of(lo_w + wchunk);
  }
  TileStats S;
#ifdef YTTM_K4_PROF
  S.t_last = (unsigned long long)clock64();
#endif
  for (;;) {  // (once; fused: once per list of claimed words)
  unsigned long long n_all = n_glob, item0 = (unsigned long long)blockIdx.x * WPB + (unsigned long long)wave, istride = (unsigned long long)gridDim.x * WPB;
  if (listed) {
    while (f_pos < f_hi) {  // gather: FUSE_PASS records of my share at a time, until the list could not take another pass
    const unsigned long long pend = f_pos + FUSE_PASS < f_hi ? f_pos + FUSE_PASS : f_hi;
    constexpr int NIT = FUSE_PASS / (WPB * 64);
    uint32_t cw[NIT];
    bool cok[NIT];
#pragma unroll
    for (int it = 0; it < NIT; it++) {  // the records: all loads of the pass in flight together
      const unsigned long long i = f_pos + (unsigned long long)(it * WPB * 64) + (unsigned long long)threadIdx.x;
      cw[it] = 0;
      cok[it] = false;
      if (i < pend) {
        uint32_t a = 0, b = ba.k;  // the rule whose run holds record i: the last j with f_pref[j] <= i
        while (b - a > 1) {
          const uint32_t mid = (a + b) >> 1;
          if (f_pref[mid] <= i) a = mid; else b = mid;
        }
        const unsigned long long at = f_base[a] + (i - f_pref[a]);
        const uint32_t mode = f_mode[a];
        if (mode == 0) {
          cw[it] = g.ix.post[at];
          cok[it] = true;
        } else {
          cok[it] = (mode == 1 ? g.tl.rec_l[at] : g.tl.rec_r[at]) == f_filt[a];
          if (cok[it]) cw[it] = g.tl.rec_word[at];
        }
      }
    }
#pragma unroll
    for (int it = 0; it < NIT; it++)  // each word once per round
      if (cok[it]) cok[it] = atomicExch(&g.stamp[cw[it]], g.round_id) != g.round_id;
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const unsigned long long hm = __ballot(cok[it]);
      if (hm) {
        unsigned int b0 = 0;
        const int fl = __ffsll((long long)hm) - 1;
        if (lane == fl) b0 = atomicAdd(&f_n, (unsigned int)__popcll(hm));
        b0 = (unsigned int)__shfl((int)b0, fl);
        if (cok[it]) f_list[b0 + (unsigned int)__popcll(hm & lanemask_lt())] = cw[it];  // (there is room: see the loop's end)
      }
    }
    f_pos = pend;
    __syncthreads();
    const unsigned int n_now = f_n;
    __syncthreads();  // (every thread has read the length before the next pass adds to it: the decision is the workgroup's)
    if (n_now > (unsigned int)(FUSE_LIST - FUSE_PASS)) break;
    }
    n_all = f_n;
    if (!n_all) break;  // (my share is done)
    item0 = (unsigned long long)wave;
    istride = WPB;
  }
  const unsigned long long n_items = (n_all + wpi - 1ull) / wpi;
  for (unsigned long long item = item0; item < n_items; item += istride) {
    // ---- my word (lane l: entry l of the run)
    bool have;
    uint32_t wid = 0;
    {
      const unsigned long long wi = item * (unsigned long long)wpi + (unsigned long long)lane;
      have = (unsigned int)lane < wpi && wi < n_all;
      wid = (uint32_t)wi;
      if (listed) {
        if (have) wid = f_list[wi];
      } else if (worklist && have) {
        wid = worklist[wi];
      }
    }
    unsigned long long meta = 0;
    uint32_t wfreq = 0;
    if (have) {
      meta = ws.wmeta[wid];
      wfreq = ws.wcnt[wid];
    }
    const uint32_t wlen = (uint32_t)(meta & 0xffffull);
    const unsigned long long woff = meta >> 16;
    const uint32_t incl = wave_incl_scan(wlen);
    int first = 0;
    while (first < 64) {
      // ---- the next words of the run that fit one tile: lanes first .. first + nw - 1
      const uint32_t before = first ? (uint32_t)__shfl((int)incl, first - 1) : 0u;
      const bool in = lane >= first && have && incl - before <= (uint32_t)SLOT;
      const unsigned long long inm = __ballot(in);
      const int nw = __popcll(inm);
      if (!nw) break;
      const int n = (int)((uint32_t)__shfl((int)incl, first + nw - 1) - before);
      const uint32_t my_start = incl - wlen - before;
      wave_sync();  // (the tile state of the words before these is no longer needed)
      if (lane < SLOT / 64) X.lin[lane] = 0ull;
      wave_sync();
      if (in && wlen) atomicOr(&X.lin[my_start >> 6], 1ull << (my_start & 63u));
      wave_sync();
      if (n == 0) { first += nw; continue; }
      // position-major gather: lane l takes positions 64 c + l; the word of a position from the start bits, its address from the
      // word's lane; all loads of the tile in flight together
      uint32_t v[SLOT / 64];
      {
        uint32_t cb = 0;
#pragma unroll
        for (int c = 0; c < SLOT / 64; c++) {
          v[c] = 0;
          if (c * 64 >= n) continue;  // (uniform)
          const unsigned long long m = uni64(X.lin[c]);
          const int p = c * 64 + lane;
          // (words of length 0 -- none exist: every word keeps its first token -- would break the rank below)
          const uint32_t rank = cb + lanes_below(m) + (lane_bit(m) ? 1u : 0u);  // words that start at or before p
          // the rank-th word with tokens: ranks count non-empty words only, and they are exactly the lanes of the run (wlen >= 1)
          const int src = first + (int)rank - 1;
          const uint32_t o_lo = (uint32_t)__shfl((int)(uint32_t)woff, src), o_hi = (uint32_t)__shfl((int)(uint32_t)(woff >> 32), src);
          const uint32_t st0 = (uint32_t)__shfl((int)my_start, src);
          if (p < n) v[c] = ws.tok[(((unsigned long long)o_hi << 32) | o_lo) + (unsigned long long)((uint32_t)p - st0)];
          cb += (uint32_t)__popcll(m);
        }
      }
#pragma unroll
      for (int c = 0; c < SLOT / 64; c++) W.tk[c * 64 + lane] = v[c];
      wave_sync();
      K4_MARK(9);  // (PROF=2: worklist -> word headers -> tokens in LDS)
      uint4 r[SLOT / 256];
#pragma unroll
      for (int j = 0; j < SLOT / 256; j++) r[j] = reinterpret_cast<const uint4 *>(W.tk)[lane + 64 * j];
      WReg<SLOT> wq{};
      wq.v[0] = (uint32_t)__shfl((int)wfreq, (first + lane) & 63);  // lane i: frequency of word i of the tile
      uint32_t my_cnt = 0, my_site = 0;
      const int site_state = reg_find_sites<SLOT, LDSR>(W, r, n, A.flagbits, self_x, rtab, my_cnt, my_site);
      S.scanned += (unsigned long long)n;
      K4_MARK(0);
      if (site_state) {
        K4_COUNT(8);
        stage_ws_masks<SLOT>(W, r, n);
        if (lane == 0) {
          W.tk[n] = TOK_WS;
          W.tk[n + 1] = TOK_WS;
          W.tk[n + 2] = TOK_WS;
        }
        wave_sync();
        process_tile<SLOT, true, LDSR, true>(W, A, ts, pt, db, rtab, self_x, self_z, z_base, 0u, n, 0u, wq, S, (site_state & 2) != 0, false, &dout);
        wave_sync();
        const int nsites = (int)(uni(W.sctl[0]) & 0xffffu);
        if (nsites) {
          // ---- compact the tile where it is (LDS); remember which of the new positions hold a new token
          const int nchunks = (n + 63) >> 6;
          const int fsc = (int)(uni(W.sctl[1]) >> 6);
          if (lane < SLOT / 64) X.newsite[lane] = 0ull;
          wave_sync();
          uint32_t abase = (uint32_t)fsc * 64u;
          unsigned long long sm_prev = 0ull;
          for (int c = fsc; c < nchunks; c++) {
            const int p = c * 64 + lane;
            const unsigned long long smc = uni64(W.sitemask[c]);
            const int left = n - c * 64;
            const unsigned long long am = (left >= 64 ? ~0ull : (1ull << left) - 1ull) & ~((smc << 1) | (sm_prev >> 63));
            const bool surv = lane_bit(am), site = lane_bit(smc);
            const uint32_t np = abase + lanes_below(am);
            uint32_t val = 0;
            if (surv) {
              const uint32_t t0 = W.tk[p];
              val = site ? ((z_base + (uint32_t)W.ridx[p]) | (t0 & TOK_WS)) : (t0 & ~(L_ISX | L_ISY));
            }
            wave_sync();  // (every lane has read its token before any lane overwrites one: np <= p)
            if (surv) {
              W.tk[np] = val;
              if (site) atomicOr(&X.newsite[np >> 6], 1ull << (np & 63u));
            }
            abase += (uint32_t)__popcll(am);
            sm_prev = smc;
          }
          const int n2 = (int)abase;
          wave_sync();
          // ---- where the words start now (every word keeps its first token, so word i of the tile is still the i-th start)
          {
            uint32_t cb = 0;
            for (int c = 0; c < ((n2 + 63) >> 6); c++) {
              const int q = c * 64 + lane;
              const bool wsb = q < n2 && (W.tk[q] & TOK_WS);
              const unsigned long long m = __ballot(wsb);
              if (wsb) X.wnew[cb + (uint32_t)__popcll(m & lanemask_lt())] = (uint32_t)q;
              cb += (uint32_t)__popcll(m);
            }
            if (lane == 0) X.wnew[nw] = (uint32_t)n2;
          }
          wave_sync();
          // my word's new length (lane first + i: word i)
          uint32_t newlen = wlen;
          if (in) newlen = X.wnew[lane - first + 1] - X.wnew[lane - first];
          const bool changed = in && newlen != wlen;
          // ---- tokens of the changed words back to their slots; records of the new instances
          {
            uint32_t cb = 0;
            for (int c = 0; c < ((n2 + 63) >> 6); c++) {
              const int q = c * 64 + lane;
              const uint32_t tq = q < n2 ? W.tk[q] : 0u;
              const bool wsb = q < n2 && (tq & TOK_WS);
              const unsigned long long m = __ballot(wsb);
              const uint32_t wi = cb + (uint32_t)__popcll(m & lanemask_lt()) + (wsb ? 1u : 0u) - 1u;  // my word of the tile
              const int src = (first + (int)wi) & 63;
              const uint32_t o_lo = (uint32_t)__shfl((int)(uint32_t)woff, src), o_hi = (uint32_t)__shfl((int)(uint32_t)(woff >> 32), src);
              const bool ch = __shfl((int)changed, src) != 0;
              const uint32_t word_id = (uint32_t)__shfl((int)wid, src);
              if (q < n2 && ch) ws.tok[(((unsigned long long)o_hi << 32) | o_lo) + (unsigned long long)((uint32_t)q - X.wnew[wi])] = tq;
              const bool isnew = q < n2 && ((X.newsite[c] >> lane) & 1ull);
              const unsigned long long nm = __ballot(isnew);
              if (nm) {
                unsigned int b0 = 0;
                const int fl = __ffsll((long long)nm) - 1;
                if (lane == fl) b0 = atomicAdd(&rn, (unsigned int)__popcll(nm));
                b0 = (unsigned int)__shfl((int)b0, fl);
                const uint32_t z = isnew ? (tq & L_ID) : 0u;
                uint32_t lnb = NBR_NONE, rnb = NBR_NONE;
                bool direct = false;
                if (isnew) {
                  if (!(tq & TOK_WS)) lnb = W.tk[q - 1] & L_ID;
                  const uint32_t tr = q + 1 < n2 ? W.tk[q + 1] : TOK_WS;
                  if (!(tr & TOK_WS)) rnb = tr & L_ID;
                  const unsigned int pos = b0 + (unsigned int)__popcll(nm & lanemask_lt());
                  if (pos < drec_cap) {
                    my_irec[pos] = make_uint4(z - z_base, word_id, lnb, rnb);
                  } else {
                    direct = true;
                  }
                }
                // the buffer is full (a round with many sites per workgroup): straight to the lists, one bump of a token's fill count for
                // all the lanes that hold an instance of it (a handful of rules with thousands of sites each: one address per rule)
                unsigned long long dm = __ballot(direct);
                while (dm) {
                  const int ld = __ffsll((long long)dm) - 1;
                  const uint32_t z0 = (uint32_t)__shfl((int)z, ld);
                  const unsigned long long same = __ballot(direct && z == z0);
                  uint32_t at0 = 0;
                  if (lane == ld) at0 = atomicAdd(&tl.fill[z0], (uint32_t)__popcll(same));
                  at0 = (uint32_t)__shfl((int)at0, ld);
                  if (direct && z == z0) {
                    const uint32_t at = at0 + (uint32_t)__popcll(same & lanemask_lt());
                    if (at < (FUSED ? f_lcap[z0 - z_base] : tl.cap[z0])) {
                      const unsigned long long o = (FUSED ? f_lbase[z0 - z_base] : tl.base[z0]) + at;
                      tl.rec_word[o] = word_id;
                      tl.rec_l[o] = lnb;
                      tl.rec_r[o] = rnb;
                    } else {
                      __hip_atomic_store(tl.broken, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                  }
                  dm &= ~same;
                }
              }
              cb += (uint32_t)__popcll(m);
            }
          }
          if (changed) {
            for (uint32_t i = newlen; i < wlen; i++) ws.tok[woff + i] = TOK_HOLE;
            ws.wmeta[wid] = (woff << 16) | (unsigned long long)newlen;
          }
          S.touched++;
          S.touched_tok += (unsigned long long)n;
          K4_MARK(6);  // (compaction, write-back, records)
        }
      }
      first += nw;
    }
  }
  if (!listed) break;
  __syncthreads();  // (every wave is done with the list)
  if (threadIdx.x == 0) f_n = 0;
  __syncthreads();
  }
  {
    S.sites = wave_sum_u64(S.sites);
    if (lane == 0) {
      if (S.sites) atomicAdd(&A.st[0], S.sites);
      if (S.touched) atomicAdd(&A.st[1], S.touched);
      if (S.touched_tok) atomicAdd(&A.st[3], S.touched_tok);
    }
  }
#ifdef YTTM_K4_PROF
  K4_MARK(7);
  __syncthreads();
  K4_MARK(11);
#endif
  __syncthreads();
  for (int sl = (int)threadIdx.x; sl < AGG_SLOTS; sl += WPB * 64) {  // the aggregator's sums are records too
    const unsigned long long k = A.key[sl];
    const long long v = k != PT_EMPTY ? (long long)A.val[sl] : 0;
    const unsigned long long ks[1] = {k};
    const long long ds[1] = {v};
    const bool ms[1] = {v != 0};
    rec_emit_batch<1>(dout, pt, db, ks, ds, ms, &A.new_keys);
  }
#ifdef YTTM_K4_PROF
  __syncthreads();
  K4_MARK(12);
#endif
  // ---- the workgroup's records: one bump of a token's fill count per workgroup (the tile buffers are free: counts per rule live there)
  {
    static_assert(sizeof(WL) >= WGATHER_MAXK * sizeof(uint32_t), "per-rule counters of the record flush");
    uint32_t *rcnt = reinterpret_cast<uint32_t *>(&WL[0]);
    __syncthreads();
    const unsigned int nrec = rn < drec_cap ? rn : drec_cap;
    if (nrec) {  // (uniform)
      const uint32_t kk = k_rules < WGATHER_MAXK ? k_rules : WGATHER_MAXK;
      for (uint32_t j = threadIdx.x; j < kk; j += WPB * 64) rcnt[j] = 0;
      __syncthreads();
      for (unsigned int i = threadIdx.x; i < nrec; i += WPB * 64) {  // rank of the record among the workgroup's records of its token
        const uint32_t zr = my_irec[i].x & 0xfffu;
        my_irec[i].x = zr | (atomicAdd(&rcnt[zr], 1u) << 12);
      }
      __syncthreads();
      for (uint32_t j = threadIdx.x; j < kk; j += WPB * 64)
        if (rcnt[j]) rcnt[j] = atomicAdd(&tl.fill[z_base + j], rcnt[j]);
      __syncthreads();
      for (unsigned int i = threadIdx.x; i < nrec; i += WPB * 64) {
        const uint4 rec = my_irec[i];
        const uint32_t zr = rec.x & 0xfffu, z = z_base + zr;
        const uint32_t at = rcnt[zr] + (rec.x >> 12);
        if (at < (FUSED ? f_lcap[zr] : tl.cap[z])) {
          const unsigned long long o = (FUSED ? f_lbase[zr] : tl.base[z]) + at;
          tl.rec_word[o] = rec.y;
          tl.rec_l[o] = rec.z;
          tl.rec_r[o] = rec.w;
        } else {
          __hip_atomic_store(tl.broken, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
    }
  }
  __syncthreads();
#ifdef YTTM_K4_PROF
  K4_MARK(10);  // (record flush)
  if (lane == 0)
    for (int i = 0; i < 16; i++)
      if (S.pt[i]) atomicAdd(&stats[8 + i], S.pt[i]);
  if (threadIdx.x == 0 && A.miss_n) {
    atomicAdd(&stats[8 + 14], A.miss_n);
    atomicAdd(&stats[8 + 15], A.miss_cyc);
  }
#endif
  if (!inline_apply) {  // the usual way: k_delta_apply takes the records from here (and runs the round's candidate scan)
    if (threadIdx.x == 0) {
      blk_add(stats, 4, A.new_keys);
      for (int i = 0; i < 4; i++) blk_add(stats, i, A.st[i]);
      drec_n[blockIdx.x] = dn < drec_cap ? dn : drec_cap;
    }
    return;
  }
  // A small round (a few thousand sites): this workgroup puts its own records into the pair table -- they are in L2, the wait is short
  // when the chip is nearly idle -- and the round's candidate scan rides in this launch: one kernel less on the round's critical path.
  {
    const unsigned int nd = dn < drec_cap ? dn : drec_cap;
    for (unsigned int i = threadIdx.x; i < nd; i += WPB * 64) {
      const DeltaRec rec = dout.recs[i];
      global_emit(pt, db, rec.key, rec.delta, &A.new_keys);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      blk_add(stats, 4, A.new_keys);
      for (int i = 0; i < 4; i++) blk_add(stats, i, A.st[i]);
    }
  }
  if (sa.on != 3u) {  // the candidate scan, by the last workgroup to get here (as in k_tiles); it also leaves the worklist's length at zero
    __shared__ unsigned int is_last;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0) is_last = atomicAdd(sa.done_ctr, 1u) == gridDim.x - 1;
    __syncthreads();
    if (is_last) {
      __threadfence();
      if (threadIdx.x <= WL_PARTS + 1) const_cast<unsigned int *>(work_n)[threadIdx.x] = 0;  // (every workgroup has read it)
      if (sa.on == 2u) {  // multi-GPU: the scan comes behind the exchange (k_fold_list)
        if (threadIdx.x == 0) *sa.done_ctr = 0;
      } else {
        const RuleProbe zprobe{LDSR ? rkeys : nullptr, LDSR ? nullptr : rules, rule_mask};
        scan_top<WPB * 64>(pt, sa, stats, zprobe, self_x != 0xffffffffu ? pair_key(self_x, self_x) : PT_EMPTY, reinterpret_cast<unsigned int *>(&WL[0]), nullptr);
      }
    }
  }
}

// The records of a word-mode round -> the pair table: region r (k_words' workgroup r) is shared by `parts` workgroups, one thread per
// record.  The round's candidate scan rides in this launch (the last workgroup to finish, as in k_tiles).
constexpr int DAPPLY_NT = 256;
__global__ __launch_bounds__(DAPPLY_NT) void k_delta_apply(PairTable pt, DeltaBuf db, const DeltaRec *__restrict__ drec, unsigned int drec_cap,
                                                          const unsigned int *__restrict__ drec_n, unsigned int parts,
                                                          unsigned int *__restrict__ work_n /* the round's worklist is done with: left at zero for the next gather */,
                                                          unsigned long long *__restrict__ stats, const RuleSlot *__restrict__ zrules, unsigned int zmask,
                                                          unsigned long long zself, BatchArgs zba, ScanArgs sa) {
  __shared__ unsigned int new_keys, is_last;
  __shared__ unsigned long long zkeys[FILTER_LDS_KEYS];
  __shared__ unsigned int scratch[CAND_BINS + 160];
  if (threadIdx.x == 0) new_keys = 0;
  if (blockIdx.x == 0 && threadIdx.x <= WL_PARTS + 1) work_n[threadIdx.x] = 0;
  __syncthreads();
  const unsigned int r = blockIdx.x / parts, p = blockIdx.x % parts;
  const unsigned int n = drec_n[r];
  const DeltaRec *reg = drec + (size_t)r * drec_cap;
  for (unsigned int i = p * DAPPLY_NT + threadIdx.x; i < n; i += parts * DAPPLY_NT) {
    const DeltaRec rec = reg[i];
    global_emit(pt, db, rec.key, rec.delta, &new_keys);
  }
  __syncthreads();
  if (threadIdx.x == 0) blk_add(stats, 4, new_keys);
  if (!sa.on) return;
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  if (threadIdx.x == 0) is_last = atomicAdd(sa.done_ctr, 1u) == gridDim.x - 1;
  __syncthreads();
  if (!is_last) return;
  __threadfence();
  if (sa.on == 2u) {  // multi-GPU: the scan comes behind the exchange (k_fold_list)
    if (threadIdx.x == 0) *sa.done_ctr = 0;
    return;
  }
  bool zkeys_in_lds = zrules && zmask < FILTER_LDS_KEYS;  // (the finished batch's pairs, to be zeroed: as in k_top_scan)
  if (zba.k) {
    zmask = 4 * BATCH_ARGS_MAX - 1;
    zkeys_in_lds = true;
    for (unsigned int sl = threadIdx.x; sl <= zmask; sl += DAPPLY_NT) zkeys[sl] = PT_EMPTY;
    __syncthreads();
    if (threadIdx.x < zba.k && zba.xy[2 * threadIdx.x] != zba.xy[2 * threadIdx.x + 1]) {  // (BATCH_ARGS_MAX <= the block size)
      const unsigned long long key = pair_key(zba.xy[2 * threadIdx.x], zba.xy[2 * threadIdx.x + 1]);
      unsigned int h = pair_hash32(key) & zmask;
      while (atomicCAS(&zkeys[h], PT_EMPTY, key) != PT_EMPTY) h = (h + 1) & zmask;
    }
  } else if (zkeys_in_lds) {
    for (unsigned int sl = threadIdx.x; sl <= zmask; sl += DAPPLY_NT) zkeys[sl] = zrules[sl].key;
  }
  __syncthreads();
  const RuleProbe zprobe{zkeys_in_lds ? zkeys : nullptr, zrules, zmask};
  scan_top<DAPPLY_NT>(pt, sa, stats, zprobe, zself, scratch, nullptr);
}

// ------------------------------------------------------------------------------------------------- tile repack
// Merges shrink tiles in place; once the average fill is low the fixed per-tile cost dominates a pass, so the live
// words are re-dealt into fresh tiles (same word order, so wcnt stays valid).  off[t] = live tokens before tile t.
template <int SLOT>
__global__ __launch_bounds__(BLOCK) void k_repack_mark(TileSet ts, const unsigned long long *__restrict__ off, unsigned int nom,
                                                       unsigned long long *__restrict__ gstart, uint32_t *__restrict__ gword0) {
  const int lane = lane_id();
  const uint32_t stride = gridDim.x * NWAVES;
  for (uint32_t t = uni(blockIdx.x * NWAVES + (uint32_t)(threadIdx.x >> 6)); t < ts.n_tiles; t += stride) {
    const int n = (int)ts.tile_len[t];
    const uint32_t *src = ts.tok + (size_t)t * SLOT;
    uint32_t wbase = 0;
    unsigned long long g_carry = ~0ull;  // new tile of the last word start seen in the earlier chunks of this tile
    const unsigned long long off_t = off[t];
    for (int c = 0; c < ((n + 63) >> 6); c++) {
      const int p = c * 64 + lane;
      const bool ws = p < n && (src[p] & TOK_WS);
      const unsigned long long m = __ballot(ws);
      const unsigned long long woff = off_t + (unsigned long long)p;
      const unsigned long long g = woff / nom;
      // Only the first word of a new tile decides its start (the minimum): a word whose predecessor in this tile goes to the same
      // new tile needs no atomic -- 2 per new tile and old tile instead of 2 per word (3.2e7 at 1 GB, 1.6 ms per repack).
      const unsigned long long before = m & lanemask_lt();
      const int src_lane = before ? 63 - __clzll((long long)before) : 0;
      const unsigned long long g_lane = ((unsigned long long)(uint32_t)__shfl((int)(g >> 32), src_lane) << 32) | (uint32_t)__shfl((int)(uint32_t)g, src_lane);
      const unsigned long long g_prev = before ? g_lane : g_carry;
      if (ws && g_prev != g) {
        atomicMin(&gstart[g], woff);
        atomicMin(&gword0[g], ts.tile_word0[t] + wbase + (uint32_t)__popcll(before));
      }
      if (m) {
        const int last = 63 - __clzll((long long)m);
        g_carry = ((unsigned long long)(uint32_t)__shfl((int)(g >> 32), last) << 32) | (uint32_t)__shfl((int)(uint32_t)g, last);
      }
      wbase += (uint32_t)__popcll(m);
    }
  }
}

template <int SLOT>
__global__ __launch_bounds__(BLOCK) void k_repack_copy(TileSet ts, const unsigned long long *__restrict__ off, unsigned int nom,
                                                       const unsigned long long *__restrict__ gstart, uint32_t *__restrict__ new_tok) {
  const int lane = lane_id();
  const uint32_t stride = gridDim.x * NWAVES;
  for (uint32_t t = uni(blockIdx.x * NWAVES + (uint32_t)(threadIdx.x >> 6)); t < ts.n_tiles; t += stride) {
    const int n = (int)ts.tile_len[t];
    const uint32_t *src = ts.tok + (size_t)t * SLOT;
    int carry_ws = 0;  // position of the last word start seen in earlier chunks (a tile starts with a word start)
    for (int c = 0; c < ((n + 63) >> 6); c++) {
      const int p = c * 64 + lane;
      const uint32_t tk = p < n ? src[p] : 0;
      const bool ws = p < n && (tk & TOK_WS);
      const unsigned long long m = __ballot(ws);
      int wsp = carry_ws;
      const unsigned long long le = m & ((2ull << lane) - 1ull);
      if (le) wsp = c * 64 + 63 - __clzll((long long)le);
      if (m) carry_ws = c * 64 + 63 - __clzll((long long)m);
      if (p < n) {
        const unsigned long long g = (off[t] + (unsigned long long)wsp) / nom;
        new_tok[g * SLOT + (off[t] + (unsigned long long)p - gstart[g])] = tk;
      }
    }
  }
}

__global__ __launch_bounds__(BLOCK) void k_repack_len(const unsigned long long *__restrict__ gstart, unsigned int n_new,
                                                      unsigned long long total, uint32_t *__restrict__ new_len) {
  unsigned int g = blockIdx.x * BLOCK + threadIdx.x;
  if (g >= n_new) return;
  const unsigned long long s0 = gstart[g];
  if (s0 == ~0ull) { new_len[g] = 0; return; }
  unsigned long long e = total;
  if (g + 1 < n_new && gstart[g + 1] != ~0ull) e = gstart[g + 1];
  new_len[g] = (uint32_t)(e - s0);
}

// ------------------------------------------------------------------------------------------------- launchers
void launch_repack(int cls, const TileSet &ts, const unsigned long long *off, unsigned int nom, unsigned long long total,
                   unsigned long long *gstart, unsigned int n_new, uint32_t *new_tok, uint32_t *new_len, uint32_t *new_word0, hipStream_t st) {
  unsigned int g = (ts.n_tiles + NWAVES - 1) / NWAVES;
  if (g > 256 * 8) g = 256 * 8;
  if (!g) g = 1;
  if (cls == 0) {
    hipLaunchKernelGGL((k_repack_mark<TILE_SLOT_A>), dim3(g), dim3(BLOCK), 0, st, ts, off, nom, gstart, new_word0);
    hipLaunchKernelGGL((k_repack_copy<TILE_SLOT_A>), dim3(g), dim3(BLOCK), 0, st, ts, off, nom, gstart, new_tok);
  } else {
    hipLaunchKernelGGL((k_repack_mark<TILE_SLOT_B>), dim3(g), dim3(BLOCK), 0, st, ts, off, nom, gstart, new_word0);
    hipLaunchKernelGGL((k_repack_copy<TILE_SLOT_B>), dim3(g), dim3(BLOCK), 0, st, ts, off, nom, gstart, new_tok);
  }
  hipLaunchKernelGGL(k_repack_len, dim3((n_new + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st, gstart, n_new, total, new_len);
}

// class-A apply kernel: waves per workgroup x workgroups per CU (6 x 4 = 6 waves per SIMD: <= 80 VGPRs, <= 40 KB LDS per workgroup)
constexpr int APPLY_WPB = 8, APPLY_BPC = 3;
// ------------------------------------------------------------------------------------------------- K3, small alphabets
// The pair count of class-A tiles when the alphabet has at most K3D_MAX_IDS symbols (any corpus of one script; 'abcd ': 5): no
// LDS staging, no hash.  A wave holds its tile in registers position-major (lane l: tokens 64 c + l), one tile ahead; the
// right neighbour comes by a lane shift, the word of a position from the ballot of the word-start bits, its frequency from
// the tile's window of word counts (registers, one ds_bpermute), and a run of equal tokens contributes floor(L/2)
// (bpe.cpp:461-475) through its pairs at an even offset from the run's start -- picked with carry arithmetic on the ballot of
// "equal to the right neighbour" (scalar unit) instead of a walk along the run.  Every adjacency is then ONE ds_add_u64 into
// a dense n x n table, kept in as many lane-indexed copies as fit (the 64 lanes of an instruction hit ~25 addresses on
// 'abcd ').  Measured on the 1 GB 'abcd ' table (546 k tiles, 252 M tokens, 16 M words): the general kernel (k_tiles<.., false, ..>)
// issues 610 VALU + 420 SALU + 125 LDS instructions per tile and is bound by them (0.87 ms, 16 % of HBM by the algorithmic bytes);
// this one 222 + 220 + 17 and takes 0.43 ms (33 %), of which 0.27 ms are its loads alone (the same loop with the arithmetic
// taken out; tools/micro/stream_bw reads the same pattern from a hot array in 0.17 ms).
constexpr uint32_t K3D_MAX_IDS = 64;
// the table's row stride: the alphabet size rounded up to a power of two (index = a << sh | b, no multiply)
__host__ __device__ inline uint32_t k3d_shift(uint32_t n) {
  uint32_t sh = 0;
  while ((1u << sh) < n) sh++;
  return sh;
}
// COUNTERS: 2048 (16 KB: LDS does not limit the waves per CU) for up to 32 symbols, 4096 beyond
__host__ __device__ inline uint32_t k3d_copies(uint32_t n, uint32_t counters) {
  uint32_t c = counters >> (2 * k3d_shift(n));
  if (c > 64u) c = 64u;
  uint32_t p = 1;
  while (2 * p <= c) p *= 2;
  return p;
}
template <int SLOT, uint32_t COUNTERS>
__global__ __launch_bounds__(256) void k_pair_count_dense(TileSet ts, PairTable pt, DeltaBuf db, uint32_t id_min, uint32_t n_ids) {
  constexpr int NC = SLOT / 64, NW = WReg<SLOT>::N;
  __shared__ unsigned long long dense[COUNTERS];  // [1 << 2 sh][copies]
  __shared__ uint32_t wwin[4][64 * NW];           // per wave: the word counts of its tile
  __shared__ unsigned int new_keys;
  const uint32_t sh = k3d_shift(n_ids), nn = 1u << (2 * sh), copies = k3d_copies(n_ids, COUNTERS);
  for (unsigned int i = threadIdx.x; i < nn * copies; i += 256) dense[i] = 0;
  if (threadIdx.x == 0) new_keys = 0;
  __syncthreads();
  const int lane = lane_id();
  // (the copies of one counter are neighbours -- lanes adding to the same pair hit different banks, and two lanes share a bank only
  // through lane and lane + 32.  Index of pair (a, b) for this lane: base + ((a << sh) + b) * copies with the raw ids; the base takes
  // id_min off both.)
  const uint32_t lc = k3d_shift(copies);
  const uint32_t base = ((uint32_t)lane & (copies - 1)) - (((id_min << sh) + id_min) << lc);
  uint32_t *lw = wwin[threadIdx.x >> 6];
  const uint32_t n_waves = gridDim.x * 4u;
  uint32_t t = uni(blockIdx.x * 4u + (threadIdx.x >> 6));  // (uniform: lengths and first words come by scalar loads)
  uint32_t r[NC], rn[NC];
  WReg<SLOT> w, wn;
  // (length and first word of a tile are read two tiles ahead, so that the loads of the tile itself never wait for them)
  auto head = [&](uint32_t tile, int &n, uint32_t &w0) {
    n = tile < ts.n_tiles ? (int)ts.tile_len[tile] : 0;
    w0 = tile < ts.n_tiles ? ts.tile_word0[tile] : 0u;
  };
  auto fetch = [&](uint32_t (&dst)[NC], WReg<SLOT> &wd, uint32_t tile, int n, uint32_t w0) {
    const uint32_t *src = ts.tok + (size_t)tile * SLOT;
#pragma unroll
    for (int c = 0; c < NC; c++) {  // (the whole slot is readable; behind the end of the tile: "a word starts here")
      const uint32_t v = src[64 * c + lane];
      dst[c] = 64 * c + lane < n ? v : TOK_WS;
    }
    wreg_load<SLOT>(wd, ts.wcnt, w0);
  };
  int n1, n2;
  uint32_t w01, w02;
  head(t, n1, w01);
  head(t + n_waves, n2, w02);
  if (t < ts.n_tiles) fetch(r, w, t, n1, w01);
  for (; t < ts.n_tiles; t += n_waves) {
    n1 = n2;
    w01 = w02;
    head(t + 2 * n_waves, n2, w02);
    if (t + n_waves < ts.n_tiles) fetch(rn, wn, t + n_waves, n1, w01);
    wave_sync();  // (the previous tile's reads of the window are done: DS operations of a wave execute in order)
#pragma unroll
    for (int i = 0; i < NW; i++) lw[lane + 64 * i] = w.v[i];
    wave_sync();
    uint32_t wbase = 0xffffffffu;  // word starts so far, minus one
    bool cont = false, cont_even = false;  // the run of equal tokens at the end of the previous chunk goes on / its next pair is at an even offset
#pragma unroll
    for (int c = 0; c < NC; c++) {
      const uint32_t t0 = r[c];
      uint32_t t1 = from_lane_right(t0);
      const uint32_t first_next = c + 1 < NC ? from_lane0(r[c + 1 < NC ? c + 1 : c]) : TOK_WS;
      if (lane == 63) t1 = first_next;
      const unsigned long long m_ws = ballot_b((int)t0 < 0);
      const uint32_t k = wbase + lanes_below(m_ws) + (t0 >> 31);  // word of this position (word starts <= p, minus one)
      wbase += (uint32_t)__popcll(m_ws);
      const bool adj = (int)t1 >= 0;  // the right neighbour belongs to the same word
      const uint32_t a = t0 & TOK_MASK;
      const bool eq = a == t1;  // (t1 without the word-start bit is its id)
      const unsigned long long E = ballot_b(adj && eq);
      // runs of E = pairs inside a run of equal tokens.  S: the runs' first bits; those at an even position (or going on from the
      // previous chunk at an even offset) make their whole run carry out in E + S_e; such runs take their even positions, the
      // others their odd ones.
      const unsigned long long S = E & ~((E << 1) | (cont ? 1ull : 0ull));
      const unsigned long long S_e = (S & 0x5555555555555555ull) | (cont_even ? (E & 1ull) : 0ull);
      const unsigned long long D = (E + S_e) ^ E;
      const unsigned long long sel = (D & E & 0x5555555555555555ull) | (~D & E & 0xaaaaaaaaaaaaaaaaull);
      cont = (E >> 63) != 0ull;
      cont_even = cont && !(sel >> 63);
      if (adj && (!eq || lane_bit(sel))) atomicAdd(&dense[base + (((a << sh) + t1) << lc)], (unsigned long long)lw[k]);
    }
    if (t + n_waves < ts.n_tiles) {
#pragma unroll
      for (int c = 0; c < NC; c++) r[c] = rn[c];
      w = wn;
    }
  }
  __syncthreads();
  for (unsigned int i = threadIdx.x; i < nn; i += 256) {
    const uint32_t x = i >> sh, y = i & ((1u << sh) - 1u);
    unsigned long long v = 0;
    for (uint32_t c = 0; c < copies; c++) v += dense[(i << lc) + c];
    if (v) global_emit(pt, db, pair_key(id_min + x, id_min + y), (long long)v, &new_keys);
  }
  __syncthreads();
  if (threadIdx.x == 0 && new_keys) atomicAdd(pt.n_keys, new_keys);
}

void pm_bloom_host(uint32_t *bloom, const uint32_t *xyz, uint32_t k) {  // (a batch that does not travel in the kernel arguments: its pair filter, built by the host)
  for (int i = 0; i < PM_BLOOM_WORDS; i++) bloom[i] = 0;
  for (uint32_t j = 0; j < k; j++) {
    const uint32_t x = xyz[3 * j], y = xyz[3 * j + 1];
    if (x == y) continue;
    const uint32_t h = pm_hash(x, y);
    bloom[pm_word(h)] |= pm_bits(h);
  }
}

static inline unsigned int tile_grid(unsigned int n_tiles, unsigned int wpb, unsigned int blocks_per_cu) {
  unsigned int need = (n_tiles + wpb - 1) / wpb;
  unsigned int g = 256u * blocks_per_cu;
  if (g > need) g = need;
  return g ? g : 1u;
}

void launch_pair_count(int cls, const TileSet &ts, const PairTable &pt, const DeltaBuf &db, uint32_t id_min, uint32_t n_ids, hipStream_t st) {
  if (!ts.n_tiles) return;
  unsigned int bpc = 4;
  if (const char *e = getenv("YTTM_K3_BPC")) bpc = (unsigned int)atoi(e);  // (tuning aids; one launch per training)
  const bool general = getenv("YTTM_K3_GENERAL") != nullptr;
  if (cls == 0 && n_ids && n_ids <= K3D_MAX_IDS && !general) {
    if (n_ids <= 32u)
      hipLaunchKernelGGL((k_pair_count_dense<TILE_SLOT_A, 2048u>), dim3(tile_grid(ts.n_tiles, 4, bpc)), dim3(256), 0, st, ts, pt, db, id_min, n_ids);
    else
      hipLaunchKernelGGL((k_pair_count_dense<TILE_SLOT_A, 4096u>), dim3(tile_grid(ts.n_tiles, 4, bpc)), dim3(256), 0, st, ts, pt, db, id_min, n_ids);
    return;
  }
  if (n_ids > 32) n_ids = 0;  // (k_tiles' dense table holds 32 x 32 counts; larger alphabets go through the LDS hash)
  if (cls == 0)
    hipLaunchKernelGGL((k_tiles<TILE_SLOT_A, 4, false, false>), dim3(tile_grid(ts.n_tiles, 4, bpc)), dim3(256), 0, st, ts, pt, db,
                       (const RuleSlot *)nullptr, 0u, (const uint8_t *)nullptr, (const uint32_t *)nullptr, 0xffffffffu, id_min, n_ids,
                       (const uint32_t *)nullptr, (const unsigned int *)nullptr, (unsigned long long *)nullptr, BatchArgs{}, ScanArgs{});
  else
    hipLaunchKernelGGL((k_tiles<TILE_SLOT_B, 1, false, false>), dim3(tile_grid(ts.n_tiles, 1, 4)), dim3(64), 0, st, ts, pt, db,
                       (const RuleSlot *)nullptr, 0u, (const uint8_t *)nullptr, (const uint32_t *)nullptr, 0xffffffffu, id_min, n_ids,
                       (const uint32_t *)nullptr, (const unsigned int *)nullptr, (unsigned long long *)nullptr, BatchArgs{}, ScanArgs{});
}
void launch_merge_apply(int cls, const TileSet &ts, const PairTable &pt, const DeltaBuf &db, const RuleSlot *rules, unsigned int rule_mask,
                        uint32_t self_x, uint32_t self_z, uint32_t z_base, unsigned long long *stats, const BatchArgs *ba, const ScanArgs *scan,
                        const uint32_t *bloom_g, hipStream_t st) {
  if (!ts.n_tiles) return;
  const BatchArgs bargs = ba ? *ba : BatchArgs{};
  const ScanArgs sargs = scan ? *scan : ScanArgs{};  // (the caller hands the scan to the round's LAST launch)
  // One pass: the apply kernel takes every tile and dismisses the clean ones itself, in registers (a separate filter pass with a
  // worklist of dirty tiles was measured slower at every share of dirty tiles and is gone, like the worklists of tiles from the pair
  // index: class A leaves the tiles for word mode before either could pay).
  // class-A grid: APPLY_BPC workgroups per CU when there are tiles for all of them; a small tile set (natural-language corpora:
  // a few thousand tiles) gets fewer workgroups with several tiles per wave -- every workgroup costs a prologue (44 KB of LDS set-up)
  // and a serialised ticket at the end (~11 ns each), which a round of ~15 us notices.  YTTM_APPLY_GRID overrides (tuning hook).
  unsigned int grid_a = tile_grid(ts.n_tiles, APPLY_WPB, APPLY_BPC);
  {
    static const char *g_env = getenv("YTTM_APPLY_GRID");
    const unsigned int small = g_env ? (unsigned int)atoi(g_env) : 256u;
    if (ts.n_tiles <= 16384 && small && grid_a > small) grid_a = small;
  }
  const uint8_t *no_flags = nullptr;
  const uint32_t *no_list = nullptr;
  const unsigned int *no_n = nullptr;
  if (cls == 0) {
    if (bargs.k && bargs.direct_v && rule_mask < APPLY_LDS_RULES)
      hipLaunchKernelGGL((k_tiles<TILE_SLOT_A, APPLY_WPB, true, true, true>), dim3(grid_a), dim3(64 * APPLY_WPB), 0, st, ts, pt, db, rules, rule_mask,
                         no_flags, bloom_g, self_x, self_z, z_base, no_list, no_n, stats, bargs, sargs);
    else if (rule_mask < APPLY_LDS_RULES)
      hipLaunchKernelGGL((k_tiles<TILE_SLOT_A, APPLY_WPB, true, true>), dim3(grid_a), dim3(64 * APPLY_WPB), 0, st, ts, pt, db, rules, rule_mask,
                         no_flags, bloom_g, self_x, self_z, z_base, no_list, no_n, stats, bargs, sargs);
    else
      hipLaunchKernelGGL((k_tiles<TILE_SLOT_A, APPLY_WPB, true, false>), dim3(grid_a), dim3(64 * APPLY_WPB), 0, st, ts, pt, db, rules, rule_mask,
                         no_flags, bloom_g, self_x, self_z, z_base, no_list, no_n, stats, bargs, sargs);
  } else {
    if (rule_mask < APPLY_LDS_RULES)
      hipLaunchKernelGGL((k_tiles<TILE_SLOT_B, 1, true, true>), dim3(tile_grid(ts.n_tiles, 1, 4)), dim3(64), 0, st, ts, pt, db, rules, rule_mask,
                         no_flags, bloom_g, self_x, self_z, z_base, no_list, no_n, stats, bargs, sargs);
    else
      hipLaunchKernelGGL((k_tiles<TILE_SLOT_B, 1, true, false>), dim3(tile_grid(ts.n_tiles, 1, 4)), dim3(64), 0, st, ts, pt, db, rules, rule_mask,
                         no_flags, bloom_g, self_x, self_z, z_base, no_list, no_n, stats, bargs, sargs);
  }
}
void launch_cand_scan(const PairTable &pt, unsigned long long tau_cnt, uint32_t tau_mx, CandRec *out, unsigned int cap,
                      unsigned int *n_out, unsigned long long *hist, hipStream_t st) {
  unsigned long long n_slots = pt.mask + 1;
  unsigned long long b = (n_slots + BLOCK - 1) / BLOCK;
  if (b > 256 * 16) b = 256 * 16;
  hipLaunchKernelGGL(k_cand_scan, dim3((unsigned int)b), dim3(BLOCK), 0, st, pt, tau_cnt, tau_mx, out, cap, n_out, hist);
}
void launch_hot_scan(const PairTable &pt, unsigned long long tau_cnt, uint32_t tau_mx, CandRec *out, unsigned int cap, unsigned int *n_out,
                     unsigned long long *hist, unsigned int *done_ctr, unsigned char *mailbox, unsigned int fast, uint32_t round_id,
                     unsigned long long *stats, const RuleSlot *zrules, unsigned int zmask, unsigned long long zself, unsigned int listed_hint,
                     const BatchArgs *zba, unsigned long long *xstat, hipStream_t st) {
  // one entry per thread; every workgroup costs ~11 ns of serialised ticket/total atomics at the end, so no more of them
  // than the list needs (the statistics rows need >= BLK_ROWS / 64 = 24)
  unsigned int g = (listed_hint + BLOCK - 1) / BLOCK;
  if (g < 32) g = 32;
  if (g > 256) g = 256;
  hipLaunchKernelGGL(k_hot_scan, dim3(g), dim3(BLOCK), 0, st, pt, tau_cnt, tau_mx, out, cap, n_out, hist, done_ctr, mailbox, fast, round_id,
                     stats, zrules, zmask, zself, zba ? *zba : BatchArgs{}, xstat);
}
void launch_dt_clean(const DeltaBuf &db, DeltaRec *other, unsigned int n_hint, unsigned long long *stats, uint32_t tiles_a, unsigned int *done_ctr, hipStream_t st) {
  unsigned int g = (n_hint + BLOCK - 1) / BLOCK;
  if (g < 4) g = 4;
  if (g > 1024) g = 1024;
  hipLaunchKernelGGL(k_dt_clean, dim3(g), dim3(BLOCK), 0, st, db, other, stats, tiles_a, done_ctr);
}
void launch_dt_init(DtSlot *slots, unsigned long long n, hipStream_t st) {
  unsigned long long b = (n + BLOCK - 1) / BLOCK;
  if (b > 256 * 16) b = 256 * 16;
  hipLaunchKernelGGL(k_dt_init, dim3((unsigned int)(b ? b : 1)), dim3(BLOCK), 0, st, slots, n);
}
void launch_pt_apply_blocks(const PairTable &pt, const DeltaRec *blocks, unsigned long long blk, int world, int rank, unsigned long long only_mask,
                            unsigned long long *xstat, unsigned long long *stats, hipStream_t st) {
  unsigned long long b = (blk + BLOCK - 1) / BLOCK;
  if (b > 256 * 4) b = 256 * 4;
  hipLaunchKernelGGL(k_pt_apply_blocks, dim3((unsigned int)b), dim3(BLOCK), 0, st, pt, blocks, blk, world, rank, only_mask, xstat, stats);
}
void launch_fold_list(const PairTable &pt, const DeltaRec *blocks, unsigned long long blk, int world, unsigned long long only_mask, const ScanArgs *scan,
                      unsigned long long *stats, const RuleSlot *zrules, unsigned int zmask, unsigned long long zself, const BatchArgs *zba,
                      unsigned long long *xstat, bool read_headers, hipStream_t st) {
  hipLaunchKernelGGL(k_fold_list, dim3(1), dim3(FOLD_NT), 0, st, pt, blocks, blk, world, only_mask, scan ? *scan : ScanArgs{}, stats, zrules, zmask, zself,
                     zba ? *zba : BatchArgs{}, xstat, read_headers ? 1 : 0);
}
void launch_top_scan(const PairTable &pt, const ScanArgs &sa, unsigned long long *stats, const RuleSlot *zrules, unsigned int zmask, unsigned long long zself,
                     const BatchArgs *zba, unsigned long long *xstat, hipStream_t st) {
  hipLaunchKernelGGL(k_top_scan, dim3(1), dim3(TOP_SCAN_NT), 0, st, pt, sa, stats, zrules, zmask, zself, zba ? *zba : BatchArgs{}, xstat);
}
void launch_top_rebuild(const PairTable &pt, unsigned int listed_hint, hipStream_t st) {
  unsigned int g = (listed_hint + BLOCK - 1) / BLOCK;
  if (g < 1) g = 1;
  if (g > 1024) g = 1024;
  hipLaunchKernelGGL(k_top_rebuild, dim3(g), dim3(BLOCK), 0, st, pt);
}
void launch_idx_seed(const PairTable &pt, const PairIndexArgs &a, unsigned int listed_hint, hipStream_t st) {
  const PairIndex ix{a.key, a.cnt, a.off, a.bloom, a.post, a.mask};
  unsigned int g = (listed_hint + BLOCK - 1) / BLOCK;
  if (g < 1) g = 1;
  if (g > 1024) g = 1024;
  hipLaunchKernelGGL(k_idx_seed, dim3(g), dim3(BLOCK), 0, st, pt, ix);
}
size_t idx_save_bytes() { return (size_t)512 * sizeof(IdxAgg); }
void launch_idx_stream(bool fill, const TileSet &ts, const PairIndexArgs &a, hipStream_t st, bool agg, void *save_) {
  IdxAgg *save = reinterpret_cast<IdxAgg *>(save_);
  if (!ts.n_tiles) return;
  const PairIndex ix{a.key, a.cnt, a.off, a.bloom, a.post, a.mask};
  unsigned int g = (ts.n_tiles + IDXA_NT / 64 - 1) / (IDXA_NT / 64);
  if (g > 512) g = 512;  // (two workgroups per CU: 72 KB of LDS each; count and fill pass MUST use the same grid -- a workgroup's shard and tiles)
  if (fill) hipLaunchKernelGGL((k_idx_stream<TILE_SLOT_A, true, true>), dim3(g), dim3(IDXA_NT), 0, st, ts, ix, agg ? 1 : 0, save);
  else hipLaunchKernelGGL((k_idx_stream<TILE_SLOT_A, false, true>), dim3(g), dim3(IDXA_NT), 0, st, ts, ix, agg ? 1 : 0, save);
}
void launch_words_init(const TileSet &ts, unsigned long long *wmeta, hipStream_t st) {
  if (!ts.n_tiles) return;
  unsigned int g = (ts.n_tiles + NWAVES - 1) / NWAVES;
  if (g > 256 * 8) g = 256 * 8;
  hipLaunchKernelGGL((k_words_init<TILE_SLOT_A>), dim3(g), dim3(BLOCK), 0, st, ts, wmeta);
}
// The grid hooks of the word-mode launchers (tests, tuning) are read when a context is made, not every round: getenv walks the whole
// environment, and a round's launch is on its critical path (yttm_kernels.h: launch_env_refresh).
static int g_wgather_grid = -1, g_words_grid = -1, g_words_wpi = -1;
void launch_env_refresh() {
  auto rd = [](const char *name) {
    const char *v = getenv(name);
    return v && *v ? atoi(v) : -1;
  };
  g_wgather_grid = rd("YTTM_WGATHER_GRID");
  g_words_grid = rd("YTTM_WORDS_GRID");
  g_words_wpi = rd("YTTM_WORDS_WPI");
}
void launch_wgather(const WGatherArgs &a, const BatchArgs *ba, unsigned int work_hint, hipStream_t st) {
  // every workgroup looks all the rules up and takes a ticket at the end: a small round (work_hint = about how many words it will visit;
  // 0: unknown) gets a small grid
  unsigned int g = 256u;
  if (work_hint) g = std::max(16u, std::min(256u, work_hint / 1024u));
  if (g_wgather_grid >= 0) g = (unsigned int)g_wgather_grid;
  hipLaunchKernelGGL(k_wgather, dim3(g ? g : 1u), dim3(WG_NT), 0, st, a, ba ? *ba : BatchArgs{});
}
bool launch_words_apply(const WordSet &ws, const PairTable &pt, const DeltaBuf &db, const RuleSlot *rules, unsigned int rule_mask, const uint32_t *bloom_g,
                        uint32_t self_x, uint32_t self_z, uint32_t z_base, uint32_t k_rules, const uint32_t *worklist, unsigned long long wl_seg,
                        const unsigned int *work_n, unsigned long long *stats, const TokLists &tl, DeltaRec *drec, unsigned int drec_cap, unsigned int *drec_n,
                        uint4 *irec, const BatchArgs *ba, const ScanArgs *scan, unsigned int work_hint, unsigned int inline_max, const WGatherArgs *ga,
                        unsigned int fuse_max, hipStream_t st) {
  if (!ws.n_words) return false;
  BatchArgs bargs = ba ? *ba : BatchArgs{};
  ScanArgs sargs = scan ? *scan : ScanArgs{};
  // the worklist first (k_wgather) -- unless the round is small enough for k_words to find its words itself (one launch a round)
  const bool fused = ga && worklist && work_hint && work_hint <= fuse_max && sargs.on && bargs.k != 0 && !ga->xyz && rule_mask < APPLY_LDS_RULES;
  if (fused && sargs.on == 2u) sargs.on = 3u;  // (multi-GPU: a fused round leaves no worklist behind -- there is no tail)
  if (ga && !fused) launch_wgather(*ga, &bargs, work_hint, st);
  if (!fused) bargs.mark = 0u;  // (the round's first launch carries the mark)
  // one run of 64 words per wave and iteration; work_hint = about how many words the round will visit (0: unknown / every word)
  const unsigned int gmax = std::min(g_words_grid >= 0 ? (unsigned int)g_words_grid : 512u, (unsigned int)WORDS_MAX_GRID);
  // words per wave: 64, or fewer when that would leave most of the chip idle (work_hint words over at most gmax workgroups)
  unsigned int wpi = 64;
  if (worklist && work_hint) {
    while (wpi > 8 && (unsigned long long)work_hint < (unsigned long long)wpi * APPLY_WPB * gmax / 2) wpi >>= 1;
    if (g_words_wpi >= 0) wpi = (unsigned int)g_words_wpi;
  }
  unsigned long long items = worklist && work_hint ? ((unsigned long long)work_hint + wpi - 1) / wpi + 1 : ((unsigned long long)ws.n_words + 63) / 64;
  unsigned long long g = (items + APPLY_WPB - 1) / APPLY_WPB;
  if (g > gmax) g = gmax;
  if (g < 1) g = 1;
  // a small round applies its records itself and carries the candidate scan (it needs that scan: its last workgroup resets the worklist)
  const bool inl = fused || (worklist && work_hint && work_hint <= inline_max && sargs.on);
  const ScanArgs none{};
  const WGatherArgs gnone{};
  if (fused)
    hipLaunchKernelGGL((k_words<APPLY_WPB, true, true>), dim3((unsigned int)g), dim3(64 * APPLY_WPB), 0, st, ws, pt, db, rules, rule_mask, bloom_g, self_x, self_z, z_base,
                       k_rules, worklist, wl_seg, work_n, stats, tl, drec, drec_cap, drec_n, irec, wpi, 1u, bargs, sargs, *ga);
  else if (rule_mask < APPLY_LDS_RULES)
    hipLaunchKernelGGL((k_words<APPLY_WPB, true, false>), dim3((unsigned int)g), dim3(64 * APPLY_WPB), 0, st, ws, pt, db, rules, rule_mask, bloom_g, self_x, self_z, z_base,
                       k_rules, worklist, wl_seg, work_n, stats, tl, drec, drec_cap, drec_n, irec, wpi, inl ? 1u : 0u, bargs, inl ? sargs : none, gnone);
  else
    hipLaunchKernelGGL((k_words<APPLY_WPB, false, false>), dim3((unsigned int)g), dim3(64 * APPLY_WPB), 0, st, ws, pt, db, rules, rule_mask, bloom_g, self_x, self_z, z_base,
                       k_rules, worklist, wl_seg, work_n, stats, tl, drec, drec_cap, drec_n, irec, wpi, inl ? 1u : 0u, bargs, inl ? sargs : none, gnone);
  if (inl) return fused;
  // the records -> the pair table, then the round's candidate scan (every workgroup owns a statistics row: at most BLK_ROWS of them)
  // (every workgroup takes a ticket at the end, ~12 ns each on one address: a small round gets a small grid)
  const bool big = !worklist || !work_hint || work_hint > (1u << 17);
  const unsigned int parts = std::max(1u, std::min(8u, (big ? (unsigned int)BLK_ROWS : 256u) / (unsigned int)g));
  hipLaunchKernelGGL(k_delta_apply, dim3((unsigned int)g * parts), dim3(DAPPLY_NT), 0, st, pt, db, (const DeltaRec *)drec, drec_cap, (const unsigned int *)drec_n, parts,
                     const_cast<unsigned int *>(work_n), stats, bargs.k ? (const RuleSlot *)nullptr : rules, rule_mask, self_x != 0xffffffffu ? pair_key(self_x, self_x) : PT_EMPTY, bargs, sargs);
  return false;
}
void launch_hot_rebuild(const PairTable &pt, hipStream_t st) {
  unsigned long long n_slots = pt.mask + 1;
  unsigned long long b = (n_slots + BLOCK - 1) / BLOCK;
  if (b > 256 * 16) b = 256 * 16;
  hipLaunchKernelGGL(k_hot_rebuild, dim3((unsigned int)b), dim3(BLOCK), 0, st, pt);
}
void launch_fold_stats(unsigned long long *stats, unsigned int *n_keys, hipStream_t st) {
  hipLaunchKernelGGL(k_fold_stats, dim3(1), dim3(BLOCK), 0, st, stats, n_keys);
}
void launch_pt_rehash(const PairTable &src, const PairTable &dst, hipStream_t st) {
  unsigned long long n_slots = src.mask + 1;
  unsigned long long b = (n_slots + BLOCK - 1) / BLOCK;
  if (b > 256 * 16) b = 256 * 16;
  hipLaunchKernelGGL(k_pt_rehash, dim3((unsigned int)b), dim3(BLOCK), 0, st, src, dst);
}
void launch_pt_zero(const PairTable &pt, const RuleSlot *rules, unsigned int n_slots, unsigned long long self_key, hipStream_t st) {
  hipLaunchKernelGGL(k_pt_zero, dim3((n_slots + 1 + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st, pt, rules, n_slots, self_key);
}
void launch_pt_query(const PairTable &pt, const unsigned long long *keys, unsigned int n, unsigned long long *out, hipStream_t st) {
  if (!n) return;
  hipLaunchKernelGGL(k_pt_query, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st, pt, keys, n, out);
}
void launch_pt_apply(const PairTable &pt, const DeltaRec *recs, unsigned long long n, hipStream_t st) {
  if (!n) return;
  unsigned long long b = (n + BLOCK - 1) / BLOCK;
  if (b > 256 * 16) b = 256 * 16;
  hipLaunchKernelGGL(k_pt_apply, dim3((unsigned int)b), dim3(BLOCK), 0, st, pt, recs, n);
}
void launch_round_begin(const RuleSlot *src_rules, unsigned int n_slots, RuleSlot *dst_rules, unsigned int *work_n_a, unsigned int *work_n_b,
                        const uint32_t *src_bloom, uint32_t *dst_bloom, hipStream_t st) {
  unsigned int work = n_slots;
  if (src_bloom && work < (unsigned int)PM_BLOOM_WORDS_H) work = PM_BLOOM_WORDS_H;
  unsigned int b = (work + BLOCK - 1) / BLOCK;
  if (b < 1) b = 1;
  if (b > 64) b = 64;
  hipLaunchKernelGGL(k_round_begin, dim3(b), dim3(BLOCK), 0, st, src_rules, n_slots, dst_rules, work_n_a, work_n_b, src_bloom, dst_bloom);
}
__global__ __launch_bounds__(BLOCK) void k_pt_clear(uint4 *__restrict__ slots, unsigned long long n) {
  unsigned long long i = (unsigned long long)blockIdx.x * BLOCK + threadIdx.x;
  const unsigned long long stride = (unsigned long long)gridDim.x * BLOCK;
  const uint4 e{0xffffffffu, 0xffffffffu, 0u, 0u};  // { PT_EMPTY, 0 }
  for (; i < n; i += stride) slots[i] = e;
}
void launch_pt_clear(const PairTable &pt, hipStream_t st) {
  unsigned long long n = pt.mask + 1, b = (n + BLOCK - 1) / BLOCK;
  if (b > 256 * 16) b = 256 * 16;
  hipLaunchKernelGGL(k_pt_clear, dim3((unsigned int)b), dim3(BLOCK), 0, st, reinterpret_cast<uint4 *>(pt.slots), n);
}
void launch_fill_u64(unsigned long long *p, unsigned long long v, unsigned long long n, hipStream_t st) {
  if (!n) return;
  unsigned long long b = (n + BLOCK - 1) / BLOCK;
  if (b > 256 * 16) b = 256 * 16;
  hipLaunchKernelGGL(k_fill_u64, dim3((unsigned int)b), dim3(BLOCK), 0, st, p, v, n);
}

}  // namespace yttm
