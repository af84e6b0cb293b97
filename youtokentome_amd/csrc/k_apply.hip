// k_apply.hip -- K4 for class-A tiles (words of <= 256 tokens): batched merge-apply + exact count deltas, position-parallel.
//
// Replaces worker_doing_merge (bpe.cpp:491-812: list splice, +-pair2cnt, run handling :625-691 / :719-785, new-pair reports
// :789-804) for a whole batch of mutually non-intersecting rules at once (host_trainer.cpp picks the batch).
//
// One WAVEFRONT owns a tile.  Per tile:
//   (1) in registers, 16 B per lane as loaded: every adjacency is hashed into a Bloom filter of the batch's pairs (64 Kbit in
//       LDS, two bits per rule in one word) -- a tile without a hit is dismissed here; the filter is on PAIRS, not on tokens, so
//       late in training (random text: every tile holds both halves of nearly every rule) a dismissed tile really is most tiles.
//   (2) a tile with a hit is staged into LDS and looked at position-major (lane l <-> position 64 c + l): exact rule look-up for
//       the hits -> the chunk's site mask (a ballot), the mask of equal neighbours and, from it by carry arithmetic on the
//       scalar unit (run_select.h), which pairs of a run count.  Sites of x != y rules cannot overlap (no token is the x of one
//       rule and the y of another in a batch); the sites of the x x rule are the pairs of its runs at an even offset.
//   (3) every lane that holds a site works out ITS count deltas from its four neighbours in LDS and the neighbouring bits of
//       the masks -- no list of sites, no walk along a run: (L,x) -> (L,z), (y,R) -> (z,R'), a run of x's or y's that loses a
//       member (the pair at the run's END tells: it knows the parity and, through a carry, that the run's first token went into
//       a merge), and runs of the new token (links of a chain at stride 2).  The (pair, delta) records go to a per-wave queue in
//       LDS and are applied 64 at a time -- one per lane -- to the workgroup's LDS aggregator, whatever tile they came from.
//   (4) the tile is compacted in place from its first site on (ballot + mbcnt).
// The merged pairs themselves are not retracted site by site: every occurrence goes, their counts are zeroed after the round.
// HBM-bound integer work, no MFMA.
#include <stdlib.h>

#include "k_merge_shared.h"
#include "run_select.h"

namespace yttm {

void pm_bloom_host(uint32_t *bloom, const uint32_t *xyz, uint32_t k) {  // (batches too large for the LDS rule hash: built by the host)
  for (int i = 0; i < PM_BLOOM_WORDS; i++) bloom[i] = 0;
  for (uint32_t j = 0; j < k; j++) {
    const uint32_t x = xyz[3 * j], y = xyz[3 * j + 1];
    if (x == y) continue;
    const uint32_t h = pm_hash(x, y);
    bloom[pm_word(h)] |= pm_bits(h);
  }
}

constexpr unsigned int PM_LDS_RULES = 512;  // slots of the batch's rule hash kept in LDS (batches of up to 256 rules)
constexpr int PM_QCAP = 128;                // site records in a wave's queue (applied 64 at a time)

struct PmShared {
  unsigned long long key[AGG_SLOTS];
  unsigned long long val[AGG_SLOTS];
  uint32_t bloom[PM_BLOOM_WORDS];
  unsigned int new_keys;
  unsigned long long st[4];
};
// A site record: what ONE merge site (or one reporting run end) changes, worked out position-parallel where the site is, applied
// later by whichever lane draws it: {L, x, y, R}, {z, R', f, flags}.
constexpr uint32_t PMF_L0 = 1u,    // (L,x) -= f   [PMF_XX: (x,x) -= f instead: a run of x's lost a member]
                   PMF_XX = 2u,
                   PMF_L1 = 4u,    // (L,z) += f
                   PMF_R0 = 8u,    // (y,R) -= f
                   PMF_R1 = 16u,   // (z,R') += f
                   PMF_ZZ = 32u;   // (z,z) += f
template <int SLOT>
struct PmWave {
  uint32_t tk[SLOT + 8];   // staged tokens: position p at tk[4 + p]; sentinels behind the end
  uint4 qa[PM_QCAP];       // site records waiting for the aggregator
  uint4 qb[PM_QCAP];
};

template <bool IN_LDS>
struct PmRules {
  const unsigned long long *lds_keys;
  const uint16_t *lds_ridx;
  const RuleSlot *g;
  unsigned int mask;
  uint32_t z_base;
  __device__ uint32_t find(uint32_t a, uint32_t b) const {  // index of the rule in the batch, or 0xffff
    const unsigned long long key = pair_key(a, b);
    unsigned int h = pair_hash32(key) & mask;
    for (;;) {
      unsigned long long k;
      if (IN_LDS) k = __hip_atomic_load(&lds_keys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else k = g[h].key;
      if (k == key) return IN_LDS ? (uint32_t)lds_ridx[h] : g[h].z - z_base;
      if (k == PT_EMPTY) return 0xffffu;
      h = (h + 1) & mask;
    }
  }
};

// one count update into the workgroup's LDS aggregator; false: no room within 8 probes (the caller sends it to the HBM table)
__device__ inline bool pm_agg_try(PmShared &A, unsigned long long key, long long delta) {
  unsigned int h = (pair_hash32(key) >> 7) & (AGG_SLOTS - 1);
#pragma unroll
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

// 16 B per lane, one tile ahead of use (lane l: tokens 256 j + 4 l + {0..3}; nothing is read behind the live prefix: zeros)
template <int SLOT>
__device__ inline void pm_fetch(uint4 (&r)[SLOT / 256], const TileSet &ts, uint32_t t, int n) {
  const uint4 *src = reinterpret_cast<const uint4 *>(ts.tok + (size_t)t * SLOT);
#pragma unroll
  for (int j = 0; j < SLOT / 256; j++) {
    const int i = lane_id() + 64 * j;
    r[j] = (4 * i < n) ? src[i] : make_uint4(0, 0, 0, 0);
  }
}

// (1) does any adjacency of the tile pass the Bloom filter (or is it an x x of the self rule)?  In registers, as loaded.
template <int SLOT>
__device__ inline bool pm_candidate(const uint4 (&r)[SLOT / 256], int n, const uint32_t *bloom, uint32_t self_x) {
  const int lane = lane_id();
  uint32_t hit = 0;
  const bool has_self = self_x != 0xffffffffu;
#define PM_TEST(HA, HB, T1)                                         \
  {                                                                 \
    const uint32_t h_ = (HA) ^ (HB);                                \
    const uint32_t bits_ = pm_bits(h_);                             \
    hit |= (uint32_t)((bloom[pm_word(h_)] & bits_) == bits_) & ~((T1) >> 31); \
  }
#pragma unroll
  for (int j = 0; j < SLOT / 256; j++) {
    if (256 * j < n) {
      uint32_t nx = from_lane_right(r[j].x);
      uint32_t nx0 = TOK_WS;
      if (j + 1 < SLOT / 256) nx0 = from_lane0(r[j + 1 < SLOT / 256 ? j + 1 : j].x);
      if (lane == 63) nx = nx0;
      const uint32_t a0 = r[j].x & TOK_MASK, a1 = r[j].y & TOK_MASK, a2 = r[j].z & TOK_MASK, a3 = r[j].w & TOK_MASK, a4 = nx & TOK_MASK;
      PM_TEST(pm_mul24(a0, PM_K1), pm_mul24(a1, PM_K2), r[j].y)
      PM_TEST(pm_mul24(a1, PM_K1), pm_mul24(a2, PM_K2), r[j].z)
      PM_TEST(pm_mul24(a2, PM_K1), pm_mul24(a3, PM_K2), r[j].w)
      PM_TEST(pm_mul24(a3, PM_K1), pm_mul24(a4, PM_K2), nx)
      if (has_self) {
        hit |= (uint32_t)(a0 == self_x && a1 == self_x) & ~(r[j].y >> 31);
        hit |= (uint32_t)(a1 == self_x && a2 == self_x) & ~(r[j].z >> 31);
        hit |= (uint32_t)(a2 == self_x && a3 == self_x) & ~(r[j].w >> 31);
        hit |= (uint32_t)(a3 == self_x && a4 == self_x) & ~(nx >> 31);
      }
    }
  }
#undef PM_TEST
  return ballot_b((hit & 1u) != 0u) != 0ull;
}

struct PmStats {
  unsigned long long sites = 0, touched = 0, scanned = 0, touched_tok = 0;
};

// what (2) finds in one 64-token chunk (masks: wave-uniform, in scalar registers)
struct PmChunk {
  unsigned long long sm = 0, eq = 0, sel = 0, wsm = 0;  // sites; equal neighbours of one word; those at an even offset of their run; word starts
  uint32_t wsbase = 0;                                  // word starts before the chunk
  uint32_t t0 = TOK_WS, t1 = TOK_WS, zi = 0xffffu;      // per lane: my token, the next one, my site's rule index
};

// the value lane + 2 holds; lanes 62 and 63 get lanes 0 and 1 of `next`
__device__ inline uint32_t pm_from_lane_p2(uint32_t v, uint32_t next) {
  uint32_t r = __shfl_down(v, 2);
  const uint32_t n0 = from_lane0(next);
  const uint32_t n1 = (uint32_t)__builtin_amdgcn_readlane((int)next, 1);
  const int lane = lane_id();
  if (lane == 62) r = n0;
  if (lane == 63) r = n1;
  return r;
}

// (2)-(4): everything that happens to a tile that passed the filter -- or, with n == 0 and flush set, only the queue's remainder.
// r = the tile as loaded, wq = its word frequencies (lane l: words l, l + 64, ...).
template <int SLOT, bool LDSR>
__device__ inline void pm_process(PmWave<SLOT> &W, PmShared &A, const TileSet &ts, const PairTable &pt, const DeltaBuf &db, const PmRules<LDSR> &rtab,
                                  uint32_t self_x, uint32_t self_z, uint32_t z_base, uint32_t t, int n, uint32_t word0, const uint4 (&r)[SLOT / 256],
                                  const uint32_t (&wq)[SLOT / 128], uint32_t &qn, PmStats &S, bool flush) {
  constexpr int NW = SLOT / 128;
  const int lane = lane_id();
  uint32_t *tk = W.tk + 4;  // tk[-1] .. tk[SLOT + 3] are addressable
  const bool has_self = self_x != 0xffffffffu;
  const uint32_t self_ri = self_z - z_base;
  // ---- stage: tokens and sentinels ("a word starts here" behind the end: no adjacency, no neighbour)
#pragma unroll
  for (int j = 0; j < SLOT / 256; j++)
    if (256 * j < n) reinterpret_cast<uint4 *>(tk)[lane + 64 * j] = r[j];
  wave_sync();
  if (lane < 4) tk[n + lane] = TOK_WS;
  wave_sync();
  const int nch = (n + 63) >> 6;
  // ---- (2) the sites of one chunk, position-major
  RunCarry rc;
  uint32_t wbase = 0;
  auto find_sites = [&](int c) {
    PmChunk k;
    const int p = 64 * c + lane;
    const bool valid = p < n;
    k.t0 = valid ? tk[p] : TOK_WS;
    k.t1 = valid ? tk[p + 1] : TOK_WS;
    const uint32_t a = k.t0 & TOK_MASK, b = k.t1 & TOK_MASK;
    const bool adj = !(k.t1 >> 31);
    k.eq = ballot_b(adj && a == b);
    if (k.eq != 0ull || rc.cont) k.sel = even_offset_select(k.eq, rc);
    const uint32_t h = pm_hash(a, b);
    const uint32_t bits = pm_bits(h);
    const bool hit = adj && (A.bloom[pm_word(h)] & bits) == bits;
    if (ballot_b(hit) != 0ull) {
      if (hit) k.zi = rtab.find(a, b);
    }
    if (has_self) {
      if (adj && a == self_x && b == self_x && lane_bit(k.sel)) k.zi = self_ri;
    }
    k.sm = ballot_b(k.zi != 0xffffu);
    k.wsm = ballot_b(valid && (k.t0 >> 31) != 0u);
    k.wsbase = wbase;
    wbase += (uint32_t)__popcll(k.wsm);
    return k;
  };
  uint32_t *dst = ts.tok + (size_t)t * SLOT;
  uint32_t abase = 0;
  bool started = false;  // a site has been seen: from its chunk on the tile is rewritten
  bool cons = false;     // the run of equal tokens at the top of the previous chunk goes on, and its first token went into a merge
  Chain2Carry zc;
  int nsites = 0;
  PmChunk prv, cur, nxt;
  for (int c = -1;; c++) {  // (chunk c is worked on when the sites of chunk c + 1 are known: one pass, one chunk of look-ahead)
    if (c < nch) {
      nxt = PmChunk{};
      if (c + 1 < nch) nxt = find_sites(c + 1);
    }
    if (c >= 0 && c < nch) {
      const unsigned long long sm = cur.sm, smp = prv.sm, smn = nxt.sm;
      if (sm != 0ull && !started) {
        started = true;
        abase = 64u * (uint32_t)c;
      }
      const int p = 64 * c + lane;
      const uint32_t t0 = cur.t0;
      const bool site = lane_bit(sm);
      const uint32_t z = z_base + cur.zi;
      if ((sm | (smp >> 63)) != 0ull || cons) {
        // ---- (3) count deltas: every lane with a site (or at a reporting run end) makes its record
        nsites += __popcll(sm);
        const unsigned long long Eq = cur.eq, Eqp = prv.eq, Eqn = nxt.eq;
        const unsigned long long selpm = (cur.sel << 1) | (prv.sel >> 63);  // bit l: the pair (l-1, l) is at an even offset of its run
        const uint32_t t1 = cur.t1, t2 = tk[p + 2], tl = tk[p - 1];
        const uint32_t x = t0 & TOK_MASK, y = t1 & TOK_MASK;
        const uint32_t Lt = tl & TOK_MASK, Rt = t2 & TOK_MASK;
        const bool hasL = !(t0 >> 31), hasR = !(t2 >> 31);
        const bool s_m2 = hasL && lane_bit((sm << 2) | (smp >> 62));   // the token to the left is the y of another site
        const unsigned long long smp2 = (sm >> 2) | (smn << 62);
        const bool s_p2 = hasR && lane_bit(smp2);                      // the token behind the site is the x of another site
        const bool eqprev = lane_bit((Eq << 1) | (Eqp >> 63));         // same word and L == x
        const bool eqnext = lane_bit((Eq >> 1) | (Eqn << 63));         // same word and R == y
        const bool selprev = lane_bit(selpm);
        const bool self = has_self && site && x == self_x;
        // runs of equal tokens whose first token went into a merge (as the y of the site right before the run): the run's LAST
        // token reports the lost self pair, if the run's last pair sat at an even offset (= the run's length was even)
        const unsigned long long endm = marked_run_ends(Eq, (sm << 1) | (smp >> 63), (Eqp >> 63) != 0ull, cons);
        const unsigned long long rem = endm & selpm;
        const bool run_end = lane_bit(rem);
        // the new token two positions on (sites in a row: x y x y -> z z)
        uint32_t zn2 = 0xffffffffu;
        unsigned long long ZZ = 0ull;
        if ((sm & smp2) != 0ull) {
          zn2 = z_base + pm_from_lane_p2(cur.zi, nxt.zi);
          ZZ = ballot_b(site && s_p2 && zn2 == z);
        }
        unsigned long long selz = 0ull;
        if (ZZ != 0ull || zc.ce || zc.co) selz = stride2_select(ZZ, zc);
        const unsigned long long recm = sm | rem;
        if (recm != 0ull) {
          // word frequency: word starts at or before me, minus one; the frequencies travel in registers (lane j: words j, j + 64, ..)
          const uint32_t kw = cur.wsbase + lanes_below(cur.wsm) + (t0 >> 31) - 1u;
          uint32_t f = 0;
#pragma unroll
          for (int i = 0; i < NW; i++) {
            const uint32_t fi = __shfl(wq[i], (int)(kw & 63u));
            if ((kw >> 6) == (uint32_t)i) f = fi;
          }
          if (lane_bit(recm) && kw >= 64u * NW) f = ts.wcnt[word0 + kw];  // (a re-dealt tile of one- and two-token words can hold more)
          const bool lside = site && hasL && !s_m2;
          const bool xx = site && !self && eqprev && selprev;
          uint32_t fl = 0;
          if ((lside && !eqprev) || xx || run_end) fl |= PMF_L0;
          if (xx || run_end) fl |= PMF_XX;
          if (lside) fl |= PMF_L1;
          if (site && hasR && !eqnext) fl |= PMF_R0;
          const uint32_t Bt = s_p2 ? zn2 : Rt;
          if (site && hasR && Bt != z) fl |= PMF_R1;
          if (lane_bit(selz)) fl |= PMF_ZZ;
          if (lane_bit(recm)) {
            const uint32_t slot = qn + lanes_below(recm);
            uint4 va, vb;
            va.x = Lt; va.y = x; va.z = y; va.w = Rt;
            vb.x = z; vb.y = Bt; vb.z = f; vb.w = fl;
            W.qa[slot] = va;
            W.qb[slot] = vb;
          }
          qn += (uint32_t)__popcll(recm);
        }
      }
      // ---- (4) survivors of the chunk = its live positions that are not the y of a site
      if (started) {
        const int left = n - 64 * c;
        const unsigned long long am = (left >= 64 ? ~0ull : (1ull << left) - 1ull) & ~((sm << 1) | (smp >> 63));
        if (lane_bit(am)) dst[abase + lanes_below(am)] = site ? (z | (t0 & TOK_WS)) : t0;
        abase += (uint32_t)__popcll(am);
      }
    }
    if (c < nch) {
      prv = cur;
      cur = nxt;
    }
    // ---- site records are applied 64 at a time, one per lane, whatever tile they came from (the ONLY place that touches the aggregator)
    const bool last = c + 1 >= nch;
    while (qn >= 64u || (flush && last && qn != 0u)) {
      wave_sync();
      const uint32_t take = qn < 64u ? qn : 64u;
      const bool have = (uint32_t)lane < take;
      const uint4 va = W.qa[qn - take + (uint32_t)lane], vb = W.qb[qn - take + (uint32_t)lane];
      const uint32_t fl = have ? vb.w : 0u;
      const long long f = (long long)vb.z;
      for (int kk = 0; kk < 5; kk++) {  // (a loop, not five copies: the aggregator code is long)
        bool v;
        unsigned long long key;
        long long d;
        if (kk == 0) { v = fl & PMF_L0; key = (fl & PMF_XX) ? pair_key(va.y, va.y) : pair_key(va.x, va.y); d = -f; }
        else if (kk == 1) { v = fl & PMF_L1; key = pair_key(va.x, vb.x); d = f; }
        else if (kk == 2) { v = fl & PMF_R0; key = pair_key(va.z, va.w); d = -f; }
        else if (kk == 3) { v = fl & PMF_R1; key = pair_key(vb.x, vb.y); d = f; }
        else { v = fl & PMF_ZZ; key = pair_key(vb.x, vb.x); d = f; }
        if (ballot_b(v) == 0ull) continue;
        bool miss = false;
        if (v) miss = !pm_agg_try(A, key, d);
        if (ballot_b(miss) != 0ull) {
          if (miss) global_emit(pt, db, key, d, &A.new_keys);
        }
      }
      qn -= take;
      wave_sync();
    }
    if (last) break;
  }
  if (n > 0) {
    S.scanned += (unsigned long long)n;
    if (started) {
      // invariant: slots behind the live prefix hold zeros
      for (int p = (int)abase + lane; p < n; p += 64) dst[p] = 0;
      if (lane == 0) {
        ts.tile_len[t] = abase;
        S.sites += (unsigned long long)nsites;
      }
      S.touched++;
      S.touched_tok += (unsigned long long)n;
    }
  }
  wave_sync();  // everyone is done with this tile's LDS state before the next one is staged
}

template <int SLOT, int WPB, bool LDSR>
__global__ __launch_bounds__(WPB * 64) void k_apply_pm(TileSet ts, PairTable pt, DeltaBuf db, const RuleSlot *__restrict__ rules, unsigned int rule_mask,
                                                       const uint32_t *__restrict__ bloom_g, uint32_t self_x, uint32_t self_z, uint32_t z_base,
                                                       const uint32_t *__restrict__ worklist, const unsigned int *__restrict__ work_n,
                                                       unsigned long long *__restrict__ stats, BatchArgs ba, ScanArgs sa, uint32_t eager_w) {
  constexpr int NW = SLOT / 128;
  __shared__ PmWave<SLOT> WL[WPB];
  __shared__ PmShared A;
  __shared__ unsigned long long rkeys[LDSR ? PM_LDS_RULES : 1];
  __shared__ uint16_t rridx[LDSR ? PM_LDS_RULES : 1];
  const bool from_args = LDSR && ba.k != 0;  // tables built from the kernel argument, nothing read from HBM
  for (int s = (int)threadIdx.x; s < AGG_SLOTS; s += WPB * 64) {
    A.key[s] = PT_EMPTY;
    A.val[s] = 0;
  }
  for (int s = (int)threadIdx.x; s < PM_BLOOM_WORDS; s += WPB * 64) A.bloom[s] = LDSR ? 0u : bloom_g[s];
  if (LDSR)
    for (unsigned int i = threadIdx.x; i <= rule_mask; i += WPB * 64) {
      rkeys[i] = from_args ? PT_EMPTY : rules[i].key;
      if (!from_args) rridx[i] = (uint16_t)(rules[i].z - z_base);
    }
  if (threadIdx.x == 0) {
    A.new_keys = 0;
    A.st[0] = A.st[1] = A.st[2] = A.st[3] = 0;
  }
  __syncthreads();
  if (from_args) {
    if (threadIdx.x < ba.k) {
      const uint32_t x = ba.xy[2 * threadIdx.x], y = ba.xy[2 * threadIdx.x + 1];
      if (x != y) {
        const unsigned long long key = pair_key(x, y);
        unsigned int h = pair_hash32(key) & rule_mask;
        for (;;) {
          if (atomicCAS(&rkeys[h], PT_EMPTY, key) == PT_EMPTY) {
            rridx[h] = (uint16_t)threadIdx.x;
            break;
          }
          h = (h + 1) & rule_mask;
        }
        const uint32_t bh = pm_hash(x, y);
        atomicOr(&A.bloom[pm_word(bh)], pm_bits(bh));
      }
    }
  } else if (LDSR) {
    for (unsigned int i = threadIdx.x; i <= rule_mask; i += WPB * 64) {
      const unsigned long long key = rkeys[i];
      if (key != PT_EMPTY) {
        const uint32_t bh = pm_hash((uint32_t)(key >> 32), (uint32_t)key);
        atomicOr(&A.bloom[pm_word(bh)], pm_bits(bh));
      }
    }
  }
  const PmRules<LDSR> rtab{rkeys, rridx, rules, rule_mask, z_base};
  __syncthreads();
  const int wave = uni((int)(threadIdx.x >> 6)), lane = lane_id();
  PmWave<SLOT> &W = WL[wave];
  const uint32_t stride = gridDim.x * WPB;
  // the tiles of this launch: all of them, or the worklist gathered from the pair index (k_merge.hip k_gather; a worklist that
  // could not be completed there -- work_n[WL_PARTS + 1] -- is not used)
  uint32_t wn[WL_PARTS];
  uint32_t NT = ts.n_tiles;
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
  // Headers (live length, first word) of the wave's next 64 tiles are loaded with ONE vector load each (lane j: item i + j stride)
  // and handed out by shuffles; the tokens of item i + 1 are fetched while item i is processed.
  uint32_t t = blockIdx.x * WPB + wave;
  int hn = 0;
  uint32_t hw = 0, ht = 0;
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
      hw = ts.tile_word0[ht];
    }
  };
  uint4 r[SLOT / 256];
  uint32_t wq[NW];
  PmStats S;
  uint32_t qn = 0;  // records in this wave's queue (uniform)
  const bool eager = eager_w != 0u || worklist != nullptr;  // (nearly) every tile of the launch holds a site: the word frequencies travel with the tokens
  auto wload = [&](uint32_t w0) {
#pragma unroll
    for (int i = 0; i < NW; i++) wq[i] = ts.wcnt[w0 + (uint32_t)(lane + 64 * i)];  // (wcnt is padded by 64 NW)
  };
  int j = 0;
  if (t < NT) {
    load_headers(t);
    pm_fetch<SLOT>(r, ts, uni(from_lane0(ht)), uni(from_lane0((uint32_t)hn)));
    if (eager && uni(from_lane0((uint32_t)hn)) != 0u) wload(uni(from_lane0(hw)));  // (an empty tile's first word is not defined)
  }
  while (t < NT) {
    const int n0 = uni(__shfl(hn, j));
    const uint32_t w0 = uni(__shfl(hw, j));
    const uint32_t tile = uni(__shfl(ht, j));
    // a dense round (the host: most tiles held a site last round) skips the filter: (2) finds the sites or finds none
    const bool cand = n0 > 0 && (eager_w != 0u || pm_candidate<SLOT>(r, n0, A.bloom, self_x));
    if (cand && !eager) wload(w0);
    uint4 rc[SLOT / 256];
    uint32_t wc[NW];
#pragma unroll
    for (int i = 0; i < SLOT / 256; i++) rc[i] = r[i];
#pragma unroll
    for (int i = 0; i < NW; i++) wc[i] = wq[i];
    // next item of this wave: header from the batch (reloaded every 64 items), tokens prefetched now
    const uint32_t t_next = t + stride;
    j++;
    if (j == 64 && t_next < NT) {
      j = 0;
      load_headers(t_next);
    }
    if (t_next < NT) {
      const int n_next = uni(__shfl(hn, j));
      pm_fetch<SLOT>(r, ts, uni(__shfl(ht, j)), n_next);
      if (eager && n_next != 0) wload(uni(__shfl(hw, j)));
    }
    // (one call site: the last pass of a wave -- past its last tile -- only applies what is left in its queue)
    const bool flush = t_next >= NT;
    if (cand || flush) pm_process<SLOT, LDSR>(W, A, ts, pt, db, rtab, self_x, self_z, z_base, tile, cand ? n0 : 0, w0, rc, wc, qn, S, flush);
    if (!cand) S.scanned += (unsigned long long)n0;
    t = t_next;
  }
  S.sites = wave_sum_u64(S.sites);
  if (lane == 0) {
    if (S.sites) atomicAdd(&A.st[0], S.sites);
    if (S.touched) atomicAdd(&A.st[1], S.touched);
    if (S.scanned && !worklist) atomicAdd(&A.st[2], S.scanned);
    if (S.touched_tok) atomicAdd(&A.st[3], S.touched_tok);
  }
  __syncthreads();
  for (int s = (int)threadIdx.x; s < AGG_SLOTS; s += WPB * 64) {
    const unsigned long long k = A.key[s];
    if (k != PT_EMPTY) {
      const long long v = (long long)A.val[s];
      if (v != 0) global_emit(pt, db, k, v, &A.new_keys);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    blk_add(stats, 4, A.new_keys);
    for (int i = 0; i < 4; i++) blk_add(stats, i, A.st[i]);
  }
  if (sa.on) {  // the round's candidate scan, by the last workgroup to get here (k_merge_shared.h scan_top)
    // Everything this workgroup leaves for the tail went out as device-scope atomics or write-through (sc1) stores -- pair table,
    // candidate lists, statistics row -- so publishing needs them COMPLETE, not a cache write-back: every wave drains its memory
    // operations (s_waitcnt vmcnt(0), written out: the compiler may drop the wait of a fence it thinks has nothing to wait for),
    // then one lane takes the ticket with an agent-scope atomic.  (MI355X_MICROARCH.md: "sc1 payload -> vmcnt(0) -> flag" is a
    // valid hand-off; a release fence at agent scope would also write the XCD's L2 back, once per workgroup: +150 us per round.)
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

constexpr int PM_WPB = 8, PM_BPC = 2;
void launch_apply_pm(const TileSet &ts, const PairTable &pt, const DeltaBuf &db, const RuleSlot *rules, unsigned int rule_mask, const uint32_t *bloom_g,
                     uint32_t self_x, uint32_t self_z, uint32_t z_base, const uint32_t *worklist, const unsigned int *work_n, unsigned long long *stats,
                     const BatchArgs *ba, const ScanArgs *scan, bool eager_w, hipStream_t st) {
  if (!ts.n_tiles) return;
  const BatchArgs bargs = ba ? *ba : BatchArgs{};
  const ScanArgs sargs = scan ? *scan : ScanArgs{};
  unsigned int need = (ts.n_tiles + PM_WPB - 1) / PM_WPB;
  unsigned int grid = 256u * PM_BPC;
  {
    // a small tile set (natural-language corpora: a few thousand tiles) gets fewer workgroups with several tiles per wave: every
    // workgroup costs a prologue and a serialised ticket at the end.  YTTM_APPLY_GRID overrides (tuning hook).
    static const char *g_env = getenv("YTTM_APPLY_GRID");
    const unsigned int small = g_env ? (unsigned int)atoi(g_env) : 256u;
    if (ts.n_tiles <= 16384 && small && grid > small) grid = small;
  }
  if (grid > need) grid = need;
  if (!grid) grid = 1;
  if (rule_mask < PM_LDS_RULES)
    hipLaunchKernelGGL((k_apply_pm<TILE_SLOT_A, PM_WPB, true>), dim3(grid), dim3(64 * PM_WPB), 0, st, ts, pt, db, rules, rule_mask, bloom_g, self_x, self_z,
                       z_base, worklist, work_n, stats, bargs, sargs, eager_w ? 1u : 0u);
  else
    hipLaunchKernelGGL((k_apply_pm<TILE_SLOT_A, PM_WPB, false>), dim3(grid), dim3(64 * PM_WPB), 0, st, ts, pt, db, rules, rule_mask, bloom_g, self_x, self_z,
                       z_base, worklist, work_n, stats, bargs, sargs, eager_w ? 1u : 0u);
}

}  // namespace yttm
