// k_giant.hip -- K3/K4 for class C: words of more than TILE_NOM_B tokens (minified code, base64 blobs, ...).
//
// The reference has no length limit (its lists are per word, bpe.cpp:436-478); the wavefront-per-tile kernels of
// k_tiles.hip keep a whole tile in LDS and stop at 2048 tokens per word.  Class C uses the same tile layout with a
// run-time slot size (nominal = the longest word, slot = twice that) and ONE WORKGROUP per tile working in HBM.  Such
// words are rare and their weight is almost always 1, so this path is written for exactness and simplicity, not speed:
// a tile that contains a merge site is re-counted -- every adjacency of the old token sequence (but the merged pairs, which
// are zeroed after the round) is retracted and every adjacency of the new one is added, which nets out to exactly the deltas the
// tile kernels compute around the sites
// (worker_doing_merge, bpe.cpp:491-812) without any of their case analysis.
#include "yttm_device.h"
#include "yttm_kernels.h"

namespace yttm {

namespace {
// phases of a workgroup hand data over through HBM: agent-scope accesses bypass the (non-coherent) per-CU L1
__device__ inline uint32_t ld32(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void st32(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ inline void giant_emit(const PairTable &pt, const DeltaBuf &db, unsigned long long key, long long delta, unsigned int *new_keys) {
  pt_add(pt, key, delta, new_keys);
  dt_add(db, key, delta);
}

// weighted adjacency counts of tok[0..n) (a run of L equal tokens counts floor(L/2) for its self pair,
// emitted by the run's first token), each times sign * (frequency of its word)
// `rules` / `self_x` (retraction pass only): the pairs of the batch being applied are left alone -- every occurrence of them is
// merged and their counts are zeroed after the round, like in the tile kernels; retracting them here as well would depend on
// the retraction reaching the table before that zeroing (multi-GPU: a delta block repeated after the first scan would not).
__device__ inline void count_pairs(const uint32_t *tok, const uint32_t *wgt, int n, long long sign, const PairTable &pt, const DeltaBuf &db,
                                   unsigned int *new_keys, const RuleSlot *__restrict__ rules = nullptr, unsigned int rule_mask = 0,
                                   uint32_t self_x = 0xffffffffu) {
  for (int p = (int)threadIdx.x; p + 1 < n; p += (int)blockDim.x) {
    const uint32_t t0 = ld32(&tok[p]), t1 = ld32(&tok[p + 1]);
    if (t1 & TOK_WS) continue;
    const uint32_t a = t0 & TOK_MASK, b = t1 & TOK_MASK;
    const long long f = sign * (long long)ld32(&wgt[p]);
    if (a != b) {
      if (rules) {
        const unsigned long long key = pair_key(a, b);
        unsigned int h = pair_hash32(key) & rule_mask;
        bool in_batch = false;
        for (;;) {
          const unsigned long long k = rules[h].key;
          if (k == key) { in_batch = true; break; }
          if (k == PT_EMPTY) break;
          h = (h + 1) & rule_mask;
        }
        if (in_batch) continue;
      }
      giant_emit(pt, db, pair_key(a, b), f, new_keys);
    } else if (a == self_x) {
      continue;
    } else {
      const bool run_start = (t0 & TOK_WS) || p == 0 || (ld32(&tok[p - 1]) & TOK_MASK) != a;
      if (run_start) {
        int q = p + 1;
        while (q + 1 < n) {
          const uint32_t tq = ld32(&tok[q + 1]);
          if ((tq & TOK_WS) || (tq & TOK_MASK) != a) break;
          q++;
        }
        const long long len = q - p + 1;
        giant_emit(pt, db, pair_key(a, a), (len / 2) * f, new_keys);
      }
    }
  }
}

// wgt[p] = frequency of the word that contains position p (block-wide running count of word starts)
__device__ inline void word_weights(const uint32_t *tok, uint32_t *wgt, int n, const uint32_t *__restrict__ wcnt, uint32_t word0, uint32_t *scan_lds) {
  __shared__ uint32_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int p0 = 0; p0 < n; p0 += (int)blockDim.x) {
    const int p = p0 + (int)threadIdx.x;
    const uint32_t ws = (p < n && (ld32(&tok[p]) & TOK_WS)) ? 1u : 0u;
    uint32_t total;
    const uint32_t before = block_excl_scan(ws, scan_lds, &total);
    const uint32_t k = carry + before + ws;  // word starts up to and including p (>= 1: a tile begins with a word)
    if (p < n) st32(&wgt[p], wcnt[word0 + k - 1]);
    __syncthreads();
    if (threadIdx.x == 0) carry += total;
    __syncthreads();
  }
}
}  // namespace

// One workgroup per class-C tile.  MERGE=false: K3 (initial counts).  MERGE=true: apply the batch.
// scratch per tile: site[slot] (rule index + 1 of the merge starting at p), wgt[slot], wgt2[slot], tok2[slot]
template <bool MERGE>
__global__ __launch_bounds__(BLOCK) void k_giant(TileSet ts, unsigned int slot, PairTable pt, DeltaBuf db, const RuleSlot *__restrict__ rules,
                                                 unsigned int rule_mask, uint32_t self_x, uint32_t self_z, uint32_t *__restrict__ scratch,
                                                 unsigned long long *__restrict__ stats) {
  __shared__ uint32_t scan_lds[NWAVES];
  __shared__ unsigned int new_keys, any_site, n_sites;
  __shared__ uint32_t carry;
  for (uint32_t t = blockIdx.x; t < ts.n_tiles; t += gridDim.x) {
    uint32_t *tok = ts.tok + (size_t)t * slot;
    uint32_t *site = scratch + (size_t)t * 4 * slot, *wgt = site + slot, *wgt2 = wgt + slot, *tok2 = wgt2 + slot;
    const int n = (int)ts.tile_len[t];
    const uint32_t word0 = ts.tile_word0[t];
    if (threadIdx.x == 0) { new_keys = 0; any_site = 0; n_sites = 0; }
    __syncthreads();
    if (!MERGE) {
      word_weights(tok, wgt, n, ts.wcnt, word0, scan_lds);
      __syncthreads();
      count_pairs(tok, wgt, n, +1, pt, db, &new_keys);
    } else {
      // ---- merge sites: adjacent pairs that are rules of the batch; x x of the self rule at even offsets of its run
      unsigned int mine = 0;
      for (int p = (int)threadIdx.x; p < n; p += (int)blockDim.x) {
        uint32_t s = 0;
        if (p + 1 < n) {
          const uint32_t t0 = ld32(&tok[p]), t1 = ld32(&tok[p + 1]);
          if (!(t1 & TOK_WS)) {
            const uint32_t a = t0 & TOK_MASK, b = t1 & TOK_MASK;
            if (a == self_x && b == self_x) {
              int q = p;
              while (q > 0 && !(ld32(&tok[q]) & TOK_WS) && (ld32(&tok[q - 1]) & TOK_MASK) == a) q--;
              if (((p - q) & 1) == 0) s = 0xffffffffu;  // marks the self rule
            } else if (rules) {
              const unsigned long long key = pair_key(a, b);
              unsigned int h = pair_hash32(key) & rule_mask;
              for (;;) {
                const unsigned long long k = rules[h].key;
                if (k == key) { s = rules[h].z + 1u; break; }
                if (k == PT_EMPTY) break;
                h = (h + 1) & rule_mask;
              }
            }
          }
        }
        st32(&site[p], s);
        mine += s != 0;
      }
      if (mine) { atomicOr(&any_site, 1u); atomicAdd(&n_sites, mine); }
      __syncthreads();
      if (any_site) {  // uniform
        word_weights(tok, wgt, n, ts.wcnt, word0, scan_lds);
        __syncthreads();
        count_pairs(tok, wgt, n, -1, pt, db, &new_keys, rules, rule_mask, self_x);
        __syncthreads();
        // ---- apply + compact into tok2 / wgt2 (rules of a batch cannot overlap; self-rule sites are two apart)
        if (threadIdx.x == 0) carry = 0;
        __syncthreads();
        for (int p0 = 0; p0 < n; p0 += (int)blockDim.x) {
          const int p = p0 + (int)threadIdx.x;
          uint32_t alive = 0, nt = 0, w = 0;
          if (p < n) {
            const bool dead = p > 0 && ld32(&site[p - 1]) != 0;
            alive = dead ? 0u : 1u;
            const uint32_t s = ld32(&site[p]), t0 = ld32(&tok[p]);
            nt = s ? ((s == 0xffffffffu ? self_z : s - 1u) | (t0 & TOK_WS)) : t0;
            w = ld32(&wgt[p]);
          }
          uint32_t total;
          const uint32_t before = block_excl_scan(alive, scan_lds, &total);
          if (alive) {
            st32(&tok2[carry + before], nt);
            st32(&wgt2[carry + before], w);
          }
          __syncthreads();
          if (threadIdx.x == 0) carry += total;
          __syncthreads();
        }
        const int n2 = (int)carry;
        for (int p = (int)threadIdx.x; p < n2; p += (int)blockDim.x) st32(&tok[p], ld32(&tok2[p]));
        __syncthreads();
        count_pairs(tok, wgt2, n2, +1, pt, db, &new_keys);
        if (threadIdx.x == 0) {
          ts.tile_len[t] = (uint32_t)n2;
          atomicAdd(&stats[0], (unsigned long long)n_sites);
          atomicAdd(&stats[1], 1ull);
          atomicAdd(&stats[3], (unsigned long long)n);
        }
      }
      if (threadIdx.x == 0) atomicAdd(&stats[2], (unsigned long long)n);
    }
    __syncthreads();
    if (threadIdx.x == 0 && new_keys) atomicAdd(pt.n_keys, new_keys);
    __syncthreads();
  }
}

void launch_giant(bool merge, const TileSet &ts, unsigned int slot, const PairTable &pt, const DeltaBuf &db_in, const RuleSlot *rules,
                  unsigned int rule_mask, uint32_t self_x, uint32_t self_z, uint32_t *scratch, unsigned long long *stats, hipStream_t st) {
  if (!ts.n_tiles) return;
  DeltaBuf db = db_in;
  unsigned int g = ts.n_tiles < 1024u ? ts.n_tiles : 1024u;
  if (merge)
    hipLaunchKernelGGL(k_giant<true>, dim3(g), dim3(BLOCK), 0, st, ts, slot, pt, db, rules, rule_mask, self_x, self_z, scratch, stats);
  else
    hipLaunchKernelGGL(k_giant<false>, dim3(g), dim3(BLOCK), 0, st, ts, slot, pt, db, rules, rule_mask, self_x, self_z, scratch, stats);
}

}  // namespace yttm
