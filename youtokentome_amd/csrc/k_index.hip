// k_index.hip -- the pair index (pair -> the words / tiles that hold it; the reference's pair2pos, bpe.cpp:438/:626/:694): seeded from the hot
// list, built by two streaming passes over the token slots (count, fill).  (Until round 4 part of k_merge.hip.)
#include "k_tile_core.h"
#include "k_index_core.h"

namespace yttm {

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
}  // namespace yttm
