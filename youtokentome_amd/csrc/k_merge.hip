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
// non-intersecting rules; one streaming pass over the token tiles applies all of them at once: a
// workgroup stages its tile in LDS, finds merge sites with one cached hash lookup per adjacency, resolves x==y runs by
// parity from the run start, emits exact count deltas only around the sites (aggregated in an LDS hash, then 64-bit
// atomics into the HBM pair table), compacts the tile with wave ballots and writes it back in place.
// HBM-bound integer work: no MFMA.
#include "yttm_device.h"
#include "yttm_kernels.h"

namespace yttm {

constexpr int AGG_SLOTS = 1024;  // LDS delta aggregator (per workgroup)

struct TileLds {
  uint32_t tk[TILE_MAX + 4];
  unsigned long long wsmask[TILE_CHUNKS];
  uint32_t wsbase[TILE_CHUNKS];
  uint32_t tmp[TILE_CHUNKS];
  unsigned long long akey[AGG_SLOTS];
  unsigned long long aval[AGG_SLOTS];
  unsigned int agg_fill;
};

__device__ inline void global_emit(const PairTable &pt, const DeltaBuf &db, unsigned long long key, long long delta) {
  pt_add(pt, key, delta);
  if (db.recs) {
    unsigned long long i = atomicAdd(db.n, 1ull);
    if (i < db.cap) {
      db.recs[i].key = key;
      db.recs[i].delta = delta;
    }
  }
}

// LDS-staged partial counts: most deltas of a tile hit few distinct pairs early in training (small alphabet), so they
// are summed in LDS first and only the per-workgroup totals go to HBM atomics (cdna guide, Guideline 12).
__device__ inline void agg_emit(TileLds &L, const PairTable &pt, const DeltaBuf &db, unsigned long long key, long long delta) {
  unsigned int h = (unsigned int)(mix64(key) >> 24) & (AGG_SLOTS - 1);
  for (int probe = 0; probe < 8; probe++) {
    unsigned long long k = ((volatile unsigned long long *)L.akey)[h];
    if (k == PT_EMPTY) {
      k = atomicCAS(&L.akey[h], PT_EMPTY, key);
      if (k == PT_EMPTY) {
        atomicAdd(&L.agg_fill, 1u);
        k = key;
      }
    }
    if (k == key) {
      atomicAdd(&L.aval[h], (unsigned long long)delta);
      return;
    }
    h = (h + 1) & (AGG_SLOTS - 1);
  }
  global_emit(pt, db, key, delta);
}

__device__ inline void agg_init(TileLds &L) {
  for (int s = (int)threadIdx.x; s < AGG_SLOTS; s += BLOCK) {
    L.akey[s] = PT_EMPTY;
    L.aval[s] = 0;
  }
  if (threadIdx.x == 0) L.agg_fill = 0;
}

__device__ inline void agg_flush(TileLds &L, const PairTable &pt, const DeltaBuf &db) {
  __syncthreads();
  for (int s = (int)threadIdx.x; s < AGG_SLOTS; s += BLOCK) {
    unsigned long long k = L.akey[s];
    if (k != PT_EMPTY) {
      long long v = (long long)L.aval[s];
      if (v != 0) global_emit(pt, db, k, v);
      L.akey[s] = PT_EMPTY;
      L.aval[s] = 0;
    }
  }
  if (threadIdx.x == 0) L.agg_fill = 0;
  __syncthreads();
}

// Stage tile `t` into LDS and build the word-start masks / per-chunk word index bases.  Returns live length n.
__device__ inline int tile_load(TileLds &L, const TileSet &ts, uint32_t t) {
  const int n = (int)ts.tile_len[t];
  const unsigned long long base = ts.tile_start[t];
  for (int p = (int)threadIdx.x; p < n; p += BLOCK) L.tk[p] = ts.tok[base + p];
  if (threadIdx.x == 0) {
    L.tk[n] = TOK_WS;  // sentinel: "next token starts a word" => no adjacency past the end
    L.tk[n + 1] = TOK_WS;
    L.tk[n + 2] = TOK_WS;
  }
  __syncthreads();
  const int nchunks = (n + 63) >> 6;
  const int wave = (int)(threadIdx.x >> 6), lane = lane_id();
  for (int c = wave; c < nchunks; c += NWAVES) {
    int p = c * 64 + lane;
    bool ws = p < n && (L.tk[p] & TOK_WS);
    unsigned long long m = __ballot(ws);
    if (lane == 0) {
      L.wsmask[c] = m;
      L.tmp[c] = (uint32_t)__popcll(m);
    }
  }
  __syncthreads();
  if (wave == 0) {
    uint32_t v = lane < nchunks ? L.tmp[lane] : 0;
    uint32_t inc = wave_incl_scan(v);
    if (lane < nchunks) L.wsbase[lane] = inc - v;
  }
  __syncthreads();
  return n;
}

// frequency of the word that contains tile position p
__device__ inline long long tile_weight(const TileLds &L, const TileSet &ts, uint32_t t, int p) {
  int c = p >> 6;
  unsigned long long le = (2ull << (p & 63)) - 1ull;  // bits 0..(p&63)
  uint32_t k = L.wsbase[c] + (uint32_t)__popcll(L.wsmask[c] & le);
  return (long long)ts.wcnt[ts.tile_word0[t] + k - 1];
}

// ------------------------------------------------------------------------------------------------- K3: pair count
// Weighted bigram histogram of the whole token table: every adjacency counts the word frequency;
// inside a run of L equal tokens the self pair counts floor(L/2) (emitted once by the run's first token).
__global__ __launch_bounds__(BLOCK) void k3_pair_count(TileSet ts, PairTable pt, DeltaBuf db) {
  __shared__ TileLds L;
  agg_init(L);
  __syncthreads();
  for (uint32_t t = blockIdx.x; t < ts.n_tiles; t += gridDim.x) {
    const int n = tile_load(L, ts, t);
    for (int p = (int)threadIdx.x; p < n; p += BLOCK) {
      const uint32_t t0 = L.tk[p], t1 = L.tk[p + 1];
      if (t1 & TOK_WS) continue;
      const uint32_t a = t0 & TOK_MASK, b = t1 & TOK_MASK;
      if (a != b) {
        agg_emit(L, pt, db, pair_key(a, b), tile_weight(L, ts, t, p));
      } else {
        const bool run_start = (t0 & TOK_WS) || p == 0 || (L.tk[p - 1] & TOK_MASK) != a;
        if (run_start) {
          int q = p + 1;
          while (!(L.tk[q + 1] & TOK_WS) && (L.tk[q + 1] & TOK_MASK) == a) q++;
          const long long len = q - p + 1;
          agg_emit(L, pt, db, pair_key(a, a), (len / 2) * tile_weight(L, ts, t, p));
        }
      }
    }
    __syncthreads();
    if (L.agg_fill > AGG_SLOTS / 2) agg_flush(L, pt, db);
    __syncthreads();
  }
  agg_flush(L, pt, db);
}

// ------------------------------------------------------------------------------------------------- K4: merge apply
struct MergeLds {
  TileLds t;
  uint32_t nz[TILE_MAX];                     // new token (z | inherited TOK_WS) at merge-site positions
  unsigned long long sitemask[TILE_CHUNKS];  // bit p: a merge (tk[p],tk[p+1]) -> nz[p] starts at p
  unsigned long long amask[TILE_CHUNKS];     // bit p: position p survives
  uint32_t abase[TILE_CHUNKS];
  int any_site;
  uint32_t new_len;
};

__global__ __launch_bounds__(BLOCK) void k4_merge_apply(TileSet ts, PairTable pt, DeltaBuf db, const RuleSlot *__restrict__ rules,
                                                        unsigned int rule_mask, const uint8_t *__restrict__ tokflag,
                                                        uint32_t self_x, uint32_t self_z,
                                                        unsigned long long *__restrict__ stats /* [0]=sites [1]=tiles touched [2]=tokens scanned [3]=tokens in touched tiles */) {
  __shared__ MergeLds M;
  TileLds &L = M.t;
  agg_init(L);
  __syncthreads();
  const int wave = (int)(threadIdx.x >> 6), lane = lane_id();
  for (uint32_t t = blockIdx.x; t < ts.n_tiles; t += gridDim.x) {
    if (threadIdx.x == 0) M.any_site = 0;
    const int n = tile_load(L, ts, t);  // contains barriers
    const int nchunks = (n + 63) >> 6;

    // ---- phase 1: merge sites -------------------------------------------------------------------------------------
    for (int c = wave; c < nchunks; c += NWAVES) {
      const int p = c * 64 + lane;
      bool s = false;
      if (p < n) {
        const uint32_t t0 = L.tk[p], t1 = L.tk[p + 1];
        if (!(t1 & TOK_WS)) {
          const uint32_t a = t0 & TOK_MASK, b = t1 & TOK_MASK;
          uint32_t z = 0;
          if (a == self_x && b == self_x) {
            // x==y rule: left-to-right greedy inside the run = positions at even offset from the run start
            int r = p;
            while (r > 0 && !(L.tk[r] & TOK_WS) && (L.tk[r - 1] & TOK_MASK) == a) r--;
            if (((p - r) & 1) == 0) { s = true; z = self_z; }
          } else if ((tokflag[a] & 1u) && (tokflag[b] & 2u)) {
            const unsigned long long key = pair_key(a, b);
            unsigned int h = (unsigned int)mix64(key) & rule_mask;
            for (;;) {
              const unsigned long long k = rules[h].key;
              if (k == key) { s = true; z = rules[h].z; break; }
              if (k == PT_EMPTY) break;
              h = (h + 1) & rule_mask;
            }
          }
          if (s) M.nz[p] = z | (t0 & TOK_WS);
        }
      }
      const unsigned long long m = __ballot(s);
      if (lane == 0) {
        M.sitemask[c] = m;
        if (m) M.any_site = 1;
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&stats[2], (unsigned long long)n);
    if (!M.any_site) {
      __syncthreads();  // keep any_site stable until everyone has read it
      continue;
    }

#define SITE(q) ((q) >= 0 && (((M.sitemask[(q) >> 6] >> ((q)&63)) & 1ull) != 0))
    // ---- phase 2: count deltas around the sites + survivor masks -----------------------------------------------------
    unsigned long long my_sites = 0;
    for (int c = wave; c < nchunks; c += NWAVES) {
      const int p = c * 64 + lane;
      bool alive = false;
      if (p < n) {
        const uint32_t t0 = L.tk[p], t1 = L.tk[p + 1];
        const uint32_t a = t0 & TOK_MASK;
        const bool sp = SITE(p);
        const bool dp = SITE(p - 1);
        const bool adj1 = !(t1 & TOK_WS);
        alive = !dp;
        if (sp || dp || (adj1 && SITE(p + 1))) {
          const long long f = tile_weight(L, ts, t, p);
          if (sp) {
            my_sites++;
            const uint32_t b = t1 & TOK_MASK;
            const uint32_t z = M.nz[p] & TOK_MASK;
            agg_emit(L, pt, db, pair_key(a, b), -f);  // the merged pair itself
            // run of new z tokens (x y x y ... or the halves of an x-run): counted floor(Lz/2) by its first site
            const bool prev_same = p >= 2 && !(t0 & TOK_WS) && SITE(p - 2) && (M.nz[p - 2] & TOK_MASK) == z;
            if (!prev_same) {
              int q = p, lz = 1;
              while (!(L.tk[q + 2] & TOK_WS) && q + 2 < n && SITE(q + 2) && (M.nz[q + 2] & TOK_MASK) == z) { q += 2; lz++; }
              if (lz >= 2) agg_emit(L, pt, db, pair_key(z, z), (long long)(lz / 2) * f);
            }
            // new adjacency (z, right neighbour)
            const int q = p + 2;
            if (q < n && !(L.tk[q] & TOK_WS)) {
              const uint32_t B = SITE(q) ? (M.nz[q] & TOK_MASK) : (L.tk[q] & TOK_MASK);
              if (B != z) agg_emit(L, pt, db, pair_key(z, B), f);
            }
            // x != y rule whose x is the last token of a run of a's: the run shrinks by one
            if (a != self_x && p > 0 && !(t0 & TOK_WS) && (L.tk[p - 1] & TOK_MASK) == a) {
              int r = p;
              while (r > 0 && !(L.tk[r] & TOK_WS) && (L.tk[r - 1] & TOK_MASK) == a) r--;
              const int len = p - r + 1;
              if ((len & 1) == 0) agg_emit(L, pt, db, pair_key(a, a), -f);
            }
          } else if (!dp) {
            // unmerged token whose right neighbour starts a site: (a,x) -> (a,z)
            const uint32_t x_ = t1 & TOK_MASK;
            const uint32_t z = M.nz[p + 1] & TOK_MASK;
            if (a != x_) agg_emit(L, pt, db, pair_key(a, x_), -f);
            agg_emit(L, pt, db, pair_key(a, z), f);
          }
          if (dp && adj1) {
            // p was the y of the site at p-1: its old right adjacency disappears
            const uint32_t b_ = t1 & TOK_MASK;
            if (a != b_) {
              agg_emit(L, pt, db, pair_key(a, b_), -f);
            } else if (a != self_x) {
              // x != y rule whose y is the first token of a run of a's: the run shrinks by one
              int q = p;
              while (!(L.tk[q + 1] & TOK_WS) && (L.tk[q + 1] & TOK_MASK) == a) q++;
              const int len = q - p + 1;
              if ((len & 1) == 0) agg_emit(L, pt, db, pair_key(a, a), -f);
            }
          }
        }
      }
      const unsigned long long am = __ballot(alive);
      if (lane == 0) {
        M.amask[c] = am;
        L.tmp[c] = (uint32_t)__popcll(am);
      }
    }
    __syncthreads();
    if (wave == 0) {
      uint32_t v = lane < nchunks ? L.tmp[lane] : 0;
      uint32_t inc = wave_incl_scan(v);
      if (lane < nchunks) M.abase[lane] = inc - v;
      if (lane == 63) M.new_len = inc;
    }
    __syncthreads();
    // ---- phase 3: compact in place (all reads come from LDS, so overwriting the tile in HBM is safe) ------------------
    const unsigned long long base = ts.tile_start[t];
    for (int p = (int)threadIdx.x; p < n; p += BLOCK) {
      const int c = p >> 6;
      const unsigned long long am = M.amask[c];
      if ((am >> (p & 63)) & 1ull) {
        const uint32_t np = M.abase[c] + (uint32_t)__popcll(am & ((1ull << (p & 63)) - 1ull));
        ts.tok[base + np] = SITE(p) ? M.nz[p] : L.tk[p];
      }
    }
    if (threadIdx.x == 0) {
      ts.tile_len[t] = M.new_len;
      atomicAdd(&stats[1], 1ull);
      atomicAdd(&stats[3], (unsigned long long)n);
    }
    my_sites = wave_sum_u64(my_sites);
    if (lane == 0 && my_sites) atomicAdd(&stats[0], my_sites);
#undef SITE
    __syncthreads();
    if (L.agg_fill > AGG_SLOTS / 2) agg_flush(L, pt, db);
    __syncthreads();
  }
  agg_flush(L, pt, db);
}

// ------------------------------------------------------------------------------------------------- pair table kernels
__device__ inline int cand_bin(unsigned long long c) {
  if (c < 256) return (int)c;
  int e = 63 - __clzll((long long)c);  // >= 8
  int m3 = (int)((c >> (e - 3)) & 7ull);
  return 256 + (e - 8) * 8 + m3;
}

// Candidate filter: appends every pair with (count > tau_cnt) or (count == tau_cnt and max(x,y) <= tau_mx) and
// histograms all live counts (CAND_BINS log-ish bins) so the host can choose the next threshold.
__global__ __launch_bounds__(BLOCK) void k_cand_scan(PairTable pt, unsigned long long tau_cnt, uint32_t tau_mx,
                                                     CandRec *__restrict__ out, unsigned int cap, unsigned int *__restrict__ n_out,
                                                     unsigned long long *__restrict__ hist) {
  __shared__ unsigned int lh[CAND_BINS];
  for (int b = (int)threadIdx.x; b < CAND_BINS; b += BLOCK) lh[b] = 0;
  __syncthreads();
  const unsigned long long n_slots = pt.mask + 1;
  const unsigned long long n_iter = (n_slots + BLOCK - 1) / BLOCK;
  for (unsigned long long it = blockIdx.x; it < n_iter; it += gridDim.x) {
    const unsigned long long i = it * BLOCK + threadIdx.x;
    bool pass = false;
    unsigned long long k = PT_EMPTY, c = 0;
    if (i < n_slots) {
      k = pt.keys[i];
      c = pt.cnts[i];
      if (k != PT_EMPTY && c > 0) {
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
      base = __shfl(base, 0);
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

__global__ __launch_bounds__(BLOCK) void k_pt_rehash(PairTable src, PairTable dst) {
  const unsigned long long n_slots = src.mask + 1;
  unsigned long long i = (unsigned long long)blockIdx.x * BLOCK + threadIdx.x;
  const unsigned long long stride = (unsigned long long)gridDim.x * BLOCK;
  for (; i < n_slots; i += stride) {
    unsigned long long k = src.keys[i];
    if (k == PT_EMPTY) continue;
    unsigned long long c = src.cnts[i];
    if (c) pt_add(dst, k, (long long)c);  // dead pairs (count 0) can never come back: drop them
  }
}

__global__ __launch_bounds__(BLOCK) void k_pt_query(PairTable pt, const unsigned long long *__restrict__ keys, unsigned int n,
                                                    unsigned long long *__restrict__ out) {
  unsigned int i = blockIdx.x * BLOCK + threadIdx.x;
  if (i < n) out[i] = pt_get(pt, keys[i]);
}

// multi-GPU: fold the count deltas received from the other ranks into the local replica of the global pair table
__global__ __launch_bounds__(BLOCK) void k_pt_apply(PairTable pt, const DeltaRec *__restrict__ recs, unsigned long long n) {
  unsigned long long i = (unsigned long long)blockIdx.x * BLOCK + threadIdx.x;
  const unsigned long long stride = (unsigned long long)gridDim.x * BLOCK;
  for (; i < n; i += stride) pt_add(pt, recs[i].key, recs[i].delta);
}

__global__ __launch_bounds__(BLOCK) void k_set_tokflag(uint8_t *__restrict__ tokflag, const uint32_t *__restrict__ upd, unsigned int n) {
  unsigned int i = blockIdx.x * BLOCK + threadIdx.x;
  if (i < n) tokflag[upd[2 * i]] = (uint8_t)upd[2 * i + 1];
}

__global__ __launch_bounds__(BLOCK) void k_fill_u64(unsigned long long *__restrict__ p, unsigned long long v, unsigned long long n) {
  unsigned long long i = (unsigned long long)blockIdx.x * BLOCK + threadIdx.x;
  const unsigned long long stride = (unsigned long long)gridDim.x * BLOCK;
  for (; i < n; i += stride) p[i] = v;
}

// ------------------------------------------------------------------------------------------------- launchers
static inline unsigned int tile_grid(unsigned int n_tiles, unsigned int per_cu) {
  unsigned int g = 256u * per_cu;
  if (g > n_tiles) g = n_tiles;
  return g ? g : 1u;
}

void launch_pair_count(const TileSet &ts, const PairTable &pt, const DeltaBuf &db, hipStream_t st) {
  if (!ts.n_tiles) return;
  hipLaunchKernelGGL(k3_pair_count, dim3(tile_grid(ts.n_tiles, 8)), dim3(BLOCK), 0, st, ts, pt, db);
}
void launch_merge_apply(const TileSet &ts, const PairTable &pt, const DeltaBuf &db, const RuleSlot *rules, unsigned int rule_mask,
                        const uint8_t *tokflag, uint32_t self_x, uint32_t self_z, unsigned long long *stats, hipStream_t st) {
  if (!ts.n_tiles) return;
  hipLaunchKernelGGL(k4_merge_apply, dim3(tile_grid(ts.n_tiles, 6)), dim3(BLOCK), 0, st, ts, pt, db, rules, rule_mask, tokflag, self_x,
                     self_z, stats);
}
void launch_cand_scan(const PairTable &pt, unsigned long long tau_cnt, uint32_t tau_mx, CandRec *out, unsigned int cap,
                      unsigned int *n_out, unsigned long long *hist, hipStream_t st) {
  unsigned long long n_slots = pt.mask + 1;
  unsigned long long b = (n_slots + BLOCK - 1) / BLOCK;
  if (b > 256 * 16) b = 256 * 16;
  hipLaunchKernelGGL(k_cand_scan, dim3((unsigned int)b), dim3(BLOCK), 0, st, pt, tau_cnt, tau_mx, out, cap, n_out, hist);
}
void launch_pt_rehash(const PairTable &src, const PairTable &dst, hipStream_t st) {
  unsigned long long n_slots = src.mask + 1;
  unsigned long long b = (n_slots + BLOCK - 1) / BLOCK;
  if (b > 256 * 16) b = 256 * 16;
  hipLaunchKernelGGL(k_pt_rehash, dim3((unsigned int)b), dim3(BLOCK), 0, st, src, dst);
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
void launch_set_tokflag(uint8_t *tokflag, const uint32_t *upd, unsigned int n, hipStream_t st) {
  if (!n) return;
  hipLaunchKernelGGL(k_set_tokflag, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st, tokflag, upd, n);
}
void launch_fill_u64(unsigned long long *p, unsigned long long v, unsigned long long n, hipStream_t st) {
  if (!n) return;
  unsigned long long b = (n + BLOCK - 1) / BLOCK;
  if (b > 256 * 16) b = 256 * 16;
  hipLaunchKernelGGL(k_fill_u64, dim3((unsigned int)b), dim3(BLOCK), 0, st, p, v, n);
}

}  // namespace yttm
