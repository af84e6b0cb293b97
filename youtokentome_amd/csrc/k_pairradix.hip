// k_pairradix.hip -- K3 for LARGE alphabets (round 6): the weighted bigram histogram of the class-A tiles (pair2cnt of build_linked_list,
// bpe.cpp:461-475, summed over the threads :1076-1088) without one random 64-bit atomic per adjacency.
//
// With a few thousand symbols (CJK: 4 096 ideographs) the pairs are millions, no per-workgroup LDS table holds more than a fraction of
// them, and the general kernel (k_tiles<.., false, ..>: a 256-slot LDS hash, then pt_add) ends up at the chip's random-atomic rate:
// 3.4e8 adjacencies in 29 ms = 1.2e10 atomics/s, 0.006 of the HBM roofline (HISTORY.md, "Measured, round 5").  Here every adjacency
// becomes an 8-byte RECORD {a << 16 | b, weight} (ids relative to id_min: the path takes alphabets of up to K3R_MAX_IDS symbols) and the
// records are partitioned by their FIRST token in two levels -- streaming passes whose writes are runs, not single words -- until all
// pairs (a, *) of one `a` lie together; then a workgroup sums them in a DENSE LDS array indexed by b (ds_add_u64, no hash, no probe) and only
// the distinct pairs reach the pair table (global_emit: one probe + one atomic per DISTINCT pair and chunk, multi-GPU deltas included).
//
//   k3r_hist      tiles -> records per first token (LDS histogram per workgroup, one global add per token and workgroup)
//   k3r_offsets   exclusive scan: the final run of every `a` (offA), the level-1 run of every group of 2^s1 consecutive a's
//   k3r_wgoffsets every workgroup's OWN run of every level-1 group (k3r_hist left its counts per group: the scatter below visits the same tiles)
//   k3r_scatter1  tiles -> buf1, grouped by a >> s1 (<= 256 groups): a record's place comes from a cursor in LDS -- no global atomic at all
//                 (the first cut reserved a run per batch of four tiles and group with a global add: 4.2e7 returning adds on 129 addresses,
//                 8.8 ms of the path's 13.5; profiles/r6_k3_radix.txt)
//   k3r_scatter2  buf1 -> buf2, grouped by a: a chunk of buf1 holds one or two groups, so a few dozen distinct a's -- same scheme
//   k3r_final     buf2 -> pair table: per chunk and `a`, dense LDS counts over b, flushed (and cleared) through global_emit
//
// Algorithmic bytes stay 4T + 8U; what the path MOVES is 2 x 4T (tiles, twice) + 4 x 8T (records written and read, twice).
// The adjacency enumeration (word of a position from the ballot of the word-start bits, a run of L equal tokens counting floor(L/2)
// through carry arithmetic on the ballot) is k_pair_count_dense's (k_tiles.hip), token for token.
#include "k_tile_core.h"

namespace yttm {

constexpr int K3R_NT = 256, K3R_WAVES = K3R_NT / 64;
constexpr uint32_t K3R_CHUNK = 4096;        // records per work item of k3r_scatter2 (16 per thread, in registers)
constexpr uint32_t K3R_FCHUNK = 32768;      // records per work item of k3r_final
constexpr uint32_t K3R_DIRECT_MAX = 256;    // a segment of at most this many records goes to the pair table one by one (no dense pass)

__host__ __device__ inline uint32_t k3r_shift1(uint32_t n_ids) {  // level 1: at most 256 groups of 2^s1 consecutive first tokens
  uint32_t s = 0;
  while (((n_ids - 1u) >> s) + 1u > 256u) s++;
  return s;
}

// The adjacencies of one tile, in the dense kernel's own order: emit(c, a, b, weight) for the adjacency that starts at position 64 c + lane.
template <int SLOT, class Emit>
__device__ inline void k3r_tile(const TileSet &ts, uint32_t tile, uint32_t *lw /* [64 * WReg::N], this wave's */, Emit &&emit) {
  constexpr int NC = SLOT / 64, NW = WReg<SLOT>::N;
  const int lane = lane_id();
  const int n = (int)ts.tile_len[tile];
  const uint32_t w0 = ts.tile_word0[tile];
  const uint32_t *src = ts.tok + (size_t)tile * SLOT;
  uint32_t r[NC];
#pragma unroll
  for (int c = 0; c < NC; c++) {
    const uint32_t v = src[64 * c + lane];
    r[c] = 64 * c + lane < n ? v : TOK_WS;
  }
  WReg<SLOT> w;
  wreg_load<SLOT>(w, ts.wcnt, w0);
  wave_sync();  // (the previous tile's reads of the window are done)
#pragma unroll
  for (int i = 0; i < NW; i++) lw[lane + 64 * i] = w.v[i];
  wave_sync();
  uint32_t wbase = 0xffffffffu;
  bool cont = false, cont_even = false;
#pragma unroll
  for (int c = 0; c < NC; c++) {
    const uint32_t t0 = r[c];
    uint32_t t1 = from_lane_right(t0);
    const uint32_t first_next = c + 1 < NC ? from_lane0(r[c + 1 < NC ? c + 1 : c]) : TOK_WS;
    if (lane == 63) t1 = first_next;
    const unsigned long long m_ws = ballot_b((int)t0 < 0);
    const uint32_t k = wbase + lanes_below(m_ws) + (t0 >> 31);
    wbase += (uint32_t)__popcll(m_ws);
    const bool adj = (int)t1 >= 0;
    const uint32_t a = t0 & TOK_MASK;
    const bool eq = a == t1;
    const unsigned long long E = ballot_b(adj && eq);
    const unsigned long long S = E & ~((E << 1) | (cont ? 1ull : 0ull));
    const unsigned long long S_e = (S & 0x5555555555555555ull) | (cont_even ? (E & 1ull) : 0ull);
    const unsigned long long D = (E + S_e) ^ E;
    const unsigned long long sel = (D & E & 0x5555555555555555ull) | (~D & E & 0xaaaaaaaaaaaaaaaaull);
    cont = (E >> 63) != 0ull;
    cont_even = cont && !(sel >> 63);
    if (adj && (!eq || lane_bit(sel))) emit(c, a, t1, lw[k]);
  }
}

template <int SLOT>
__global__ __launch_bounds__(K3R_NT) void k3r_hist(TileSet ts, uint32_t id_min, uint32_t n_ids, uint32_t *__restrict__ hA, uint32_t s1, uint32_t nb1,
                                                   uint32_t *__restrict__ wgcnt /* [nb1][gridDim.x]: this workgroup's records per level-1 group */) {
  __shared__ uint32_t h[K3R_MAX_IDS];
  __shared__ uint32_t wwin[K3R_WAVES][64 * WReg<SLOT>::N];
  for (uint32_t i = threadIdx.x; i < n_ids; i += K3R_NT) h[i] = 0;
  __syncthreads();
  const uint32_t n_waves = gridDim.x * K3R_WAVES;
  for (uint32_t t = uni(blockIdx.x * K3R_WAVES + (uint32_t)(threadIdx.x >> 6)); t < ts.n_tiles; t += n_waves)
    k3r_tile<SLOT>(ts, t, wwin[threadIdx.x >> 6], [&](int, uint32_t a, uint32_t, uint32_t) { atomicAdd(&h[a - id_min], 1u); });
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < n_ids; i += K3R_NT)
    if (h[i]) atomicAdd(&hA[i], h[i]);
  for (uint32_t gq = threadIdx.x; gq < nb1; gq += K3R_NT) {
    uint32_t c = 0;
    for (uint32_t i = gq << s1; i < ((gq + 1u) << s1) && i < n_ids; i++) c += h[i];
    wgcnt[(size_t)gq * gridDim.x + blockIdx.x] = c;
  }
}
// workgroup g of this launch: level-1 group g -- where each workgroup of k3r_hist / k3r_scatter1 puts its records of the group
__global__ __launch_bounds__(K3R_NT) void k3r_wgoffsets(const uint32_t *__restrict__ wgcnt, uint32_t n_wg, uint32_t s1, const uint32_t *__restrict__ offA,
                                                        uint32_t *__restrict__ wgoff) {
  __shared__ uint32_t part[K3R_NT];
  const uint32_t *c = wgcnt + (size_t)blockIdx.x * n_wg;
  uint32_t *o = wgoff + (size_t)blockIdx.x * n_wg;
  const uint32_t per = (n_wg + K3R_NT - 1) / K3R_NT, lo = threadIdx.x * per, hi = lo + per < n_wg ? lo + per : n_wg;
  uint32_t sum = 0;
  for (uint32_t i = lo; i < hi; i++) sum += c[i];
  part[threadIdx.x] = sum;
  __syncthreads();
  uint32_t before = offA[blockIdx.x << s1];
  for (uint32_t j = 0; j < threadIdx.x; j++) before += part[j];
  for (uint32_t i = lo; i < hi; i++) {
    o[i] = before;
    before += c[i];
  }
}

// one workgroup: offA[i] = records of the first tokens below i (offA[n_ids] = all of them); cur2 = the cursors of the second scatter pass
__global__ __launch_bounds__(K3R_NT) void k3r_offsets(const uint32_t *__restrict__ hA, uint32_t n_ids, uint32_t *__restrict__ offA,
                                                      uint32_t *__restrict__ cur2) {
  __shared__ uint32_t part[K3R_NT];
  const uint32_t per = (n_ids + K3R_NT - 1) / K3R_NT, lo = threadIdx.x * per, hi = lo + per < n_ids ? lo + per : n_ids;
  uint32_t s = 0;
  for (uint32_t i = lo; i < hi; i++) s += hA[i];
  part[threadIdx.x] = s;
  __syncthreads();
  uint32_t before = 0;
  for (uint32_t j = 0; j < threadIdx.x; j++) before += part[j];
  for (uint32_t i = lo; i < hi; i++) {
    offA[i] = before;
    cur2[i] = before;
    before += hA[i];
  }
  if (hi == n_ids && lo < n_ids) offA[n_ids] = before;
  if (n_ids == 0 && threadIdx.x == 0) offA[0] = 0;
}

template <int SLOT>
__global__ __launch_bounds__(K3R_NT) void k3r_scatter1(TileSet ts, uint32_t id_min, uint32_t s1, uint32_t nb1, const uint32_t *__restrict__ wgoff,
                                                       unsigned long long *__restrict__ buf1) {
  __shared__ uint32_t cur[256];  // next place of this workgroup's records of each group (its runs were laid out by k3r_wgoffsets)
  __shared__ uint32_t wwin[K3R_WAVES][64 * WReg<SLOT>::N];
  for (uint32_t gq = threadIdx.x; gq < nb1; gq += K3R_NT) cur[gq] = wgoff[(size_t)gq * gridDim.x + blockIdx.x];
  __syncthreads();
  const uint32_t n_waves = gridDim.x * K3R_WAVES;
  // (the same tiles as this workgroup of k3r_hist took: same grid, same loop)
  for (uint32_t t = uni(blockIdx.x * K3R_WAVES + (uint32_t)(threadIdx.x >> 6)); t < ts.n_tiles; t += n_waves)
    k3r_tile<SLOT>(ts, t, wwin[threadIdx.x >> 6], [&](int, uint32_t a, uint32_t b, uint32_t w) {
      const uint32_t ar = a - id_min;
      buf1[atomicAdd(&cur[ar >> s1], 1u)] = ((unsigned long long)((ar << 16) | (b - id_min)) << 32) | w;
    });
}

// buf1 (grouped by a >> s1) -> buf2 (grouped by a).  A chunk of buf1 lies in a few consecutive groups: its first tokens span [a_lo, a_hi).
__global__ __launch_bounds__(K3R_NT) void k3r_scatter2(const unsigned long long *__restrict__ buf1, const uint32_t *__restrict__ offA,
                                                       uint32_t n_ids, uint32_t s1, uint32_t *__restrict__ cur2, unsigned long long *__restrict__ buf2) {
  constexpr int PER = K3R_CHUNK / K3R_NT;
  __shared__ uint32_t cntA[K3R_MAX_IDS], cntC[K3R_MAX_IDS];
  __shared__ uint32_t span[2];
  const uint32_t total = offA[n_ids];
  const uint32_t n_chunks = (total + K3R_CHUNK - 1) / K3R_CHUNK;
  for (uint32_t i = threadIdx.x; i < n_ids; i += K3R_NT) { cntA[i] = 0; cntC[i] = 0; }
  __syncthreads();
  for (uint32_t ch = blockIdx.x; ch < n_chunks; ch += gridDim.x) {
    const uint32_t lo = ch * K3R_CHUNK, hi = lo + K3R_CHUNK < total ? lo + K3R_CHUNK : total;
    unsigned long long rec[PER];
#pragma unroll
    for (int j = 0; j < PER; j++) {
      const uint32_t i = lo + (uint32_t)j * K3R_NT + threadIdx.x;
      rec[j] = i < hi ? buf1[i] : ~0ull;
    }
    if (threadIdx.x == 0) {  // (records are ordered by group: the first and the last record bound the first tokens of the chunk)
      span[0] = (uint32_t)(((uint32_t)(buf1[lo] >> 48) >> s1) << s1);
      const uint32_t e = (((uint32_t)(buf1[hi - 1] >> 48) >> s1) + 1u) << s1;
      span[1] = e < n_ids ? e : n_ids;
    }
#pragma unroll
    for (int j = 0; j < PER; j++)
      if (rec[j] != ~0ull) atomicAdd(&cntA[(uint32_t)(rec[j] >> 48)], 1u);
    __syncthreads();
    const uint32_t a_lo = span[0], a_hi = span[1];
    for (uint32_t a = a_lo + threadIdx.x; a < a_hi; a += K3R_NT) {
      const uint32_t c = cntA[a];
      // (cntC[a] becomes the START of the workgroup's run of a; the scatter below bumps it)
      cntC[a] = c ? atomicAdd(&cur2[a], c) : 0u;
      cntA[a] = 0;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PER; j++)
      if (rec[j] != ~0ull) buf2[atomicAdd(&cntC[(uint32_t)(rec[j] >> 48)], 1u)] = rec[j];
    __syncthreads();  // (span and cntC are rewritten by the next chunk)
  }
}

// buf2 (grouped by first token) -> the pair table
__global__ __launch_bounds__(K3R_NT) void k3r_final(const unsigned long long *__restrict__ buf2, const uint32_t *__restrict__ offA, uint32_t id_min,
                                                    uint32_t n_ids, PairTable pt, DeltaBuf db) {
  __shared__ unsigned long long dense[K3R_MAX_IDS];
  __shared__ unsigned int new_keys;
  for (uint32_t i = threadIdx.x; i < n_ids; i += K3R_NT) dense[i] = 0;
  if (threadIdx.x == 0) new_keys = 0;
  __syncthreads();
  const uint32_t total = offA[n_ids];
  const uint32_t n_chunks = (total + K3R_FCHUNK - 1) / K3R_FCHUNK;
  for (uint32_t ch = blockIdx.x; ch < n_chunks; ch += gridDim.x) {
    const uint32_t lo = ch * K3R_FCHUNK, hi = lo + K3R_FCHUNK < total ? lo + K3R_FCHUNK : total;
    uint32_t a = (uint32_t)(buf2[lo] >> 48);  // (uniform: every thread reads the same word)
    for (uint32_t pos = lo; pos < hi;) {
      while (offA[a + 1] <= pos) a++;  // (first tokens without records)
      const uint32_t end = offA[a + 1] < hi ? offA[a + 1] : hi;
      if (end - pos <= K3R_DIRECT_MAX) {
        for (uint32_t i = pos + threadIdx.x; i < end; i += K3R_NT) {
          const unsigned long long r = buf2[i];
          global_emit(pt, db, pair_key(id_min + a, id_min + ((uint32_t)(r >> 32) & 0xffffu)), (long long)(uint32_t)r, &new_keys);
        }
      } else {
        for (uint32_t i = pos + threadIdx.x; i < end; i += K3R_NT) {
          const unsigned long long r = buf2[i];
          atomicAdd(&dense[(uint32_t)(r >> 32) & 0xffffu], (unsigned long long)(uint32_t)r);
        }
        __syncthreads();
        for (uint32_t b = threadIdx.x; b < n_ids; b += K3R_NT) {
          const unsigned long long v = dense[b];
          if (v) {
            dense[b] = 0;
            global_emit(pt, db, pair_key(id_min + a, id_min + b), (long long)v, &new_keys);
          }
        }
        __syncthreads();
      }
      pos = end;
      a++;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0 && new_keys) atomicAdd(pt.n_keys, new_keys);
}

constexpr unsigned int K3R_GRID_MAX = 1024;  // (tile_grid(.., 4 waves, 4 per CU))
size_t pair_count_radix_scratch_u32(uint32_t n_ids) { return (size_t)3 * n_ids + 256 + 8 + (size_t)2 * 256 * K3R_GRID_MAX; }  // hA | offA (+1) | cur2 | wgcnt | wgoff
bool pair_count_radix_takes(uint32_t n_ids, unsigned long long n_tokens) { return n_ids > 64u && n_ids <= K3R_MAX_IDS && n_tokens < (1ull << 31); }
void launch_pair_count_radix(const TileSet &ts, const PairTable &pt, const DeltaBuf &db, uint32_t id_min, uint32_t n_ids, uint32_t *scratch,
                             unsigned long long *buf1, unsigned long long *buf2, unsigned long long n_tokens, hipStream_t st) {
  if (!ts.n_tiles || !n_ids) return;
  const uint32_t s1 = k3r_shift1(n_ids), nb1 = ((n_ids - 1u) >> s1) + 1u;
  uint32_t *hA = scratch, *offA = scratch + n_ids, *cur2 = offA + n_ids + 1, *wgcnt = cur2 + n_ids + 7, *wgoff = wgcnt + (size_t)256 * K3R_GRID_MAX;
  // (hA = scratch[0 .. n_ids) arrives zeroed: the caller's memset)
  const unsigned int g = tile_grid(ts.n_tiles, K3R_WAVES, 4);  // (<= K3R_GRID_MAX)
  static_assert(K3R_GRID_MAX == 256u * 4u, "tile_grid(n, waves, 4 per CU) <= 1024");
  hipLaunchKernelGGL((k3r_hist<TILE_SLOT_A>), dim3(g), dim3(K3R_NT), 0, st, ts, id_min, n_ids, hA, s1, nb1, wgcnt);
  hipLaunchKernelGGL(k3r_offsets, dim3(1), dim3(K3R_NT), 0, st, (const uint32_t *)hA, n_ids, offA, cur2);
  hipLaunchKernelGGL(k3r_wgoffsets, dim3(nb1), dim3(K3R_NT), 0, st, (const uint32_t *)wgcnt, g, s1, (const uint32_t *)offA, wgoff);
  hipLaunchKernelGGL((k3r_scatter1<TILE_SLOT_A>), dim3(g), dim3(K3R_NT), 0, st, ts, id_min, s1, nb1, (const uint32_t *)wgoff, buf1);
  // (records < tokens: grids no larger than the chunks there can be)
  const unsigned int g2 = (unsigned int)std::min<unsigned long long>(2048, n_tokens / K3R_CHUNK + 1), g3 = (unsigned int)std::min<unsigned long long>(2048, n_tokens / K3R_FCHUNK + 1);
  hipLaunchKernelGGL(k3r_scatter2, dim3(g2), dim3(K3R_NT), 0, st, (const unsigned long long *)buf1, (const uint32_t *)offA, n_ids, s1, cur2, buf2);
  hipLaunchKernelGGL(k3r_final, dim3(g3), dim3(K3R_NT), 0, st, (const unsigned long long *)buf2, (const uint32_t *)offA, id_min, n_ids, pt, db);
}
}  // namespace yttm
