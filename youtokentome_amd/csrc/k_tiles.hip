// k_tiles.hip -- K3 (pair-frequency count) and K4 (batched merge-apply + count deltas) over LDS-staged token tiles for gfx950; the tile
// repack.  The device code the kernels are made of is k_tile_core.h; word mode, the pair index and the pair table's kernels have files of
// their own (k_words.hip, k_index.hip, k_pairtable.hip; one file, k_merge.hip, until round 4).
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
#include "k_tile_core.h"
#include "k_index_core.h"

namespace yttm {

template <int SLOT, int WPB, bool MERGE, bool LDSR, bool DIRECT = false>
__global__ __launch_bounds__(WPB * 64, SLOT == TILE_SLOT_B ? 1 : MERGE ? (WPB == 4 ? 5 : WPB == 8 ? 6 : WPB) : WPB) void k_tiles(TileSet ts, PairTable pt, DeltaBuf db, const RuleSlot *__restrict__ rules,
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
    // (The full argument -- which store is ordered before which load, by what -- is in k_merge_shared.h, "ORDERING OF A FUSED TAIL"; this is its P2.)
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
    if (is_last && sa.on == 4u) {  // beside another launch of the round, which has the tail: say that this one is over (ScanArgs::peer_flag)
      if (threadIdx.x == 0) {
        *sa.done_ctr = 0;
        __hip_atomic_store(sa.peer_flag, sa.round_id, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else if (is_last) {
      __threadfence();  // (acquire: nothing stale in this CU's caches)
      static_assert(sizeof(WL) >= (CAND_BINS + 160) * sizeof(unsigned int), "tile buffers double as the tail's scratch");
      const RuleProbe zprobe{LDSR ? rkeys : nullptr, LDSR ? nullptr : rules, rule_mask};
      scan_top<WPB * 64>(pt, sa, stats, zprobe, self_x != 0xffffffffu ? pair_key(self_x, self_x) : PT_EMPTY, reinterpret_cast<unsigned int *>(&WL[0]), nullptr);
    }
  }
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


void launch_pair_count(int cls, const TileSet &ts, const PairTable &pt, const DeltaBuf &db, uint32_t id_min, uint32_t n_ids, hipStream_t st) {
  if (!ts.n_tiles) return;
  const std::shared_ptr<const Config> C = cfg();  // (tuning aids; one launch per training)
  const unsigned int bpc = 4;  // (workgroups per CU; measured best, round 2)
  const bool general = C->k3_general.set;
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
    const unsigned int small = (unsigned int)g_apply_grid;  // (YTTM_APPLY_GRID, read by launch_env_refresh when the context was made)
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
    // class B (round 6): APPLY_WPB_B waves -- tiles -- per workgroup share one set-up of the batch's tables in LDS (one wave per workgroup spent
    // most of a 20 us launch on it: 315 workgroups each clearing and filling 22 KB with 64 threads; profiles/r6_classb_waves.txt)
    if (rule_mask < APPLY_LDS_RULES)
      hipLaunchKernelGGL((k_tiles<TILE_SLOT_B, APPLY_WPB_B, true, true>), dim3(tile_grid(ts.n_tiles, APPLY_WPB_B, 4)), dim3(64 * APPLY_WPB_B), 0, st, ts, pt, db, rules, rule_mask,
                         no_flags, bloom_g, self_x, self_z, z_base, no_list, no_n, stats, bargs, sargs);
    else
      hipLaunchKernelGGL((k_tiles<TILE_SLOT_B, APPLY_WPB_B, true, false>), dim3(tile_grid(ts.n_tiles, APPLY_WPB_B, 4)), dim3(64 * APPLY_WPB_B), 0, st, ts, pt, db, rules, rule_mask,
                         no_flags, bloom_g, self_x, self_z, z_base, no_list, no_n, stats, bargs, sargs);
  }
}
}  // namespace yttm
