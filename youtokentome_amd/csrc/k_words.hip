// k_words.hip -- word mode (worker_doing_merge bpe.cpp:491-812 on the words that hold a site instead of every tile): the switch, the gather of
// a round's words, k_words (the tile kernel's own code on gathered words; FUSED: the whole round in one launch), the application of its records.
// (Until round 4 part of k_merge.hip.)
#include "k_tile_core.h"
#include "k_index_core.h"

namespace yttm {

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
        const uint32_t nb = mode == 1 ? g.tl.rec_l[at] : g.tl.rec_r[at], ww = g.tl.rec_word[at];  // (both loads in flight together)
        match = nb == s_filt[a];
        w = ww;
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
  __shared__ uint32_t rcnt_a[FUSED ? BATCH_ARGS_MAX : 1];  // FUSED: the workgroup's records per new token, counted as they are appended (a record's rank rides in it)
  __shared__ unsigned long long rkeys[LDSR ? APPLY_LDS_RULES : 1];
  __shared__ uint16_t rridx[LDSR ? APPLY_LDS_RULES : 1];
#ifdef YTTM_K4_PROF
  const unsigned long long wall0_ = wall_clock64();  // (tuning build: this workgroup's timeline, 100 MHz: start | set-up done, words done | end, words)
  unsigned long long wall_setup_ = 0, wall_words_ = 0, wall_m1_ = 0, wall_m2_ = 0, wall_m3_ = 0;  // (m1: look-ups issued, m2: tables set up + first barrier, m3: runs scanned)
#endif
  const bool from_args = LDSR && ba.k != 0;
  // FUSED, round 6: the rules' runs are looked up FIRST -- thread j: rule j -- and every load of a look-up is issued HERE at once, none behind
  // another: the rule's first probe slot of the index WITH that slot's run (the offsets of its first and of the next key's first shard), the
  // list header of the pair's younger token, the lists' cursor.  Nothing waits for them before the LDS set-up below is done (no barrier waits for
  // a load in flight: a workgroup-scope fence drains LDS operations only); a probe that misses its first slot -- rare at the index's load of
  // at most a half -- goes on from there, behind the set-up.  (The first cut called idx_find here: its probe loop waits for every key it loads,
  // and a workgroup sat 3.2 us in front of its set-up -- tools/dbg/words_blocks.py, profiles/r6_words_blocks.txt.)
  unsigned long long fq_base = 0, fq_len = 0, fq_cur = 0;
  uint32_t fq_filt = 0, fq_mode = 0;
  bool fq_found = true;
  unsigned long long sp_key = PT_EMPTY, sp_o0 = 0, sp_o1 = 0, sp_lb = 0, q_key = 0;
  uint32_t sp_f = 0, sp_c = 0, sp_slot = 0, q_x = 0, q_y = 0, q_cnt = 0;
  bool q_old = false;
  if (FUSED) {
    fq_cur = g.tl.cursor[g.round_id & 1u];
    if (threadIdx.x < ba.k) {
      q_x = ba.xy[2 * threadIdx.x];
      q_y = ba.xy[2 * threadIdx.x + 1];
      q_cnt = g.cnt[threadIdx.x];  // (a dynamically indexed kernel argument is a load from the argument buffer in HBM: with the others, not behind the barrier)
      const uint32_t m = q_x > q_y ? q_x : q_y;
      q_old = m < g.z_static;
      q_key = pair_key(q_x, q_y);
      if (g.ix_valid) {  // (uniform)
        sp_slot = enc_hash(q_x, q_y) & g.ix.mask;
        sp_key = g.ix.key[sp_slot];
        sp_o0 = g.ix.off[(size_t)sp_slot * IDX_SHARDS];
        sp_o1 = g.ix.off[((size_t)sp_slot + 1) * IDX_SHARDS];
      }
      sp_lb = g.tl.base[m];
      sp_f = g.tl.fill[m];
      sp_c = g.tl.cap[m];
    }
  }
#ifdef YTTM_K4_PROF
  wall_m1_ = wall_clock64();
#endif
  agg_init<WPB * 64>(A, from_args ? nullptr : bloom_g);  // (A.flagbits holds the batch's pair filter)
  if (threadIdx.x == 0) rn = 0;
  if (threadIdx.x == 0) dn = 0;
  if (FUSED && threadIdx.x < BATCH_ARGS_MAX) rcnt_a[threadIdx.x] = 0;
  if (from_args) {
    for (int s = (int)threadIdx.x; s < (int)(FLAG_LDS_IDS / 16); s += WPB * 64) A.flagbits[s] = 0;
    for (unsigned int i = threadIdx.x; i <= rule_mask; i += WPB * 64) rkeys[i] = PT_EMPTY;
    __syncthreads();
    for (unsigned int j = threadIdx.x; j < ba.k; j += WPB * 64) {
      // (FUSED: the thread's rule was fetched at the top -- a dynamically indexed kernel argument is a load from the argument buffer)
      const uint32_t x = FUSED && j == threadIdx.x ? q_x : ba.xy[2 * j], y = FUSED && j == threadIdx.x ? q_y : ba.xy[2 * j + 1];
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
#ifdef YTTM_K4_PROF
  wall_m2_ = wall_clock64();
#endif
  const int wave = uni((int)(threadIdx.x >> 6)), lane = lane_id();
  if (FUSED) {  // where each rule's candidates are (as k_wgather: a posting run of the index, or the younger token's instance list): looked up above
    const uint32_t j = threadIdx.x;
    unsigned long long len = 0, lcap = 0;
    bool f_mode_of_mine = false;
    if (j < ba.k) {
      if (q_old) {  // a pair of two tokens the index covers: its postings, if it is a key
        const PairIndex ix{g.ix.key, g.ix.cnt, g.ix.off, g.ix.bloom, g.ix.post, g.ix.mask};
        uint32_t sl = 0xffffffffu;
        if (g.ix_valid) {
          if (sp_key == q_key) sl = sp_slot;
          else if (sp_key != PT_EMPTY) sl = idx_find(ix, q_key, sp_slot + 1u);  // (the first slot held another key: on from the next one)
        }
        if (sl == 0xffffffffu) {
          fq_found = false;
        } else if (sl == sp_slot) {
          fq_base = sp_o0;
          fq_len = sp_o1 - sp_o0;
        } else {
          fq_base = ix.off[(size_t)sl * IDX_SHARDS];
          fq_len = ix.off[((size_t)sl + 1) * IDX_SHARDS] - fq_base;
        }
      } else {  // the instance list of the pair's younger token, filtered by the other one
        fq_base = sp_lb;
        fq_len = sp_f < sp_c ? sp_f : sp_c;
        if (q_x > q_y) { fq_mode = 2; fq_filt = q_y; } else { fq_mode = 1; fq_filt = q_x; }
      }
      len = fq_len;
      if (!fq_found) f_every = 1u;  // not in the index: this round takes every word
      f_base[j] = fq_base;
      f_filt[j] = fq_filt;
      f_mode[j] = (uint8_t)fq_mode;
      f_mode_of_mine = fq_mode != 0;
      const unsigned long long cj = q_cnt;
      lcap = fq_found && len < cj ? len : cj;
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
    const unsigned long long cur = fq_cur;
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
#ifdef YTTM_K4_PROF
  wall_m3_ = wall_clock64();
#endif
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
    // (both ends of the stretch at once: odd lanes search the upper one -- two binary searches over LDS one behind the other were a microsecond)
    const unsigned long long r_q = rec_of(lo_w + ((lane & 1) ? wchunk : 0ull));
    f_pos = __shfl(r_q, 0);
    f_hi = __shfl(r_q, 1);
  }
  TileStats S;
#ifdef YTTM_K4_PROF
  S.t_last = (unsigned long long)clock64();
  wall_setup_ = wall_clock64();
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
        } else {  // (the record's word is loaded WITH its neighbour, not behind the comparison: one dependent trip less per pass)
          const uint32_t nb = mode == 1 ? g.tl.rec_l[at] : g.tl.rec_r[at], ww = g.tl.rec_word[at];
          cok[it] = nb == f_filt[a];
          cw[it] = ww;
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
                    // (FUSED: the record's rank among the workgroup's records of its token is taken here, not by a pass over the region at the end)
                    const uint32_t rank = FUSED ? atomicAdd(&rcnt_a[z - z_base], 1u) : 0u;
                    my_irec[pos] = make_uint4((z - z_base) | (rank << 12), word_id, lnb, rnb);
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
#ifdef YTTM_K4_PROF
  wall_words_ = wall_clock64();
#endif
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
  DeltaRec drec_first{};  // (FUSED: this thread's first count record, loaded with its first instance record -- below)
  {
    static_assert(sizeof(WL) >= WGATHER_MAXK * sizeof(uint32_t), "per-rule counters of the record flush");
    uint32_t *rcnt = FUSED ? rcnt_a : reinterpret_cast<uint32_t *>(&WL[0]);
    __syncthreads();
    const unsigned int nrec = rn < drec_cap ? rn : drec_cap;
    // Round 6, the late rounds' latency: a thread's first instance record and its first count record are loaded HERE, ahead of the returning
    // adds on the lists' fill counts below -- three trips to L2 side by side instead of one behind the other (the records were stored by this
    // workgroup: a barrier orders them; a FUSED round already knows every record's rank, so nothing has to be read before the adds).
    uint4 rec_first = make_uint4(0u, 0u, 0u, 0u);
    if (FUSED && threadIdx.x < nrec) rec_first = my_irec[threadIdx.x];
    const unsigned int nd_pre = dn < drec_cap ? dn : drec_cap;
    if (FUSED && inline_apply && threadIdx.x < nd_pre) drec_first = dout.recs[threadIdx.x];
    if (nrec) {  // (uniform)
      const uint32_t kk = k_rules < WGATHER_MAXK ? k_rules : WGATHER_MAXK;
      if (!FUSED) {
        for (uint32_t j = threadIdx.x; j < kk; j += WPB * 64) rcnt[j] = 0;
        __syncthreads();
        for (unsigned int i = threadIdx.x; i < nrec; i += WPB * 64) {  // rank of the record among the workgroup's records of its token
          const uint32_t zr = my_irec[i].x & 0xfffu;
          my_irec[i].x = zr | (atomicAdd(&rcnt[zr], 1u) << 12);
        }
        __syncthreads();
      }
      for (uint32_t j = threadIdx.x; j < kk; j += WPB * 64)
        if (rcnt[j]) rcnt[j] = atomicAdd(&tl.fill[z_base + j], rcnt[j]);
      __syncthreads();
      for (unsigned int i = threadIdx.x; i < nrec; i += WPB * 64) {
        const uint4 rec = FUSED && i == threadIdx.x ? rec_first : my_irec[i];
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
      const DeltaRec rec = FUSED && i == threadIdx.x ? drec_first : dout.recs[i];  // (FUSED: the first one came with the instance records, above)
      global_emit(pt, db, rec.key, rec.delta, &A.new_keys);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      blk_add(stats, 4, A.new_keys);
      for (int i = 0; i < 4; i++) blk_add(stats, i, A.st[i]);
    }
  }
#ifdef YTTM_K4_PROF
  if (threadIdx.x == 0) {
    unsigned long long *row = stats + BLK_BASE + 8 * (blockIdx.x % BLK_ROWS);
    row[5] = wall0_;
    // (set-up marks: word 5 of row 512 + b -- k_words runs at most 512 workgroups, the rows behind are nobody's in a word-mode round)
    stats[BLK_BASE + 8 * (512 + blockIdx.x % 512) + 5] = ((wall_m1_ - wall0_) & 0xffffull) | (((wall_m2_ - wall0_) & 0xffffull) << 16) | (((wall_m3_ - wall0_) & 0xffffull) << 32);
    row[6] = ((wall_setup_ - wall0_) & 0xffffffffull) | ((wall_words_ - wall0_) << 32);
    row[7] = ((wall_clock64() - wall0_) & 0xffffffffull) | ((unsigned long long)A.st[1] << 32);
  }
#endif
  // (ordering: k_merge_shared.h "ORDERING OF A FUSED TAIL" -- P1: the inline apply above is atomics only, blk_add atomics; P2 here; C1 - C3 in the branch)
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

void launch_words_init(const TileSet &ts, unsigned long long *wmeta, hipStream_t st) {
  if (!ts.n_tiles) return;
  unsigned int g = (ts.n_tiles + NWAVES - 1) / NWAVES;
  if (g > 256 * 8) g = 256 * 8;
  hipLaunchKernelGGL((k_words_init<TILE_SLOT_A>), dim3(g), dim3(BLOCK), 0, st, ts, wmeta);
}
// The grid hooks of the word-mode launchers (tests, tuning) are read when a context is made, not every round: getenv walks the whole
// environment, and a round's launch is on its critical path (yttm_kernels.h: launch_env_refresh).
static int g_wgather_grid = -1, g_words_grid = -1, g_words_wpi = -1;
int g_apply_grid = 256;  // (k_tiles.hip: launch_merge_apply)
void launch_env_refresh() {
  const std::shared_ptr<const Config> C = cfg();
  auto rd = [](const Hook &h) { return h.set && !h.raw.empty() ? (int)h.i : -1; };
  g_wgather_grid = rd(C->wgather_grid);
  g_words_grid = rd(C->words_grid);
  g_words_wpi = rd(C->words_wpi);
  g_apply_grid = (int)C->apply_grid.i;
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
                        unsigned int fuse_max, hipStream_t st, unsigned int avg_word_tokens) {
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
    // Long words (round 6): a wave's LDS tile holds 512 tokens, and a work item whose words do not fit it takes a second (third ...) gather-and-merge
    // pass, one behind the other -- CJK-shaped text, clauses of ~41 tokens: 16 words per item were 656 tokens, two passes; 8 words per item, one
    // pass and twice the waves at work: merge loop 0.432 -> 0.385 s (profiles/r6_words_per_item.txt).
    while (wpi > 4 && avg_word_tokens && (unsigned long long)wpi * avg_word_tokens > 600ull) wpi >>= 1;
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
}  // namespace yttm
