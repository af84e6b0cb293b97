// k_pairtable.hip -- the HBM pair table's kernels (pair2cnt_g bpe.cpp:891, check_cnt :1099-1108, PriorityQueue :271-314: the final ordered
// pick stays on the host, host_trainer.cpp): the candidate scans over the table, the hot list and the top list, their rebuilds; rehash, query,
// zero, clear; the multi-GPU round's delta blocks (apply, fold, clean); the round's first small kernel.  (Until round 4 part of k_merge.hip.)
#include "k_tile_core.h"
#include "k_index_core.h"

namespace yttm {

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
// Ordering (k_merge_shared.h, "ORDERING OF A FUSED TAIL"): k_fold_list is ONE workgroup launched behind the apply kernels, the collective and phase
// 1 on the same stream -- what those wrote is ordered by kernel boundaries, not by tickets.  Inside the workgroup, the list appends below are atomics and
// `sc1` stores by some threads that scan_top's threads read back: __syncthreads() (which waits for each wave's outstanding memory operations) + the
// agent-scope loads of scan_top order them; no cross-workgroup hand-off happens in this kernel.
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
