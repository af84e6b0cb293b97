// k_merge_shared.h -- device code shared by the merge-loop kernels (k_tiles.hip: K3, K4 on tiles; k_words.hip: word mode; k_pairtable.hip: the pair table; k_giant.hip: the
// class-A K4): the HBM side of a count update, per-workgroup statistics rows, the batch's exact rule probe, and the candidate
// scan that the last workgroup of an apply kernel runs (scan_top).
#pragma once
#include "yttm_device.h"
#include "yttm_kernels.h"

namespace yttm {

// ---- Bloom filter of the batch's pairs.  24-bit multiplies (full rate; v_mul_lo_u32 is a quarter-rate instruction): ids beyond
// 2^24 only lose selectivity.  Word = top 11 bits, two bit positions from the next 10.
constexpr int PM_BLOOM_WORDS = PM_BLOOM_WORDS_H;
static_assert(PM_BLOOM_WORDS * 16 == (int)FLAG_LDS_IDS, "the filter lives where k_tiles keeps its flag bitmap");
__host__ __device__ inline uint32_t pm_mul24(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul24(a, b);
#else
  return (a & 0xffffffu) * (b & 0xffffffu);
#endif
}
constexpr uint32_t PM_K1 = 0x9E3779u, PM_K2 = 0x85EBCBu;
__host__ __device__ inline uint32_t pm_hash(uint32_t a, uint32_t b) { return pm_mul24(a, PM_K1) ^ pm_mul24(b, PM_K2); }
__host__ __device__ inline uint32_t pm_word(uint32_t h) { return h >> 21; }
__host__ __device__ inline uint32_t pm_bits(uint32_t h) { return (1u << ((h >> 16) & 31u)) | (1u << ((h >> 11) & 31u)); }

constexpr int AGG_SLOTS = 512;   // LDS delta aggregator shared by the waves of a workgroup (hot pairs).  256 slots: 1.05e8 emits of rounds 12-100 at 1 GB
                                 // found no room and went to the HBM table one by one (K4 135 ms); 512: 123 ms; 704: 124; 1024 (two workgroups per CU): 156.
                                 // The probe loop stays at 8, unrolled: 12 -> 127 ms, 24 -> 253 (!), not unrolled -> 135
// The worklist of dirty tiles is kept in WL_PARTS sub-lists (workgroup b of the filter appends to list b % WL_PARTS, each
// list WL_SEG(n_tiles) entries apart): one cursor bumped by all 1280 workgroups of a launch cost 14 us per round.
constexpr uint32_t WL_PARTS = 8;
__host__ __device__ inline size_t WL_SEG(uint32_t n_tiles) { return (size_t)n_tiles + 64; }

__device__ inline void global_emit(const PairTable &pt, const DeltaBuf &db, unsigned long long key, long long delta, unsigned int *new_keys) {
  pt_add(pt, key, delta, new_keys);
  dt_add(db, key, delta);  // (multi-GPU: the same update, for the other ranks)
}

// Per-workgroup statistics.  A launch of >= 1024 workgroups that each bump the same global counters serialises at
// ~11 ns per atomic (measured: +10 us per launch at 1024 workgroups, +47 us at 4096), a floor under every short kernel of
// a late round.  So workgroup b adds to its own row stats[BLK_BASE + 8 b + j] with plain stores (launches on the stream
// are serial), and one workgroup folds the rows into the totals when somebody needs them (fold_blk_stats).
constexpr int BLK_BASE = 32, BLK_ROWS = 1536;  // >= the largest grid of k_filter / k_tiles<.., true>  // j: 0..3 = the K4 counters, 4 = pair-table slots claimed
__device__ inline void blk_add(unsigned long long *stats, int j, unsigned long long v) {
  // (the row is this workgroup's alone; the store is write-through because the round's tail may read it from another XCD before
  // any cache write-back -- see round_tail)
  // (round 5: an atomic add, not load + store: two launches of a round may run side by side -- ScanArgs::peer_flag -- and share a row, and
  // the tail of the round before may still be folding the rows; an atomic on a line nobody else touches costs what the write-through store did)
  unsigned long long *p = &stats[BLK_BASE + 8 * (blockIdx.x % BLK_ROWS) + j];
  if (v) atomicAdd(p, v);
}
// called by ONE workgroup of 256 threads, all threads; ends with the totals in stats[0..3] and *n_keys
__device__ inline void fold_blk_stats(unsigned long long *stats, unsigned int *n_keys) {
  __shared__ unsigned long long fold_acc[5];
  if (threadIdx.x < 5) fold_acc[threadIdx.x] = 0;
  __syncthreads();
  unsigned long long a[5] = {0, 0, 0, 0, 0};
  for (int b = (int)threadIdx.x; b < BLK_ROWS; b += (int)blockDim.x) {
    unsigned long long *row = stats + BLK_BASE + 8 * b;  // written by earlier kernels: plain 16-byte loads
    const uint4 v01 = *reinterpret_cast<const uint4 *>(row), v23 = *reinterpret_cast<const uint4 *>(row + 2);
    const unsigned long long v4 = row[4];
    const unsigned long long v[5] = {((unsigned long long)v01.y << 32) | v01.x, ((unsigned long long)v01.w << 32) | v01.z,
                                     ((unsigned long long)v23.y << 32) | v23.x, ((unsigned long long)v23.w << 32) | v23.z, v4};
    bool any = false;
#pragma unroll
    for (int j = 0; j < 5; j++) {
      a[j] += v[j];
      any = any || v[j] != 0;
    }
    if (any) {
      const uint4 z{0u, 0u, 0u, 0u};
      *reinterpret_cast<uint4 *>(row) = z;
      *reinterpret_cast<uint4 *>(row + 2) = z;
      row[4] = 0;
    }
  }
#pragma unroll
  for (int j = 0; j < 5; j++)
    if (a[j]) atomicAdd(&fold_acc[j], a[j]);
  __syncthreads();
  if (threadIdx.x < 4 && fold_acc[threadIdx.x]) stats[threadIdx.x] += fold_acc[threadIdx.x];
  if (threadIdx.x == 4 && fold_acc[4]) *n_keys += (unsigned int)fold_acc[4];
  __threadfence();
  __syncthreads();
}

constexpr unsigned int FILTER_LDS_KEYS = 1024;
struct RuleProbe {
  const unsigned long long *lds_keys;  // [mask+1] or nullptr
  const RuleSlot *g;                   // the same hash in HBM; nullptr = no exact test
  unsigned int mask;
  __device__ bool has(uint32_t a, uint32_t b) const {
    const unsigned long long key = pair_key(a, b);
    unsigned int h = pair_hash32(key) & mask;
    for (;;) {
      const unsigned long long k = lds_keys ? __hip_atomic_load(&lds_keys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : g[h].key;
      if (k == key) return true;
      if (k == PT_EMPTY) return false;
      h = (h + 1) & mask;
    }
  }
};

// ------------------------------------------------------------------------------------------------- multi-GPU pieces of a round
// What the fold of this round's exchange left in xstat (yttm_kernels.h: XSTAT_WORDS) goes to the host with the scan's result: the verdict
// words [0..3] are consumed (zeroed), the sums [4..7] stay.  Called with k = 0 .. 7 by eight threads; xstat == nullptr: zeros (single GPU).
__device__ inline void xstat_forward(unsigned char *mailbox, unsigned long long *__restrict__ xstat, int k) {
  if (k < 0 || k >= XSTAT_WORDS || (!xstat && k >= 4)) return;
  unsigned long long v = 0;
  if (xstat) {
    v = ld_agent(&xstat[k]);
    if (k < 4) xstat[k] = 0;
  }
  if (k < 4) *reinterpret_cast<unsigned long long *>(mailbox + 56 + 8 * k) = v;
  else *reinterpret_cast<unsigned long long *>(mailbox + MB_XSUM + 8 * (k - 4)) = v;
}

// ------------------------------------------------------------------------------------------------- fused candidate scan
__device__ inline int cand_bin(unsigned long long c) {
  if (c < 256) return (int)c;
  int e = 63 - __clzll((long long)c);  // >= 8
  int m3 = (int)((c >> (e - 3)) & 7ull);
  return 256 + (e - 8) * 8 + m3;
}


// ---- ORDERING OF A FUSED TAIL (the memory-model argument; gfx950).  Sites: k_tiles.hip (tile rounds), k_words.hip (k_words<FUSED>, k_delta_apply,
// k_wgather's allotment), k_pairtable.hip (k_fold_list, k_hot_scan).  tests/test_gpu_parity.py::test_zz_fused_tail_ordering backs it, it does not replace it.
// The hand-off.  Workgroups P (any CU, any XCD) update the pair table, the hot / top lists' slots and lengths, their statistics row and (multi-GPU)
// the send block; the LAST workgroup C of the same launch then reads all of that (scan_top).  Nothing orders P and C but what is written here.
// Hardware facts relied on (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility"): (F1) per-XCD L2s are not
// coherent for plain accesses and a CU's vector L1 is never refreshed by another CU's stores; (F2) an agent-scope atomic RMW and an `sc1`
// (write-through, __hip_atomic_store relaxed/agent) store leave the issuing XCD's L2 -- they are performed where every XCD sees them -- and count
// in the wave's vmcnt until acknowledged; (F3) an agent-scope atomic load (`sc1`) bypasses the reader's L1; "`sc1` payload -> s_waitcnt vmcnt(0)
// -> flag" with `sc1` loads (or an agent acquire) on the consumer is one of the guide's valid cross-CU hand-offs; (F4) a workgroup-scope fence
// is nothing another CU can observe, an agent-scope RELEASE writes the whole XCD L2 back (1.7 - 6.5 us each: +150 us per round at 768 workgroups).
// Producer side, every workgroup of the launch:
//   P1  every store C may read is an atomic RMW (pt_add, list appends, blk_add, dt_add) or an `sc1` store -- never a plain store.  (Plain stores
//       exist, into data C does NOT read in this launch: token tiles, wmeta, record regions read back by the writer itself; the kernel boundary
//       orders those for the next launch.)
//   P2  each wave executes `s_waitcnt vmcnt(0)` (asm volatile: the compiler may drop the wait of a fence it believes has nothing to wait for), so by
//       F2 all of P1 is performed; __builtin_amdgcn_fence(release, "workgroup") keeps the compiler from sinking stores below; __syncthreads()
//       joins the waves; THEN one lane takes the ticket, atomicAdd(done_ctr) -- agent scope, performed after everything before it in program order
//       because the wave waited for vmcnt(0) first.  No agent-scope release: there is nothing dirty in L2 that C reads (P1), see F4.
// Consumer side, the workgroup whose ticket is gridDim.x - 1:
//   C1  the ticket values form one modification order on one address, so every other workgroup's P1 + P2 happened before C's RMW returned;
//   C2  __threadfence() (agent acq_rel: buffer_inv sc1) drops this CU's L1 lines, which may predate the producers' updates (F1); then
//       __syncthreads() before any thread reads;
//   C3  everything C reads of P1 is read with agent-scope atomic loads (ld_agent, __hip_atomic_load relaxed/agent): F3.  Plain loads in the tail
//       touch only kernel arguments, LDS and data no workgroup of this launch wrote.
// Two launches of one round side by side (class-B tiles on a second stream, ScanArgs::peer_flag): the peer's last workgroup -- after the same P1/P2
// and its own ticket -- stores the round's number with an agent-scope RELEASE store; C polls it with agent-scope ACQUIRE loads (bounded) before C2.
// The peer's plain tile stores are NOT covered by that; they are ordered for later launches by an event the host records on the second stream and
// waits for before the main stream touches class-B tiles again (gpu_ctx.cpp).
// Towards the host: candidates, histogram and header are plain stores into pinned host memory, then __threadfence_system() + __syncthreads(),
// then ONE system-scope release store of the round id; the host spins on that word and issues an acquire fence before reading the rest.
// After the publish C folds the statistics rows with atomic exchanges (the NEXT round's class-B launch may already be adding to them).
//
// The candidate scan of a merge round: ONE workgroup reads the top list (PairTable::top_slots, about a thousand entries).
// Run by the LAST workgroup of the apply kernel to finish (ScanArgs, yttm_kernels.h; every other workgroup has published its
// updates as device-scope atomics or write-through stores and then taken its ticket, nothing else touches the pair table), or as
// a kernel of its own (k_top_scan) where a round is more than one launch.
//   1. the top list: zero the pairs of the batch just applied (every occurrence was merged), histogram the live counts, collect
//      the candidates above the host's threshold, and COMPACT the list in place: an entry whose count fell below top_tau leaves
//      the list (PT_TOP cleared, so it can come back).  TAIL_E entries per thread and pass, two dependent memory round trips
//      per pass (slot numbers, then records) with all loads of a round trip in flight together.
//   2. header, histogram and the first `fast` candidates go to `box` -- the host's pinned mailbox (then the round id is
//      published there, system-scope release: the host polls instead of copying and synchronising).  Multi-GPU: the same -- the lists
//      hold the same pairs on every rank (k_fold_list), so no verdict on them has to be exchanged first; `xstat` (the fold's report on
//      the exchange itself) rides along.
//   3. last, off the critical path: the per-workgroup statistics rows are folded into the totals and the key count
// Box layout: [0] candidates, [4] keys in the table, [8] top-list entries before the scan, [12] of those still >= top_tau,
// [16] hot-list entries (overflow check), [24] the round's duration on the device (ScanArgs::timed), [88] merge sites so far, [32] round id, [40] tokens streamed so far, [48] tiles with a site so far, [56..87]
// xstat verdicts (multi-GPU; the sums at MB_XSUM), [96..127] timing marks (100 MHz), [MB_HIST..) histogram, [8192..) candidates.
// lds = at least (CAND_BINS + 160) words of scratch (the apply kernel's tile buffers are free by now).
template <int NT>
__device__ inline void scan_top(const PairTable &pt, const ScanArgs &sa, unsigned long long *__restrict__ stats, const RuleProbe &zprobe,
                                unsigned long long zself, unsigned int *lds, unsigned long long *__restrict__ xstat) {
  constexpr int NW = NT / 64;
  unsigned int *lh = lds;                     // [CAND_BINS]
  unsigned int *wcount = lds + CAND_BINS;     // [NW] kept entries per wave of this pass
  unsigned int *ctl = lds + CAND_BINS + 32;   // [1] candidates, [2] live entries, [3] highest bin in use, [4 + (pass & 1)] entries kept up to and including that pass
  unsigned long long *facc = reinterpret_cast<unsigned long long *>(lds + CAND_BINS + 40);  // [5] fold accumulators
  unsigned int *sub = lds + CAND_BINS + 64;   // [64] counts inside the boundary bin
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (sa.peer_flag) {
    // The round's other launch (the class-B tiles, on a second stream) must be over: its last workgroup stores the round's number behind an
    // agent-scope release.  It was submitted BEFORE this launch, so it runs whatever this workgroup does; the spin is bounded all the same
    // -- a peer that never signals leaves the round unpublished, which the host reports (poll_mailbox), instead of a hung device.
    if (tid == 0) {
      unsigned int ok = 0;
      for (unsigned int spins = 0; spins < (1u << 24); spins++) {
        if (__hip_atomic_load(sa.peer_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == sa.round_id) { ok = 1; break; }
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_s_sleep(8);
#endif
      }
      lh[0] = ok;
    }
    __syncthreads();
    const unsigned int ok = lh[0];
    __syncthreads();
    if (!ok) return;
    __threadfence();  // (acquire for every thread: nothing of the peer's is stale in this CU's caches)
  }
  const unsigned long long tm0 = (unsigned long long)wall_clock64();
  for (int b = tid; b < CAND_BINS; b += NT) lh[b] = 0;
  if (tid < 68) sub[tid] = 0;
  if (tid < 6) ctl[tid] = 0;
  if (tid < 5) facc[tid] = 0;
  __syncthreads();
  const unsigned long long tm1 = (unsigned long long)wall_clock64();
  // ---- 2. the top list
  const unsigned int tn_raw = __hip_atomic_load(pt.top_n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned int hot_raw = __hip_atomic_load(pt.hot_n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const bool overflow = tn_raw > pt.top_cap;  // entries were dropped: the host refills the list, nothing to scan
  const unsigned int tn = overflow ? 0u : tn_raw;
  uint4 *box_out = reinterpret_cast<uint4 *>(sa.mailbox + 8192);
  constexpr int TAIL_E = 8;
  unsigned int my_live = 0;
  unsigned int pass = 0;  // (ctl[4 + ((pass - 1) & 1)] = entries kept by the passes before this one; 0 for the first)
  // The host wants about `want` candidates (four times its recent batch), and the count histogram it picks its threshold from is too
  // coarse for that (8 bins per power of two): on natural text the top list's counts sit in one or two bins, and every round sent the
  // WHOLE list -- 570 candidates on Zipf text for batches of 8 -- through the mailbox and the host's heap.  When the list fits one pass the
  // scan refines the threshold itself: boundary bin from the histogram, 64 sub-bins inside it, candidates = every entry at or above the
  // refined count (a complete prefix of the order, as the host needs; it reads the threshold back as the smallest count it got).
  const bool refine = sa.want != 0 && tn != 0 && tn <= (unsigned int)(NT * TAIL_E);
#if defined(YTTM_K4_PROF) && defined(__HIP_DEVICE_COMPILE__)
  // tuning build: where the scan's time goes (100 MHz ticks, summed over the rounds into stats[24..30]; YTTM_TRACE prints them)
  unsigned long long tq_[7];
#define TAIL_MARK(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); tq_[i] = (unsigned long long)wall_clock64(); } while (0)
  TAIL_MARK(0);
#else
#define TAIL_MARK(i) ((void)0)
#endif
  for (unsigned int base = 0; base < tn; base += NT * TAIL_E, pass++) {
    uint32_t sl[TAIL_E];
    unsigned long long k[TAIL_E], c[TAIL_E];
#pragma unroll
    for (int e = 0; e < TAIL_E; e++) {
      const unsigned int i = base + (unsigned int)(e * NT + tid);
      sl[e] = i < tn ? __hip_atomic_load(&pt.top_slots[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xffffffffu;
    }
    if (base == 0) TAIL_MARK(1);
#pragma unroll
    for (int e = 0; e < TAIL_E; e++) {
      k[e] = PT_EMPTY;
      c[e] = 0;
      if (sl[e] != 0xffffffffu) {
        k[e] = ld_agent(pt.key_p(sl[e]));
        c[e] = ld_agent(pt.cnt_p(sl[e]));
      }
    }
    if (base == 0) TAIL_MARK(2);
    uint32_t keepm = 0;  // bit e: entry e stays on the list
    unsigned int my_maxbin = 0;
#pragma unroll
    for (int e = 0; e < TAIL_E; e++) {
      if (sl[e] == 0xffffffffu) continue;
      unsigned long long cc = c[e] & PT_CNT;
      if (cc && (k[e] == zself || zprobe.has((uint32_t)(k[e] >> 32), (uint32_t)k[e]))) cc = 0;  // a pair of the finished batch
      const bool keep = cc >= pt.top_tau && cc > 0;
      const unsigned long long want = (c[e] & PT_HOT) | (keep ? PT_TOP : 0ull) | cc;  // a dropped entry loses PT_TOP and can come back
      if (want != c[e]) *pt.cnt_p(sl[e]) = want;
      if (keep) {
        keepm |= 1u << e;
        my_live++;
        const int bin = cand_bin(cc);
        atomicAdd(&lh[bin], 1u);
        my_maxbin = (unsigned int)bin > my_maxbin ? (unsigned int)bin : my_maxbin;
        c[e] = cc;  // (the live count, for the emission below)
      }
    }
    {  // the highest bin in use: ONE LDS atomic per wave (round 5: one per entry, a thousand on one address, was 1 - 2 us of every scan)
      unsigned int m = my_maxbin;
      for (int o = 32; o > 0; o >>= 1) {  // (lane 0 ends up with the maximum)
        const unsigned int t = (unsigned int)__shfl_down((int)m, o);
        if (lane + o < 64) m = t > m ? t : m;
      }
      if (lane == 0 && m) atomicMax(&ctl[3], m);
    }
    if (base == 0) TAIL_MARK(3);
    unsigned long long tau_ref = 0;  // candidates must also reach this count
    if (refine) {  // (uniform; one pass: the histogram is complete after the barrier)
      // Round 5: EVERY wave works the threshold out for itself from the histogram in LDS -- the same numbers in every wave -- instead of wave 0
      // doing it between two barriers while the others wait: two barriers (histogram complete, sub-bins complete) instead of four.
      __syncthreads();
      static_assert(CAND_BINS == 64 * 11, "eleven bins per lane");
      // the bin in which the `want`-th largest count lies: lane l owns bins [11 l, 11 l + 11), summed from the top lane down
      unsigned int mine_b = 0;
      for (int b = 0; b < 11; b++) mine_b += lh[lane * 11 + b];
      unsigned int incl_b = mine_b;  // entries in my bins and those of the lanes above me
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned int t = __shfl_down(incl_b, o);
        if (lane + o < 64) incl_b += t;
      }
      const unsigned int above = incl_b - mine_b;
      const unsigned long long owner_m = __ballot(above < sa.want && incl_b >= sa.want);  // (exactly one lane, if the list holds `want` entries at all)
      if (owner_m) {
        const int owner = __ffsll((long long)owner_m) - 1;
        unsigned int bnd = 0, acc_above = 0;  // (meaningful in the owner lane)
        if (lane == owner) {
          unsigned int acc = above;
          for (int b = 10; b >= 0; b--) {
            const unsigned int h = lh[lane * 11 + b];
            if (acc + h >= sa.want) {
              bnd = (unsigned int)(lane * 11 + b);
              acc_above = acc;
              break;
            }
            acc += h;
          }
        }
        const int B = __shfl((int)bnd, owner);
        const unsigned int n_above = (unsigned int)__shfl((int)acc_above, owner);
        unsigned long long lo = (unsigned long long)B;
        int shift = 0;
        bool wide = false;
        if (B >= 256) {
          const int ex = 8 + (B - 256) / 8, m3 = (B - 256) % 8;
          lo = (unsigned long long)(8 + m3) << (ex - 3);
          wide = true;
          shift = ex - 3 > 6 ? ex - 3 - 6 : 0;  // (the bin is 2^(ex-3) counts wide: 64 sub-bins)
        }
        tau_ref = lo;
        if (wide) {  // (uniform over the workgroup: every wave found the same bin)
#pragma unroll
          for (int e = 0; e < TAIL_E; e++)
            if (((keepm >> e) & 1u) && cand_bin(c[e]) == B) atomicAdd(&sub[(unsigned int)((c[e] - lo) >> shift) & 63u], 1u);
          __syncthreads();
          // lane s: entries of the sub-bins s .. 63 plus those above the bin; the highest s that reaches `want` is the threshold
          unsigned int v = sub[lane];
          for (int o = 1; o < 64; o <<= 1) {
            const unsigned int t = __shfl_down(v, o);
            if (lane + o < 64) v += t;
          }
          const unsigned long long ok = __ballot(n_above + v >= sa.want);  // (lane 0 always: the bin was chosen so)
          const int sb = ok ? 63 - __clzll((long long)ok) : 0;
          tau_ref = lo + ((unsigned long long)sb << shift);
        }
      }
    }
    if (base == 0) TAIL_MARK(4);
#pragma unroll
    for (int e = 0; e < TAIL_E; e++) {
      const unsigned long long cc = c[e];
      const uint32_t x = (uint32_t)(k[e] >> 32), y = (uint32_t)k[e];
      const uint32_t mx = x > y ? x : y;
      const bool is_cand = ((keepm >> e) & 1u) && cc >= tau_ref && (cc > sa.tau_cnt || (cc == sa.tau_cnt && mx <= sa.tau_mx));
      // a wave's candidates of this step take their places with ONE add (a returning add per candidate on one LDS address serialises)
      const unsigned long long cm = __ballot(is_cand);
      if (!cm) continue;
      unsigned int o0 = 0;
      const int leader = __ffsll((long long)cm) - 1;
      if (lane == leader) o0 = atomicAdd(&ctl[1], (unsigned int)__popcll(cm));
      o0 = (unsigned int)__shfl((int)o0, leader);
      if (is_cand) {
        const unsigned int o = o0 + (unsigned int)__popcll(cm & ((1ull << lane) - 1ull));
        if (o < sa.cap) {
          sa.out[o].key = k[e];
          sa.out[o].cnt = cc;
        }
        if (o < sa.fast) {
          uint4 vv;
          vv.x = (uint32_t)k[e]; vv.y = (uint32_t)(k[e] >> 32); vv.z = (uint32_t)cc; vv.w = (uint32_t)(cc >> 32);
          box_out[o] = vv;
        }
      }
    }
    if (base == 0) TAIL_MARK(5);
    // compaction: the kept entries of this pass move down behind those of the earlier passes (any order)
    const uint32_t mine = (uint32_t)__popc(keepm);
    const uint32_t incl = wave_incl_scan(mine);
    if (lane == 63) wcount[wave] = incl;
    __syncthreads();  // every entry of this pass has been read
    // (the running total is kept in two alternating words: this pass reads the one the pass before wrote -- ordered by the barrier above --
    // and thread 0 writes the other; what follows needs no barrier of its own: the next pass, or the publish, has one before it reads)
    const unsigned int kept_before = pass ? ctl[4 + ((pass - 1) & 1)] : 0u;
    unsigned int pos = kept_before + incl - mine;
    for (int w = 0; w < wave; w++) pos += wcount[w];
#pragma unroll
    for (int e = 0; e < TAIL_E; e++)
      if ((keepm >> e) & 1u) pt.top_slots[pos++] = sl[e];
    if (tid == 0) {
      unsigned int t = kept_before;
      for (int w = 0; w < NW; w++) t += wcount[w];
      ctl[4 + (pass & 1)] = t;
    }
    if (base + (unsigned int)(NT * TAIL_E) < tn) __syncthreads();  // (wcount is rewritten by the next pass)
  }
  const unsigned int kept_total_at = pass ? 4 + ((pass - 1) & 1) : 0;  // (no pass at all: ctl[0] is zero)
  {
    const unsigned long long t = wave_sum_u64((unsigned long long)my_live);
    if (lane == 0 && t) atomicAdd(&ctl[2], (unsigned int)t);
  }
  __syncthreads();
#if defined(YTTM_K4_PROF) && defined(__HIP_DEVICE_COMPILE__)
  TAIL_MARK(6);
  if (tid == 0 && tn != 0) {
    stats[24] += tq_[0] - tm1;  // list lengths
    for (int i = 1; i < 7; i++) stats[24 + i] += tq_[i] - tq_[i - 1];  // slots | keys + counts | keep / zero / histogram | refine | emit | compaction + the later passes
    stats[31] += 1;
  }
#endif
  // ---- 3. publish
  const unsigned long long tm2 = (unsigned long long)wall_clock64();
  unsigned int *hdr = reinterpret_cast<unsigned int *>(sa.mailbox);
  unsigned long long *box_hist = reinterpret_cast<unsigned long long *>(sa.mailbox + MB_HIST);
  if (tid == 0) {
    hdr[0] = ctl[1];
    hdr[1] = __hip_atomic_load(pt.n_keys, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    hdr[2] = tn_raw;
    hdr[3] = ctl[2];
    hdr[4] = hot_raw;
    hdr[5] = ctl[3];  // the host reads the histogram from here down (every line of the pinned mailbox it touches is a cache miss)
    *reinterpret_cast<unsigned long long *>(sa.mailbox + 88) = stats[0];  // merge sites so far (word-mode switch)
    *reinterpret_cast<unsigned long long *>(sa.mailbox + 24) = sa.timed ? (unsigned long long)wall_clock64() - ld_agent(&stats[STAT_T0]) : 0ull;  // the round on the device
    *reinterpret_cast<unsigned long long *>(sa.mailbox + 40) = stats[2];  // tokens streamed so far (repack trigger)
    *reinterpret_cast<unsigned long long *>(sa.mailbox + 48) = stats[1];  // tiles that held a merge site so far
    if (!overflow) *pt.top_n = ctl[kept_total_at];
    if (sa.done_ctr) *sa.done_ctr = 0;
    unsigned long long *tmark = reinterpret_cast<unsigned long long *>(sa.mailbox + 96);
    tmark[0] = tm0; tmark[1] = tm1; tmark[2] = tm2; tmark[3] = (unsigned long long)wall_clock64();
  }
  xstat_forward(sa.mailbox, xstat, tid - 6);  // (multi-GPU: the exchange's report)
  if (xstat && tid == 14) *reinterpret_cast<unsigned long long *>(sa.mailbox + MB_XSUM + 32) = sa.timed ? ld_agent(&stats[STAT_T1]) - ld_agent(&stats[STAT_T0]) : 0ull;
  for (int b = tid; b < CAND_BINS; b += NT) box_hist[b] = (unsigned long long)lh[b];
  if (sa.round_id) {
    __threadfence_system();
    __syncthreads();
    if (tid == 0) __hip_atomic_store(&hdr[8], sa.round_id, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // ---- 4. the statistics rows, AFTER the host has its candidates: the fold is off the round's critical path (the key count
  // and the token totals in the header are therefore one round old; the host allows for that)
  {  // statistics rows (left by the workgroups of this and earlier launches, write-through): all loads in flight together
    constexpr int RPT = NT >= 512 ? (BLK_ROWS + NT - 1) / NT : 3;  // rows per thread in flight together (a one-wave workgroup -- class-B tiles -- takes its rows in groups)
    unsigned long long a[5] = {0, 0, 0, 0, 0};
    for (int b0 = 0; b0 < BLK_ROWS; b0 += RPT * NT) {
      unsigned long long v[RPT][5];
#pragma unroll
      for (int r = 0; r < RPT; r++) {
        const int b = b0 + tid + r * NT;
#pragma unroll
        for (int jj = 0; jj < 5; jj++)
          v[r][jj] = b < BLK_ROWS ? __hip_atomic_load(&stats[BLK_BASE + 8 * b + jj], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
      }
#pragma unroll
      for (int r = 0; r < RPT; r++) {
        const int b = b0 + tid + r * NT;
#pragma unroll
        for (int jj = 0; jj < 5; jj++) {
          // (taken by an exchange: the NEXT round's class-B launch may already be adding to the row -- gpu_ctx.cpp merge_apply puts it on a
          // second stream without waiting for this kernel's end -- and what it adds after the exchange belongs to the next fold)
          if (v[r][jj]) a[jj] += atomicExch(&stats[BLK_BASE + 8 * b + jj], 0ull);
        }
      }
    }
#pragma unroll
    for (int jj = 0; jj < 5; jj++) {
      const unsigned long long t = wave_sum_u64(a[jj]);
      if (lane == 0 && t) atomicAdd(&facc[jj], t);
    }
    __syncthreads();
    if (tid < 4 && facc[tid]) stats[tid] += facc[tid];
    if (tid == 4 && facc[4]) atomicAdd(pt.n_keys, (unsigned int)facc[4]);
  }
}

}  // namespace yttm
