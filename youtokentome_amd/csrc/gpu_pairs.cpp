// gpu_pairs.cpp -- the pair table (K3) and the three tiers of candidate lists behind GpuCtx::candidates(): hot list, top list, the pinned mailbox.
// (Round 5: cut out of gpu_ctx.cpp, code motion only; gpu_ctx_internal.h says what went where.)
#include "gpu_ctx_internal.h"

namespace yttm {

// ------------------------------------------------------------------------------------------------- pair table
void GpuCtx::alloc_table(PairTable &pt, unsigned long long cap) {
  pt.slots = dmalloc<unsigned long long>(2 * cap);
  pt.n_keys = dmalloc<unsigned int>(4);
  pt.mask = cap - 1;
  pt.hot_tau = ~0ull;  // no hot list until rebuild_hot()
  pt.hot_slots = d_hot_slots_;
  pt.hot_n = d_hot_n_;
  pt.hot_cap = hot_cap_;
  pt.top_tau = ~0ull;
  pt.top_slots = d_top_slots_;
  pt.top_n = d_top_n_;
  pt.top_cap = top_cap_;
  hot_state_ = HOT_INVALID;
  top_state_ = TOP_INVALID;
  launch_pt_clear(pt, strm());
  HIP_CHECK(hipMemsetAsync(pt.n_keys, 0, 16, strm()));
}
void GpuCtx::free_table(PairTable &pt) {
  DFREE(pt.slots);
  DFREE(pt.n_keys);
  pt.mask = 0;
}

void GpuCtx::ensure_table_capacity(unsigned long long need_keys) {
  if (pt_cap_ && need_keys * 2 <= pt_cap_) return;
  chain_event_ = nullptr;
  // load stays below 1/2; growth is by 4x (a rehash also costs a rebuild of the hot list)
  unsigned long long new_cap = pow2_at_least(std::max<unsigned long long>(1ull << 16, need_keys * 4));
  if (!pt_cap_) {
    alloc_table(pt_, new_cap);
    pt_cap_ = new_cap;
    return;
  }
  PairTable nt{};
  alloc_table(nt, new_cap);
  launch_pt_rehash(pt_, nt, strm());
  rehashes++;
  unsigned int nk = 0;
  HIP_CHECK(hipMemcpyAsync(&nk, nt.n_keys, 4, hipMemcpyDeviceToHost, strm()));
  sync();
  free_table(pt_);
  pt_ = nt;
  pt_cap_ = new_cap;
  n_keys_host = nk;
}

// keys the pair table is sized for before the first merge: distinct initial pairs <= adjacencies <= tokens, and -- the candidate filter
// does not stream the table, so its size costs nothing per round, while every growth step is a rehash plus a hot-list rebuild -- the
// size a corpus of this many tokens typically ends with
unsigned long long GpuCtx::initial_table_keys(unsigned long long n_tok) const {
  unsigned long long bound = std::min<unsigned long long>(n_tok + 16, ((unsigned long long)n_alpha_ + 1) * (n_alpha_ + 1));
  bound = std::min<unsigned long long>(bound, 1ull << 26);
  return std::max(bound, std::min<unsigned long long>(n_tok / 16, 1ull << 25));
}

void GpuCtx::pair_count() {
  HIP_CHECK(hipSetDevice(device_));
  tl_stream = strm();
  tl_device = device_;
  n_keys_host = 0;
  unsigned long long bound = initial_table_keys(n_tokens0);
  if (multi()) {
    bound = std::min<unsigned long long>(bound * comm_->world, 1ull << 27);
    if (!d_send2_[0]) {
      // distinct pairs a round of this rank can touch: bounded by its updates (a handful per live token); sized for a
      // quarter of that -- dense rounds touch few distinct pairs, sparse rounds few tokens -- and checked: a rank whose table or
      // send block overflowed says so in its block's header and every rank stops
      unsigned long long cap = 1ull << 20;
      while (cap < n_tokens0 / 2 && cap < (1ull << 27)) cap <<= 1;
      // (... and never less than twice the distinct pairs K3 itself can produce on this rank: a large alphabet on a small corpus)
      while (cap < 2 * initial_table_keys(n_tokens0) && cap < (1ull << 28)) cap <<= 1;
      if (const unsigned int forced = (unsigned int)cfg_->xchg_table_cap.u) {  // (tests: a table that overflows)
        cap = pow2_at_least(std::max(forced, 4u));
        delta_cap_forced_ = true;
      }
      alloc_delta_table(cap);
      d_xstat_ = dmalloc<unsigned long long>(XSTAT_WORDS);
      HIP_CHECK(hipMemsetAsync(d_xstat_, 0, XSTAT_WORDS * 8, strm()));
      maybe_cap_ = std::max(1u, (unsigned int)cfg_->xchg_notes.u);  // (tests shrink it: the fold then walks every record)
      d_maybe_ = dmalloc<uint32_t>(maybe_cap_);
      d_maybe_n_ = dmalloc<unsigned int>(4);
      HIP_CHECK(hipMemsetAsync(d_maybe_n_, 0, 16, strm()));
      blk_min_ = std::max(2u * XHDR, (unsigned int)cfg_->xchg_blk_min.u);  // (tests shrink it to force the repeat path)
      blk_ = blk_min_;
      grow_recv(std::max<unsigned long long>(send_cap_, blk_ * (unsigned long long)comm_->world));
    }
  }
  if (multi()) bound = std::max(bound, std::min<unsigned long long>(n_tokens0 / 16 * (unsigned long long)comm_->world, 1ull << 25));
  if (!pt_fresh_ || bound * 2 > pt_cap_) {  // (normally build_class(0) has put a cleared table of this size in place)
    free_table(pt_);
    pt_cap_ = 0;
  }
  pt_fresh_ = false;
  ensure_table_capacity(bound);
  t_begin(KT_PAIR_COUNT);
  const uint32_t n_ids = id_max_ >= id_min_ ? id_max_ - id_min_ + 1 : 0;
  // class A of a large alphabet (CJK: thousands of symbols, millions of pairs): records partitioned by their first token instead of one random
  // atomic per adjacency (k_pairradix.hip; 16 B of scratch per class-A token for the length of the count)
  uint32_t *rx_scratch = nullptr;
  unsigned long long *rx_buf1 = nullptr, *rx_buf2 = nullptr;
  // (its scratch -- two 8-byte records per class-A token -- must fit what is free with room to spare: else the general kernel, never an out-of-memory)
  const bool radix = cls_[0].n_tiles && pair_count_radix_takes(n_ids, cls_[0].n_tokens0) && cls_[0].n_tokens0 >= cfg_->k3_radix_min.u && !cfg_->k3_general.set &&
                     16ull * (cls_[0].n_tokens0 + 1) + (64ull << 20) <= free_device_bytes() / 4 * 3;
  if (radix) {
    rx_scratch = dmalloc<uint32_t>(pair_count_radix_scratch_u32(n_ids));
    rx_buf1 = dmalloc<unsigned long long>(cls_[0].n_tokens0 + 1);
    rx_buf2 = dmalloc<unsigned long long>(cls_[0].n_tokens0 + 1);
    HIP_CHECK(hipMemsetAsync(rx_scratch, 0, (size_t)n_ids * 4, strm()));
    launch_pair_count_radix(cls_[0].ts, pt_, db_, id_min_, n_ids, rx_scratch, rx_buf1, rx_buf2, cls_[0].n_tokens0, strm());
    k3_radix = 1;
  }
  for (int ci = radix ? 1 : 0; ci < 2; ci++) launch_pair_count(ci, cls_[ci].ts, pt_, db_, id_min_, n_ids, strm());
  launch_giant(false, cls_[2].ts, cls_[2].slot, pt_, db_, nullptr, 0, 0xffffffffu, 0, cls_[2].d_scratch, d_stats_, strm());
  t_end(KT_PAIR_COUNT, 4 * n_tokens0 + 8 * n_unique);
  unsigned int nk = 0;
  HIP_CHECK(hipMemcpyAsync(&nk, pt_.n_keys, 4, hipMemcpyDeviceToHost, strm()));
  sync();
  DFREE(rx_scratch); DFREE(rx_buf1); DFREE(rx_buf2);
  n_keys_host = nk;
  exchange_deltas();
}

void GpuCtx::download_pairs(std::vector<unsigned long long> &keys, std::vector<unsigned long long> &cnts) {
  std::vector<CandRec> out;
  uint32_t n = scan_full(0, 0xffffffffu, out, nullptr);
  if (n > out.size()) throw GpuError{"download_pairs: more than 2^20 live pairs"};
  keys.resize(n);
  cnts.resize(n);
  for (uint32_t i = 0; i < n; i++) { keys[i] = out[i].key; cnts[i] = out[i].cnt; }
}

uint32_t GpuCtx::scan_full(unsigned long long tau_cnt, uint32_t tau_mx, std::vector<CandRec> &out, unsigned long long *hist) {
  HIP_CHECK(hipSetDevice(device_));
  tl_stream = strm();
  tl_device = device_;
  out.clear();
  if (!pt_cap_) {
    if (hist) memset(hist, 0, CAND_BINS * 8);
    memset(hist_buf_, 0, sizeof hist_buf_);
    last_hist_ = hist_buf_;
    last_live_ = 0;
    return 0;
  }
  flush_pending_zero();
  launch_fold_stats(d_stats_, pt_.n_keys, strm());
  HIP_CHECK(hipMemsetAsync(d_cand_n_, 0, 16, strm()));
  HIP_CHECK(hipMemsetAsync(d_cand_hist_, 0, CAND_BINS * 8, strm()));
  t_begin(KT_CAND);
  launch_cand_scan(pt_, tau_cnt, tau_mx, d_cand_, cand_cap_, d_cand_n_, d_cand_hist_, strm());  // (always with the histogram: last_hist())
  t_end(KT_CAND, 16 * pt_cap_);
  // ONE device-to-host copy per round: header + histogram + the first CAND_FAST candidates; a second copy only when
  // more candidates passed (the host rarely looks past a few thousand)
  constexpr unsigned int CAND_FAST = 4096;
  unsigned char *h = (unsigned char *)h_pin_;
  HIP_CHECK(hipMemcpyAsync(h, d_round_, 8192 + (size_t)CAND_FAST * sizeof(CandRec), hipMemcpyDeviceToHost, strm()));
  sync();
  const unsigned int n = *(unsigned int *)h;
  n_keys_host = *(unsigned int *)(h + 4);
  if (hist) memcpy(hist, h + 64, CAND_BINS * 8);
  memcpy(hist_buf_, h + 64, CAND_BINS * 8);
  last_hist_ = hist_buf_;
  last_live_ = 0;
  last_top_bin_ = CAND_BINS - 1;
  for (int b = 1; b < CAND_BINS; b++) last_live_ += hist_buf_[b];
  const unsigned int take = std::min(n, cand_cap_);
  CandRec *h_c = (CandRec *)(h + 8192);
  if (take > CAND_FAST) {
    HIP_CHECK(hipMemcpyAsync(h_c + CAND_FAST, d_cand_ + CAND_FAST, (size_t)(take - CAND_FAST) * sizeof(CandRec), hipMemcpyDeviceToHost, strm()));
    sync();
  }
  out.assign(h_c, h_c + take);
  HIP_CHECK(hipMemsetAsync(d_cand_n_, 0, 16, strm()));  // the hot-list filter expects its counters cleared
  HIP_CHECK(hipMemsetAsync(d_cand_hist_, 0, CAND_BINS * 8, strm()));
  return n;
}

// Choose hot_tau from the histogram of the whole table (about HOT_TARGET pairs at or above it, never more than half the
// list) and list those slots.  Huge ties that do not fit switch the filter back to whole-table scans for a while.
void GpuCtx::rebuild_hot() {
  std::vector<CandRec> none;
  unsigned long long hist[CAND_BINS];
  pt_.hot_tau = ~0ull;
  pt_.top_tau = ~0ull;  // (the top list is refilled from the new hot list; k_hot_rebuild clears every PT_TOP)
  top_state_ = TOP_INVALID;
  idx_valid_ = false;  // (the pair index holds the OLD list's pairs)
  idx_pending_ = true;
  scan_full(~0ull >> 1, 0, none, hist);
  unsigned long long acc = 0;
  int chosen = -1;
  for (int b = CAND_BINS - 1; b >= 1; b--) {
    if (acc + hist[b] > hot_cap_ / 2) break;
    acc += hist[b];
    chosen = b;
    // (word mode: a rebuilt list means a rebuilt pair index -- two passes over the words.  word_global_, not word_mode_: the threshold this
    // picks shapes the candidate lists, which must come out alike on every rank of a sharded training)
    if (acc >= (word_global_ ? std::max(hot_target_, hot_target_words_) : hot_target_)) break;
  }
  hot_rebuilds++;
  if (chosen < 0 || (acc < hot_min_ && chosen > 1)) {  // ties too large for the list right below the few top pairs
    hot_state_ = HOT_FULLSCAN;
    fullscan_rounds_ = 0;
    return;
  }
  pt_.hot_tau = std::max<unsigned long long>(1, cand_bin_lower(chosen));
  HIP_CHECK(hipMemsetAsync(d_hot_n_, 0, 4, strm()));
  t_begin(KT_CAND);
  launch_hot_rebuild(pt_, strm());
  t_end(KT_CAND, 8 * pt_cap_);
  hot_state_ = HOT_ACTIVE;
  hot_just_rebuilt_ = true;
}

// Waits for `round_id` in the pinned mailbox: the kernel that publishes it writes header + histogram + first candidates there
// first (a copy + stream synchronisation would cost tens of microseconds per round).
void GpuCtx::poll_mailbox(uint32_t round_id) {
  unsigned char *h = (unsigned char *)h_pin_;
  volatile uint32_t *flag = (volatile uint32_t *)(h + 32);
  for (unsigned long long spins = 0; *flag != round_id; spins++) {
    if ((spins & 0x3fff) == 0x3fff) {
      const hipError_t q = hipStreamQuery(st_raw_);
      if (q == hipSuccess) {
        if (*flag != round_id) throw GpuError{"candidate mailbox was not published"};
      } else if (q != hipErrorNotReady) {
        throw GpuError{std::string("candidate filter: ") + hipGetErrorString(q)};
      }
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  const unsigned long long cum = *(const unsigned long long *)(h + 40);
  if (cum != scanned_cum_) {  // a merge round ran since the last call: that is how many tokens its filters streamed
    live_tokens_last_ = cum - scanned_cum_;
    scanned_cum_ = cum;
    const unsigned long long touched = *(const unsigned long long *)(h + 48);
    touched_last_ = touched - touched_cum_;
    touched_cum_ = touched;
  }
  st_touched_ = false;  // (whatever was queued before the kernel that published is over; that kernel is past everything but its statistics fold)
  const unsigned long long sites = *(const unsigned long long *)(h + 88);  // (published by scan_top only; one round old, like the token counts)
  if (sites > sites_cum_) {
    sites_last_ = sites - sites_cum_;
    sites_cum_ = sites;
  }
  if (multi()) {  // the same numbers summed over the ranks' block headers: what the switch to word mode is decided from
    const unsigned long long *xs = (const unsigned long long *)(h + MB_XSUM);
    if (xs[3] == (unsigned long long)comm_->world) {  // (a scan that ran before any exchange leaves zeros)
      if (xs[0] > g_sites_cum_) { g_sites_last_ = xs[0] - g_sites_cum_; g_sites_cum_ = xs[0]; }
      if (xs[1] > g_tokens_cum_) { g_tokens_last_ = xs[1] - g_tokens_cum_; g_tokens_cum_ = xs[1]; }
      g_tiles_a_ = xs[2];
    }
  }
}

// One scan of the hot list (L1) by k_hot_scan -- every listed slot, many workgroups: candidates above (t, tm), histogram of the
// live counts, the pending zeroing of the finished batch's pairs.  Leaves the result in the mailbox.  False: the exchange of
// this round had to be completed first (multi-GPU), scan again.
bool GpuCtx::scan_hot(unsigned long long t, uint32_t tm) {
  constexpr unsigned int CAND_FAST = 4096;
  unsigned char *h = (unsigned char *)h_pin_;
  const uint32_t round_id = ++mail_round_;
  t_begin(KT_CAND);
  launch_hot_scan(pt_, t, tm, d_cand_, cand_cap_, d_cand_n_, d_cand_hist_, d_hot_n_ + 1, h, CAND_FAST, round_id, d_stats_,
                  pending_zero_ && !zero_ba_.k ? d_rules_ : nullptr, zero_cap_ - 1, zero_self_key_, listed_last_ ? listed_last_ + 4096 : hot_cap_,
                  pending_zero_ && zero_ba_.k ? &zero_ba_ : nullptr, multi() ? d_xstat_ : nullptr, strm());
  pending_zero_ = false;
  t_end(KT_CAND, 20ull * listed_last_);  // (not chained: the host round trip that follows belongs to no kernel family)
  poll_mailbox(round_id);
  const unsigned int *hdr = (const unsigned int *)h;
  n_keys_host = hdr[1];
  listed_last_ = std::min(hdr[2], hot_cap_);
  if (multi() && settle_exchange(*(const unsigned long long *)(h + 56), *(const unsigned long long *)(h + 64), *(const unsigned long long *)(h + 80))) return false;
  return true;
}

// Refill of the top list (L2) from the hot list (L1): one scan of L1 for the histogram of its live counts, the threshold that
// puts about top_target_ of them on the top list, one pass that lists them.  False: L1 itself has to be rebuilt first (it
// overflowed or ran dry; hot_state_ says so) or the scan has to be repeated.
bool GpuCtx::refill_top() {
  unsigned char *h = (unsigned char *)h_pin_;
  if (!scan_hot(~0ull >> 2, 0)) return false;
  const unsigned int *hdr = (const unsigned int *)h;
  const unsigned int listed = hdr[2], live = hdr[3];
  const bool over = listed > hot_cap_;  // (multi-GPU: the lists hold the same pairs on every rank -- k_fold_list -- so this verdict is every rank's)
  if (over || (live < hot_min_ && pt_.hot_tau > 1 && !hot_just_rebuilt_)) {
    hot_state_ = HOT_INVALID;  // overflowed, or running dry: relist with a new threshold (a list that is short right after its
    return false;              // rebuild stays: ties kept the threshold up)
  }
  hot_just_rebuilt_ = false;
  const unsigned long long *hist = (const unsigned long long *)(h + MB_HIST);
  unsigned long long acc = 0;
  int chosen = -1;
  for (int b = CAND_BINS - 1; b >= 1; b--) {
    if (acc + hist[b] > top_cap_ / 2) break;
    acc += hist[b];
    chosen = b;
    if (acc >= top_target_) break;
  }
  top_refills++;
  if (chosen < 0 || (acc < top_min_ && acc < live)) {  // ties too large for the top list right below its first few entries: scan the hot list itself for a while
    pt_.top_tau = ~0ull;
    top_state_ = TOP_BYPASS;
    bypass_rounds_ = 0;
    return true;
  }
  pt_.top_tau = std::max<unsigned long long>(pt_.hot_tau, cand_bin_lower(chosen));
  HIP_CHECK(hipMemsetAsync(d_top_n_, 0, 4, strm()));
  t_begin(KT_CAND);
  launch_top_rebuild(pt_, listed_last_, strm());
  t_end(KT_CAND, 20ull * listed_last_);
  top_state_ = TOP_ACTIVE;
  return true;
}

// Candidates for the host's pick (see gpu_ctx.h).  Three tiers: the top list (about a thousand slots, read by one workgroup --
// in the tail of the round's apply kernel when the round is one launch), refilled from the hot list (tens of thousands, read by a
// kernel of its own) when it runs dry or overflows, which is rebuilt from the whole table when IT runs dry or overflows.
uint32_t GpuCtx::candidates(unsigned long long tau_cnt, uint32_t tau_mx, std::vector<CandRec> &out, unsigned long long *hist) {
  HIP_CHECK(hipSetDevice(device_));
  tl_stream = strm();
  tl_device = device_;
  out.clear();
  if (!pt_cap_) {
    if (hist) memset(hist, 0, CAND_BINS * 8);
    memset(hist_buf_, 0, sizeof hist_buf_);
    last_hist_ = hist_buf_;
    last_live_ = 0;
    return 0;
  }
  constexpr unsigned int CAND_FAST = 4096;
  unsigned char *h = (unsigned char *)h_pin_;
  const bool fused_ok = fused_pending_ && fused_tau_ == tau_cnt && fused_mx_ == tau_mx;
  fused_pending_ = false;  // (a scan that was fused but is not wanted any more is simply ignored)
  int dry_refills = 0, dry_rebuilds = 0;
  for (int attempt = 0;; attempt++) {
    if (attempt > 24) throw GpuError{"candidate lists do not settle"};
    const bool use_fused = fused_ok && attempt == 0 && top_state_ == TOP_ACTIVE && hot_state_ == HOT_ACTIVE;
    if (multi() && hot_state_ != HOT_ACTIVE) {
      // whole-table scans ahead (list rebuild, or no list at all): they synchronise anyway, so the verdict of this round's
      // exchange is fetched directly instead of travelling with the mailbox
      unsigned long long x[4] = {0, 0, 0, 0};
      HIP_CHECK(hipMemcpyAsync(x, d_xstat_, 32, hipMemcpyDeviceToHost, strm()));
      HIP_CHECK(hipMemsetAsync(d_xstat_, 0, 32, strm()));
      sync();
      settle_exchange(x[0], x[1], x[3]);
    }
    if (hot_state_ == HOT_FULLSCAN && ++fullscan_rounds_ >= 64) hot_state_ = HOT_INVALID;  // ties may have dissolved
    if (hot_state_ == HOT_INVALID) {
      rebuild_hot();
      top_state_ = TOP_INVALID;
      pt_.top_tau = ~0ull;
    }
    if (hot_state_ == HOT_FULLSCAN) return scan_full(tau_cnt, tau_mx, out, hist);
    if (top_state_ == TOP_BYPASS && ++bypass_rounds_ >= 64) top_state_ = TOP_INVALID;
    if (top_state_ == TOP_INVALID && !refill_top()) continue;
    unsigned long long t = tau_cnt;
    uint32_t tm = tau_mx;
    const unsigned long long floor_tau = top_state_ == TOP_ACTIVE ? pt_.top_tau : pt_.hot_tau;
    if (t < floor_tau) {  // a list is complete only from its threshold up
      t = floor_tau;
      tm = 0xffffffffu;
    }
    unsigned int n = 0, live = 0;
    bool hot_over = false, top_over = false;
    if (top_state_ == TOP_BYPASS) {
      if (!scan_hot(t, tm)) continue;
      const unsigned int *hdr = (const unsigned int *)h;
      n = hdr[0];
      live = hdr[3];
      hot_over = hdr[2] > hot_cap_;
      if (hot_over || (live < hot_min_ && pt_.hot_tau > 1 && dry_rebuilds < 1)) {
        if (!hot_over) dry_rebuilds++;
        hot_state_ = HOT_INVALID;
        continue;
      }
    } else {
      const uint32_t round_id = use_fused ? fused_round_ : ++mail_round_;
      if (!use_fused) {
        ScanArgs sa{};
        sa.on = 1;
        sa.tau_cnt = t;
        sa.tau_mx = tm;
        sa.out = d_cand_;
        sa.cap = cand_cap_;
        sa.fast = CAND_FAST;
        sa.done_ctr = nullptr;
        sa.mailbox = h;
        sa.round_id = round_id;
        t_begin(KT_CAND);
        launch_top_scan(pt_, sa, d_stats_, pending_zero_ && !zero_ba_.k ? d_rules_ : nullptr, zero_cap_ - 1, zero_self_key_,
                        pending_zero_ && zero_ba_.k ? &zero_ba_ : nullptr, multi() ? d_xstat_ : nullptr, strm());
        pending_zero_ = false;
        t_end(KT_CAND, 20ull * top_listed_last_);
      }
      poll_mailbox(round_id);
      const unsigned int *hdr = (const unsigned int *)h;
      n = hdr[0];
      n_keys_host = hdr[1];
      const unsigned int top_listed = hdr[2], hot_listed = hdr[4];
      live = hdr[3];
      top_listed_last_ = std::min(live, top_cap_);
      hot_over = hot_listed > hot_cap_;
      top_over = top_listed > top_cap_;
      if (use_fused) {
        fused_rounds++;
        if (dev_timing_pending_) {  // the round's duration by the device's 100 MHz clock (merge_apply: dev_timing)
          double ms = (double)*(const unsigned long long *)(h + 24) * 1e-5;
          if (multi()) {  // the apply kernels, and what follows them (pack, all-gather, fold, scan), apart
            const double k4 = std::min(ms, (double)*(const unsigned long long *)(h + MB_XSUM + 32) * 1e-5);
            kt.ms[KT_XCHG] += ms - k4;
            kt.launches[KT_XCHG]++;
            ms = k4;
          }
          kt.ms[KT_MERGE] += ms;
          if (word_mode_) { merge_ms_words += ms; merge_launches_words++; }
          dev_round_ms_.push_back((float)ms);
          last_round_dev_ms = ms;
        }
        const unsigned long long *tmk = (const unsigned long long *)(h + 96);  // scan_top's marks (100 MHz wall clock)
        tail_ticks[0] += tmk[1] - tmk[0];
        tail_ticks[1] += tmk[2] - tmk[1];
        tail_ticks[2] += tmk[3] - tmk[2];
        tail_listed += top_listed;
        if (top_over) fused_overflows++;
      }
      // a scan that found its list overflowed read nothing -- and so did not zero the finished batch's pairs: k_pt_zero /
      // the next scan does it (the batch is still described by the zero_* members)
      if (top_over) pending_zero_ = zero_valid_;
      if (multi() && settle_exchange(*(const unsigned long long *)(h + 56), *(const unsigned long long *)(h + 64), *(const unsigned long long *)(h + 80))) continue;
      if (hot_over) {
        hot_state_ = HOT_INVALID;
        continue;
      }
      if (top_over) {
        top_state_ = TOP_INVALID;
        continue;
      }
      if (live < top_min_) {  // running dry: a lower threshold for the top list; at the hot list's own threshold, for that one
        if (pt_.top_tau > pt_.hot_tau && dry_refills < 1) {  // (once per call: large ties can leave a refilled list short)
          dry_refills++;
          top_state_ = TOP_INVALID;
          continue;
        }
        if (pt_.top_tau <= pt_.hot_tau && live < hot_min_ && pt_.hot_tau > 1 && dry_rebuilds < 1) {
          dry_rebuilds++;
          hot_state_ = HOT_INVALID;
          continue;
        }
      }
    }
    if (hist) memcpy(hist, h + MB_HIST, CAND_BINS * 8);
    last_hist_ = (const unsigned long long *)(h + MB_HIST);
    last_live_ = live;
    last_top_bin_ = top_state_ == TOP_ACTIVE ? std::min<unsigned int>(((const unsigned int *)h)[5], CAND_BINS - 1) : CAND_BINS - 1;
    const unsigned int take = std::min(n, cand_cap_);
    CandRec *h_c = (CandRec *)(h + 8192);
    if (take > CAND_FAST) {
      HIP_CHECK(hipMemcpyAsync(h_c + CAND_FAST, d_cand_ + CAND_FAST, (size_t)(take - CAND_FAST) * sizeof(CandRec), hipMemcpyDeviceToHost, strm()));
      sync();
    }
    out.assign(h_c, h_c + take);
    if (const char *dbg = dbg_cand_) {  // debugging aid: one line per scan, comparable across scan implementations
      static FILE *f = nullptr;
      if (!f) f = fopen(dbg, "w");
      unsigned long long hx = 0;
      for (unsigned int i = 0; i < take; i++) hx ^= mix64(out[i].key * 31 + out[i].cnt);
      if (f) fprintf(f, "r=%llu fused=%d tau=%llu mx=%u hot_tau=%llu top_tau=%llu n=%u live=%u nkeys=%llu cand=%016llx\n", merge_rounds, (int)use_fused, t, tm,
                     pt_.hot_tau, pt_.top_tau, n, live, n_keys_host, hx);
      if (f) fflush(f);
    }
    return n;
  }
}

void GpuCtx::pair_query(const unsigned long long *keys, uint32_t n, unsigned long long *outv) {
  if (!n) return;
  flush_pending_zero();
  unsigned long long *d_k = dmalloc<unsigned long long>(n), *d_o = dmalloc<unsigned long long>(n);
  HIP_CHECK(hipMemcpyAsync(d_k, keys, (size_t)n * 8, hipMemcpyHostToDevice, strm()));
  launch_pt_query(pt_, d_k, n, d_o, strm());
  HIP_CHECK(hipMemcpyAsync(outv, d_o, (size_t)n * 8, hipMemcpyDeviceToHost, strm()));
  sync();
  DFREE(d_k);
  DFREE(d_o);
}

void GpuCtx::flush_pending_zero() {
  if (!pending_zero_) return;
  chain_event_ = nullptr;
  if (zero_ba_.k) {  // the batch never went to HBM: upload its rule hash for k_pt_zero (rare: only readers other than the hot scan)
    std::vector<RuleSlot> tab(zero_cap_);
    for (auto &r : tab) { r.key = PT_EMPTY; r.z = 0; r.pad = 0; }
    for (uint32_t j = 0; j < zero_ba_.k; j++) {
      const uint32_t x = zero_ba_.xy[2 * j], y = zero_ba_.xy[2 * j + 1];
      if (x == y) continue;
      const unsigned long long key = pair_key(x, y);
      unsigned int h = pair_hash32(key) & (zero_cap_ - 1);
      while (tab[h].key != PT_EMPTY) h = (h + 1) & (zero_cap_ - 1);
      tab[h].key = key;
    }
    HIP_CHECK(hipMemcpyAsync(d_rules_, tab.data(), tab.size() * sizeof(RuleSlot), hipMemcpyHostToDevice, strm()));
    sync();
  }
  launch_pt_zero(pt_, d_rules_, zero_cap_, zero_self_key_, strm());
  pending_zero_ = false;
}

}  // namespace yttm
