// gpu_ctx.cpp -- the trainer's device context (gpu_ctx.h): construction, kernel-family timers, and the merge round (K4) with its one launch and one
// mailbox poll.  The rest of GpuCtx lives in gpu_pool.cpp, gpu_upload.cpp, gpu_frontend.cpp, gpu_pairs.cpp, gpu_exchange.cpp, gpu_words.cpp
// (gpu_ctx_internal.h: what went where).
#include "gpu_ctx_internal.h"

namespace yttm {

GpuCtx::GpuCtx(int device) : device_(device) {
  cfg_refresh();  // the environment hooks are read here, once per context (yttm_config.h); nothing below this constructor calls getenv
  cfg_ = cfg();
  const Config &C = *cfg_;
  xchg_margin_ = C.xchg_margin.d;
  pool_reset_peak();
  HIP_CHECK(hipSetDevice(device_));
  tl_stream = strm();
  tl_device = device_;
  st_raw_ = pool_take_stream(device_);  // (a finished context's: creating and destroying one costs ~2 ms of a training)
  if (!st_raw_) HIP_CHECK(hipStreamCreateWithFlags(&st_raw_, hipStreamNonBlocking));
  tl_stream = strm();
  tl_device = device_;
  d_counters_ = dmalloc<unsigned long long>(64);
  d_stats_ = dmalloc<unsigned long long>(STATS_WORDS);  // [0..3] K4 counters, [8..23] per-phase cycles of a YTTM_K4_PROF build, [32..) per-workgroup rows
  HIP_CHECK(hipMemsetAsync(d_stats_, 0, STATS_WORDS * sizeof(unsigned long long), strm()));  // (stream-ordered like everything that uses them)
  // one block for everything the host reads back per round, so that it is ONE device-to-host copy:
  // [0] n_cand, [4] n_keys | [64..) count histogram | [8192..) candidates
  d_round_ = dmalloc<unsigned char>(8192 + (size_t)CAND_CAP * sizeof(CandRec));
  hot_cap_ = std::min((unsigned int)C.hot_cap.u, HOT_CAP);
  hot_target_ = (unsigned int)C.hot_target.u;  // measured at 1 GB: 4096..16384 equal on the abcd corpus, 8192 best on Zipf text (4279 rounds)
  hot_min_ = (unsigned int)C.hot_min.u;
  fuse_enabled_ = C.no_fuse.u == 0;
  // class-B tiles of a word-mode round: one stream (default, round 6) or beside k_words on a second one (YTTM_CLASSB_BESIDE=1: round 5's protocol)
  classb_overlap_ = C.classb_beside.set && C.classb_beside.u == 1;
  idx_enabled_ = C.no_index.u == 0;  // (no pair index: no word mode either)
  idx_agg_min_ = C.index_agg_min.u;  // (fill pass of an index build: postings from which on a workgroup sums them per key in LDS first; tests: 0)
  hot_target_words_ = (unsigned int)C.hot_target_words.u;  // (measured at 1 GB, word mode: 8192 -> 6 rebuilds, candidate family 21.0 ms; 32768 -> 3, 16.9 ms; round 4: 32768 -> 3, 14.6 ms; 65536 -> 2, 12.5; 131072 -> 2, 15.0)
  // rounds of at most this many words (by the hint) whose batch travels in the kernel arguments are ONE launch, k_words<FUSED>; 0: never.
  // (1 GB random text, wall / K4 ms: never 138.3 / 86.0, 32 k 136.6 / 83.3, 256 k 133.1 / 80.2, 2 M 125.9 / 73.5, every round 125.0 / 72.5)
  words_fuse_max_ = (unsigned int)C.words_fuse_max.u;
  word_hint_floor_ = (unsigned int)C.word_hint_floor.u;  // (the words a round is sized for beyond twice the last round's sites; 1 GB random text, K4 ms on the device clock: 1024 -> 73.6, 4096 -> 72.7, 16384 -> 72.1)
  words_inline_max_ = (unsigned int)C.words_inline_max.u;  // (measured at 1 GB, K4 ms: 16 k -> 97.2, 64 k -> 95.4, 256 k -> 94.1)
  profile_events_ = C.profile_events.u != 0;
  words_enabled_ = C.word_mode.u != 0;   // (0: tiles to the end)
  direct_enabled_ = C.k4_direct.u != 0;  // (0: the pair filter + rule hash from the first round on; A/B runs)
  word_div_ = (unsigned int)C.word_div.u;  // (measured at 1 GB: 96 -> K4 135 ms, 150 -> 107, 200 -> 103.7, 300 -> 103.7, 500 -> 104; round 4, merge loop ms of random 'abcd ': 80 / 100 -> 98.7 (switch at
                                               // round 13), 120 -> 96.0 (round 20), 150 -> 96.4, 200 -> 99.4 (round 29), 400 -> 101.6 -- but the CJK-shaped corpus: 120 -> 485 ms, 200 -> 472: left at 200;
                                               // round 5, with class B's repack looks no longer a sync every second word-mode round: CJK 120 / 200 / 300 -> 493 / 493 / 492 ms, 'abcd ' 100 / 120 / 150 / 200 ->
                                               // 113.4 / 110.8 / 110.8 / 112.7 ms: 150)
  word_min_tiles_ = (unsigned int)C.word_min_tiles.u;  // (tests: 0 = switch as soon as the hot list is active)
  // a pass over the tiles must cost more than word mode's three launches: 1 GB enwik-like text (25 M tokens, 48 us per dense round) got 15 % slower
  // in word mode, the 1 GB CJK-shaped corpus (337 M tokens) 21 % faster, random 'abcd ' (94 M tokens at the switch) 10 % faster
  word_min_tokens_ = C.word_min_tokens.u;
  no_batch_args_ = C.no_batch_args.set;
  launch_env_refresh();
  trace_rounds_ = C.trace_rounds.c_str();  // (points into cfg_, which this context keeps)
  dbg_cand_ = C.dbg_cand.c_str();
  d_hot_slots_ = dmalloc<uint32_t>(HOT_CAP);
  d_hot_n_ = dmalloc<unsigned int>(4);  // [0] list length, [1] k_hot_scan's finished-workgroup ticket, [2..3] overflow verdict (u64)
  HIP_CHECK(hipMemsetAsync(d_hot_n_, 0, 16, strm()));
  top_cap_ = std::max(16u, std::min((unsigned int)C.top_cap.u, TOP_CAP));
  top_target_ = (unsigned int)C.top_target.u;  // about four times what the host looks at per round
  top_min_ = (unsigned int)C.top_min.u;
  d_top_slots_ = dmalloc<uint32_t>(TOP_CAP);
  d_top_n_ = dmalloc<unsigned int>(4);
  HIP_CHECK(hipMemsetAsync(d_top_n_, 0, 16, strm()));
  HIP_CHECK(hipMemsetAsync(d_round_, 0, 8192, strm()));  // k_hot_scan leaves its counters zeroed for the next call
  d_cand_n_ = (unsigned int *)d_round_;
  d_cand_hist_ = (unsigned long long *)(d_round_ + 64);
  d_cand_ = (CandRec *)(d_round_ + 8192);
  cand_cap_ = CAND_CAP;
  d_rules_ = dmalloc<RuleSlot>(RULES_CAP);
  rules_cap_ = RULES_CAP;
  h_pin_ = pool_take_pin();  // pinned staging: one buffer is kept across contexts (hipHostMalloc of 17 MB costs milliseconds)
  if (!h_pin_) HIP_CHECK(hipHostMalloc(&h_pin_, PIN_BYTES, hipHostMallocDefault));
  h_pin_bytes_ = PIN_BYTES;
  memset(h_pin_, 0, 8192);  // mailbox header (k_hot_scan publishes its round id at byte 32)
}

GpuCtx::~GpuCtx() {
  (void)hipSetDevice(device_);
  tl_stream = strm();
  tl_device = device_;
  (void)hipStreamSynchronize(strm());
  if (st_b_) (void)hipStreamSynchronize(st_b_);
  drop_spec();
  for (hipEvent_t e : all_events_) (void)hipEventDestroy(e);
  if (ev_fork_) (void)hipEventDestroy(ev_fork_);
  if (ev_join_) (void)hipEventDestroy(ev_join_);
  DFREE(d_text_owned_); DFREE(d_hist_); DFREE(d_chunk_segs_); DFREE(d_counters_); DFREE(d_cpmap_); DFREE(d_rules_);
  free_class(cls_[0]); free_class(cls_[1]); free_class(cls_[2]);
  DFREE(d_stats_); DFREE(d_round_); DFREE(d_recv_); DFREE(d_hot_slots_); DFREE(d_hot_n_); DFREE(d_top_slots_); DFREE(d_top_n_);
  DFREE(d_xstat_); DFREE(d_bloom_); DFREE(d_maybe_); DFREE(d_maybe_n_); DFREE(d_bsync_);
  DFREE(db_.keys); DFREE(db_.touched); DFREE(d_send2_[0]); DFREE(d_send2_[1]);
  free_table(pt_);
  free_index();
  free_words();
  pool_quiesce(strm());
  if (h_pin_ && pool_give_pin(h_pin_)) h_pin_ = nullptr;
  if (h_pin_) (void)hipHostFree(h_pin_);
  if (st_raw_ && pool_give_stream(device_, st_raw_)) st_raw_ = nullptr;  // (synchronised above: nothing is pending on it)
  if (st_raw_) (void)hipStreamDestroy(st_raw_);
  if (st_b_ && pool_give_stream(device_, st_b_)) st_b_ = nullptr;
  if (st_b_) (void)hipStreamDestroy(st_b_);
}

void GpuCtx::sync() { HIP_CHECK(hipStreamSynchronize(strm())); }
void GpuCtx::join_class_b() {
  if (!classb_unjoined_ || !st_b_) return;
  if (!ev_join_) HIP_CHECK(hipEventCreateWithFlags(&ev_join_, hipEventDisableTiming));
  HIP_CHECK(hipEventRecord(ev_join_, st_b_));
  HIP_CHECK(hipStreamWaitEvent(strm(), ev_join_, 0));
  classb_unjoined_ = false;
}
void GpuCtx::read_stats(int first, int n, unsigned long long *out) {
  HIP_CHECK(hipMemcpyAsync(out, d_stats_ + first, (size_t)n * 8, hipMemcpyDeviceToHost, strm()));
  sync();
}

// Kernel-family timers (profile mode): HIP events on the context's stream.  Events come from a process-wide pool, and an
// interval that starts where the previous one ended shares that event (t_end(..., chain=true) followed by t_begin): a
// merge round costs two hipEventRecord calls instead of four -- on Zipf text (4279 rounds) the four cost 17 % of a step.
static std::vector<hipEvent_t> g_event_pool;
static hipEvent_t event_get() {
  {
    std::lock_guard<std::mutex> g(pool_mutex());
    if (!g_event_pool.empty()) {
      hipEvent_t e = g_event_pool.back();
      g_event_pool.pop_back();
      return e;
    }
  }
  hipEvent_t e;
  HIP_CHECK(hipEventCreate(&e));
  return e;
}
void GpuCtx::t_begin(int which) {
  (void)which;
  if (!profile) return;
  if (chain_event_) {  // nothing was enqueued since the interval that ended there
    cur_a_ = chain_event_;
    chain_event_ = nullptr;
    return;
  }
  cur_a_ = event_get();
  all_events_.push_back(cur_a_);
  HIP_CHECK(hipEventRecord(cur_a_, strm()));
}
void GpuCtx::t_end(int which, unsigned long long bytes, bool chain) {
  kt.launches[which]++;
  kt.bytes[which] += bytes;
  if (!profile) return;
  hipEvent_t b = event_get();
  all_events_.push_back(b);
  HIP_CHECK(hipEventRecord(b, strm()));
  evs_.push_back(Ev{cur_a_, b, which});
  cur_a_ = nullptr;
  chain_event_ = chain ? b : nullptr;
}
void GpuCtx::resolve_timers() {
  sync();
  {
    // K4 algorithmic traffic: every live token is read once (4 B); the tiles that had a merge site are
    // counted once more as re-read and rewritten (8 B per token of those -- an upper bound since single-site tiles are
    // rewritten from the registers they were loaded into)
    unsigned long long st[8] = {0};
    if (pt_cap_) launch_fold_stats(d_stats_, pt_.n_keys, strm());
    sync();
    if (hipMemcpy(st, d_stats_, sizeof st, hipMemcpyDeviceToHost) == hipSuccess) {
      merge_sites = st[0];
      kt.bytes[KT_MERGE] = 4 * st[2] + 8 * st[3];
      touched_tiles = st[1];
      touched_tile_tokens = st[3];
      touched_words = st[4];
      touched_word_tokens = st[5];
    }
  }
  FILE *trace = cfg_->trace.set ? fopen(cfg_->trace.raw.c_str(), "w") : nullptr;  // per-launch times for tuning
  for (auto &e : evs_) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) kt.ms[e.which] += ms;
    if (trace) fprintf(trace, "%d %.4f\n", e.which, ms);
  }
  evs_.clear();
  if (trace)  // (rounds timed by the device: already in kt.ms; listed after the event-timed ones -- nearly every round of a single-GPU training)
    for (float ms : dev_round_ms_) fprintf(trace, "%d %.4f\n", (int)KT_MERGE, ms);
  dev_round_ms_.clear();
  {
    std::lock_guard<std::mutex> g(pool_mutex());
    g_event_pool.insert(g_event_pool.end(), all_events_.begin(), all_events_.end());
  }
  all_events_.clear();
  chain_event_ = nullptr;
  if (trace) fclose(trace);
}

// ------------------------------------------------------------------------------------------------- K4
void GpuCtx::merge_apply(const uint32_t *xyz, uint32_t k, const unsigned long long *rule_counts, const unsigned long long *next_tau_cnt,
                         uint32_t next_tau_mx, uint32_t next_want) {
  HIP_CHECK(hipSetDevice(device_));
  tl_stream = st_raw_;  // (not strm(): naming the stream queues nothing -- see st_touched_)
  tl_device = device_;
  if (!k) return;
  if (!n_tiles && !multi()) return;  // a rank without words still takes part in the exchange
  if (k > RULES_CAP / 2) throw GpuError{"merge_apply: batch too large"};
  // new pairs this round: every site adds <= 2 neighbours (+ the z,z run pair); distinct new keys per rule are also
  // bounded by the number of live token types on either side
  uint32_t vmax = 0, z_base = xyz[2];
  for (uint32_t j = 0; j < k; j++) {
    vmax = std::max(vmax, xyz[3 * j + 2]);
    if (xyz[3 * j + 2] != z_base + j) throw GpuError{"merge_apply: the new ids of a batch must be consecutive"};
  }
  if (vmax >= (1u << 29)) throw GpuError{"merge_apply: token ids must be below 2^29"};
  unsigned long long bound_new = 0;
  for (uint32_t j = 0; j < k; j++) {
    unsigned long long by_tokens = 2ull * (vmax + 1) + 1;
    unsigned long long by_count = rule_counts ? 3 * rule_counts[j] : by_tokens;
    bound_new += std::min(by_tokens, by_count);
  }
  // (the key count the scans report is one round old -- they fold the statistics after publishing: the previous round's bound covers it)
  ensure_table_capacity(n_keys_host + bound_prev_ + bound_new);
  bound_prev_ = bound_new;
  if (multi() && !delta_cap_forced_) {
    // distinct pairs this rank's round can touch: per rule at most five updates per site (sites <= the pair's global count) and at most
    // four pairs per token type ((a,x), (a,z), (y,b), (z,b)) plus the self pairs.  The same number on every rank (it is computed from the
    // batch and the global counts), so every rank regrows its table in the same round -- an overflow is a bug, not a workload.
    unsigned long long dt_bound = 0;
    for (uint32_t j = 0; j < k; j++) dt_bound += std::min<unsigned long long>(rule_counts ? 5 * rule_counts[j] : ~0ull >> 8, 4ull * (vmax + 1) + 4);
    if (dt_bound > send_cap_) {
      unsigned long long cap = db_.mask + 1;
      while (cap / 2 < dt_bound) cap <<= 1;
      chain_event_ = nullptr;
      alloc_delta_table(cap);
      delta_regrows++;
    }
  }

  // Word mode: on when the last round's merge sites are few against the tokens a pass over the tiles streams, and a pass is expensive
  // (and then for good).  Single GPU: this context's numbers.  Multi-GPU: the ranks' numbers summed (block headers -> mailbox), per rank on
  // average -- the same verdict on every rank in the same round; a rank without class-A tiles follows the decision without switching.
  if (!word_global_ && words_enabled_ && idx_enabled_ && !instrument && hot_state_ == HOT_ACTIVE) {
    bool go;
    if (multi()) {
      const unsigned long long W = (unsigned long long)comm_->world;
      go = g_tiles_a_ >= (unsigned long long)word_min_tiles_ * W && g_tiles_a_ && g_sites_last_ != ~0ull && g_tokens_last_ && g_tokens_last_ >= word_min_tokens_ * W &&
           (word_div_ == 0 || g_sites_last_ * (unsigned long long)word_div_ < g_tokens_last_);
    } else {
      go = cls_[0].n_tiles >= word_min_tiles_ && cls_[0].n_tiles && sites_last_ != ~0ull && live_tokens_last_ && live_tokens_last_ >= word_min_tokens_ &&
           (word_div_ == 0 || sites_last_ * (unsigned long long)word_div_ < live_tokens_last_);
    }
    if (go) {
      word_global_ = true;
      if (cls_[0].n_tiles) enter_word_mode(z_base);
      else word_switch_round = merge_rounds;
    }
  }
  if (word_mode_) {
    if (*(volatile unsigned int *)tl_.broken) idx_pending_ = true;  // (the last round is over: its mailbox has been read)
    if (idx_pending_) build_index(z_base);
  }

  // rule hash (x != y rules) + at most one x == y rule passed by value
  unsigned int cap = 64;
  while (cap < 2 * k) cap <<= 1;
  char *pin = (char *)h_pin_ + (1u << 16) + (size_t)CAND_CAP * sizeof(CandRec);  // after the read-back area of candidates()
  RuleSlot *h_rules = (RuleSlot *)pin;
  uint32_t self_x = 0xffffffffu, self_z = 0;
  for (uint32_t j = 0; j < k; j++) {
    const uint32_t x = xyz[3 * j], y = xyz[3 * j + 1], z = xyz[3 * j + 2];
    if (x >= id_cap_ || y >= id_cap_ || z >= id_cap_) throw GpuError{"merge_apply: token id out of range"};
    if (x == y) {
      if (self_x != 0xffffffffu) throw GpuError{"merge_apply: more than one x==y rule in a batch"};
      self_x = x;
      self_z = z;
    }
  }
  // A small batch goes to the kernels as an argument and nothing is uploaded (yttm_kernels.h: BatchArgs); a larger one (or any, with
  // class-C tiles: k_giant reads the rule hash from HBM) travels through k_round_begin.
  const bool by_args = k <= (uint32_t)BATCH_ARGS_MAX && !cls_[2].n_tiles && !no_batch_args_;
  if (!by_args) {  // (the common small batch needs none of this: the host's share of a round is on the critical path)
    for (unsigned int i = 0; i < cap; i++) { h_rules[i].key = PT_EMPTY; h_rules[i].z = 0; h_rules[i].pad = 0; }
    for (uint32_t j = 0; j < k; j++) {
      const uint32_t x = xyz[3 * j], y = xyz[3 * j + 1], z = xyz[3 * j + 2];
      if (x == y) continue;
      const unsigned long long key = pair_key(x, y);
      unsigned int h = pair_hash32(key) & (cap - 1);
      while (h_rules[h].key != PT_EMPTY) h = (h + 1) & (cap - 1);
      h_rules[h].key = key;
      h_rules[h].z = z;
    }
  }
  BatchArgs ba{};
  ba.instr = instrument ? 1u : 0u;
  const uint32_t max_in = max_id_;  // largest id a tile can hold BEFORE this round (the ids the site search looks up)
  max_id_ = std::max(max_id_, vmax);
  if (by_args) {
    ba.k = k;
    ba.direct_v = direct_enabled_ && max_in + 1 <= DIRECT_MAX_V ? max_in + 1 : 0u;  // (the first rounds of a small alphabet: k_tiles<.., DIRECT>)
    for (uint32_t j = 0; j < k; j++) { ba.xy[2 * j] = xyz[3 * j]; ba.xy[2 * j + 1] = xyz[3 * j + 1]; }
  }
  // One launch per round: the candidate scan rides in the tail of the round's last kernel -- single GPU: the apply kernel of class A (class
  // B goes first: the one-wave workgroups of class B took longer over the tail than the launch it saved; class-C tiles -- words of more
  // than 2048 tokens -- keep the separate scan); multi-GPU: the fold kernel behind the all-gather (exchange_round), whatever the classes.
  ScanArgs sa{};
  fused_pending_ = false;
  const int last_cls = cls_[0].n_tiles ? 0 : 1;
  if (next_tau_cnt && fuse_enabled_ && hot_state_ == HOT_ACTIVE && top_state_ == TOP_ACTIVE && !instrument &&
      (multi() || ((cls_[0].n_tiles || cls_[1].n_tiles) && !cls_[2].n_tiles))) {
    sa.on = 1u;
    sa.tau_cnt = *next_tau_cnt;
    sa.tau_mx = next_tau_mx;
    if (sa.tau_cnt < pt_.top_tau) {  // the list is complete only from top_tau up (as in candidates())
      sa.tau_cnt = pt_.top_tau;
      sa.tau_mx = 0xffffffffu;
    }
    sa.out = d_cand_;
    sa.cap = cand_cap_;
    sa.fast = 4096;
    sa.want = next_want;  // (the scan may raise the threshold to about this many candidates: scan_top)
    sa.done_ctr = d_hot_n_ + 1;
    sa.mailbox = (unsigned char *)h_pin_;
    sa.round_id = ++mail_round_;
    fused_pending_ = true;
    fused_tau_ = *next_tau_cnt;
    fused_mx_ = next_tau_mx;
    fused_round_ = sa.round_id;
  }
  if (multi()) {
    // this round's blocks: sized for what the busiest rank will send, predicted from the batch -- its summed pair counts are its merge sites
    // over all ranks -- and the records per site of the last two rounds, with a margin of three (a block that is too small costs a second
    // exchange and a scan of its own, ~60 us; one that is too large costs bytes on the links); never more than the round can touch at all
    // (dt_bound above: the same formula).  Every input is the same on every rank.
    unsigned long long sites = 0, bound = 0;
    for (uint32_t j = 0; j < k; j++) {
      sites += rule_counts ? rule_counts[j] : 0;
      bound += std::min<unsigned long long>(rule_counts ? 5 * rule_counts[j] : ~0ull >> 8, 4ull * (vmax + 1) + 4);
    }
    xch_sites_ = sites;
    const double margin = xchg_margin_;  // (YTTM_XCHG_MARGIN; tests: a margin below one forces the repeat path)
    const double pred = rule_counts ? std::max(xrate_[0], xrate_[1]) * (double)sites * margin : (double)bound;
    unsigned long long need = (unsigned long long)std::min((double)std::min<unsigned long long>(bound, send_cap_), pred) + XHDR;
    unsigned long long b2 = blk_min_;
    while (b2 < need) b2 <<= 1;
    blk_ = b2;
  }
  // Multi-GPU: what the word-mode launches get instead of the scan (ScanArgs::on == 2): the round's last workgroup only leaves the
  // worklist counters at zero; a one-launch round has not even that to do (launch_words_apply: on = 3).  (Packing the delta table in this
  // tail as well was measured: ONE workgroup walking ten thousand claimed slots across XCDs took 45 us -- k_dt_pack's hundred take 6.)
  ScanArgs xa{};
  if (multi()) {
    xa.on = 2u;
    xa.done_ctr = d_hot_n_ + 1;
  }
  // A fused round is timed by the device itself (its first launch notes the time, the tail reports the difference in the mailbox):
  // no hipEventRecord on the round's critical path (two per round were 4 us of host time: 8 % of a Zipf step).  YTTM_PROFILE_EVENTS=1
  // keeps the events (cross-check).
  const bool dev_timing = profile && sa.on && !profile_events_;
  bool marked = false;
  auto first_ba = [&]() {  // the BatchArgs of the round's next launch: the first one carries the mark
    BatchArgs b = ba;
    if (dev_timing && !marked) { b.mark = 1u; marked = true; }
    return b;
  };
  sa.timed = dev_timing ? 1u : 0u;
  dev_timing_pending_ = dev_timing;
  if (!dev_timing) t_begin(KT_MERGE);
  if (!by_args) {
    uint32_t *h_bloom = (uint32_t *)(pin + (size_t)RULES_CAP * sizeof(RuleSlot) + 8 * (size_t)RULES_CAP * sizeof(uint32_t));
    pm_bloom_host(h_bloom, xyz, k);  // the batch's pair filter for the apply kernels (built here: a few hundred hashes)
    if (!d_bloom_) d_bloom_ = dmalloc<uint32_t>(PM_BLOOM_WORDS_H);
    launch_round_begin(h_rules, cap, d_rules_, cls_[0].n_tiles ? cls_[0].d_work_n : nullptr, cls_[1].n_tiles ? cls_[1].d_work_n : nullptr, h_bloom, d_bloom_, strm());
  }
  const PairTable kpt = multi() ? pt_nolist() : pt_;  // (multi-GPU: the lists are filled behind the exchange, by the final counts -- k_fold_list)
  // the tail of class ci's launch: the scan (single GPU) or the exchange tail (multi-GPU) in the round's last tile-class launch, nothing elsewhere
  auto tail_of = [&](int ci) -> const ScanArgs * {
    if (ci != last_cls) return nullptr;
    if (multi()) return ci == 0 && word_mode_ ? &xa : nullptr;  // (the tile kernels have nothing to do in a tail)
    return sa.on ? &sa : nullptr;
  };
  // Word mode on one GPU with class-B tiles (words of 257 .. 2 048 tokens: long clauses of unsegmented scripts): the round is two launches
  // in a row, the class-B tiles (~31 us on the CJK-shaped corpus) and then k_words (~120 us).  DEFAULT since round 6: just that, on the main
  // stream.  YTTM_CLASSB_BESIDE=1 (`beside` below) is round 5's protocol, kept and tested: they share nothing until the tail, so class B
  // goes to a second stream and its last workgroup raises a flag the tail waits for (ScanArgs::peer_flag; tools/micro/two_streams.hip: the
  // whole of the shorter kernel comes off the round IN A MICRO-BENCHMARK; in the trainer it measured 14 ms slower per CJK training than one
  // stream once the second stream was a real one and not, by accident, the NULL stream: profiles/r6_classb_order.txt).  No join: the main stream's next kernel starts after k_words has ended, k_words' tail
  // has waited for the flag, and the flag is stored behind everything class B wrote -- stream order on the main stream IS the join.  The
  // fork: when nothing was queued on the main stream since the host read the last round's mailbox (st_clean: the common round), all that
  // can still run there is that round's tail folding the statistics rows -- by exchanges, so that this launch may add to them meanwhile --
  // and class B starts at once; otherwise (a list refill, a repack, a rule table on its way) it waits for an event, which costs the
  // round ~35 us of cross-queue latency and is why it is not the rule.
  const bool beside = classb_overlap_ && !multi() && sa.on == 1u && word_mode_ && cls_[0].n_tiles && cls_[1].n_tiles && !cls_[2].n_tiles;
  auto prep_b_beside = [&]() {  // the second stream, and the fork event when the main stream holds more than the last round's tail
      bool st_clean = !st_touched_;
      if (!st_b_) {  // first class-B launch beside k_words of this context: the second stream, the fork event, the flag block -- each once
        st_b_ = pool_take_stream(device_);
        // (the pool holds a stream only if a finished context gave one back: never the legacy NULL stream -- it would serialise with the
        // embedding application's default-stream work and the tail's bounded spin on peer_flag could run out)
        if (!st_b_) HIP_CHECK(hipStreamCreateWithFlags(&st_b_, hipStreamNonBlocking));
        if (!ev_fork_)
          HIP_CHECK(hipEventCreateWithFlags(&ev_fork_, hipEventDisableTiming));  // (destroyed by ~GpuCtx: not one of the pooled timing events)
        if (!d_bsync_) {
          d_bsync_ = dmalloc<unsigned int>(4);
          HIP_CHECK(hipMemsetAsync(d_bsync_, 0, 16, strm()));
        }
        st_clean = false;
      }
      if (!st_clean) {
        HIP_CHECK(hipEventRecord(ev_fork_, strm()));
        HIP_CHECK(hipStreamWaitEvent(st_b_, ev_fork_, 0));
      }
  };
  auto launch_b_beside = [&]() {
      ScanArgs sb{};
      sb.on = 4u;
      sb.done_ctr = d_bsync_;
      sb.peer_flag = d_bsync_ + 1;
      sb.round_id = sa.round_id;
      const BatchArgs tba = first_ba();
      launch_merge_apply(1, cls_[1].ts, kpt, db_, d_rules_, cap - 1, self_x, self_z, z_base, d_stats_, &tba, &sb, d_bloom_, st_b_);
      classb_overlapped++;
      classb_unjoined_ = true;
  };
  for (int ci = 1; ci >= 0; ci--) {
    if (!cls_[ci].n_tiles) continue;
    if (ci == 1 && beside) {
      if (!d_bsync_) {
        d_bsync_ = dmalloc<unsigned int>(4);
        HIP_CHECK(hipMemsetAsync(d_bsync_, 0, 16, strm()));
      }
      sa.peer_flag = d_bsync_ + 1;
      // Submitted BEFORE k_words, whose tail waits for the flag: two HIP streams may share one hardware queue, where kernels run in submission
      // order (measured, round 6: submitted behind k_words the class-B launch never started -- the tail's bounded spin ran out and the round
      // was reported unpublished; profiles/r6_classb_order.txt).
      prep_b_beside();
      launch_b_beside();
      continue;
    }
    if (ci == 1) join_class_b();  // (class B on the main stream again behind rounds that ran it beside)
    if (ci == 0 && word_mode_) {
      // the batch's rules -> worklist of words (k_wgather; it also allots the new tokens' instance lists), then the words (k_words)
      WordClass &c = cls_[0];
      const uint32_t *d_xyz = nullptr;
      if (!by_args) {
        uint32_t *h_xyz = (uint32_t *)(pin + (size_t)RULES_CAP * sizeof(RuleSlot) + 8 * (size_t)RULES_CAP * sizeof(uint32_t) + 16384);
        memcpy(h_xyz, xyz, (size_t)k * 12);
        HIP_CHECK(hipMemcpyAsync(d_xyz_, h_xyz, (size_t)k * 12, hipMemcpyHostToDevice, strm()));
        d_xyz = d_xyz_;
      }
      if (k > WGATHER_MAXK) throw GpuError{"merge_apply: batch too large for the word-mode gather"};
      WGatherArgs ga{};  // (the worklist's length is at zero: enter_word_mode, then every round's k_delta_apply)
      ga.ix = idx_;
      ga.ix_valid = idx_valid_ ? 1u : 0u;
      ga.z_static = idx_valid_ ? idx_zbuild_ : 0xffffffffu;  // (no index: every rule is "not found", the round takes every word)
      ga.tl = tl_;
      ga.stamp = d_stamp_;
      ga.round_id = (uint32_t)(merge_rounds + 1);
      ga.worklist = d_wworklist_;
      ga.wl_seg = c.n_unique + 64;
      ga.work_n = c.d_work_n;
      ga.gm = d_gm_;
      ga.done_ctr = d_gm_ + WGATHER_MAXK;
      ga.xyz = d_xyz;
      ga.k = k;
      ga.z_base = z_base;
      if (by_args)
        for (uint32_t j = 0; j < k; j++) ga.cnt[j] = rule_counts ? (uint32_t)std::min<unsigned long long>(rule_counts[j], 0xffffffffull) : 0xffffffffu;
      if (!d_stamp_) {  // (the index could not be built yet: no stamps either -- the gather must not claim words)
        stamp_cap_ = (unsigned int)(c.n_unique + c.n_unique / 8 + 64);
        d_stamp_ = dmalloc<uint32_t>(stamp_cap_);
        HIP_CHECK(hipMemsetAsync(d_stamp_, 0, (size_t)stamp_cap_ * 4, strm()));
        ga.stamp = d_stamp_;
      }
      // live tokens per class-A word, about: what the class held when it left the tiles, less a token per merge site since (a round or two behind)
      if (sites_last_ != ~0ull && word_sites_seen_ != sites_cum_) {
        word_live_tokens_ -= std::min(word_live_tokens_, sites_cum_ - std::min(sites_cum_, word_sites_seen_));
        word_sites_seen_ = sites_cum_;
      }
      const unsigned int avg_word_tokens = (unsigned int)std::min<unsigned long long>(std::max<unsigned long long>(1, word_live_tokens_ / std::max<unsigned long long>(1, c.n_unique)), 1u << 16);
      const unsigned int work_hint = sites_last_ != ~0ull && idx_valid_ ? (unsigned int)std::min<unsigned long long>(2 * sites_last_ + word_hint_floor_, 1ull << 30) : 0u;
      ga.stats = d_stats_;
      const BatchArgs gba = first_ba();
      const WordSet wset{c.d_tok, d_wmeta_, c.d_wcnt, (uint32_t)c.n_unique};
      if (launch_words_apply(wset, kpt, db_, d_rules_, cap - 1, d_bloom_, self_x, self_z, z_base, k, d_wworklist_, c.n_unique + 64, c.d_work_n, d_stats_, tl_, d_drec_, drec_cap_, d_drec_n_, d_irec_, &gba,
                             tail_of(0), work_hint, words_inline_max_, &ga, words_fuse_max_, strm(), avg_word_tokens))
        word_fused_rounds++;
      word_rounds++;
      if (!idx_valid_) word_all_rounds++;
      continue;
    }
    const BatchArgs tba = first_ba();
    launch_merge_apply(ci, cls_[ci].ts, kpt, db_, d_rules_, cap - 1, self_x, self_z, z_base, d_stats_, &tba, tail_of(ci), d_bloom_, strm());
  }
  launch_giant(true, cls_[2].ts, cls_[2].slot, kpt, db_, d_rules_, cap - 1, self_x, self_z, cls_[2].d_scratch, d_stats_, strm());
  if (dev_timing) kt.launches[KT_MERGE]++;
  else t_end(KT_MERGE, 0, /*chain=*/!sa.on);  // (a fused round is followed by the host's turn, not by another kernel: its end event must not start the next interval)
  merge_rounds++;
  const char *trace_rounds = trace_rounds_;
  if (trace_rounds) {
    chain_event_ = nullptr;  // tuning aid: cumulative device stats after every round (adds a sync)
    unsigned long long stt[24];
    launch_fold_stats(d_stats_, pt_.n_keys, strm());
    HIP_CHECK(hipMemcpyAsync(stt, d_stats_, sizeof stt, hipMemcpyDeviceToHost, strm()));
    sync();
    if (cfg_->trace_blocks.set && merge_rounds % 50 == 0) {  // PROF build: per-workgroup start / end / dirty tiles of this round
      std::vector<unsigned long long> rows(STATS_WORDS);
      HIP_CHECK(hipMemcpy(rows.data(), d_stats_, STATS_WORDS * 8, hipMemcpyDeviceToHost));
      std::string name = cfg_->trace_blocks.raw + "." + std::to_string(merge_rounds);
      if (FILE *fb = fopen(name.c_str(), "w")) {
        for (int b = 0; b < 1536; b++) fprintf(fb, "%d %llu %llu %llu\n", b, rows[32 + 8 * b + 5], rows[32 + 8 * b + 6], rows[32 + 8 * b + 7]);
        fclose(fb);
      }
    }
    FILE *f = fopen(trace_rounds, merge_rounds == 1 ? "w" : "a");
    if (f) {
      fprintf(f, "%llu %u %llu %llu %llu %llu %u", merge_rounds, k, stt[0], stt[1], stt[2], stt[3], cls_[0].n_tiles);
      for (int i = 8; i < 24; i++) fprintf(f, " %llu", stt[i]);
      fprintf(f, "\n");
      fclose(f);
    }
  }
  if (instrument && split_round && merge_rounds == split_round) {  // (measurement pass only: a sync does not matter)
    unsigned long long st[8] = {0};
    launch_fold_stats(d_stats_, pt_.n_keys, strm());
    HIP_CHECK(hipMemcpyAsync(st, d_stats_, sizeof st, hipMemcpyDeviceToHost, strm()));
    sync();
    split_sites = st[0];
    split_touched_words = st[4];
    split_touched_word_tokens = st[5];
  }
  {  // the sites this round may add to what the mailbox has reported so far (maybe_repack)
    unsigned long long s3 = 0;
    for (uint32_t j = 0; j < k; j++) s3 += rule_counts ? rule_counts[j] : (~0ull >> 8);
    rp_recent_[2] = rp_recent_[1];
    rp_recent_[1] = rp_recent_[0];
    rp_recent_[0] = std::min<unsigned long long>(s3, ~0ull >> 4);
  }
  pending_zero_ = !sa.on || multi();  // (a fused round zeroes its batch's pairs itself; multi-GPU: exchange_round hands the batch to the fold's scan)
  zero_valid_ = true;
  zero_ba_ = ba;
  zero_cap_ = cap;
  zero_self_key_ = self_x != 0xffffffffu ? pair_key(self_x, self_x) : PT_EMPTY;
  // repack when the tiles are less than half full.  With the hot-list filter the fill is known for free (the previous
  // round's filters report the tokens they streamed); otherwise look every 8 rounds.
  if (hot_state_ == HOT_ACTIVE && live_tokens_last_) {
    const unsigned long long nominal = (unsigned long long)cls_[0].n_tiles * cls_[0].nom + (unsigned long long)cls_[1].n_tiles * cls_[1].nom;
    // (word mode: class A no longer lives in tiles, and "tokens streamed last round" is small against the nominal size of everything whatever
    // the fill of class B: its looks are spaced out -- a late repack of the few long words costs less than a look every other round)
    if (live_tokens_last_ * 2 <= nominal && ++rounds_since_check_ >= (word_mode_ ? 64u : 2u)) {
      rounds_since_check_ = 0;
      for (int ci = 0; ci < 2; ci++) maybe_repack(ci);
    }
  } else if (++rounds_since_check_ >= 8) {
    rounds_since_check_ = 0;
    for (int ci = 0; ci < 2; ci++) maybe_repack(ci);
  }
  if (multi()) {  // (stream-ordered; the scan in the fold's tail reports blocks that were too small)
    exchange_round(0, sa.on ? &sa : nullptr);
  }
  // single GPU: no sync here -- the candidate filter that always follows reads n_keys back together with its results
  // (its sync also makes the pinned rule staging reusable for the next round)
  // every occurrence of the batch's pairs has been merged (on every rank): their counts are exactly zero.  The candidate
  // filter that follows zeroes them while it reads the hot list; any other reader goes through flush_pending_zero().
}

}  // namespace yttm
