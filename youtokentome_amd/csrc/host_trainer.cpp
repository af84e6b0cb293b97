// host_trainer.cpp -- the host half of BPE training: alphabet, the exact ordered pick of the next merges, rename, dump.
//
// Mirrors train_bpe / learn_bpe_from_string (bpe.cpp:1368-1388, :859-1293).  The reference keeps a lazy priority
// queue on the host and two rules in flight (:1121-1282); here the host keeps NO pair state at all: every round the
// device filters the pair table for candidates above a count threshold, the host orders them exactly as
// MergeCandidate::operator< does (bpe.cpp:110-126) and takes the longest prefix of mutually non-intersecting rules
// (rule_intersection, bpe.cpp:145-147; an x==y rule closes the batch, cf. :1160) -- provably the same sequence as the
// one-at-a-time greedy of the -DDETERMINISTIC_QUEUE build / learn_bpe_slow -- and one K4 pass applies
// the whole batch.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>

#include "gpu_ctx.h"
#include "host_core.h"

namespace yttm {

using clk = std::chrono::steady_clock;
static double since(clk::time_point t0) { return std::chrono::duration<double>(clk::now() - t0).count(); }

namespace {
struct Cand {
  unsigned long long cnt;
  uint32_t x, y;
};
// true if a must be picked before b: larger count; then smaller max(x,y); then smaller min(x,y); then larger x
inline bool before(const Cand &a, const Cand &b) {
  if (a.cnt != b.cnt) return a.cnt > b.cnt;
  const uint32_t amx = std::max(a.x, a.y), bmx = std::max(b.x, b.y);
  if (amx != bmx) return amx < bmx;
  const uint32_t amn = std::min(a.x, a.y), bmn = std::min(b.x, b.y);
  if (amn != bmn) return amn < bmn;
  return a.x > b.x;
}
struct HeapCmp {  // std heap keeps the "largest" on top: largest = picked first
  bool operator()(const Cand &a, const Cand &b) const { return before(b, a); }
};

// threshold such that about `target` pairs have count >= tau (from the device histogram of live counts, read from the top:
// only its first few lines of the pinned mailbox are touched)
unsigned long long choose_tau(const unsigned long long *hist, unsigned long long target, int top_bin) {
  unsigned long long acc = 0;
  for (int b = top_bin; b >= 1; b--) {
    acc += hist[b];
    if (acc >= target) return std::max<unsigned long long>(1, cand_bin_lower(b));
  }
  return 1;
}
}  // namespace

Status learn_bpe(GpuCtx &g, int vocab_size, const std::string &model_path, const BpeConfig &cfg, BPEState *state_out,
                 TrainReport *rep) {
  const auto t_all = clk::now();
  Comm *comm = g.comm();
  const Config &C = g.config();  // (the YTTM_* hooks, read when the context was made)
  const bool root = !comm || comm->rank == 0;
  bool replicated = false;  // multi-GPU, small word tables: the merge loop runs on every rank alone (see below)
  // ---- K1 + alphabet (bpe.cpp:941-944, :1013-1021)
  std::vector<uint32_t> cps;
  std::vector<unsigned long long> cnts;
  unsigned long long data_len = 0;
  g.char_hist(cps, cnts, data_len);
  // deterministic order for the host sort (the device compaction order is arbitrary)
  {
    std::vector<size_t> idx(cps.size());
    for (size_t i = 0; i < idx.size(); i++) idx[i] = i;
    std::sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return cps[a] < cps[b]; });
    std::vector<uint32_t> c2(cps.size());
    std::vector<unsigned long long> n2(cps.size());
    for (size_t i = 0; i < idx.size(); i++) { c2[i] = cps[idx[i]]; n2[i] = cnts[idx[i]]; }
    cps.swap(c2);
    cnts.swap(n2);
  }
  std::vector<std::pair<uint32_t, uint32_t>> alpha;  // insertion order, compact ids
  uint64_t n_removed = 0;
  compute_alphabet(cps, cnts, data_len, cfg, alpha, n_removed);
  const uint32_t n_special = (uint32_t)cfg.special_tokens.n_special_tokens();
  uint64_t used_ids = alpha.size() + n_special;
  if (used_ids > (uint64_t)vocab_size) {  // bpe.cpp:1051-1062
    return Status(1, "Incorrect arguments. Vocabulary size too small. Set vocab_size>=" + std::to_string(used_ids) +
                         ".  Current value for vocab_size=" + std::to_string(vocab_size));
  }
  // ---- K2 + K3
  {
    std::vector<uint32_t> a_cp(alpha.size()), a_id(alpha.size());
    for (size_t i = 0; i < alpha.size(); i++) { a_cp[i] = alpha[i].first; a_id[i] = alpha[i].second; }
    g.build_word_table(a_cp.data(), a_id.data(), (uint32_t)alpha.size(), /*space_id=*/n_special, (uint32_t)vocab_size);
    // Multi-GPU, two modes.  A LARGE word table (random text: hundreds of millions of dedup tokens) stays sharded: every
    // round's apply pass is divided by the ranks and pays one exchange of count deltas.  A SMALL one (natural text: a few million tokens,
    // thousands of latency-bound rounds) would only get slower -- there the reference's own scheme applies (bpe.cpp:1029-1044: merge the
    // shards' word maps once, then loop): the shards are gathered (the ranks' byte ranges in rank order are the file), every rank dedups
    // the whole text and runs the merge loop alone, with NO collective per round; rank 0 writes the model.  The choice is made from the
    // summed local token counts (an upper bound of the merged table), the same number on every rank.
    if (comm) {
      const unsigned long long rep_max = C.replicate_max_tokens.u;
      const unsigned long long t_sum = g.allreduce_scalar(g.n_tokens0);
      // ... and only if the whole corpus fits beside the shard on EVERY rank: the gather holds the shard, the padded blocks of all ranks and
      // the assembled text at once (about twice the whole corpus), the second dedup its segment starts and word table on top.  A large,
      // repetitive corpus (logs; natural text of tens of GB over 8 ranks) has few dedup tokens and would qualify by tokens alone.  The
      // verdict is collective (summed "does not fit" flags): a rank that would run out of memory must not leave the others waiting in
      // the gather.
      bool fits_everywhere = true;
      if (t_sum <= rep_max) {
        const unsigned long long total_bytes = g.allreduce_scalar(g.corpus_bytes);
        // (a rank that took its shard in chunks no longer holds the text: nothing to gather -- the sharded loop then, on every rank)
        const unsigned long long need = 3 * total_bytes + total_bytes / 2 + (64ull << 20);
        const unsigned long long no_fit = g.allreduce_scalar((need > g.free_device_bytes() / 10 * 9 || !g.corpus_resident()) ? 1ull : 0ull);
        fits_everywhere = no_fit == 0;
      }
      if (t_sum <= rep_max && fits_everywhere) {
        g.gather_full_corpus();
        g.set_comm(nullptr);
        replicated = true;
        std::vector<uint32_t> cps2;
        std::vector<unsigned long long> cnts2;
        unsigned long long len2 = 0;
        g.char_hist(cps2, cnts2, len2);  // (the segment count of the whole text; the histogram is the reduced one again)
        if (len2 != data_len) return Status(2, "multi-GPU: the gathered corpus differs from the sum of the shards");
        g.build_word_table(a_cp.data(), a_id.data(), (uint32_t)alpha.size(), /*space_id=*/n_special, (uint32_t)vocab_size);
      }
    }
  }
  g.pair_count();
  if (rep) rep->seconds_frontend = since(t_all);
  const auto t_merge = clk::now();

  // ---- merge loop
  std::vector<BPE_Rule> rules;
  rules.reserve((size_t)vocab_size);
  // Candidates asked for per scan: about four times the recent batch size -- what the host does not look at still travels
  // through the mailbox and the heap (natural text: batches of ~8 rules, random text: ~50).  YTTM_CAND_TARGET fixes it (tuning hook).
  const unsigned long long target_fixed = C.cand_target.u;
  unsigned long long TARGET = target_fixed ? target_fixed : 256;
  const double target_max = C.cand_max.d;
  double batch_ema = 64;
  const uint32_t MX_ALL = 0xffffffffu;
  unsigned long long tau = 1;
  uint32_t tau_mx = MX_ALL;
  std::vector<CandRec> recs;
  std::vector<Cand> heap;
  std::vector<uint32_t> batch_xyz;
  std::vector<unsigned long long> batch_cnt;
  unsigned long long rounds = 0, rescans = 0, rounds_exhausted = 0, batch_extensions = 0, batch_splits = 0;
  const bool extend_on = !C.no_extend.set;  // (tuning hook / A-B runs)
  const bool split_on = !C.no_batch_split.set;
  const bool refine_on = !C.no_refine.set;  // (the same: the fused scan keeps the host's threshold)
  std::vector<unsigned long long> batch_keys;
  double w_cand = 0, w_pick = 0, w_apply = 0, w_pick_a = 0, w_pick_b = 0;  // (YTTM_TRACE: pick = threshold + heap build | pops)
  const bool trace_pick = C.trace.set;
  unsigned long long n_cand_sum = 0;
  struct RoundLine { float wait_us, pick_us, apply_us, dev_us; uint32_t k; unsigned long long sites, touched, tokens; };  // (YTTM_TRACE: where a round's wall time goes, by ranges of rounds)
  std::vector<RoundLine> round_lines;
  std::vector<uint8_t> in_batch((size_t)vocab_size + 1, 0);  // bit0: token is the x of a batch rule, bit1: the y
  while (used_ids < (uint64_t)vocab_size) {
    // Candidate set = every pair with count > tau, or count == tau and max(x,y) <= tau_mx: a complete prefix of the
    // global order, so the batch built from it is exact.  The threshold only trades list length against early batch ends.
    auto tw0 = clk::now();
    g.last_round_dev_ms = 0;
    uint32_t n = g.candidates(tau, tau_mx, recs, nullptr);
    const double w_wait_this = since(tw0);
    w_cand += w_wait_this;
    const double dev_ms_this = g.last_round_dev_ms;
    if (n && n <= recs.size()) {
      // The fused scan may have raised the threshold (ScanArgs::want): what came back is every pair at or above the smallest count listed --
      // a complete prefix of the order all the same -- and that count is the threshold the rest of this round reasons with.
      unsigned long long mn = ~0ull;
      for (uint32_t i = 0; i < n; i++) mn = std::min<unsigned long long>(mn, recs[i].cnt);
      if (mn > tau) { tau = mn; tau_mx = MX_ALL; }
    }
    auto tw1 = clk::now();
    const unsigned long long *hist = g.last_hist();  // (of the counts the scan looked at; valid until the next scan)
    const unsigned long long total_pairs = g.last_live();
    const unsigned long long tau_hint = choose_tau(hist, TARGET, g.last_top_bin());
    if (total_pairs == 0) {
      if (root) fprintf(stderr, "WARNING merged only: %llu pairs of tokens\n", (unsigned long long)used_ids);  // bpe.cpp:1139
      break;
    }
    if (n == 0 || (n < TARGET / 8 && tau_hint < tau)) {  // threshold too high: use the histogram's
      tau = std::min(tau, tau_hint);
      tau_mx = MX_ALL;
      rescans++;
      if (n == 0) continue;
      n = g.candidates(tau, tau_mx, recs, nullptr);
      hist = g.last_hist();
    }
    if (n > recs.size()) {
      // More candidates than the buffer holds (rare: huge ties).  Raise the count threshold by the histogram, then
      // bisect the exact count, then bisect the max(x,y) bound inside the overflowing count.
      rescans++;
      if (tau_hint > tau) { tau = tau_hint; tau_mx = MX_ALL; continue; }
      int top = g.last_top_bin();
      while (top > 1 && hist[top] == 0) top--;
      unsigned long long lo = tau, hi = cand_bin_lower(top + 1 < CAND_BINS ? top + 1 : top) * 2 + 1;
      while (hi - lo > 1) {
        const unsigned long long mid = lo + (hi - lo) / 2;
        rescans++;
        if (g.candidates(mid, MX_ALL, recs, nullptr) > recs.size()) lo = mid; else hi = mid;
      }
      uint32_t mlo = 0, mhi = (uint32_t)used_ids;
      while (mhi - mlo > 1) {
        const uint32_t mid = mlo + (mhi - mlo) / 2;
        rescans++;
        if (g.candidates(lo, mid, recs, nullptr) > recs.size()) mhi = mid; else mlo = mid;
      }
      tau = lo;
      tau_mx = mlo;
      n = g.candidates(tau, tau_mx, recs, nullptr);
      if (n == 0 || n > recs.size()) return Status(2, "candidate filter could not be fitted (more than 2^20 exact ties)");
    }
    if (trace_pick) w_pick_a += since(tw1);
    n_cand_sum += n;
    const auto tw1b = clk::now();
    heap.resize(n);
    for (uint32_t i = 0; i < n; i++) heap[i] = Cand{recs[i].cnt, (uint32_t)(recs[i].key >> 32), (uint32_t)recs[i].key};
    std::make_heap(heap.begin(), heap.end(), HeapCmp());
    if (trace_pick) w_pick_b += since(tw1b);
    batch_xyz.clear();
    batch_cnt.clear();
    const size_t max_batch = 4096;
    bool closed = false;  // by an intersection, an x x rule, or because the vocabulary is full
    int extensions_this_round = 0;
    for (;;) {
    while (!heap.empty() && used_ids + batch_cnt.size() < (uint64_t)vocab_size && batch_cnt.size() < max_batch) {
      std::pop_heap(heap.begin(), heap.end(), HeapCmp());
      const Cand c = heap.back();
      heap.pop_back();
      // rule_intersection (bpe.cpp:145-147) against every earlier rule of the batch: x == some y_j or y == some x_j
      const bool intersects = (in_batch[c.x] & 2) || (in_batch[c.y] & 1);
      if (intersects) { closed = true; break; }  // its exact count after the earlier rules is unknown: close the batch here
      in_batch[c.x] |= 1;
      in_batch[c.y] |= 2;
      const uint32_t z = (uint32_t)(used_ids + batch_cnt.size());
      batch_xyz.push_back(c.x);
      batch_xyz.push_back(c.y);
      batch_xyz.push_back(z);
      batch_cnt.push_back(c.cnt);
      if (c.x == c.y) { closed = true; break; }  // a self-pair rule must be the last of its batch
    }
      if (used_ids + batch_cnt.size() >= (uint64_t)vocab_size || batch_cnt.size() >= max_batch) closed = true;
      if (closed || !heap.empty()) break;
      // The candidates ran out before an intersection closed the batch: the rules picked so far are exact, and so is every further
      // one taken from a LARGER complete prefix of the order -- the counts have not changed, nothing was applied yet.  One more scan
      // with a lower threshold (a launch of its own, ~20 us) instead of a whole merge round for what may be a handful of rules.
      if (heap.empty() && extend_on && !closed && extensions_this_round < 4 && used_ids + batch_cnt.size() < (uint64_t)vocab_size &&
          batch_cnt.size() < max_batch && total_pairs > n) {
        const unsigned long long want = std::max<unsigned long long>(4ull * n, 2048ull);
        const unsigned long long tau2 = choose_tau(g.last_hist(), want, g.last_top_bin());
        if (tau2 < tau) {
          const uint32_t n2 = g.candidates(tau2, MX_ALL, recs, nullptr);
          extensions_this_round++;
          if (n2 > n && n2 <= recs.size()) {
            batch_extensions++;
            batch_keys.clear();
            for (size_t j = 0; j < batch_cnt.size(); j++) batch_keys.push_back(((unsigned long long)batch_xyz[3 * j] << 32) | batch_xyz[3 * j + 1]);
            std::sort(batch_keys.begin(), batch_keys.end());
            for (uint32_t i = 0; i < n2; i++)
              if (!std::binary_search(batch_keys.begin(), batch_keys.end(), recs[i].key))
                heap.push_back(Cand{recs[i].cnt, (uint32_t)(recs[i].key >> 32), (uint32_t)recs[i].key});
            std::make_heap(heap.begin(), heap.end(), HeapCmp());
            n = n2;
            tau = tau2;
          }
        }
      }
      if (heap.empty()) break;
    }
    uint32_t k = (uint32_t)batch_cnt.size();
    for (uint32_t j = 0; j < k; j++) in_batch[batch_xyz[3 * j]] = in_batch[batch_xyz[3 * j + 1]] = 0;
    // Word mode runs a batch that fits the kernel arguments as ONE launch (~70 us a round late in training) and a larger one as four
    // (~200 us): a batch of up to twice that many rules goes as two rounds -- its first BATCH_ARGS_MAX rules now (a prefix of a batch
    // is a batch), the others come back with the next scan, their counts untouched (nothing of this batch intersects them).
    if (split_on && g.one_launch_rounds() && k > (uint32_t)BATCH_ARGS_MAX && k <= 2u * (uint32_t)BATCH_ARGS_MAX) {
      k = (uint32_t)BATCH_ARGS_MAX;
      batch_xyz.resize(3 * (size_t)k);
      batch_cnt.resize(k);
      closed = true;
      batch_splits++;
    }
    if (root)
      for (uint32_t j = 0; j < k; j++)
        if (batch_xyz[3 * j + 2] % 1000 == 0)
          fprintf(stderr, "id: %u=%u+%u  freq: %llu\n", batch_xyz[3 * j + 2], batch_xyz[3 * j], batch_xyz[3 * j + 1], batch_cnt[j]);  // cf. bpe.cpp:1198-1219
    const bool exhausted = !closed;  // the batch ended for lack of candidates, not at an intersection: ask for more next time
    if (exhausted) rounds_exhausted++;
    w_pick += since(tw1);
    auto tw2 = clk::now();
    g.merge_apply(batch_xyz.data(), k, batch_cnt.data(), &tau_hint, MX_ALL, refine_on ? (uint32_t)TARGET : 0u);  // (the next scan's threshold rides along)
    w_apply += since(tw2);
    if (trace_pick) {
      unsigned long long lc[3];
      g.last_round_counts(lc);  // (of the round whose mailbox this round waited for, like the device time)
      round_lines.push_back(RoundLine{(float)(w_wait_this * 1e6), (float)(std::chrono::duration<double>(tw2 - tw1).count() * 1e6), (float)(since(tw2) * 1e6), (float)(dev_ms_this * 1e3), k, lc[0], lc[1], lc[2]});
    }
    for (uint32_t j = 0; j < k; j++) rules.push_back(BPE_Rule{batch_xyz[3 * j], batch_xyz[3 * j + 1], batch_xyz[3 * j + 2]});
    used_ids += k;
    rounds++;
    if (!target_fixed) {
      batch_ema = exhausted ? std::max(batch_ema, 2.0 * (double)k) : 0.9 * batch_ema + 0.1 * (double)k;
      TARGET = (unsigned long long)std::min(target_max, std::max(32.0, 4.0 * batch_ema));
    }
    // next threshold: keep about TARGET candidates above it (any threshold is valid, see above)
    tau = tau_hint;
    tau_mx = MX_ALL;
  }
  {
    if (rep) rep->seconds_merge = since(t_merge);
    if (trace_pick && g.fused_rounds)
      fprintf(stderr, "[yttm] fused rounds %llu: tail set-up %.2f us, top-list scan %.2f us (%.0f entries), publish %.2f us per round\n", g.fused_rounds,
              g.tail_ticks[0] * 0.01 / g.fused_rounds, g.tail_ticks[1] * 0.01 / g.fused_rounds, (double)g.tail_listed / g.fused_rounds,
              g.tail_ticks[2] * 0.01 / g.fused_rounds);
    if (trace_pick && !round_lines.empty()) {
      // line i: the wait for round i-1's mailbox (its device time is beside it), then round i's pick and launch
      const size_t cuts[] = {0, 11, 28, 46, 100, 200, 300, 450, 700, 1500, 3000, 1u << 30};
      for (size_t c = 0; c + 1 < sizeof cuts / sizeof cuts[0] && cuts[c] < round_lines.size(); c++) {
        const size_t a = cuts[c], b = std::min(cuts[c + 1], round_lines.size());
        double w = 0, p = 0, ap = 0, d = 0, kk = 0, apmax = 0, si = 0, to = 0, tk = 0;
        for (size_t i = a; i < b; i++) {
          w += round_lines[i].wait_us; p += round_lines[i].pick_us; ap += round_lines[i].apply_us; d += round_lines[i].dev_us; kk += round_lines[i].k; apmax = std::max<double>(apmax, round_lines[i].apply_us);
          if (i + 1 < round_lines.size()) { si += (double)round_lines[i + 1].sites; to += (double)round_lines[i + 1].touched; tk += (double)round_lines[i + 1].tokens; }  // (line i + 1 holds round i's counts)
        }
        const double m = (double)(b - a);
        fprintf(stderr, "[yttm] rounds %zu-%zu: per round %.1f us = wait for the mailbox %.1f (device %.1f) + pick %.1f + merge_apply call %.1f (max %.0f); batch %.1f rules, %.0f merge sites in %.0f tiles / words, %.0f tokens streamed\n", a + 1, b,
                (w + p + ap) / m, w / m, d / m, p / m, ap / m, apmax, kk / m, si / m, to / m, tk / m);
      }
    }
#ifdef YTTM_K4_PROF
    if (trace_pick) {
      unsigned long long tq[8];
      g.read_stats(24, 8, tq);
      if (tq[7])
        fprintf(stderr, "[yttm] scan_top, first pass, us per scan over %llu scans: list lengths %.2f, slot numbers %.2f, keys + counts %.2f, keep / zero / histogram %.2f, refine %.2f, emit %.2f, compaction %.2f\n", tq[7],
                tq[0] * 0.01 / tq[7], tq[1] * 0.01 / tq[7], tq[2] * 0.01 / tq[7], tq[3] * 0.01 / tq[7], tq[4] * 0.01 / tq[7], tq[5] * 0.01 / tq[7], tq[6] * 0.01 / tq[7]);
    }
#endif
    if (trace_pick) fprintf(stderr, "[yttm] host pick: threshold %.1f ms, heap build %.1f ms of the %.1f; %.0f candidates per round\n", w_pick_a * 1e3, w_pick_b * 1e3, w_pick * 1e3, (double)n_cand_sum / (double)std::max<unsigned long long>(rounds, 1));
    if (trace_pick) fprintf(stderr, "[yttm] merge loop wall: candidates %.1f ms, host pick %.1f ms, merge_apply %.1f ms, repacks %llu (%llu looks), hot rebuilds %llu, top refills %llu, index builds %llu (%llu rounds in word mode from round %llu on, %llu of them over every word), pair table %llu keys in %llu slots (%llu rehashes)\n",
                                       w_cand * 1e3, w_pick * 1e3, w_apply * 1e3, g.repacks, g.repack_looks, g.hot_rebuilds, g.top_refills, g.index_builds, g.word_rounds, g.word_switch_round, g.word_all_rounds, g.n_keys_host, g.table_capacity(), g.rehashes);
  }
  if (rep) {
    rep->rounds = rounds;
    rep->cand_rescans = rescans;
    rep->rounds_exhausted = rounds_exhausted;
    rep->replicated_merge_loop = replicated ? 1 : 0;
    rep->batch_extensions = batch_extensions;
    rep->batch_splits = batch_splits;
    rep->hot_rebuilds = g.hot_rebuilds;
    rep->fused_rounds = g.fused_rounds;
    rep->fused_overflows = g.fused_overflows;
    rep->exchange_retries = g.exchange_retries;
    rep->word_table_retries = g.word_table_retries;
    rep->front_end_overlapped = g.front_end_overlapped ? 1 : 0;
    rep->front_end_chunks = g.front_end_chunks;
    rep->classb_overlapped = g.classb_overlapped;
    rep->k3_radix = g.k3_radix;
    rep->peak_device_bytes = g.peak_device_bytes();
    rep->top_refills = g.top_refills;
    rep->index_builds = g.index_builds;
    rep->word_rounds = g.word_rounds;
    rep->word_switch_round = g.word_switch_round;
    rep->word_all_rounds = g.word_all_rounds;
    rep->word_fused_rounds = g.word_fused_rounds;
    rep->rules = rules.size();
    rep->n_unique = g.n_unique;
    rep->n_tokens = g.n_tokens0;
    rep->corpus_bytes = g.corpus_bytes;
  }

  // ---- rename_tokens (bpe.cpp:814-837) + dump (utils.cpp:50-66)
  const auto t_io = clk::now();
  std::vector<uint32_t> ren((size_t)vocab_size + 1, 0);
  {
    uint32_t cur = n_special;
    for (int i = 0; i < vocab_size; i++)
      if (!cfg.special_tokens.taken_id(i)) ren[cur++] = (uint32_t)i;
  }
  BPEState st;
  st.special_tokens = cfg.special_tokens;
  {
    std::vector<uint32_t> keys(alpha.size());
    for (size_t i = 0; i < alpha.size(); i++) keys[i] = alpha[i].first;
    std::vector<uint32_t> order = flat_hash_map_order(keys);
    std::vector<std::pair<uint32_t, uint32_t>> sorted(alpha);
    std::sort(sorted.begin(), sorted.end());
    for (uint32_t cp : order) {
      auto it = std::lower_bound(sorted.begin(), sorted.end(), std::make_pair(cp, 0u));
      st.char2id.emplace_back(cp, ren[it->second]);
    }
  }
  for (auto &r : rules) st.rules.push_back(BPE_Rule{ren[r.x], ren[r.y], ren[r.z]});
  Status s;
  if (root && !model_path.empty()) {
    s = st.dump(model_path);
    if (s.ok()) fprintf(stderr, "model saved to: %s\n", model_path.c_str());
  }
  if (state_out) *state_out = st;
  if (rep) {
    rep->seconds_io = since(t_io);
    rep->seconds_total = since(t_all);
    g.resolve_timers();
    rep->repacks = g.repacks;
    rep->merge_sites = g.merge_sites;
    rep->touched_tiles = g.touched_tiles; rep->touched_tile_tokens = g.touched_tile_tokens;
    rep->touched_words = g.touched_words; rep->touched_word_tokens = g.touched_word_tokens;
    rep->split_round = g.split_round; rep->split_touched_words = g.split_touched_words; rep->split_touched_word_tokens = g.split_touched_word_tokens;
    rep->split_sites = g.split_sites; rep->merge_ms_words = g.merge_ms_words; rep->merge_launches_words = g.merge_launches_words;
    for (int i = 0; i < 8; i++) { rep->kt_ms[i] = g.kt.ms[i]; rep->kt_launches[i] = g.kt.launches[i]; rep->kt_bytes[i] = g.kt.bytes[i]; }
  }
  return s;
}

static void print_config(const std::string &input_path, const std::string &model_path, int vocab_size, const BpeConfig &c) {
  // bpe.cpp:1352-1366
  fprintf(stderr, "Training parameters\n  input: %s\n  model: %s\n  vocab_size: %d\n  n_threads: %d\n  character_coverage: %g\n", input_path.c_str(),
          model_path.c_str(), vocab_size, c.n_threads, c.character_coverage);
  fprintf(stderr, "  pad: %d\n  unk: %d\n  bos: %d\n  eos: %d\n\n", c.special_tokens.pad_id, c.special_tokens.unk_id, c.special_tokens.bos_id,
          c.special_tokens.eos_id);
}

template <class F>
static Status guarded(F &&f) {
  try {
    return f();
  } catch (const GpuError &e) {
    return Status(2, "GPU error: " + e.msg);
  } catch (const std::exception &e) {
    return Status(2, std::string("error: ") + e.what());
  }
}

Status train_bpe_from_device(const void *d_text, unsigned long long n, const std::string &model_path, int vocab_size, BpeConfig cfg,
                             int device, TrainReport *report, Comm *comm, int profile) {
  Status st = check_config(cfg, vocab_size);
  if (!st.ok()) return st;
  return guarded([&]() {
    GpuCtx g(device);
    const CfgBind bind(g.config_ptr());  // (the launchers' own hook reads see this training's snapshot whatever another thread refreshes)
    g.profile = profile && !g.config().no_profile.set;  // (tuning hook: what do the timing events themselves cost?)
    g.instrument = profile == 2;
    if (g.instrument && g.config().measure_split_round.set) g.split_round = g.config().measure_split_round.u;
    g.set_comm(comm);
    g.attach_corpus(d_text, n);
    return learn_bpe(g, vocab_size, model_path, cfg, nullptr, report);
  });
}

Status train_bpe_from_memory(const uint8_t *text, unsigned long long n, const std::string &model_path, int vocab_size, BpeConfig cfg,
                             int device, TrainReport *report, Comm *comm) {
  Status st = check_config(cfg, vocab_size);
  if (!st.ok()) return st;
  return guarded([&]() {
    GpuCtx g(device);
    const CfgBind bind(g.config_ptr());  // (the launchers' own hook reads see this training's snapshot whatever another thread refreshes)
    g.set_comm(comm);
    g.upload_corpus(text, n);
    return learn_bpe(g, vocab_size, model_path, cfg, nullptr, report);
  });
}

Status train_bpe(const std::string &input_path, const std::string &model_path, int vocab_size, BpeConfig cfg, int device,
                 TrainReport *report, Comm *comm, int profile) {
  Status st = check_config(cfg, vocab_size);
  if (!st.ok()) return st;
  print_config(input_path, model_path, vocab_size, cfg);
  fprintf(stderr, "reading file...\n");
  int fd = open(input_path.c_str(), O_RDONLY);
  if (fd < 0) return Status(1, "Failed to open file: " + input_path);  // bpe.cpp:72
  struct stat sb;
  if (fstat(fd, &sb) != 0) { close(fd); return Status(1, "Failed to open file: " + input_path); }
  const unsigned long long size = (unsigned long long)sb.st_size;
  // multi-GPU: each rank takes the byte range [size*r/W, size*(r+1)/W), advanced to the next ASCII space like the
  // reference's per-thread split (bpe.cpp:864-873)
  unsigned long long lo = 0, hi = size;
  if (comm && comm->world > 1) {
    auto split = [&](int i) {
      if (i == 0) return 0ull;  // split_pos[0] = 0 (bpe.cpp:865)
      unsigned long long c = size * (unsigned long long)i / (unsigned long long)comm->world;
      uint8_t buf[4096];
      while (c < size) {
        const ssize_t got = pread(fd, buf, sizeof buf, (off_t)c);
        if (got <= 0) return size;
        for (ssize_t k = 0; k < got; k++)
          if (buf[k] == 32 || (buf[k] >= 9 && buf[k] <= 13)) return c + (unsigned long long)k;
        c += (unsigned long long)got;
      }
      return c;
    };
    lo = split(comm->rank);
    hi = split(comm->rank + 1);
  }
  fprintf(stderr, "learning bpe...\n");
  const auto t_call = clk::now();
  double s_ctor = 0, s_body = 0;
  Status r = guarded([&]() {
    GpuCtx g(device);
    const CfgBind bind(g.config_ptr());  // (the launchers' own hook reads see this training's snapshot whatever another thread refreshes)
    s_ctor = since(t_call);
    g.profile = profile && !g.config().no_profile.set;
    g.set_comm(comm);
    const auto t_up = clk::now();
    g.upload_corpus_fd(fd, lo, hi - lo);  // file -> pinned chunks -> HBM (replaces fast_read_file_utf8, bpe.cpp:67-84)
    const double s_up = since(t_up);
    Status st2 = learn_bpe(g, vocab_size, model_path, cfg, nullptr, report);
    if (report) { report->seconds_upload = s_up; report->seconds_total += s_up; }
    s_body = since(t_call);
    return st2;
  });
  if (yttm::cfg()->trace.set) fprintf(stderr, "[yttm] train_bpe: context set-up %.2f ms, upload + training %.2f ms, context tear-down %.2f ms\n", s_ctor * 1e3, (s_body - s_ctor) * 1e3, (since(t_call) - s_body) * 1e3);
  close(fd);
  return r;
}

}  // namespace yttm
