// gpu_words.cpp -- word mode (class-A words processed from worklists) and the pair index behind it: set-up and rebuilds; the rounds themselves are GpuCtx::merge_apply.
// (Round 5: cut out of gpu_ctx.cpp, code motion only; gpu_ctx_internal.h says what went where.)
#include "gpu_ctx_internal.h"

namespace yttm {

void GpuCtx::free_words() {
  DFREE(d_wmeta_); DFREE(d_gm_); DFREE(d_xyz_); DFREE(d_wworklist_); DFREE(d_drec_); DFREE(d_drec_n_); DFREE(d_irec_);
  DFREE(tl_.base); DFREE(tl_.cap); DFREE(tl_.fill); DFREE(tl_.rec_word); DFREE(tl_.rec_l); DFREE(tl_.rec_r); DFREE(tl_.cursor);
  tl_ = TokLists{};
  word_mode_ = false;
  word_global_ = false;
  sites_last_ = ~0ull;
  g_sites_last_ = ~0ull;
  g_sites_cum_ = g_tokens_cum_ = g_tokens_last_ = g_tiles_a_ = 0;
}

// The switch to word mode (k_words.hip): from here on class-A words live in the slots they have now and a round visits the words
// that hold a merge site.  Called between rounds.
void GpuCtx::enter_word_mode(uint32_t z_next) {
  WordClass &c = cls_[0];
  chain_event_ = nullptr;
  d_wmeta_ = dmalloc<unsigned long long>(c.n_unique + 1);
  launch_words_init(c.ts, d_wmeta_, strm());
  d_wworklist_ = dmalloc<uint32_t>(c.n_unique + 64);
  HIP_CHECK(hipMemsetAsync(c.d_work_n, 0, 64, strm()));
  d_gm_ = dmalloc<unsigned int>(WGATHER_MAXK + 4);
  HIP_CHECK(hipMemsetAsync(d_gm_, 0, (WGATHER_MAXK + 4) * 4, strm()));
  d_xyz_ = dmalloc<uint32_t>(3 * (size_t)RULES_CAP);
  drec_cap_ = (unsigned int)cfg_->word_drec.u;  // (tests: a region that overflows)
  d_drec_ = dmalloc<DeltaRec>((size_t)WORDS_MAX_GRID * drec_cap_);
  d_drec_n_ = dmalloc<unsigned int>(WORDS_MAX_GRID);
  d_irec_ = dmalloc<uint4>((size_t)WORDS_MAX_GRID * drec_cap_);
  tl_.base = dmalloc<unsigned long long>(id_cap_);
  tl_.cap = dmalloc<uint32_t>(id_cap_);
  tl_.fill = dmalloc<uint32_t>(id_cap_);
  HIP_CHECK(hipMemsetAsync(tl_.base, 0, (size_t)id_cap_ * 8, strm()));
  HIP_CHECK(hipMemsetAsync(tl_.cap, 0, (size_t)id_cap_ * 4, strm()));
  HIP_CHECK(hipMemsetAsync(tl_.fill, 0, (size_t)id_cap_ * 4, strm()));
  tl_.cursor = dmalloc<unsigned long long>(2);
  HIP_CHECK(hipMemsetAsync(tl_.cursor, 0, 16, strm()));
  // every record ever matched is a site at most once through each of its two neighbours, and a site removes a token: a few records per
  // live token bound the log between two index builds; should it fill up all the same, the round says so and the index is rebuilt
  const unsigned long long live = std::max<unsigned long long>(live_tokens_last_, 1ull << 16);
  const unsigned long long log_env = cfg_->word_log.u;  // (tests: a log that overflows)
  tl_.log_cap = log_env ? log_env : 2 * live + (1ull << 20);
  tl_.rec_word = dmalloc<uint32_t>(tl_.log_cap);
  tl_.rec_l = dmalloc<uint32_t>(tl_.log_cap);
  tl_.rec_r = dmalloc<uint32_t>(tl_.log_cap);
  tl_.broken = (unsigned int *)((unsigned char *)h_pin_ + PIN_BYTES - 64);  // (the last line of the pinned block -- behind the mailbox, the candidates' read-back
                                                                            // area and the batch staging; the kernels write it with system-scope stores)
  *(volatile unsigned int *)tl_.broken = 0;
  word_mode_ = true;
  word_global_ = true;
  word_live_tokens_ = std::min<unsigned long long>(c.n_tokens0, live_tokens_last_ ? live_tokens_last_ : c.n_tokens0);
  word_sites_seen_ = sites_cum_;
  word_switch_round = merge_rounds;
  idx_valid_ = false;
  idx_pending_ = true;
  if (cfg_->trace.set) fprintf(stderr, "[yttm] word mode from round %llu on: %llu words, last round %llu sites, %llu tokens streamed; log %llu records\n", merge_rounds,
                                  c.n_unique, sites_last_, live_tokens_last_, tl_.log_cap);
  build_index(z_next);
}

void GpuCtx::free_index() {
  DFREE(idx_.key); DFREE(idx_.cnt); DFREE(idx_.off); DFREE(idx_.bloom); DFREE(idx_.post); DFREE(d_stamp_); DFREE(idx_scan_tmp_); DFREE(idx_save_);
  idx_cap_ = post_cap_ = 0;
  stamp_cap_ = 0;
  idx_valid_ = false;
}

// (Re)builds the pair index of word mode from the hot list as it is now and the class-A words as they are now (see k_index_core.h PairIndex).
// Called between rounds.
void GpuCtx::build_index(uint32_t z_next) {
  idx_pending_ = false;
  idx_valid_ = false;
  WordClass &c = cls_[0];
  if (!c.n_tiles || hot_state_ != HOT_ACTIVE || !word_mode_) return;
  chain_event_ = nullptr;
  unsigned int listed = 0;  // (the list has grown since the scan that last reported its length)
  HIP_CHECK(hipMemcpyAsync(&listed, d_hot_n_, 4, hipMemcpyDeviceToHost, strm()));
  sync();
  if (listed > hot_cap_) return;  // overflowed: the next scan rebuilds the list, and the index after it
  unsigned long long want = 1024;
  while (want < 2ull * ((unsigned long long)listed + 256)) want <<= 1;
  if (want > idx_cap_) {
    DFREE(idx_.key); DFREE(idx_.cnt); DFREE(idx_.off);
    idx_.key = dmalloc<unsigned long long>(want);
    idx_.cnt = dmalloc<uint32_t>(want * IDX_SHARDS + 1);
    idx_.off = dmalloc<unsigned long long>(want * IDX_SHARDS + 2);
    DFREE(idx_scan_tmp_);
    idx_scan_tmp_ = dmalloc<unsigned long long>(scan_scratch_blocks(want * IDX_SHARDS + 1));
    idx_cap_ = want;
  }
  if (!idx_.bloom) idx_.bloom = dmalloc<uint32_t>(ENC_BLOOM_WORDS);
  idx_.mask = (unsigned int)(want - 1);
  launch_fill_u64(idx_.key, PT_EMPTY, want, strm());
  HIP_CHECK(hipMemsetAsync(idx_.cnt, 0, want * IDX_SHARDS * 4, strm()));
  HIP_CHECK(hipMemsetAsync(idx_.bloom, 0, ENC_BLOOM_WORDS * 4, strm()));
  t_begin(KT_CAND);
  launch_idx_seed(pt_, idx_, listed, strm());
  if (!idx_save_) idx_save_ = dmalloc<unsigned char>(idx_save_bytes());
  launch_idx_stream(false, c.ts, idx_, strm(), true, idx_save_);
  // offsets = exclusive scan of the counts (one extra zero count behind the last slot: off[mask + 1] = the total)
  HIP_CHECK(hipMemsetAsync(idx_.cnt + want * IDX_SHARDS, 0, 4, strm()));
  launch_exclusive_scan(idx_.cnt, want * IDX_SHARDS + 1, idx_.off, idx_scan_tmp_, d_counters_ + 56, strm());
  HIP_CHECK(hipMemsetAsync(idx_.cnt, 0, want * IDX_SHARDS * 4, strm()));  // the fill pass's cursors
  unsigned long long total = 0;
  HIP_CHECK(hipMemcpyAsync(&total, d_counters_ + 56, 8, hipMemcpyDeviceToHost, strm()));
  sync();
  index_builds++;
  if (cfg_->trace.set) fprintf(stderr, "[yttm] index build at round %llu: %u listed pairs, %llu postings, %u tiles, last round touched %llu tiles\n", merge_rounds, listed, total, c.n_tiles, touched_last_);
  if (total == 0 || total > 0xfffffff0ull) {  // (no postings, or more than the 32-bit run offsets hold: the rounds take every word)
    t_end(KT_CAND, 4ull * c.n_tiles * c.nom);
    return;
  }
  if (total > post_cap_) {
    DFREE(idx_.post);
    post_cap_ = total + total / 4 + 1024;
    idx_.post = dmalloc<uint32_t>(post_cap_);
  }
  launch_idx_stream(true, c.ts, idx_, strm(), /*agg=*/total > idx_agg_min_, idx_save_);
  t_end(KT_CAND, 8ull * c.n_tiles * c.nom);
  const unsigned long long stamps = c.n_unique;  // (a posting is a word, and a round claims words)
  if (stamps > stamp_cap_) {
    DFREE(d_stamp_);
    stamp_cap_ = (unsigned int)(stamps + stamps / 8 + 64);
    d_stamp_ = dmalloc<uint32_t>(stamp_cap_);
  }
  HIP_CHECK(hipMemsetAsync(d_stamp_, 0, (size_t)stamp_cap_ * 4, strm()));
  {  // every token that exists now is covered by the postings: the instance lists start over
    HIP_CHECK(hipMemsetAsync(tl_.cursor, 0, 16, strm()));
    sync();
    *(volatile unsigned int *)tl_.broken = 0;
  }
  idx_valid_ = true;
  idx_zbuild_ = z_next;
}

}  // namespace yttm
