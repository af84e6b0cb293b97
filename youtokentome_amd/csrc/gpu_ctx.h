// gpu_ctx.h -- device context of the BPE trainer: owns every HBM buffer, one HIP stream, and (multi-GPU) the RCCL
// communicator.  The C-ABI in include/yttm_gpu.h is a thin wrapper over this class.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <functional>
#include <string>
#include <vector>

#include "yttm_config.h"
#include "yttm_kernels.h"

namespace yttm {

struct GpuError {
  std::string msg;
};

// n bytes between HBM and the host through pinned chunks that several host threads fill (to_device) or drain (gpu_ctx.cpp): what a single
// hipMemcpy from / to pageable memory does on one thread through one internal buffer.  host_side(chunk, off, len) -> false: give up.
// arrived(off, len) (to_device only, may be empty): bytes [off, off + len) are in HBM -- called from the workers' threads, chunks in no order.
// host_side and arrived must be THREAD-SAFE: several workers call them at once, for different chunks (the upload of a corpus: pread / memcpy
// of disjoint ranges; GpuCtx::upload_overlapped also samples the source through the same callback before the workers start).
void staged_transfer(int device, uint8_t *d_ptr, unsigned long long n, bool to_device,
                     const std::function<bool(void *chunk, unsigned long long off, size_t len)> &host_side,
                     const std::function<void(unsigned long long off, size_t len)> &arrived = nullptr, size_t chunk_bytes = 0 /* 0: staged_chunk_bytes() */);
size_t staged_chunk_bytes();  // the default chunk size (8 MB; YTTM_IO_CHUNK_MB / _KB)

// Exchange interface for the multi-GPU path (one process per GPU).  Implementations: RCCL over xGMI
// (comm_rccl.cpp) and a host-callback variant used by the gloo CPU tests.
struct Comm {
  int rank = 0, world = 1;
  virtual ~Comm() {}
  // ---- set-up path (once per training; may synchronise the stream)
  // in-place sum of n uint64 values living in device memory
  virtual void allreduce_sum_u64(unsigned long long *dev, size_t n, hipStream_t st) = 0;
  // every rank contributes n_local records (~0: "my send buffer overflowed"); recv (device, capacity cap records) receives the
  // records of all OTHER ranks back to back.  *need_all = records of ALL ranks -- the same number on every rank -- or ~0 if
  // some rank reported an overflow.  If it does not fit `cap` nothing is transferred (every rank takes that branch together)
  // and false is returned; the caller grows its buffer and calls again.
  virtual bool allgather_recs(const DeltaRec *send, unsigned long long n_local, DeltaRec *recv, size_t cap, hipStream_t st,
                              unsigned long long *need_all, size_t *n_remote) = 0;
  // ---- per-round path: stream-ordered, NO host synchronisation
  // block r of recv (bytes_per_rank each, every rank's including this one's) = rank r's send block
  virtual void allgather_blocks(const void *send, void *recv, size_t bytes_per_rank, hipStream_t st) = 0;
};

// returns the device (and pinned) memory cached by finished contexts to the driver
void release_device_memory();

struct KernelTimes {  // accumulated GPU time per kernel family, measured with HIP events on the ctx stream
  double ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long launches[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long bytes[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // algorithmic bytes
};
enum { KT_CHAR_HIST = 0, KT_SEGS = 1, KT_DEDUP = 2, KT_BUILD = 3, KT_PAIR_COUNT = 4, KT_MERGE = 5, KT_CAND = 6, KT_XCHG = 7 };  // XCHG (multi-GPU): pack + all-gather + fold + the round's scan, on the device clock

class GpuCtx {
 public:
  explicit GpuCtx(int device);
  ~GpuCtx();

  // ---- corpus
  void upload_corpus(const uint8_t *host, unsigned long long n);
  void attach_corpus(const void *dev, unsigned long long n);
  // bytes [lo, lo + n) of an open file, through pinned chunks filled by several host threads (gpu_ctx.cpp: upload_staged)
  void upload_corpus_fd(int fd, unsigned long long lo, unsigned long long n);
  // multi-GPU: the ranks' shards gathered into one corpus on every rank (gpu_ctx.cpp); a value summed over the ranks
  void gather_full_corpus();
  unsigned long long allreduce_scalar(unsigned long long v);
  // HBM this context could still allocate: what the driver reports free plus the blocks the pool holds (YTTM_TEST_FREE_BYTES: tests)
  unsigned long long free_device_bytes() const;

  // ---- K1
  void char_hist(std::vector<uint32_t> &cps, std::vector<unsigned long long> &cnts, unsigned long long &n_codepoints);
  // ---- K2: builds the unique-word token tiles from the char->id map (chars not listed are deleted)
  void build_word_table(const uint32_t *cp, const uint32_t *id, uint32_t n_alpha, uint32_t space_id, uint32_t n_ids_cap);
  void download_word_table(std::vector<uint32_t> &tok, std::vector<unsigned long long> &off, std::vector<uint32_t> &cnt);
  // ---- K3
  void pair_count();
  void download_pairs(std::vector<unsigned long long> &keys, std::vector<unsigned long long> &cnts);
  // ---- K4: apply a batch of mutually non-intersecting rules (x,y,z)*k
  // next_tau_cnt / next_tau_mx (optional): the threshold of the candidate scan that will follow this round.  When the round is
  // one launch (single GPU, class-A tiles only, hot list active) that scan runs inside it (k_merge_shared.h scan_top) and the
  // next candidates() call with the same threshold only waits for the mailbox.
  // next_want != 0: about that many candidates are wanted from the fused scan -- it may then raise the threshold by itself; the caller reads
  // the threshold that was used back as the smallest count among the candidates (they are every pair at or above it)
  void merge_apply(const uint32_t *xyz, uint32_t k, const unsigned long long *rule_counts, const unsigned long long *next_tau_cnt = nullptr,
                   uint32_t next_tau_mx = 0xffffffffu, uint32_t next_want = 0);
  void pair_query(const unsigned long long *keys, uint32_t n, unsigned long long *out);
  // candidate filter; returns number of candidates that passed (may exceed out.size() capacity => retry with higher tau)
  // Pairs with count > tau_cnt, or == tau_cnt and max(x,y) <= tau_mx (a complete prefix of the pick order), + histogram
  // of the counts the filter looked at.  Served from the hot list (counts >= hot_tau(); a lower tau_cnt is raised to it).
  uint32_t candidates(unsigned long long tau_cnt, uint32_t tau_mx, std::vector<CandRec> &out, unsigned long long *hist /*CAND_BINS or null*/);
  // the same over the whole table (one streaming pass)
  uint32_t scan_full(unsigned long long tau_cnt, uint32_t tau_mx, std::vector<CandRec> &out, unsigned long long *hist);
  unsigned long long hot_tau() const { return hot_state_ == HOT_ACTIVE ? pt_.hot_tau : 1; }
  // histogram (CAND_BINS bins) and number of the counts the last candidates() / scan_full() call looked at; the histogram lives in
  // the pinned mailbox until the next scan
  const unsigned long long *last_hist() const { return last_hist_; }
  unsigned long long last_live() const { return last_live_; }
  int last_top_bin() const { return (int)last_top_bin_; }  // no bin above this one is in use
  unsigned long long index_builds = 0, word_rounds = 0, word_switch_round = 0, word_all_rounds = 0, word_fused_rounds = 0;  // K4 rounds whose worklist came from the pair index
  unsigned long long classb_overlapped = 0;   // word-mode rounds whose class-B tiles ran beside k_words on a second stream
  unsigned long long k3_radix = 0;            // 1: K3 of class A ran by radix partition (k_pairradix.hip)
  unsigned long long front_end_chunks = 0;    // > 0: the corpus was taken in this many chunks (front_end_chunked)
  bool corpus_resident() const { return !chunked_; }  // false: only the distinct words' bytes are in HBM
  unsigned long long peak_device_bytes() const;       // high-water mark of the device memory pool since this context was made
  unsigned long long word_table_retries = 0;  // K2: the word table had to be redone with the worst-case size
  bool front_end_overlapped = false;          // K1, K2a, K2b ran under the upload and the word table they made was taken (upload_overlapped)
  unsigned long long hot_rebuilds = 0, top_refills = 0, rehashes = 0, exchange_retries = 0, delta_regrows = 0;
  unsigned long long tail_ticks[3] = {0, 0, 0}, tail_listed = 0;  // round_tail: fold / list scan / publish, in 10 ns ticks; entries it read
  unsigned long long fused_rounds = 0, fused_overflows = 0;  // rounds whose candidate scan ran in the apply kernel's tail; of those, with a hot-list overflow
  unsigned long long table_capacity() const { return pt_cap_; }

  void sync();
  void set_comm(Comm *c) { comm_ = c; }
  Comm *comm() const { return comm_; }
  int device() const { return device_; }
  const Config &config() const { return *cfg_; }  // the YTTM_* hooks as they stood when this context was made (yttm_config.h)
  std::shared_ptr<const Config> config_ptr() const { return cfg_; }  // (for CfgBind: the launchers' tuning hooks read cfg() on the calling thread)
  hipStream_t stream() const { return st_raw_; }

  unsigned long long n_unique = 0, n_tokens0 = 0, n_segments = 0, corpus_bytes = 0;
  unsigned int n_tiles = 0;
  unsigned long long n_keys_host = 0;
  bool profile = false;
  KernelTimes kt;
  unsigned long long merge_sites = 0, merge_rounds = 0, repacks = 0, repack_looks = 0;  // (looks: scans of a class's tile fills, each with a stream synchronisation)
  // K4 totals over the training: tiles that held a site and their tokens; with `instrument` (a measurement pass, never the
  // timed one) also the WORDS that held a site and their tokens = W_touched / T_touched of SURVEY.md 8d
  unsigned long long touched_tiles = 0, touched_tile_tokens = 0, touched_words = 0, touched_word_tokens = 0;
  bool instrument = false;
  // measurement pass: the totals above as they stood after this many rounds (bench: the round the timed run switched to word mode at, so
  // that the contract's bytes can be stated for the tile rounds and the word-mode rounds apart); 0: no snapshot
  unsigned long long split_round = 0, split_touched_words = 0, split_touched_word_tokens = 0, split_sites = 0;
  double last_round_dev_ms = 0;            // the last fused round on the device clock (0: that round was not timed so)
  // the last merge round by the mailbox's counters (YTTM_TRACE): merge sites | tiles (word mode: words) that held one | tokens its kernels streamed
  void last_round_counts(unsigned long long out[3]) const {
    out[0] = sites_last_ == ~0ull ? 0 : sites_last_;
    out[1] = touched_last_ == ~0ull >> 2 ? 0 : touched_last_;
    out[2] = live_tokens_last_;
  }
  double merge_ms_words = 0;               // device-clock time of the word-mode rounds (part of kt.ms[KT_MERGE])
  unsigned long long merge_launches_words = 0;
  void resolve_timers();
  void read_stats(int first, int n, unsigned long long *out);  // (tuning aid: words of the device statistics block, synchronising)

 private:
  void upload_staged(unsigned long long n, const std::function<bool(void *dst, unsigned long long off, size_t len)> &fill);
  void ensure_table_capacity(unsigned long long need_keys);
  void rebuild_hot();
  enum HotState { HOT_INVALID, HOT_ACTIVE, HOT_FULLSCAN };
  HotState hot_state_ = HOT_INVALID;
  enum TopState { TOP_INVALID, TOP_ACTIVE, TOP_BYPASS };  // BYPASS: ties too large for the top list, the hot list is scanned instead
  TopState top_state_ = TOP_INVALID;
  uint32_t *d_top_slots_ = nullptr;
  unsigned int *d_top_n_ = nullptr;
  unsigned int top_cap_ = 0, top_target_ = 0, top_min_ = 0, top_listed_last_ = 0, bypass_rounds_ = 0;
  void poll_mailbox(uint32_t round_id);
  bool scan_hot(unsigned long long t, uint32_t tm);
  bool refill_top();
  bool hot_just_rebuilt_ = false;
  const unsigned long long *last_hist_ = nullptr;
  unsigned long long last_live_ = 0, hist_buf_[CAND_BINS] = {0};
  unsigned int last_top_bin_ = CAND_BINS - 1;
  unsigned long long bound_prev_ = 0;  // new-key bound of the previous round (see merge_apply)
  uint32_t *d_hot_slots_ = nullptr;
  unsigned int *d_hot_n_ = nullptr;
  unsigned int fullscan_rounds_ = 0;
  uint32_t mail_round_ = 0;
  bool fused_pending_ = false;  // a fused scan is in flight / in the mailbox ...
  unsigned long long fused_tau_ = 0;  // ... for this threshold
  uint32_t fused_mx_ = 0, fused_round_ = 0;
  bool no_batch_args_ = false;
  uint32_t *d_bloom_ = nullptr;   // pair filter of a batch that does not travel in the kernel arguments
  const char *trace_rounds_ = nullptr, *dbg_cand_ = nullptr;
  bool fuse_enabled_ = true;  // YTTM_NO_FUSE=1: always the separate scan kernel (tuning hook / tests)
  // word mode, single GPU, YTTM_CLASSB_BESIDE=1: the class-B tiles' launch of a round runs on a second stream beside k_words (merge_apply; ScanArgs::peer_flag)
  bool classb_overlap_ = false;      // YTTM_CLASSB_BESIDE
  hipStream_t st_b_ = nullptr;
  hipEvent_t ev_fork_ = nullptr, ev_join_ = nullptr;
  // A class-B launch on st_b_ is joined with the main stream by the tail's wait for peer_flag: that orders what the TAIL reads (atomics,
  // write-through stores).  The tiles' plain in-place token rewrites are only guaranteed visible at that kernel's END, and nothing on the main
  // stream depends on its end -- so before the main stream next touches class-B tiles itself (a repack, a class-B launch that is not beside,
  // a download) it waits for an event recorded behind the last beside-launch.  Off the common path: a beside-round never calls it.
  bool classb_unjoined_ = false;
  void join_class_b();
  unsigned int *d_bsync_ = nullptr;  // [0] the class-B launch's ticket, [1] the round it has finished
  uint32_t id_min_ = 0, id_max_ = 0;  // id range of the alphabet (K3)
  uint32_t max_id_ = 0xffffffffu;  // largest token id in the tiles (unknown until the word table is built)
  unsigned long long scanned_cum_ = 0, live_tokens_last_ = 0, touched_cum_ = 0, touched_last_ = ~0ull >> 2;  // (first round: dense)
  unsigned int hot_cap_ = 0, hot_target_ = 0, hot_min_ = 0, listed_last_ = 0;
  void alloc_table(PairTable &pt, unsigned long long cap);
  void free_table(PairTable &pt);
  void exchange_deltas();
  void t_begin(int which);
  void t_end(int which, unsigned long long bytes, bool chain = false);

  int device_;
  std::shared_ptr<const Config> cfg_;
  double xchg_margin_ = 3.0;
  // The context's stream.  Every use goes through strm(), which notes that something may have been queued since the host last read a
  // round's mailbox (poll_mailbox clears the note: whatever was queued before the kernel that published is over, and that kernel is past
  // everything but its statistics fold) -- merge_apply asks before it puts a launch on the second stream (class-B tiles beside k_words).
  hipStream_t st_raw_ = nullptr;
  mutable bool st_touched_ = true;
  hipStream_t strm() const {
    st_touched_ = true;
    return st_raw_;
  }
  Comm *comm_ = nullptr;

  // corpus
  const uint8_t *d_text_ = nullptr;
  uint8_t *d_text_owned_ = nullptr;
  unsigned long long n_text_ = 0;
  // K1
  unsigned long long *d_hist_ = nullptr;      // [N_CODEPOINTS]
  uint32_t *d_chunk_segs_ = nullptr;          // [fe_chunks(n_text_)] segment starts per 4 KB chunk (K1 counts them, K2a places them)
  // What upload_corpus_fd has done of the front end while the file was still crossing the link (single GPU: K1, K2a and K2b on the parts that
  // had arrived; see there): char_hist() / build_word_table() take it from here instead of launching the kernels.
  struct FrontSpec {
    bool hist_done = false;             // d_hist_, d_counters_[0..1], d_chunk_segs_ hold K1's results for the whole text
    bool words_done = false;            // ht holds every segment's word, deduplicated under the code-point-as-id map (valid iff the alphabet keeps every char seen)
    unsigned long long n_segs = 0;
    unsigned long long *ht = nullptr;
    unsigned long long ht_cap = 0;
    unsigned int h_status[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool long_segments = false;
  } spec_;
  std::vector<uint32_t> seen_cps_;      // char_hist: the code points that occur
  void drop_spec();
  // ---- corpora that do not fit (gpu_ctx.cpp front_end_chunked): the text crosses the device in chunks, only the distinct words' bytes stay
  bool chunked_ = false;
  unsigned long long chunk_cap_ = 0, lex_cap_ = 0, lex_used_ = 0, chunk_src_n_ = 0;
  std::function<bool(void *dst, unsigned long long off, size_t len)> chunk_src_;  // the source again (a second pass when coverage drops chars)
  unsigned long long chunk_bytes_for(unsigned long long n) const;  // 0: the whole text at once
  void front_end_chunked(bool first_pass);
  bool overlap_front_end(unsigned long long n) const;
  void upload_overlapped(unsigned long long n, const std::function<bool(void *dst, unsigned long long off, size_t len)> &fill);
  unsigned long long *d_counters_ = nullptr;  // small scratch of u64 counters
  // K2
  uint32_t *d_cpmap_ = nullptr;  // [N_CODEPOINTS]
  uint32_t n_alpha_ = 0;
  // token tiles: class 0 = short words (slot 1024), class 1 = long words (slot 4096)
  struct WordClass {
    TileSet ts{};
    uint32_t *d_tok = nullptr, *d_tile_len = nullptr, *d_tile_word0 = nullptr, *d_wcnt = nullptr;
    uint32_t *d_scratch = nullptr;  // class C only (k_giant.hip)
    unsigned int *d_work_n = nullptr;
    unsigned long long n_unique = 0, n_tokens0 = 0;
    unsigned int n_tiles = 0, nom = 0, slot = 0;
  };
  WordClass cls_[3];  // A: words <= TILE_NOM_A tokens, B: <= TILE_NOM_B, C: longer (one workgroup per tile, k_giant.hip)
  void free_class(WordClass &c);
  void build_class(int ci, unsigned long long *uw_pos, uint32_t *uw_len, unsigned int U, uint32_t space_id);
  void maybe_repack(int ci);
  // pair index of word mode (k_index_core.h: PairIndex): keys = the hot list when it was built, postings = class-A words
  PairIndexArgs idx_{};
  unsigned long long idx_cap_ = 0, post_cap_ = 0;
  unsigned long long *idx_scan_tmp_ = nullptr;
  unsigned char *idx_save_ = nullptr;  // the index count pass's per-workgroup tables (k_idx_stream)
  uint32_t *d_stamp_ = nullptr;   // [class-A words] round that claimed the word for its worklist last
  unsigned int stamp_cap_ = 0;
  bool idx_valid_ = false, idx_pending_ = false, idx_enabled_ = true;
  uint32_t idx_zbuild_ = 0;       // token ids below this existed when the index was built
  void build_index(uint32_t z_next);
  void free_index();
  // word mode (k_words.hip): class-A words processed one by one from a worklist of the words that hold a merge site
  bool profile_events_ = false, dev_timing_pending_ = false;  // (merge_apply: dev_timing)
  std::vector<float> dev_round_ms_;
  bool word_mode_ = false, words_enabled_ = true, direct_enabled_ = true;
  // The DECISION that class A runs in word mode.  Single GPU: word_mode_ itself.  Multi-GPU: taken from numbers summed over the ranks'
  // block headers (the same on every rank, in the same round) -- everything that shapes the candidate lists or the batches (the hot
  // list's target, the batch split) follows this flag, never the rank-local word_mode_ (a rank without class-A words stays on tiles).
  bool word_global_ = false;
  unsigned long long g_sites_cum_ = 0, g_tokens_cum_ = 0, g_sites_last_ = ~0ull, g_tokens_last_ = 0, g_tiles_a_ = 0;
 public:
  // class A is in word mode, and a batch of at most this many rules is one launch there (k_words<FUSED>): the trainer's batch split
  // (the conditions merge_apply fuses a round under, as far as they are known before the batch is -- cutting a batch for a
  // round that then runs unfused anyway only adds a round.  Multi-GPU: only what is the same on every rank -- the batches must be.)
  bool one_launch_rounds() const {
    if (!(word_global_ && words_fuse_max_ != 0 && fuse_enabled_ && !instrument && hot_state_ == HOT_ACTIVE && top_state_ == TOP_ACTIVE)) return false;
    return multi() || (word_mode_ && idx_valid_ && !cls_[2].n_tiles);
  }
 private:
  unsigned int hot_target_words_ = 1u << 16, words_inline_max_ = 1u << 18, words_fuse_max_ = 1u << 30, word_hint_floor_ = 16384;
  unsigned int word_div_ = 200;    // switch when (merge sites of the last round) * word_div_ < (tokens a pass over the tiles streams)
  unsigned int word_min_tiles_ = 16384;
  unsigned long long idx_agg_min_ = 16ull << 20, word_min_tokens_ = 48ull << 20;
  unsigned long long *d_wmeta_ = nullptr;
  unsigned int *d_gm_ = nullptr;   // [WGATHER_MAXK] + the gather's ticket
  uint32_t *d_xyz_ = nullptr;      // a batch too large for BatchArgs, as (x, y, z) triples
  uint32_t *d_wworklist_ = nullptr;  // [n_unique + 64] the round's words
  DeltaRec *d_drec_ = nullptr;       // [WORDS_MAX_GRID * drec_cap_] the round's count updates, a region per workgroup of k_words
  unsigned int *d_drec_n_ = nullptr, drec_cap_ = 0;
  uint4 *d_irec_ = nullptr;          // [WORDS_MAX_GRID * drec_cap_] the round's new-instance records before they go to the tokens' lists
  TokLists tl_{};
  unsigned long long word_live_tokens_ = 0, word_sites_seen_ = 0;  // class A in word mode: its live tokens (about), the merge sites already taken off them
  unsigned long long sites_cum_ = 0, sites_last_ = ~0ull;
  void enter_word_mode(uint32_t z_next);
  void free_words();
  unsigned long long rounds_since_check_ = 0;
  // maybe_repack: live tokens of the class at its last look, the merge sites reported by then, the last three batches' summed pair counts
  bool rp_known_[2] = {false, false};
  unsigned long long rp_total_[2] = {0, 0}, rp_sites_at_[2] = {0, 0}, rp_recent_[3] = {0, 0, 0};
  bool pending_zero_ = false, zero_valid_ = false;  // valid: the zero_* members still describe the last batch
  void flush_pending_zero();
  unsigned int zero_cap_ = 0;
  BatchArgs zero_ba_{};  // the batch whose pairs are still to be zeroed, when it travelled as a kernel argument
  unsigned long long zero_self_key_ = 0;
  // pair table
  PairTable pt_{};
  unsigned long long pt_cap_ = 0;
  // per-round staging
  RuleSlot *d_rules_ = nullptr;
  unsigned int rules_cap_ = 0;
  uint32_t id_cap_ = 0;  // token ids are below this (vocab_size + slack)
  unsigned long long *d_stats_ = nullptr;
  void *h_pin_ = nullptr;  // pinned staging (rules + flag updates + candidate header)
  size_t h_pin_bytes_ = 0;
  // candidates
  unsigned char *d_round_ = nullptr;
  CandRec *d_cand_ = nullptr;
  unsigned int cand_cap_ = 0;
  unsigned int *d_cand_n_ = nullptr;
  unsigned long long *d_cand_hist_ = nullptr;
  // multi-GPU delta exchange.  db_ = the round's delta table and send block (yttm_device.h: DeltaBuf); per round the first blk_ 16-byte
  // units of every rank's send block are all-gathered into d_recv_
  DeltaBuf db_{};
  DeltaRec *d_send2_[2] = {nullptr, nullptr}, *d_recv_ = nullptr;  // the two send blocks (rounds alternate: yttm_device.h DeltaBuf)
  const DeltaRec *xch_last_ = nullptr;                           // the block of the exchange under way (a repeat gathers it again)
  void finish_block(unsigned int n_hint);
  unsigned long long recv_cap_ = 0, blk_ = 4096, blk_min_ = 4096, send_cap_ = 0;
  void alloc_delta_table(unsigned long long cap);
  bool delta_cap_forced_ = false;  // YTTM_XCHG_TABLE_CAP (tests: the overflow verdict)
  bool pt_fresh_ = false;  // build_class(0) left an empty pair table of the initial size
  unsigned long long initial_table_keys(unsigned long long n_tok) const;
  unsigned long long *d_xstat_ = nullptr;  // [XSTAT_WORDS] the fold's report on a round's exchange (yttm_kernels.h)
  unsigned int xch_parity_ = 0;            // which send block this round's updates go to
  uint32_t *d_maybe_ = nullptr;            // slots whose count an add of this round saw at or above a list threshold (PairTable::maybe)
  unsigned int *d_maybe_n_ = nullptr, maybe_cap_ = 0;
  unsigned long long xch_sites_ = 0;       // merge sites (summed pair counts of the batch) of the round whose exchange is under way
  double xrate_[2] = {5.0, 5.0};           // delta records of the busiest rank per merge site, last two rounds
  bool multi() const { return comm_ != nullptr; }  // (a communicator of world size 1 still runs the whole exchange path)
  PairTable pt_nolist() const;  // pt_ with the list thresholds off (multi-GPU: the apply kernels and phase 1 of the fold list nothing)
  void exchange_round(unsigned long long only_mask, const ScanArgs *scan);
  bool settle_exchange(unsigned long long xmask, unsigned long long xmax, unsigned long long fatal);
  void grow_recv(unsigned long long need);

  struct Ev { hipEvent_t a, b; int which; };
  std::vector<Ev> evs_;
  hipEvent_t cur_a_ = nullptr, chain_event_ = nullptr;
  std::vector<hipEvent_t> all_events_;  // every event this context took from the pool (each once)
};

#define HIP_CHECK(expr)                                                                                         \
  do {                                                                                                          \
    hipError_t _e = (expr);                                                                                     \
    if (_e != hipSuccess) throw yttm::GpuError{std::string(#expr) + ": " + hipGetErrorString(_e)};             \
  } while (0)

}  // namespace yttm
