// yttm_config.h -- every YTTM_* environment hook of the library in ONE table (round 5; until then 44 names were read by getenv at ~95 sites
// in five files, some of them once per process in function-local statics, so that a test could not change them between two trainings).
//
// The environment is read when a context is made -- GpuCtx's constructor, an encoder's creation, the CLI loops' start -- by cfg_refresh();
// everything else reads the snapshot: launchers and the round loop never call getenv (it walks the whole environment; a round's launch is
// on its critical path).  A hook is a test knob or a tuning knob, never part of the drop-in surface: the defaults are the product.  The table
// below is also the documentation: yttm_config_table() prints it (INTEGRATION.md holds that output; tests/test_abi.py keeps the two in step
// and checks that no other getenv("YTTM_...") exists in the sources).
#pragma once
#include <memory>
#include <string>

namespace yttm {

struct Hook {
  bool set = false;            // the variable exists in the environment
  std::string raw;             // its text ("" if unset)
  unsigned long long u = 0;    // strtoull(raw) -- the default if unset or empty
  long long i = 0;             // strtoll(raw)  -- likewise
  double d = 0;                // atof(raw)     -- likewise
  const char *c_str() const { return set ? raw.c_str() : nullptr; }  // what getenv returned
};

// X(field, "NAME", default as text, kind, "what it does")        kind: test = exists for the test-suite; tune = a measured default, kept
// adjustable for A/B runs; diag = diagnostics / tracing; path = selects a code path that must give the same result (differential tests)
#define YTTM_HOOKS(X)                                                                                                                         \
  /* ---- diagnostics */                                                                                                                     \
  X(trace, "YTTM_TRACE", "", "diag", "print phase / round timings to stderr; its value names the file of per-launch times")               \
  X(trace_rounds, "YTTM_TRACE_ROUNDS", "", "diag", "file: cumulative device statistics after every merge round (adds a sync per round)")  \
  X(trace_blocks, "YTTM_TRACE_BLOCKS", "", "diag", "PROF build: per-workgroup timeline of every 50th round, file prefix")                  \
  X(dbg_cand, "YTTM_DBG_CAND", "", "diag", "file: one line per candidate scan (threshold, list lengths, hash of the candidates)")         \
  X(no_profile, "YTTM_NO_PROFILE", "", "diag", "set: no per-kernel timing even when the caller asks for a profile")                         \
  X(profile_events, "YTTM_PROFILE_EVENTS", "0", "diag", "1: time merge rounds with HIP events instead of the device clock (cross-check)")  \
  X(measure_split_round, "YTTM_MEASURE_SPLIT_ROUND", "0", "diag", "measurement pass: snapshot the touched-word totals after this round")   \
  /* ---- memory, upload */                                                                                                                  \
  X(no_pool, "YTTM_NO_POOL", "0", "tune", "1: no device-memory pool between contexts (every buffer a hipMalloc)")                          \
  X(plain_upload, "YTTM_PLAIN_UPLOAD", "", "path", "set: host-memory corpus by ONE hipMemcpy instead of the pinned chunks")                \
  X(io_chunk_mb, "YTTM_IO_CHUNK_MB", "8", "tune", "size of a pinned staging chunk, MB")                                                    \
  X(io_chunk_kb, "YTTM_IO_CHUNK_KB", "0", "test", "the same in KB (tests: many chunks of a small input)")                                  \
  X(io_threads, "YTTM_IO_THREADS", "0", "tune", "upload / download workers (0: by size, at most 8)")                                       \
  X(test_free_bytes, "YTTM_TEST_FREE_BYTES", "0", "test", "pretend this many bytes of HBM are free (replication's memory check)")         \
  /* ---- front end */                                                                                                                       \
  X(k1_wide, "YTTM_K1_WIDE", "0", "path", "force K1's variant: 1 = wide chars counted in an LDS hash, 0 = global atomics")                 \
  X(fe_overlap_min, "YTTM_FE_OVERLAP_MIN", "33554432", "tune", "texts of at least this many bytes run K1/K2a/K2b under the upload")       \
  X(fe_no_overlap, "YTTM_FE_NO_OVERLAP", "", "path", "set: never run the front end under the upload")                                       \
  X(fe_no_spec, "YTTM_FE_NO_SPEC", "", "path", "set: under the upload only K1, not the speculative dedup by code points")                  \
  X(fe_chunk_mb, "YTTM_FE_CHUNK_MB", "0", "tune", "take the corpus in chunks of this many MB (0: only when the whole text would not fit the free HBM)") \
  X(fe_chunk_kb, "YTTM_FE_CHUNK_KB", "0", "test", "the same in KB (tests: many chunks of a toy corpus)")                                   \
  X(fe_chunk_serial, "YTTM_FE_CHUNK_SERIAL", "", "path", "chunked front end: upload a chunk, then work on it (no landing buffer: one chunk less of HBM)") \
  X(fe_part_kb, "YTTM_FE_PART_KB", "32768", "tune", "size of a part of the text the overlapped front end works on, KB")                    \
  X(fe_k2b_blocks, "YTTM_FE_K2B_BLOCKS", "4096", "tune", "workgroups of a part's dedup launch")                                             \
  X(word_table_full, "YTTM_WORD_TABLE_FULL", "", "path", "set: size the word table for the worst case at once (no estimate, no retry)")    \
  X(test_wcnt_max, "YTTM_TEST_WCNT_MAX", "4294967295", "test", "largest weight a word may carry (tests: words 'seen 2^32 times' at toy sizes)") \
  X(k3_radix_min, "YTTM_K3_RADIX_MIN", "4194304", "path", "K3 of alphabets of 65 .. 8192 symbols: class-A tokens from which it runs by radix partition (tests: 0; huge: never)") \
  X(k3_general, "YTTM_K3_GENERAL", "", "path", "set: K3 through the general tile kernel even on a small alphabet")                          \
  /* ---- candidate lists, pick */                                                                                                           \
  X(hot_cap, "YTTM_HOT_CAP", "262144", "test", "capacity of the hot list (tests: overflows)")                                               \
  X(hot_target, "YTTM_HOT_TARGET", "8192", "tune", "pairs the hot list is built for (tile rounds)")                                          \
  X(hot_target_words, "YTTM_HOT_TARGET_WORDS", "65536", "tune", "... in word mode")                                                          \
  X(hot_min, "YTTM_HOT_MIN", "512", "tune", "live entries below which the hot list is rebuilt")                                              \
  X(top_cap, "YTTM_TOP_CAP", "8192", "test", "capacity of the top list")                                                                     \
  X(top_target, "YTTM_TOP_TARGET", "1024", "tune", "entries the top list is refilled to")                                                    \
  X(top_min, "YTTM_TOP_MIN", "192", "tune", "live entries below which the top list is refilled")                                             \
  X(cand_target, "YTTM_CAND_TARGET", "0", "tune", "candidates asked for per scan (0: four times the recent batch)")                        \
  X(cand_max, "YTTM_CAND_MAX", "512", "tune", "upper bound of that adaptive target")                                                         \
  X(no_extend, "YTTM_NO_EXTEND", "", "path", "set: a batch that runs out of candidates is not extended by a second scan")                   \
  X(no_batch_split, "YTTM_NO_BATCH_SPLIT", "", "path", "set: word-mode batches of 129..256 rules are not cut in two")                       \
  X(no_refine, "YTTM_NO_REFINE", "", "path", "set: the fused scan keeps the host's threshold")                                              \
  X(no_fuse, "YTTM_NO_FUSE", "0", "path", "1: the candidate scan is always a kernel of its own (differential test of the fused tail)")     \
  X(classb_beside, "YTTM_CLASSB_BESIDE", "", "path", "1: word mode, the class-B tiles' launch of a round beside k_words on a second stream (round 5; default since round 6: before it on one stream)") \
  X(no_batch_args, "YTTM_NO_BATCH_ARGS", "", "path", "set: every batch travels through k_round_begin, none in the kernel arguments")       \
  /* ---- K4 */                                                                                                                              \
  X(apply_grid, "YTTM_APPLY_GRID", "256", "tune", "class-A grid cap for small tile sets (0: none)")                                          \
  X(k4_direct, "YTTM_K4_DIRECT", "1", "path", "0: no direct pair->rule table in the first rounds")                                          \
  X(word_mode, "YTTM_WORD_MODE", "1", "path", "0: tiles to the end (differential test of word mode)")                                      \
  X(no_index, "YTTM_NO_INDEX", "0", "path", "1: no pair index (and so no word mode)")                                                       \
  X(word_div, "YTTM_WORD_DIV", "150", "tune", "switch to word mode when sites * this < tokens streamed")                                    \
  X(word_min_tiles, "YTTM_WORD_MIN_TILES", "16384", "tune", "... and class A has at least this many tiles (tests: 0)")                      \
  X(word_min_tokens, "YTTM_WORD_MIN_TOKENS", "50331648", "tune", "... and a pass streams at least this many tokens")                        \
  X(word_hint_floor, "YTTM_WORD_HINT_FLOOR", "16384", "tune", "words a round is sized for beyond twice the last round's sites")            \
  X(words_inline_max, "YTTM_WORDS_INLINE_MAX", "262144", "tune", "rounds of at most this many words apply their records themselves")       \
  X(words_fuse_max, "YTTM_WORDS_FUSE_MAX", "1073741824", "path", "rounds of at most this many words are ONE launch (0: never)")            \
  X(index_agg_min, "YTTM_INDEX_AGG_MIN", "16777216", "tune", "index fill pass: postings from which a workgroup sums per key in LDS first") \
  X(word_drec, "YTTM_WORD_DREC", "32768", "test", "records per workgroup region of k_words (tests: overflow)")                              \
  X(word_log, "YTTM_WORD_LOG", "0", "test", "capacity of the instance-record log (tests: overflow; 0: sized from the tokens)")             \
  X(wgather_grid, "YTTM_WGATHER_GRID", "", "test", "grid of k_wgather")                                                                      \
  X(words_grid, "YTTM_WORDS_GRID", "", "test", "grid of k_words")                                                                            \
  X(words_wpi, "YTTM_WORDS_WPI", "", "test", "words per wave and iteration of k_words")                                                      \
  /* ---- multi-GPU */                                                                                                                       \
  X(replicate_max_tokens, "YTTM_REPLICATE_MAX_TOKENS", "67108864", "path", "word tables of at most this many tokens (summed over ranks) run the replicated merge loop; 0: always sharded") \
  X(xchg_margin, "YTTM_XCHG_MARGIN", "3.0", "test", "safety factor of a round's block size (tests: < 1 forces the repeat path)")           \
  X(xchg_table_cap, "YTTM_XCHG_TABLE_CAP", "0", "test", "delta-table slots (tests: overflow verdict)")                                      \
  X(xchg_notes, "YTTM_XCHG_NOTES", "65536", "test", "capacity of the threshold-crossing notes (tests: overflow -> full walk)")              \
  X(xchg_blk_min, "YTTM_XCHG_BLK_MIN", "4096", "test", "smallest block of the per-round all-gather, 16-byte units")                         \
  /* ---- encode */                                                                                                                          \
  X(encode_cache, "YTTM_ENCODE_CACHE", "", "path", "word cache: 0 = off, 1 = always (default: batches of >= 8 MB)")                         \
  X(encode_cache_min_mb, "YTTM_ENCODE_CACHE_MIN_MB", "", "tune", "that size, MB")                                                           \
  X(dropout_seed, "YTTM_DROPOUT_SEED", "", "test", "fixes the per-encoder salt of the BPE-dropout RNG (default: std::random_device)")      \
  X(dropout_heap_from, "YTTM_DROPOUT_HEAP_FROM", "256", "path", "words of at least this many tokens keep their dropout events in a heap")  \
  X(dropout_hbm_queues, "YTTM_DROPOUT_HBM_QUEUES", "", "path", "set: dropout event queues in the HBM scratch, not LDS")                    \
  X(dropout_no_pack, "YTTM_DROPOUT_NO_PACK", "", "path", "set: dropout merges test an event against rule_xy and read rule_z (round 4's three trips per merge)") \
  X(k5_lane_words, "YTTM_K5_LANE_WORDS", "48", "path", "one-word-per-lane for the cache's distinct words up to this many tokens (0: wave-wide rounds)") \
  X(k5_lane_sent, "YTTM_K5_LANE_SENT", "48", "path", "... for packed sentences")                                                            \
  X(k5_classes, "YTTM_K5_CLASSES", "4", "path", "length classes of the distinct-word list (<= 1: one)")                                     \
  X(dropout_pack_sent, "YTTM_DROPOUT_PACK_SENT", "0", "test", "dropout: at most this many sentences share a pack (0: as many as fit)") \
  X(k5_group, "YTTM_K5_GROUP", "0", "test", "K5: consecutive sentences a wavefront owns at a time (packs are cut from them; 0: by the batch's size)") \
  X(wc_sblk, "YTTM_WC_SBLK", "", "test", "sentences a wavefront of the word cache's walks takes at a time")                                 \
  X(wc_short_slots, "YTTM_WC_SHORT_SLOTS", "", "test", "slots of the short-word region of the word cache (rounded up to a power of two)")  \
  X(enc_staged_from, "YTTM_ENC_STAGED_FROM", "", "test", "host <-> device copies of at least this many bytes go through the pinned chunks")\
  X(enc_sub_mb, "YTTM_ENC_SUB_MB", "320", "tune", "host -> host encode: sub-batch size, MB")                                                \
  X(enc_sub_kb, "YTTM_ENC_SUB_KB", "", "test", "the same in KB")                                                                             \
  X(enc_pipe_from, "YTTM_ENC_PIPE_FROM", "536870912", "tune", "host -> host batches of at least this many bytes are pipelined in sub-batches") \
  X(cli_batch_bytes, "YTTM_CLI_BATCH_BYTES", "0", "test", "`yttm encode` batch size in bytes (0: the reference's 10 MiB)")

struct Config {
#define X(field, name, dflt, kind, doc) Hook field;
  YTTM_HOOKS(X)
#undef X
};

// The snapshot the calling thread is BOUND to (CfgBind: an encoder's or a context's own, for the length of a call into it), else the one
// taken by the last cfg_refresh() (the first call takes one itself).
std::shared_ptr<const Config> cfg();
// re-reads the environment into a new process-wide snapshot: GpuCtx's constructor, encoder creation.  Objects keep the shared_ptr they took
// then; a later refresh -- another encoder, a training on another thread -- never changes the hooks of an object that already exists.
void cfg_refresh();
// Binds a snapshot to the calling thread for a scope (nests; a null pointer binds nothing).  Every entry point of BaseEncoder and of the
// trainer binds its object's snapshot, so that the launchers deep below (k_encode.hip, k_wcache.hip, ...) read THAT one through cfg().
// Threads started inside such a call bind the same snapshot themselves.
struct CfgBind {
  explicit CfgBind(std::shared_ptr<const Config> c);
  ~CfgBind();
  CfgBind(const CfgBind &) = delete;
  CfgBind &operator=(const CfgBind &) = delete;

 private:
  std::shared_ptr<const Config> prev_;
  bool bound_;
};
// the table above as Markdown rows "| `NAME` | default | kind | what |" 
const char *config_table_markdown();

}  // namespace yttm
