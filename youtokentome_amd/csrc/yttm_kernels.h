// yttm_kernels.h -- host-callable launchers of the gfx950 kernels (k_frontend.hip; k_tiles.hip, k_words.hip, k_index.hip, k_pairtable.hip, k_giant.hip; k_encode.hip, k_wcache.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "yttm_config.h"
#include "yttm_device.h"

namespace yttm {

constexpr int CAND_BINS = 704;  // counts 0..255 exact, then 8 bins per power of two

struct CandRec {
  unsigned long long key;  // x<<32|y
  unsigned long long cnt;
};

// lower bound of a candidate-histogram bin (inverse of cand_bin in k_merge_shared.h)
inline unsigned long long cand_bin_lower(int bin) {
  if (bin < 256) return (unsigned long long)bin;
  int e = 8 + (bin - 256) / 8, m3 = (bin - 256) % 8;
  return (unsigned long long)(8 + m3) << (e - 3);
}

// ---- front end (k_frontend.hip)
// wide_chars: many chars beyond U+07FF (three- and four-byte UTF-8): they are counted in an LDS hash instead of global atomics
// chunk_segs [fe_chunks(n)]: segment starts per 4 KB chunk of the text (out); the segment pass takes their exclusive scan
// [chunk_lo, chunk_hi): the 4 KB chunks of this launch -- all of them, or those of a part of the text that has arrived while the rest is still on
// its way over the link (gpu_ctx.cpp upload_corpus_fd); counts and histogram add up over the launches
void launch_char_hist(const uint8_t *text, unsigned long long n, unsigned long long *hist, unsigned long long *counters, bool wide_chars,
                      uint32_t *chunk_segs, hipStream_t st, unsigned long long chunk_lo = 0, unsigned long long chunk_hi = ~0ull);
unsigned long long fe_chunks(unsigned long long n);
unsigned long long fe_chunk_bytes();
void launch_seg_write(const uint8_t *text, unsigned long long n, unsigned long long *seg_pos, const unsigned long long *chunk_off, hipStream_t st,
                      unsigned long long chunk_lo = 0, unsigned long long chunk_hi = ~0ull);
void launch_hist_compact(const unsigned long long *hist, uint32_t *cps, unsigned long long *cnts, unsigned int *n_out, unsigned int cap,
                         hipStream_t st);
// word table (K2b/K2c): ht[0 .. n_slots) keys, ht[n_slots .. 2 n_slots) counts, ht[2 n_slots .. 3 n_slots) positions of the short words, see k_frontend.hip
void launch_word_table_clear(unsigned long long *ht, unsigned long long n_slots, hipStream_t st);
void launch_insert_words(const uint8_t *text, unsigned long long n, const uint32_t *cpmap, const unsigned long long *seg_pos,
                         unsigned long long n_segs, unsigned long long *ht, unsigned long long ht_mask, unsigned int *status, hipStream_t st,
                         unsigned int max_blocks = 256 * 32 /* (a part of the segments: fewer workgroups, each ends with a flush of its LDS table) */);
// chunked front end (k_frontend.hip: k2b_relocate, k2b_rehash): words whose representative lies below `chunk_end` move to the lexicon at *cursor
// (device; advanced by the bytes taken -- move == false only adds them up); the table rebuilt with new_slots (a power of two, cleared) slots
void launch_words_relocate(uint8_t *text, unsigned long long chunk_end, unsigned long long chunk_len, unsigned long long *ht, unsigned long long n_slots,
                           unsigned long long *cursor, bool move, hipStream_t st);
void launch_word_table_rehash(const uint8_t *text, unsigned long long n, const uint32_t *cpmap, const unsigned long long *old_ht, unsigned long long old_slots,
                              unsigned long long *new_ht, unsigned long long new_slots, hipStream_t st);
void launch_compact_words(const uint8_t *text, unsigned long long n, const uint32_t *cpmap, const unsigned long long *ht, unsigned long long n_slots,
                          unsigned long long *posA, uint32_t *cntA, uint32_t *lenA, unsigned long long *posB, uint32_t *cntB, uint32_t *lenB,
                          unsigned long long *posC, uint32_t *cntC, uint32_t *lenC, unsigned int *cursor, unsigned int *status,
                          unsigned long long wmax /* largest weight a word may carry (2^32 - 1; tests: less) */,
                          unsigned long long *heavy /* [HEAVY_CAP][3] words seen more often than that: pos, tokens, count still to place; their number in cursor[3] */,
                          hipStream_t st);
constexpr int HEAVY_CAP = 256;  // (a word of two bytes seen 2^32 times is 8.6 GB of text: HBM holds a few dozen such words at most -- and tests lower wmax)
void launch_exclusive_scan(const uint32_t *in, unsigned long long n, unsigned long long *out, unsigned long long *block_sums,
                           unsigned long long *total_out, hipStream_t st);
unsigned long long scan_scratch_blocks(unsigned long long n);
void launch_fill_tokens(const uint8_t *text, unsigned long long n, const uint32_t *cpmap, uint32_t space_id,
                        const unsigned long long *uw_pos, const unsigned long long *uw_off, unsigned int n_words, unsigned int nom,
                        unsigned int slot, const unsigned long long *tile_start, uint32_t *tok, hipStream_t st,
                        unsigned long long total_tokens /* of these words: picks the size of the wavefronts' LDS windows */);
void launch_tiles(const unsigned long long *uw_off, unsigned int n_words, unsigned int nom, unsigned long long *tile_start,
                  uint32_t *tile_word0, hipStream_t st);
void launch_tile_len(const unsigned long long *tile_start, unsigned int n_tiles, unsigned long long total_tokens, uint32_t *tile_len,
                     hipStream_t st);

// ---- merge loop (k_tiles.hip, k_pairtable.hip)
// cls: 0 = class A tiles (slot 1024), 1 = class B tiles (slot 4096)
// A small batch travels as a kernel argument: the apply kernel and the candidate scan build their LDS tables (token flags, rule
// hash) from it, and the round needs no prologue kernel, no rule upload and no flag table in HBM.  Used when the whole round
// runs without the filter pass (small or dense tile sets), all ids fit the LDS flag bitmap and there is no class-C tile.
constexpr int BATCH_ARGS_MAX = 128;  // (1 KB of kernel arguments; 32 until round 3: the late rounds of random text have batches of 30 .. 100 rules)
constexpr uint32_t FLAG_LDS_IDS = 32768;  // (8 KB of LDS: the batch's pair filter, or the direct pair -> rule table)
constexpr uint32_t DIRECT_MAX_V = 90;     // direct table: ids below this (90 * 90 bytes <= 8 KB)
struct BatchArgs {
  uint32_t k;                        // 0: not used (tables come from HBM)
  uint32_t xy[2 * BATCH_ARGS_MAX];   // rule j of the batch merges (xy[2j], xy[2j+1]) into z_base + j; x == y: the self rule (skipped)
  uint32_t direct_v;                 // != 0 (class-A k_tiles<.., DIRECT>): every token id in the tiles is < direct_v, and direct_v^2 bytes fit the LDS table
                                     // that maps a PAIR straight to its rule -- the first rounds of a small alphabet, the most expensive ones: one byte
                                     // read per adjacency instead of a pair filter and a hash probe
  uint32_t mark;                     // this is the round's first launch: workgroup 0 notes the time in stats[STAT_T0] (the round's duration then
                                     // comes from the device's own clock, ScanArgs::timed -- two hipEventRecord calls per round cost 4 us of host time)
  uint32_t instr;                    // measurement pass (never timed): also count the WORDS that hold a merge site and their tokens
                                     // (T_touched, W_touched) into stats[4], stats[5]; single-site tiles take the general path
};
// The candidate scan that follows a merge round, done by the round's own apply kernel: the last workgroup to finish folds the
// statistics rows, zeroes the batch's pairs, scans (and compacts) the hot list and publishes header + histogram + candidates
// in the host's pinned mailbox -- one launch per round instead of two, and no second trip through the launch path.
struct ScanArgs {
  uint32_t on;  // 0: no tail in this launch; 1: the candidate scan; 4: see peer_flag; 2 (multi-GPU, word mode): the last workgroup only leaves the worklist
                // counters at zero -- the scan rides in the fold kernel behind the all-gather (k_fold_list); 3 (multi-GPU, set by
                // launch_words_apply): a one-launch word round, which leaves no worklist behind -- no ticket, no tail
  uint32_t tau_mx;
  unsigned long long tau_cnt;  // candidates: count > tau_cnt, or == tau_cnt and max(x,y) <= tau_mx
  CandRec *out;                // [cap] all candidates (the first `fast` also go to the mailbox)
  unsigned int cap, fast;
  unsigned int *done_ctr;      // ticket of finished workgroups (left at 0); nullptr: a single-workgroup launch
  unsigned char *mailbox;      // the host's pinned mailbox, or (round_id == 0) a staging block in HBM with the same layout
  uint32_t round_id;           // published in the mailbox when everything else is there; 0: nothing is published
  uint32_t want;               // != 0: about this many candidates are wanted -- the scan may raise the threshold by itself (scan_top: refine)
  uint32_t timed;              // the round's first launch left its start time in stats[STAT_T0]: the mailbox gets the duration (100 MHz ticks)
  // Two launches of one round side by side (single GPU, word mode: the class-B tiles on a second stream beside k_words; gpu_ctx.cpp merge_apply).
  // on == 4: this launch is the one WITHOUT the tail -- its last workgroup (ticket: done_ctr, its own) stores round_id to *peer_flag;
  // on == 1 with peer_flag != nullptr: the tail starts once *peer_flag == round_id (scan_top).
  unsigned int *peer_flag;
};
constexpr int STAT_T0 = 6;  // stats[6]: wall_clock64() at the start of the round's first launch
constexpr int STAT_T1 = 7;  // stats[7] (multi-GPU): ... when the round's apply kernels and the all-gather behind them were done (noted by the fold's first kernel)
void launch_top_scan(const PairTable &pt, const ScanArgs &sa, unsigned long long *stats, const RuleSlot *zrules, unsigned int zmask, unsigned long long zself,
                     const BatchArgs *zba, unsigned long long *xstat /* multi-GPU: the exchange's report, forwarded to the mailbox */, hipStream_t st);
void launch_top_rebuild(const PairTable &pt, unsigned int listed_hint, hipStream_t st);
// id_min / n_ids: the token ids in the tiles are id_min .. id_min + n_ids - 1 (K3 runs before any merge: the alphabet); n_ids <= 32
// counts pairs in a dense LDS table, 0 (unknown / larger) in the LDS hash
void launch_pair_count(int cls, const TileSet &ts, const PairTable &pt, const DeltaBuf &db, uint32_t id_min, uint32_t n_ids, hipStream_t st);
// K3 of class A for alphabets of 65 .. K3R_MAX_IDS symbols by two-level radix partition of (pair, weight) records (k_pairradix.hip): no atomic per
// adjacency.  scratch: pair_count_radix_scratch_u32(n_ids) words; buf1, buf2: one 8-byte record per class-A token each.
constexpr uint32_t K3R_MAX_IDS = 8192;
size_t pair_count_radix_scratch_u32(uint32_t n_ids);
bool pair_count_radix_takes(uint32_t n_ids, unsigned long long n_tokens);
void launch_pair_count_radix(const TileSet &ts, const PairTable &pt, const DeltaBuf &db, uint32_t id_min, uint32_t n_ids, uint32_t *scratch,
                             unsigned long long *buf1, unsigned long long *buf2, unsigned long long n_tokens, hipStream_t st);
void launch_merge_apply(int cls, const TileSet &ts, const PairTable &pt, const DeltaBuf &db, const RuleSlot *rules, unsigned int rule_mask,
                        uint32_t self_x, uint32_t self_z, uint32_t z_base, unsigned long long *stats, const BatchArgs *ba,
                        const ScanArgs *scan /* the round's last launch only */,
                        const uint32_t *bloom_g /* the batch not in ba: the batch's pair filter (pm_bloom_host) */, hipStream_t st);
constexpr int PM_BLOOM_WORDS_H = 2048;
void pm_bloom_host(uint32_t *bloom, const uint32_t *xyz, uint32_t k);
// pair index for K4's worklists (k_index_core.h: PairIndex; k_index.hip)
// A key's posting count / fill cursor is kept in IDX_SHARDS copies (a wave adds to copy (its number) % IDX_SHARDS): a pair with a million
// adjacencies is a million atomics on ONE address otherwise (~12 ns each: 6.5 ms per pass at the word-mode switch of the 1 GB corpus).
// The copies' runs are adjacent, so a key's postings are still one run: [off[s * IDX_SHARDS], off[(s + 1) * IDX_SHARDS]).
constexpr unsigned int IDX_SHARDS = 16;
struct PairIndexArgs {
  unsigned long long *key, *off;
  uint32_t *cnt, *bloom, *post;
  unsigned int mask;
};
void launch_idx_seed(const PairTable &pt, const PairIndexArgs &a, unsigned int listed_hint, hipStream_t st);
// (class-A token slots in word mode: postings are word ids, the slots may hold TOK_HOLEs)
void launch_idx_stream(bool fill, const TileSet &ts, const PairIndexArgs &a, hipStream_t st,
                       bool agg = true /* fill pass: sum a workgroup's postings per key in LDS first (worth a second sweep only when they are many) */,
                       void *save = nullptr /* [idx_save_bytes()] the count pass's per-workgroup tables, reused by an agg fill pass */);
size_t idx_save_bytes();
// ---- word mode (k_words.hip)
void launch_words_init(const TileSet &ts, unsigned long long *wmeta, hipStream_t st);
struct WGatherArgs {
  PairIndexArgs ix;
  uint32_t ix_valid, z_static;   // tokens below z_static existed when the index was built
  TokLists tl;
  uint32_t *stamp;               // [n_words] round that claimed the word last
  uint32_t round_id;
  uint32_t *worklist;            // WL_PARTS sub-lists, wl_seg entries apart; work_n[0..WL_PARTS) their lengths, work_n[WL_PARTS + 1] != 0: take every word
  unsigned long long wl_seg;
  unsigned int *work_n;
  unsigned int *gm;              // [WGATHER_MAXK] records matched per rule (left at zero)
  unsigned int *done_ctr;
  unsigned long long *stats;     // (BatchArgs::mark)
  const uint32_t *xyz;           // rule j = (xyz[3j], xyz[3j+1]) -> z_base + j, in HBM; nullptr: the batch is in the BatchArgs
  uint32_t k, z_base;
  uint32_t cnt[BATCH_ARGS_MAX];  // k_words<FUSED>: the count the host picked rule j by (saturated): no rule has more sites than that
};
constexpr unsigned int WGATHER_MAXK = 4096;
extern int g_apply_grid;    // YTTM_APPLY_GRID as launch_env_refresh() found it (k_tiles.hip: launch_merge_apply)
void launch_env_refresh();  // re-reads the launchers' environment hooks (YTTM_WORDS_GRID, YTTM_WORDS_WPI, YTTM_WGATHER_GRID): once per context
void launch_wgather(const WGatherArgs &a, const BatchArgs *ba, unsigned int work_hint, hipStream_t st);
bool launch_words_apply(const WordSet &ws, const PairTable &pt, const DeltaBuf &db, const RuleSlot *rules, unsigned int rule_mask, const uint32_t *bloom_g,
                        uint32_t self_x, uint32_t self_z, uint32_t z_base, uint32_t k_rules, const uint32_t *worklist /* nullptr: every word */, unsigned long long wl_seg,
                        const unsigned int *work_n, unsigned long long *stats, const TokLists &tl, DeltaRec *drec /* [WORDS_MAX_GRID * drec_cap] */,
                        unsigned int drec_cap, unsigned int *drec_n /* [WORDS_MAX_GRID] */, uint4 *irec /* [WORDS_MAX_GRID * drec_cap] */, const BatchArgs *ba, const ScanArgs *scan, unsigned int work_hint,
                        unsigned int inline_max /* rounds of at most this many words (by the hint) apply their records themselves */,
                        const WGatherArgs *ga /* the round's gather: launched here (k_wgather) ... */,
                        unsigned int fuse_max /* ... unless the hint is at most this and the batch is in the arguments: k_words gathers itself */, hipStream_t st,
                        unsigned int avg_word_tokens = 0 /* live tokens per word, about (0: unknown): a wave takes no more words at a time than fit its 512-token tile */);
constexpr unsigned int WORDS_MAX_GRID = 512;  // workgroups of k_words: each owns a region of the round's count-update records
void launch_repack(int cls, const TileSet &ts, const unsigned long long *off, unsigned int nom, unsigned long long total,
                   unsigned long long *gstart, unsigned int n_new, uint32_t *new_tok, uint32_t *new_len, uint32_t *new_word0, hipStream_t st);
void launch_cand_scan(const PairTable &pt, unsigned long long tau_cnt, uint32_t tau_mx, CandRec *out, unsigned int cap,
                      unsigned int *n_out, unsigned long long *hist, hipStream_t st);
void launch_hot_scan(const PairTable &pt, unsigned long long tau_cnt, uint32_t tau_mx, CandRec *out, unsigned int cap, unsigned int *n_out,
                     unsigned long long *hist, unsigned int *done_ctr, unsigned char *mailbox, unsigned int fast, uint32_t round_id,
                     unsigned long long *stats, const RuleSlot *zrules, unsigned int zmask, unsigned long long zself, unsigned int listed_hint,
                     const BatchArgs *zba, unsigned long long *xstat /* multi-GPU: the exchange's report, forwarded to the mailbox */, hipStream_t st);
// ---- multi-GPU, per round: K4 -> ncclAllGather -> k_pt_apply_blocks -> k_fold_list (+ the round's candidate scan) [-> k_dt_clean, during the host's turn]
void launch_dt_clean(const DeltaBuf &db /* .send = the block just exchanged */, DeltaRec *other /* the block of the round to come */, unsigned int n_hint,
                     unsigned long long *stats, uint32_t tiles_a, unsigned int *done_ctr, hipStream_t st);
void launch_dt_init(DtSlot *slots, unsigned long long n, hipStream_t st);
// phase 1: the OTHER ranks' count deltas into the local replica (no list appends: pass a PairTable with the thresholds off)
void launch_pt_apply_blocks(const PairTable &pt, const DeltaRec *blocks, unsigned long long blk, int world, int rank, unsigned long long only_mask,
                            unsigned long long *xstat, unsigned long long *stats, hipStream_t st);  // (not launched for a communicator of one rank: no other rank's block exists; k_fold_list reads the header then)
// phase 2, behind phase 1's kernel boundary, ONE workgroup: every slot noted by an add of this round (PairTable::maybe: pt comes with its
// real thresholds AND the notes) whose pair ended the round at or above a list threshold joins that list -- judged by the FINAL count,
// the same on every rank, so the lists hold the same pairs everywhere and no verdict on them has to be exchanged.  Then the round's
// candidate scan (scan != nullptr) straight into the mailbox.
void launch_fold_list(const PairTable &pt, const DeltaRec *blocks, unsigned long long blk, int world, unsigned long long only_mask, const ScanArgs *scan,
                      unsigned long long *stats, const RuleSlot *zrules, unsigned int zmask, unsigned long long zself, const BatchArgs *zba,
                      unsigned long long *xstat, bool read_headers /* phase 1 was not launched */, hipStream_t st);
constexpr int MB_HIST = 128;  // byte offset of the count histogram in the mailbox (header + xstat before it)
constexpr int MB_XSUM = 6144; // multi-GPU, behind the histogram: sums over the ranks' block headers -- [0] merge sites so far, [8] tokens streamed so far,
                              // [16] class-A tiles, [24] ranks; [32] the round's apply kernels on the device clock (ticks; ScanArgs::timed)
constexpr int XSTAT_WORDS = 8;  // d_xstat: [0] ranks whose block did not fit (bit r), [1] largest record count, [2] -, [3] a rank lost records, [4..7] the sums above
void launch_fold_stats(unsigned long long *stats, unsigned int *n_keys, hipStream_t st);
constexpr int STATS_WORDS = 32 + 8 * 1536;  // totals + one row per workgroup (k_merge_shared.h: BLK_BASE, BLK_ROWS)
void launch_round_begin(const RuleSlot *src_rules, unsigned int n_slots, RuleSlot *dst_rules, unsigned int *work_n_a, unsigned int *work_n_b,
                        const uint32_t *src_bloom, uint32_t *dst_bloom, hipStream_t st);
void launch_hot_rebuild(const PairTable &pt, hipStream_t st);
// class C (k_giant.hip): K3 (merge=false) / K4 (merge=true) for tiles of words longer than TILE_NOM_B tokens; scratch = 4*slot
// uint32 per tile
void launch_giant(bool merge, const TileSet &ts, unsigned int slot, const PairTable &pt, const DeltaBuf &db, const RuleSlot *rules,
                  unsigned int rule_mask, uint32_t self_x, uint32_t self_z, uint32_t *scratch, unsigned long long *stats, hipStream_t st);
void launch_pt_clear(const PairTable &pt, hipStream_t st);
void launch_pt_rehash(const PairTable &src, const PairTable &dst, hipStream_t st);
void launch_pt_zero(const PairTable &pt, const RuleSlot *rules, unsigned int n_slots, unsigned long long self_key, hipStream_t st);
void launch_pt_query(const PairTable &pt, const unsigned long long *keys, unsigned int n, unsigned long long *out, hipStream_t st);
void launch_pt_apply(const PairTable &pt, const DeltaRec *recs, unsigned long long n, hipStream_t st);
void launch_fill_u64(unsigned long long *p, unsigned long long v, unsigned long long n, hipStream_t st);

// ---- batch encode (k_encode.hip)
constexpr int ENC_BLOOM_WORDS = 8192;    // 32 KB of LDS: 8 bits per rule at 32k rules
constexpr int ENC_WAVES_PER_BLOCK = 8;   // 512 threads; 48 KB of working arrays + the 32 KB filter = 80 KB, two workgroups per CU
                                         // (measured per 1e7 sentences: 8 waves 76 ms, 10 waves x 384 tokens 104 ms, 6 waves 168 ms)
constexpr int ENC_LDS_TOKENS = 512;      // a sentence of B bytes has <= B+1 tokens: up to 511 bytes work in LDS, longer ones in HBM scratch
// One 32-bit hash of a token pair serves the rule hash (low bits = slot) and the Bloom filter (top 13 bits = word,
// 3 x 5 bits of a second multiply = bit positions).  64-bit multiplies cost ~8 VALU instructions each on CDNA; this is 9 in all.
__host__ __device__ inline uint32_t enc_hash(uint32_t a, uint32_t b) {
  uint32_t h = a * 0x9E3779B1u;
  h ^= b + 0x7F4A7C15u + (h << 6) + (h >> 2);
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}
// Blocked Bloom filter of all rule keys (one 32-bit word, 3 bits per key), resident in LDS: most adjacent pairs have no
// rule, and this answers "certainly none" without leaving the CU.  The host fills it with the same functions.
__host__ __device__ inline uint32_t enc_bloom_bits(uint32_t h) {
  const uint32_t g = h * 0x27D4EB2Fu;
  return (1u << (g >> 27)) | (1u << ((g >> 22) & 31)) | (1u << ((g >> 17) & 31));
}
__host__ __device__ inline uint32_t enc_bloom_word(uint32_t h) { return h >> 19; }  // 13 bits = ENC_BLOOM_WORDS
struct EncModel {
  const uint32_t *cpmap;      // [N_CODEPOINTS]: final token id, CP_SPACE, CP_UNK
  const RuleSlot *rules;      // hash (x<<32|y) -> RuleSlot{z, pad = rule index = priority}
  const uint32_t *rule_z;     // [n_rules] z of rule i
  const unsigned long long *rule_xy;  // [n_rules] x<<32|y of rule i
  const uint32_t *bloom;      // [ENC_BLOOM_WORDS] blocked Bloom filter of the rule keys (staged into LDS)
  unsigned int rule_mask;
  uint32_t n_rules;
  uint32_t z_affine, z_base, z_bp[4];  // rule_z[r] == z_base + r + #{k : z_bp[k] <= r} when z_affine 
  uint32_t space_id;
  int unk_id, bos_id, eos_id;
};
// offsets[n_sent + 1], ends == nullptr: sentences back to back.  ends != nullptr: item j is bytes [offsets[j], ends[j]), any order (the
// word cache's distinct words); its ids land at scratch + 2 offsets[j].
// where the ids of the word cache's distinct words are announced (k_wcache.hip): item u's table slot uslot[u] (u < n_table), or the entry
// u - n_table of the words too long to cache, becomes ids offset << 20 | count (offset, count for the long ones)
struct WordPublish {
  unsigned long long *slot;
  const uint32_t *uslot;
  unsigned long long n_table;
  unsigned long long *extra;
  unsigned long long text_bytes;  // size of the text buffer (wide loads stay inside it)
};
void launch_encode(const EncModel &m, const uint8_t *text, const unsigned long long *offsets, const unsigned long long *ends,
                   unsigned long long n_sent, int bos,
                   int eos, int reverse, int32_t *scratch_ids, uint32_t *counts, uint32_t *work, unsigned long long work_stride,
                   unsigned int n_blocks, double dropout_prob, unsigned long long seed, uint32_t *drop_scratch,
                   unsigned long long drop_stride, hipStream_t st, const WordPublish *pub = nullptr);
void launch_encode_gather(const int32_t *scratch_ids, const unsigned long long *offsets, const unsigned long long *ends,
                          const unsigned long long *out_off, unsigned long long n_sent, int32_t *ids_out, hipStream_t st);

// ---- N4: word cache of the batch encoder (k_wcache.hip) ----
struct WordCache {
  unsigned long long *slot;   // [mask + 1] key of a distinct word (PT_EMPTY = free); after launch_wcache_publish: ids offset << 20 | count
  unsigned long long *pos;    // [mask + 1] first byte of the word's first occurrence
  unsigned long long mask;
  unsigned long long short_mask;  // words of up to 7 bytes hash into the first short_mask + 1 slots (<= mask)
  uint32_t *occ;              // [(bytes + sentences) / 2 + 2] per word occurrence (index (first byte + sentence) / 2): its slot; 0xffffffff elsewhere
  unsigned long long *extra;  // [extra_cap][2] words too long to cache: begin, end; after publish: ids offset, count
  unsigned int *extra_n;
  unsigned int extra_cap;
  unsigned int *status;       // bit 0: the table was too full; bit 1: (also) for a word of up to 7 bytes
};
unsigned long long wcache_count_blocks(const WordCache &wc);
void launch_wcache_insert(const EncModel &m, const uint8_t *text, unsigned long long total, const unsigned long long *offsets, unsigned long long n_sent,
                          const WordCache &wc, hipStream_t st);
void launch_wcache_count_slots(const WordCache &wc, uint32_t *blk_cnt, hipStream_t st);
void launch_wcache_list(const WordCache &wc, const unsigned long long *blk_off, unsigned long long n_table, unsigned long long *ustart,
                        unsigned long long *uend, uint32_t *uslot, hipStream_t st);
void launch_wcache_count(const unsigned long long *offsets, unsigned long long n_sent, const WordCache &wc, int n_fixed, uint32_t *counts, hipStream_t st);
void launch_wcache_scatter(const EncModel &m, const unsigned long long *offsets, unsigned long long n_sent, const WordCache &wc, const int32_t *uids, int bos,
                           int eos, int reverse, const unsigned long long *out_off, int32_t *ids_out, hipStream_t st);

}  // namespace yttm
