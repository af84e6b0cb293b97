// host_core.h -- host-side (C++17) half of the drop-in: model state + file format, alphabet, the trainer's ordered
// pick, and the encoder object.  Mirrors the C++ surface declared in the reference's bpe.h / utils.h (same names,
// argument meaning and error strings) on top of the HIP kernels; nothing here computes the hot path on the CPU.
#pragma once
#include <stdint.h>

#include <memory>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace yttm {

constexpr uint32_t SPACE_TOKEN = 9601;  // utils.h:9

struct Status {  // utils.h:56-64
  int code = 0;
  std::string message;
  Status() = default;
  Status(int c, std::string m) : code(c), message(std::move(m)) {}
  bool ok() const { return code == 0; }
};

struct SpecialTokens {  // utils.h:22-43 (note the constructor order pad, unk, bos, eos)
  int pad_id = -1, unk_id = -1, bos_id = -1, eos_id = -1;
  bool taken_id(int id) const { return id == unk_id || id == pad_id || id == bos_id || id == eos_id; }
  uint64_t n_special_tokens() const { return (unk_id != -1) + (pad_id != -1) + (bos_id != -1) + (eos_id != -1); }
};

struct BpeConfig {  // utils.h:45-54
  double character_coverage = 1;
  int n_threads = 0;
  SpecialTokens special_tokens;
};

struct BPE_Rule {  // utils.h:11-20
  uint32_t x = 0, y = 0, z = 0;
};

struct BPEState {  // utils.h:66-74; chars are kept in MODEL-FILE order (the reference's hash-slot order)
  std::vector<std::pair<uint32_t, uint32_t>> char2id;  // (code point, id)
  std::vector<BPE_Rule> rules;
  SpecialTokens special_tokens;
  Status dump(const std::string &file_name) const;  // utils.cpp:50-66
  Status load(const std::string &file_name);        // utils.cpp:68-91
};

// Slot order of a ska::flat_hash_map<uint32_t,...> after inserting `keys` in order into an empty map and
// copy-constructing it once (bpe.cpp:1289 copies char2id into BPEState; utils.cpp:57-59 iterates the copy).
std::vector<uint32_t> flat_hash_map_order(const std::vector<uint32_t> &keys);

// bpe.cpp:316-355 compute_alphabet_helper.  Returns chars in INSERTION order (▁ first, then by descending
// (count, code point)) with compact ids n_special, n_special+1, ...
void compute_alphabet(const std::vector<uint32_t> &cps, const std::vector<unsigned long long> &cnts, unsigned long long data_len,
                      const BpeConfig &cfg, std::vector<std::pair<uint32_t, uint32_t>> &char2id_insertion, uint64_t &n_removed);

Status check_config(BpeConfig &cfg, int vocab_size);  // bpe.cpp:1295-1350

struct TrainReport {
  double seconds_total = 0, seconds_frontend = 0, seconds_merge = 0, seconds_io = 0, seconds_upload = 0;  // upload: file/host -> HBM (train_bpe only)
  unsigned long long corpus_bytes = 0, n_unique = 0, n_tokens = 0, rounds = 0, rules = 0, cand_rescans = 0, repacks = 0, merge_sites = 0, hot_rebuilds = 0, fused_rounds = 0, fused_overflows = 0, exchange_retries = 0, word_table_retries = 0, front_end_overlapped = 0, top_refills = 0, index_builds = 0, word_rounds = 0, word_switch_round = 0, word_all_rounds = 0, word_fused_rounds = 0,
                     rounds_exhausted = 0 /* batches closed for lack of candidates, not at an intersection */, batch_extensions = 0 /* extra scans that refilled the pick */, batch_splits = 0 /* batches cut to the first BATCH_ARGS_MAX rules: word mode, two one-launch rounds instead of a four-launch one */,
                     classb_overlapped = 0 /* word-mode rounds whose class-B tiles ran beside k_words on a second stream */, k3_radix = 0 /* 1: the pair count of a large alphabet ran by radix partition */, front_end_chunks = 0 /* > 0: the corpus crossed the device in this many chunks (it did not fit the HBM at once) */, peak_device_bytes = 0 /* high-water mark of the device memory pool during the call */,
                     replicated_merge_loop = 0 /* multi-GPU: the shards were gathered, every rank ran the merge loop alone (no per-round collective) */;
  // K4 totals: tiles with a merge site and their tokens; words with a merge site and their tokens (measurement pass only, else 0)
  unsigned long long touched_tiles = 0, touched_tile_tokens = 0, touched_words = 0, touched_word_tokens = 0;
  // ... the same after `split_round` rounds (YTTM_MEASURE_SPLIT_ROUND; 0: none), and the device-clock time of the word-mode rounds
  unsigned long long split_round = 0, split_touched_words = 0, split_touched_word_tokens = 0, split_sites = 0, merge_launches_words = 0;
  double merge_ms_words = 0;
  // per kernel family: ms, launches, algorithmic bytes (gpu_ctx.h KT_*)
  double kt_ms[8] = {0};
  unsigned long long kt_launches[8] = {0}, kt_bytes[8] = {0};
};

class GpuCtx;
struct Comm;

// bpe.h:19 train_bpe -- file based.  `device` = HIP device ordinal.
Status train_bpe(const std::string &input_path, const std::string &model_path, int vocab_size, BpeConfig cfg, int device = 0,
                 TrainReport *report = nullptr, Comm *comm = nullptr, int profile = 0 /* 1: per-kernel timers in the report */);
// same, corpus already in host memory / already resident in HBM (bench: timed region starts with the bytes in HBM)
Status train_bpe_from_memory(const uint8_t *text, unsigned long long n, const std::string &model_path, int vocab_size, BpeConfig cfg,
                             int device = 0, TrainReport *report = nullptr, Comm *comm = nullptr);
Status train_bpe_from_device(const void *d_text, unsigned long long n, const std::string &model_path, int vocab_size, BpeConfig cfg,
                             int device = 0, TrainReport *report = nullptr, Comm *comm = nullptr, int profile = 0 /* 1: HIP-event kernel times; 2: K4 measurement pass (touched words), never timed */);
// learn_bpe_from_string (bpe.cpp:859-1293) on an attached corpus
Status learn_bpe(GpuCtx &g, int vocab_size, const std::string &model_path, const BpeConfig &cfg, BPEState *state, TrainReport *report);

struct EncoderDevice;  // HBM-resident model tables for K5
struct Config;         // yttm_config.h

class BaseEncoder {  // bpe.h:22-82
 public:
  BPEState bpe_state;
  std::unordered_map<uint32_t, uint32_t> char2id, id2char;
  std::unordered_map<uint32_t, std::vector<uint32_t>> recipe;
  std::unordered_map<std::string, uint32_t> reversed_recipe;
  int n_threads = 1;

  BaseEncoder(const std::string &model_path, int n_threads, Status *ret_status, int device = 0);
  ~BaseEncoder();

  // packed batch API (bytes + offsets[S+1]); ids/out_off are filled on the host
  Status encode_as_ids(const uint8_t *bytes, const unsigned long long *offsets, unsigned long long n_sent, bool bos, bool eos, bool reverse,
                       double dropout_prob, std::vector<int32_t> *ids, std::vector<unsigned long long> *out_off) const;
  Status encode_as_ids_malloc(const uint8_t *bytes, const unsigned long long *offsets, unsigned long long n_sent, bool bos, bool eos, bool reverse,
                              double dropout_prob, int32_t **ids, unsigned long long **out_off) const;
  Status encode_as_subwords(const uint8_t *bytes, const unsigned long long *offsets, unsigned long long n_sent, bool bos, bool eos,
                            bool reverse, double dropout_prob, std::vector<std::string> *pieces,
                            std::vector<unsigned long long> *piece_off) const;
  // device-resident variant used by bench.py: input already in HBM, output left in HBM (ids_dev/off_dev owned by the encoder)
  Status encode_device(const void *d_bytes, const void *d_offsets, unsigned long long n_sent, unsigned long long total_bytes,
                       unsigned long long max_sentence_bytes, bool bos, bool eos, bool reverse, double dropout_prob,
                       unsigned long long *n_ids_out, double *kernel_ms) const;
  Status fetch_device_result(int32_t *ids, unsigned long long *out_off, unsigned long long n_sent) const;
  // the YTTM_* hooks as they stood when THIS encoder was made: every entry point binds them to its thread (yttm_config.h CfgBind), so that a
  // later encoder or training never changes the paths of this one
  std::shared_ptr<const Config> config() const;
  void set_cache(int mode, unsigned long long min_bytes) const;  // word cache of the batch encoder: 0 off, 1 always, 2 from min_bytes up
  unsigned long long cache_words() const;                          // distinct words of the last encode_device batch (0: not cached)

  Status id_to_subword(int id, std::string *subword, bool replace_space = false) const;  // bpe.cpp:1774-1807
  int subword_to_id(const std::string &token) const;                                      // bpe.cpp:1809-1826
  Status decode(const std::vector<int> &ids, std::string *sentence, const std::unordered_set<int> *ignore_ids) const;  // bpe.cpp:1843
  int vocab_size() const;                                                                 // bpe.cpp:1692
  std::vector<std::string> vocabulary() const;                                            // bpe.cpp:1884
  // the streaming loops of the `yttm` command line (host_cli.cpp); the reference's read std::cin and write std::cout
  Status encode_cli(const std::string &output_type, bool stream, bool bos, bool eos, bool reverse, double dropout_prob, int in_fd = 0,
                    int out_fd = 1) const;                                                // bpe.h:66-68, bpe.cpp:1942-2014
  Status decode_cli(const std::unordered_set<int> *ignore_ids, int in_fd = 0, int out_fd = 1) const;  // bpe.h:70, bpe.cpp:2016-2028
  Status vocab_cli(bool verbose, int out_fd = 1) const;                                   // bpe.h:71, bpe.cpp:1896-1940

 private:
  void fill_from_state();  // bpe.cpp:1667-1690
  EncoderDevice *dev_ = nullptr;
  int device_ = 0;
};

std::string encode_utf8(const std::vector<uint32_t> &text);  // utf8.cpp:102-109
std::vector<uint32_t> decode_utf8(const char *begin, const char *end, bool *invalid = nullptr);  // utf8.cpp:111-128
bool is_space(uint32_t ch);  // utils.cpp:99-101

}  // namespace yttm
