// vkcom_adapter.h -- the C++ surface of the reference's youtokentome/cpp/bpe.h (bpe.h:19-82), utils.h (:10-103) and the two utf8.h
// helpers its tests call, re-declared as a thin adapter over the C ABI of libyttm_mi355x.so (include/yttm_mi355x.h).
//
// Purpose: what links against bpe.h today -- the Cython binding yttm.pyx:10-49 and tests/unit_tests/stress_test.cpp -- compiles UNCHANGED
// against this header (shim/cpp/youtokentome/cpp/{bpe.h,utils.h,utf8.h,third_party/flat_hash_map.h} forward here) and runs on the MI355X:
//   * `vkcom::train_bpe`, `vkcom::BaseEncoder::*`      -> yttm_train_bpe / yttm_encoder_* / yttm_encode_as_* / yttm_decode / *_cli
//   * `learn_bpe_from_string` (stress_test.h:14-18)     -> yttm_train_bpe_from_memory + the model file read back into a BPEState
//   * BaseEncoder's public maps (bpe_state, id2char, recipe, reversed_recipe, rule2id; read by stress_test.cpp's decode_slow) are filled
//     from the model file exactly as fill_from_state does (bpe.cpp:1667-1690); they are host-side mirrors, the encoding runs on the GPU.
// The adapter holds NO algorithm of the hot path: every train / encode / decode call goes through the C ABI (dlopen'ed at first use:
// $YTTM_AMD_LIB, else the library next to this repository's package).  `flat_hash_map` is std::unordered_map here -- the byte order of
// the model file's char section (the reference's ska::flat_hash_map iteration order) is produced inside the library.
#pragma once
#include <stdint.h>

#include <iostream>
#include <iterator>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace vkcom {

template <class K, class V, class H = std::hash<K>, class E = std::equal_to<K>>
using flat_hash_map = std::unordered_map<K, V, H, E>;
template <class K, class H = std::hash<K>, class E = std::equal_to<K>>
using flat_hash_set = std::unordered_set<K, H, E>;

// ---- utils.h
const uint32_t SPACE_TOKEN = 9601;  // U+2581, the word-start marker

struct Status {  // utils.h:56-64: 0 = ok; the message is user-visible (raised as ValueError by yttm.pyx)
  int code{0};
  std::string message;
  Status() = default;
  Status(int c, std::string m) : code(c), message(std::move(m)) {}
  const std::string &error_message() const { return message; }
  bool ok() const { return code == 0; }
};

struct BPE_Rule {  // x + y -> z
  uint32_t x{0}, y{0}, z{0};
  BPE_Rule() = default;
  BPE_Rule(uint32_t x_, uint32_t y_, uint32_t z_) : x(x_), y(y_), z(z_) {}
  bool operator==(const BPE_Rule &o) const { return x == o.x && y == o.y && z == o.z; }
};

struct SpecialTokens {
  int pad_id = -1, unk_id = -1, bos_id = -1, eos_id = -1;
  SpecialTokens() = default;
  SpecialTokens(int pad, int unk, int bos, int eos) : pad_id(pad), unk_id(unk), bos_id(bos), eos_id(eos) {}
  void dump(std::ofstream &fout);
  void load(std::ifstream &fin);
  uint32_t max_id() const;
  bool taken_id(int id) const { return id == pad_id || id == unk_id || id == bos_id || id == eos_id; }
  uint64_t n_special_tokens() const { return (pad_id != -1) + (unk_id != -1) + (bos_id != -1) + (eos_id != -1); }
};

struct BpeConfig {
  double character_coverage = 1;
  int n_threads = 0;
  SpecialTokens special_tokens;
  BpeConfig() = default;
  BpeConfig(double cov, int threads, const SpecialTokens &st) : character_coverage(cov), n_threads(threads), special_tokens(st) {}
};

struct BPEState {  // the model file (utils.cpp:50-91)
  flat_hash_map<uint32_t, uint32_t> char2id;
  std::vector<BPE_Rule> rules;
  SpecialTokens special_tokens;
  void dump(const std::string &file_name);
  Status load(const std::string &file_name);
};

struct DecodeResult {
  std::vector<int> ids;
  std::vector<std::string> pieces;
};
struct EncodingConfig {
  bool bos, eos, reverse;
  double dropout_prob;
};

bool is_space(uint32_t ch);
std::vector<std::string> read_lines_from_stdin(uint64_t batch_limit, uint64_t *processed);
template <typename T>
void write_to_stdout(const std::vector<std::vector<T>> &sentences, bool flush) {
  for (const auto &s : sentences) {
    for (const auto &t : s) std::cout << t << " ";
    std::cout << "\n";
  }
  if (flush) std::cout << std::flush;
}

// ---- utf8.h (what the reference's tests use of it)
constexpr static uint32_t INVALID_UNICODE = 0x0fffffff;
uint32_t chars_to_utf8(const char *begin, uint64_t size, uint64_t *utf8_len);
void utf8_to_chars(uint32_t x, std::back_insert_iterator<std::string> it);
std::string encode_utf8(const std::vector<uint32_t> &utext);
std::vector<uint32_t> decode_utf8(const char *begin, const char *end);
std::vector<uint32_t> decode_utf8(const std::string &utf8_text);

// ---- bpe.h
const std::string UNK_TOKEN = "<UNK>";
const std::string PAD_TOKEN = "<PAD>";
const std::string BOS_TOKEN = "<BOS>";
const std::string EOS_TOKEN = "<EOS>";
enum OutputType { ID, SUBWORD };

Status train_bpe(const std::string &input_path, const std::string &model_path, int vocab_size, BpeConfig config);  // bpe.h:19

class BaseEncoder {  // bpe.h:22-82
 public:
  BPEState bpe_state;
  flat_hash_map<uint32_t, uint32_t> id2char;
  flat_hash_map<uint32_t, std::vector<uint32_t>> recipe;
  flat_hash_map<std::string, uint32_t> reversed_recipe;
  flat_hash_map<uint64_t, int> rule2id;
  int n_threads;

  explicit BaseEncoder(BPEState bpe_state, int n_threads);  // (the model goes through a temporary file: the C ABI loads models by path)
  explicit BaseEncoder(const std::string &model_path, int n_threads, Status *ret_status);
  ~BaseEncoder();
  BaseEncoder(const BaseEncoder &) = delete;
  BaseEncoder &operator=(const BaseEncoder &) = delete;

  void fill_from_state();

  Status encode_as_ids(const std::vector<std::string> &sentences, std::vector<std::vector<int>> *ids, bool bos = false, bool eos = false,
                       bool reverse = false, double dropout_prob = 0) const;
  Status encode_as_subwords(const std::vector<std::string> &sentences, std::vector<std::vector<std::string>> *subwords, bool bos = false,
                            bool eos = false, bool reverse = false, double dropout_prob = 0) const;
  Status id_to_subword(int id, std::string *subword, bool replace_space = false) const;
  int subword_to_id(const std::string &token) const;
  Status decode(const std::vector<std::vector<int>> &ids, std::vector<std::string> *sentences, const std::unordered_set<int> *ignore_ids) const;
  Status decode(const std::vector<int> &ids, std::string *sentence, const std::unordered_set<int> *ignore_ids) const;
  Status decode(const std::vector<std::string> &ids, std::vector<std::string> *sentences, const std::unordered_set<int> *ignore_ids) const;
  int vocab_size() const;
  std::vector<std::string> vocabulary() const;
  Status encode_cli(const std::string &output_type, bool stream, bool bos = false, bool eos = false, bool reverse = false,
                    double dropout_prob = 0) const;
  Status decode_cli(const std::unordered_set<int> *ignore_ids) const;
  void vocab_cli(bool verbose) const;

 private:
  void *handle_ = nullptr;  // yttm_encoder*
  Status open(const std::string &model_path);
};

// ---- the two private functions the reference's stress test reaches through tests/unit_tests/stress_test.h:8-18
flat_hash_map<uint32_t, uint32_t> compute_alphabet_helper(const flat_hash_map<uint32_t, uint64_t> &char_cnt, uint64_t data_len,
                                                          flat_hash_set<uint32_t> &removed_chars, const BpeConfig &bpe_config);
Status learn_bpe_from_string(std::string &text_utf8, int n_tokens, const std::string &output_file, BpeConfig bpe_config, BPEState *bpe_state);

}  // namespace vkcom
