// vkcom_adapter.cpp -- implementation of shim/cpp/vkcom_adapter.h: the reference's C++ surface (bpe.h:19-82) forwarded to the C ABI of
// libyttm_mi355x.so (include/yttm_mi355x.h).  Nothing of the hot path is computed here: train / encode / decode / the CLI loops are calls
// into the library; what the adapter does itself is marshalling (vector<string> <-> packed bytes + offsets), the model file's text format
// (utils.cpp:50-91) for BaseEncoder's public host-side maps, and the helpers the reference's tests call directly (utf8.cpp, is_space,
// compute_alphabet_helper: bpe.cpp:316-355).
#include "vkcom_adapter.h"

#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <fstream>
#include <mutex>

#include "../../include/yttm_mi355x.h"

namespace vkcom {
namespace {

constexpr int ERRLEN = 2048;

// ---- the library, bound at first use ($YTTM_AMD_LIB; else next to the package this file belongs to; else the loader's search path)
struct Api {
  void *so = nullptr;
  std::string why;
  decltype(&yttm_train_bpe) train_bpe = nullptr;
  decltype(&yttm_train_bpe_from_memory) train_bpe_from_memory = nullptr;
  decltype(&yttm_encoder_create) encoder_create = nullptr;
  decltype(&yttm_encoder_destroy) encoder_destroy = nullptr;
  decltype(&yttm_encode_as_ids) encode_as_ids = nullptr;
  decltype(&yttm_encode_as_subwords) encode_as_subwords = nullptr;
  decltype(&yttm_id_to_subword) id_to_subword = nullptr;
  decltype(&yttm_subword_to_id) subword_to_id = nullptr;
  decltype(&yttm_decode) decode = nullptr;
  decltype(&yttm_vocab_size) vocab_size = nullptr;
  decltype(&yttm_vocabulary) vocabulary = nullptr;
  decltype(&yttm_encode_cli) encode_cli = nullptr;
  decltype(&yttm_decode_cli) decode_cli = nullptr;
  decltype(&yttm_vocab_cli) vocab_cli = nullptr;
  decltype(&yttm_free) free_ = nullptr;
};

template <class F>
bool bind(Api &a, F &slot, const char *name) {
  slot = reinterpret_cast<F>(dlsym(a.so, name));
  if (!slot) a.why = std::string("libyttm_mi355x.so lacks ") + name;
  return slot != nullptr;
}

// where the library is when $YTTM_AMD_LIB does not say: beside the object this code was linked into, or in the youtokentome_amd/ package
// of a directory above it (the repository layout: oracle/_ref/..., shim/...), else whatever the loader finds under the plain name
std::string default_library() {
  Dl_info info;
  if (dladdr(reinterpret_cast<const void *>(&default_library), &info) && info.dli_fname) {
    std::string dir(info.dli_fname);
    for (int up = 0; up < 6; up++) {
      const size_t cut = dir.find_last_of('/');
      if (cut == std::string::npos) break;
      dir.resize(cut);
      for (const char *rel : {"/libyttm_mi355x.so", "/youtokentome_amd/libyttm_mi355x.so"}) {
        const std::string cand = dir + rel;
        if (access(cand.c_str(), R_OK) == 0) return cand;
      }
    }
  }
  return "libyttm_mi355x.so";
}

const Api &api() {
  static Api a;
  static std::once_flag once;
  std::call_once(once, [] {
    const char *env = getenv("YTTM_AMD_LIB");
    const std::string found = env && *env ? std::string(env) : default_library();
    const char *path = found.c_str();
    a.so = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!a.so) {
      a.why = std::string("cannot load the MI355X BPE library (") + path + "): " + dlerror() + " -- there is no CPU fallback";
      return;
    }
    const bool ok = bind(a, a.train_bpe, "yttm_train_bpe") && bind(a, a.train_bpe_from_memory, "yttm_train_bpe_from_memory") &&
                    bind(a, a.encoder_create, "yttm_encoder_create") && bind(a, a.encoder_destroy, "yttm_encoder_destroy") &&
                    bind(a, a.encode_as_ids, "yttm_encode_as_ids") && bind(a, a.encode_as_subwords, "yttm_encode_as_subwords") &&
                    bind(a, a.id_to_subword, "yttm_id_to_subword") && bind(a, a.subword_to_id, "yttm_subword_to_id") &&
                    bind(a, a.decode, "yttm_decode") && bind(a, a.vocab_size, "yttm_vocab_size") && bind(a, a.vocabulary, "yttm_vocabulary") &&
                    bind(a, a.encode_cli, "yttm_encode_cli") && bind(a, a.decode_cli, "yttm_decode_cli") && bind(a, a.vocab_cli, "yttm_vocab_cli") &&
                    bind(a, a.free_, "yttm_free");
    if (!ok) {
      dlclose(a.so);
      a.so = nullptr;
    }
  });
  return a;
}
Status no_library() { return Status(2, api().why); }

// vector<string> -> one blob + offsets[n + 1]
struct Packed {
  std::string bytes;
  std::vector<uint64_t> off;
  explicit Packed(const std::vector<std::string> &v) {
    size_t total = 0;
    for (const auto &s : v) total += s.size();
    bytes.reserve(total);
    off.reserve(v.size() + 1);
    off.push_back(0);
    for (const auto &s : v) {
      bytes += s;
      off.push_back(bytes.size());
    }
  }
  const uint8_t *data() const { return reinterpret_cast<const uint8_t *>(bytes.data()); }
};

Status from_rc(int rc, const char *err) { return rc == 0 ? Status() : Status(rc, err); }

std::string temp_model_path() {
  const char *dir = getenv("TMPDIR");
  std::string p = std::string(dir && *dir ? dir : "/tmp") + "/yttm_adapter_XXXXXX";
  std::vector<char> buf(p.begin(), p.end());
  buf.push_back(0);
  const int fd = mkstemp(buf.data());
  if (fd >= 0) close(fd);
  return std::string(buf.data());
}

}  // namespace

// ------------------------------------------------------------------------------------------------ utils.h
bool is_space(uint32_t ch) { return (ch < 256 && isspace((int)ch)) || ch == SPACE_TOKEN; }  // utils.cpp:99-101

uint32_t SpecialTokens::max_id() const { return (uint32_t)std::max(std::max(pad_id, unk_id), std::max(bos_id, eos_id)); }
void SpecialTokens::dump(std::ofstream &fout) { fout << unk_id << " " << pad_id << " " << bos_id << " " << eos_id << std::endl; }  // utils.cpp:10-13
void SpecialTokens::load(std::ifstream &fin) { fin >> unk_id >> pad_id >> bos_id >> eos_id; }                                    // utils.cpp:15-17

void BPEState::dump(const std::string &file_name) {  // (hash-map order of the char lines: whatever this map iterates in -- a loader does not care)
  std::ofstream fout(file_name, std::ios::out);
  if (fout.fail()) {
    std::cerr << "Can't open file: " << file_name << std::endl;
    abort();
  }
  fout << char2id.size() << " " << rules.size() << std::endl;
  for (const auto &c : char2id) fout << c.first << " " << c.second << std::endl;
  for (const auto &r : rules) fout << r.x << " " << r.y << " " << r.z << std::endl;
  special_tokens.dump(fout);
}

Status BPEState::load(const std::string &file_name) {
  char2id.clear();
  rules.clear();
  std::ifstream fin(file_name, std::ios::in);
  if (fin.fail()) return Status(1, "Can not open file with model: " + file_name);
  int n_chars = 0, n_rules = 0;
  fin >> n_chars >> n_rules;
  for (int i = 0; i < n_chars; i++) {
    uint32_t cp, id;
    fin >> cp >> id;
    char2id[cp] = id;
  }
  for (int i = 0; i < n_rules; i++) {
    uint32_t x, y, z;
    fin >> x >> y >> z;
    rules.emplace_back(x, y, z);
  }
  special_tokens.load(fin);
  return Status();
}

std::vector<std::string> read_lines_from_stdin(uint64_t batch_limit, uint64_t *processed) {  // utils.cpp:103-111
  std::vector<std::string> lines;
  std::string line;
  while (*processed < batch_limit && std::getline(std::cin, line)) {
    *processed += line.size();
    lines.push_back(std::move(line));
  }
  return lines;
}

// ------------------------------------------------------------------------------------------------ utf8.h
// One decoder for both entry points; the classification of malformed input follows utf8.cpp:37-74 (a byte that cannot start a sequence, a
// truncated or ill-formed continuation, an overlong or out-of-range value: INVALID_UNICODE with the length the reference skips).
uint32_t chars_to_utf8(const char *begin, uint64_t size, uint64_t *utf8_len) {
  const unsigned char *p = reinterpret_cast<const unsigned char *>(begin);
  const unsigned char b0 = p[0];
  if (b0 < 0x80) {
    *utf8_len = 1;
    return b0;
  }
  int need = 0;
  uint32_t lo = 0, v = 0;
  if ((b0 & 0xE0) == 0xC0) { need = 1; lo = 0x80; v = b0 & 0x1Fu; }
  else if ((b0 & 0xF0) == 0xE0) { need = 2; lo = 0x800; v = b0 & 0x0Fu; }
  else if ((b0 & 0xF8) == 0xF0) { need = 3; lo = 0x10000; v = b0 & 0x07u; }
  else {
    *utf8_len = 1;
    return INVALID_UNICODE;
  }
  if (size < (uint64_t)need + 1) {  // cut off by the end of the buffer
    *utf8_len = 1;
    return INVALID_UNICODE;
  }
  for (int i = 1; i <= need; i++) {
    if ((p[i] & 0xC0) != 0x80) {
      *utf8_len = 1;
      return INVALID_UNICODE;
    }
    v = (v << 6) | (p[i] & 0x3Fu);
  }
  if (v < lo || v > 0x10FFFF || (v >= 0xD800 && v <= 0xDFFF)) {  // overlong, out of range, a surrogate: one byte is skipped, like every other error
    *utf8_len = 1;
    return INVALID_UNICODE;
  }
  *utf8_len = (uint64_t)need + 1;
  return v;
}

void utf8_to_chars(uint32_t x, std::back_insert_iterator<std::string> it) {
  if (x == INVALID_UNICODE) return;
  if (x < 0x80) { *it++ = (char)x; return; }
  if (x < 0x800) { *it++ = (char)(0xC0 | (x >> 6)); *it++ = (char)(0x80 | (x & 0x3F)); return; }
  if (x < 0x10000) { *it++ = (char)(0xE0 | (x >> 12)); *it++ = (char)(0x80 | ((x >> 6) & 0x3F)); *it++ = (char)(0x80 | (x & 0x3F)); return; }
  *it++ = (char)(0xF0 | (x >> 18));
  *it++ = (char)(0x80 | ((x >> 12) & 0x3F));
  *it++ = (char)(0x80 | ((x >> 6) & 0x3F));
  *it++ = (char)(0x80 | (x & 0x3F));
}

std::string encode_utf8(const std::vector<uint32_t> &utext) {
  std::string out;
  for (uint32_t c : utext) utf8_to_chars(c, std::back_inserter(out));
  return out;
}

std::vector<uint32_t> decode_utf8(const char *begin, const char *end) {  // invalid sequences are dropped, like utf8.cpp:111-128
  std::vector<uint32_t> out;
  bool bad = false;
  while (begin < end) {
    uint64_t len = 0;
    const uint32_t c = chars_to_utf8(begin, (uint64_t)(end - begin), &len);
    if (c != INVALID_UNICODE) out.push_back(c);
    else bad = true;
    begin += len;
  }
  if (bad) std::cerr << "WARNING Input contains invalid unicode characters." << std::endl;
  return out;
}
std::vector<uint32_t> decode_utf8(const std::string &s) { return decode_utf8(s.data(), s.data() + s.size()); }

// ------------------------------------------------------------------------------------------------ bpe.h: training
Status train_bpe(const std::string &input_path, const std::string &model_path, int vocab_size, BpeConfig c) {
  const Api &a = api();
  if (!a.so) return no_library();
  char err[ERRLEN] = {0};
  const SpecialTokens &t = c.special_tokens;
  return from_rc(a.train_bpe(input_path.c_str(), model_path.c_str(), vocab_size, c.character_coverage, c.n_threads, t.pad_id, t.unk_id, t.bos_id, t.eos_id, err, ERRLEN), err);
}

// The reference's test entry point (bpe.cpp:859; stress_test.h:14-18): train on a string, return the state.  Here: the string goes to
// yttm_train_bpe_from_memory, the model file it writes (output_file, as the reference's does) is read back.
Status learn_bpe_from_string(std::string &text_utf8, int n_tokens, const std::string &output_file, BpeConfig c, BPEState *bpe_state) {
  const Api &a = api();
  if (!a.so) return no_library();
  char err[ERRLEN] = {0};
  const SpecialTokens &t = c.special_tokens;
  const int rc = a.train_bpe_from_memory(reinterpret_cast<const uint8_t *>(text_utf8.data()), text_utf8.size(), output_file.c_str(), n_tokens, c.character_coverage,
                                         t.pad_id, t.unk_id, t.bos_id, t.eos_id, /*device=*/0, nullptr, 0, err, ERRLEN);
  if (rc != 0) return Status(rc, err);
  return bpe_state ? bpe_state->load(output_file) : Status();
}

// bpe.cpp:316-355, for the reference's brute-force trainer in stress_test.cpp (learn_bpe_slow): sort by (count, char), drop the rarest
// while what is left still covers `character_coverage` of the text, ids by descending (count, char) behind the word-start marker.
flat_hash_map<uint32_t, uint32_t> compute_alphabet_helper(const flat_hash_map<uint32_t, uint64_t> &char_cnt, uint64_t data_len,
                                                          flat_hash_set<uint32_t> &removed_chars, const BpeConfig &cfg) {
  std::vector<std::pair<uint64_t, uint32_t>> freq;
  freq.reserve(char_cnt.size());
  for (const auto &kv : char_cnt) freq.emplace_back(kv.second, kv.first);
  std::sort(freq.begin(), freq.end());
  size_t cut = 0;
  uint64_t removed = 0;
  while (cut < freq.size() && (double)(data_len - removed - freq[cut].first) > (double)data_len * cfg.character_coverage) removed += freq[cut++].first;
  std::cerr << "number of unique characters in the training data: " << freq.size() << std::endl;
  std::cerr << "number of deleted characters: " << cut << std::endl;
  std::cerr << "number of unique characters left: " << freq.size() - cut << std::endl;
  flat_hash_map<uint32_t, uint32_t> char2id;
  uint64_t next_id = cfg.special_tokens.n_special_tokens();
  char2id[SPACE_TOKEN] = (uint32_t)next_id++;
  for (size_t i = 0; i < cut; i++) removed_chars.insert(freq[i].second);
  for (size_t i = freq.size(); i-- > cut;)
    if (!is_space(freq[i].second)) char2id[freq[i].second] = (uint32_t)next_id++;
  return char2id;
}

// ------------------------------------------------------------------------------------------------ bpe.h: BaseEncoder
Status BaseEncoder::open(const std::string &model_path) {
  const Api &a = api();
  if (!a.so) return no_library();
  char err[ERRLEN] = {0};
  yttm_encoder *e = nullptr;
  const int rc = a.encoder_create(model_path.c_str(), n_threads, /*device=*/0, &e, err, ERRLEN);
  if (rc != 0) return Status(rc, err);
  handle_ = e;
  return Status();
}

BaseEncoder::BaseEncoder(const std::string &model_path, int n_threads_, Status *ret_status) : n_threads(n_threads_) {
  Status s = bpe_state.load(model_path);  // (the same "Can not open file with model" as bpe.cpp:1645-1648)
  if (s.ok()) s = open(model_path);
  if (s.ok()) fill_from_state();
  if (ret_status) *ret_status = s;
}

BaseEncoder::BaseEncoder(BPEState state, int n_threads_) : bpe_state(std::move(state)), n_threads(n_threads_) {
  const std::string tmp = temp_model_path();
  bpe_state.dump(tmp);
  const Status s = open(tmp);
  unlink(tmp.c_str());
  if (!s.ok()) {
    std::cerr << "BaseEncoder: " << s.message << std::endl;
    abort();  // (this constructor has no Status to report through: bpe.h:29)
  }
  fill_from_state();
}

BaseEncoder::~BaseEncoder() {
  if (handle_ && api().so) api().encoder_destroy(static_cast<yttm_encoder *>(handle_));
}

void BaseEncoder::fill_from_state() {  // bpe.cpp:1667-1690: host-side views of the model (the library keeps its own on the device)
  id2char.clear(); recipe.clear(); reversed_recipe.clear(); rule2id.clear();
  for (const auto &c : bpe_state.char2id) id2char[c.second] = c.first;
  for (size_t i = 0; i < bpe_state.rules.size(); i++) {
    const BPE_Rule &r = bpe_state.rules[i];
    rule2id[((uint64_t)r.x << 32) | r.y] = (int)i;
  }
  for (const auto &c : id2char) recipe[c.first] = {c.first};
  for (const BPE_Rule &r : bpe_state.rules) {
    std::vector<uint32_t> v = recipe[r.x];
    const std::vector<uint32_t> &w = recipe[r.y];
    v.insert(v.end(), w.begin(), w.end());
    recipe[r.z] = std::move(v);
  }
  for (const auto &kv : recipe) {
    std::vector<uint32_t> cps;
    cps.reserve(kv.second.size());
    for (uint32_t id : kv.second) cps.push_back(id2char.at(id));
    reversed_recipe[encode_utf8(cps)] = kv.first;
  }
}

Status BaseEncoder::encode_as_ids(const std::vector<std::string> &sentences, std::vector<std::vector<int>> *ids, bool bos, bool eos, bool reverse,
                                  double dropout_prob) const {
  const Api &a = api();
  if (!a.so) return no_library();
  const Packed in(sentences);
  int32_t *out = nullptr;
  uint64_t *off = nullptr;
  char err[ERRLEN] = {0};
  const int rc = a.encode_as_ids(static_cast<yttm_encoder *>(handle_), in.data(), in.off.data(), sentences.size(), bos, eos, reverse, dropout_prob, &out, &off, err, ERRLEN);
  if (rc != 0) return Status(rc, err);
  ids->assign(sentences.size(), {});
  for (size_t i = 0; i < sentences.size(); i++) (*ids)[i].assign(out + off[i], out + off[i + 1]);
  a.free_(out);
  a.free_(off);
  return Status();
}

Status BaseEncoder::encode_as_subwords(const std::vector<std::string> &sentences, std::vector<std::vector<std::string>> *subwords, bool bos, bool eos,
                                       bool reverse, double dropout_prob) const {
  const Api &a = api();
  if (!a.so) return no_library();
  const Packed in(sentences);
  char *blob = nullptr;
  uint64_t *piece_off = nullptr, *sent_off = nullptr, n_pieces = 0;
  char err[ERRLEN] = {0};
  const int rc = a.encode_as_subwords(static_cast<yttm_encoder *>(handle_), in.data(), in.off.data(), sentences.size(), bos, eos, reverse, dropout_prob, &blob,
                                      &piece_off, &n_pieces, &sent_off, err, ERRLEN);
  if (rc != 0) return Status(rc, err);
  subwords->assign(sentences.size(), {});
  for (size_t i = 0; i < sentences.size(); i++) {
    auto &dst = (*subwords)[i];
    dst.reserve(sent_off[i + 1] - sent_off[i]);
    for (uint64_t p = sent_off[i]; p < sent_off[i + 1]; p++) dst.emplace_back(blob + piece_off[p], blob + piece_off[p + 1]);
  }
  a.free_(blob);
  a.free_(piece_off);
  a.free_(sent_off);
  return Status();
}

Status BaseEncoder::id_to_subword(int id, std::string *subword, bool replace_space) const {
  const Api &a = api();
  if (!a.so) return no_library();
  char *s = nullptr;
  char err[ERRLEN] = {0};
  const int rc = a.id_to_subword(static_cast<yttm_encoder *>(handle_), id, &s, err, ERRLEN);
  if (rc != 0) return Status(rc, err);
  *subword = s;
  a.free_(s);
  static const char marker[] = "\xE2\x96\x81";  // U+2581
  if (replace_space && subword->compare(0, 3, marker) == 0 && !bpe_state.special_tokens.taken_id(id)) subword->replace(0, 3, " ");  // bpe.cpp:1798-1804
  return Status();
}

int BaseEncoder::subword_to_id(const std::string &token) const {
  const Api &a = api();
  return a.so ? a.subword_to_id(static_cast<yttm_encoder *>(handle_), token.c_str()) : bpe_state.special_tokens.unk_id;
}

Status BaseEncoder::decode(const std::vector<std::vector<int>> &ids, std::vector<std::string> *sentences, const std::unordered_set<int> *ignore_ids) const {
  const Api &a = api();
  if (!a.so) return no_library();
  std::vector<int32_t> flat;
  std::vector<uint64_t> off{0};
  for (const auto &v : ids) {
    flat.insert(flat.end(), v.begin(), v.end());
    off.push_back(flat.size());
  }
  std::vector<int32_t> ign;
  if (ignore_ids) ign.assign(ignore_ids->begin(), ignore_ids->end());
  char *blob = nullptr;
  uint64_t *out_off = nullptr;
  char err[ERRLEN] = {0};
  const int rc = a.decode(static_cast<yttm_encoder *>(handle_), flat.data(), off.data(), ids.size(), ign.data(), ign.size(), &blob, &out_off, err, ERRLEN);
  if (rc != 0) return Status(rc, err);
  sentences->clear();
  for (size_t i = 0; i < ids.size(); i++) sentences->emplace_back(blob + out_off[i], blob + out_off[i + 1]);
  a.free_(blob);
  a.free_(out_off);
  return Status();
}

Status BaseEncoder::decode(const std::vector<int> &ids, std::string *sentence, const std::unordered_set<int> *ignore_ids) const {
  std::vector<std::string> one;
  Status s = decode(std::vector<std::vector<int>>{ids}, &one, ignore_ids);
  if (s.ok()) *sentence = one.empty() ? std::string() : one[0];
  return s;
}

Status BaseEncoder::decode(const std::vector<std::string> &data, std::vector<std::string> *sentences, const std::unordered_set<int> *ignore_ids) const {
  // lines of white-space separated ids (bpe.cpp:1863-1882)
  std::vector<std::vector<int>> ids;
  for (const auto &line : data) {
    std::vector<int> v;
    const char *p = line.c_str();
    char *end = nullptr;
    for (long x = strtol(p, &end, 10); end != p; x = strtol(p, &end, 10)) {
      v.push_back((int)x);
      p = end;
    }
    ids.push_back(std::move(v));
  }
  return decode(ids, sentences, ignore_ids);
}

int BaseEncoder::vocab_size() const {
  const Api &a = api();
  return a.so && handle_ ? a.vocab_size(static_cast<yttm_encoder *>(handle_)) : (int)(bpe_state.rules.size() + bpe_state.char2id.size() + bpe_state.special_tokens.n_special_tokens());
}

std::vector<std::string> BaseEncoder::vocabulary() const {
  std::vector<std::string> v;
  const Api &a = api();
  if (!a.so) return v;
  char *blob = nullptr;
  uint64_t *off = nullptr, n = 0;
  if (a.vocabulary(static_cast<yttm_encoder *>(handle_), &blob, &off, &n) != 0) return v;
  v.reserve(n);
  for (uint64_t i = 0; i < n; i++) v.emplace_back(blob + off[i], blob + off[i + 1]);
  a.free_(blob);
  a.free_(off);
  return v;
}

Status BaseEncoder::encode_cli(const std::string &output_type, bool stream, bool bos, bool eos, bool reverse, double dropout_prob) const {
  const Api &a = api();
  if (!a.so) return no_library();
  char err[ERRLEN] = {0};
  std::cout << std::flush;
  return from_rc(a.encode_cli(static_cast<yttm_encoder *>(handle_), output_type.c_str(), stream, bos, eos, reverse, dropout_prob, 0, 1, err, ERRLEN), err);
}

Status BaseEncoder::decode_cli(const std::unordered_set<int> *ignore_ids) const {
  const Api &a = api();
  if (!a.so) return no_library();
  std::vector<int32_t> ign;
  if (ignore_ids) ign.assign(ignore_ids->begin(), ignore_ids->end());
  char err[ERRLEN] = {0};
  std::cout << std::flush;
  return from_rc(a.decode_cli(static_cast<yttm_encoder *>(handle_), ign.data(), ign.size(), 0, 1, err, ERRLEN), err);
}

void BaseEncoder::vocab_cli(bool verbose) const {
  const Api &a = api();
  if (!a.so) return;
  char err[ERRLEN] = {0};
  std::cout << std::flush;
  a.vocab_cli(static_cast<yttm_encoder *>(handle_), verbose, 1, err, ERRLEN);
}

}  // namespace vkcom
