// host_model.cpp -- model state, file format and alphabet selection (host side; tiny, not on the hot path).
// Behavioural spec: SURVEY.md Appendix A.1-A.3, A.6, A.8.
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <fstream>
#include <thread>

#include "host_core.h"

namespace yttm {

// ------------------------------------------------------------------------------------------------- utf8 / spaces
bool is_space(uint32_t ch) {  // utils.cpp:99-101 (isspace in the "C" locale)
  return (ch < 256 && (ch == 32 || (ch >= 9 && ch <= 13))) || ch == SPACE_TOKEN;
}

static void utf8_append(uint32_t x, std::string &out) {  // utf8.cpp:76-100
  if (x <= 0x7f) {
    out.push_back((char)x);
  } else if (x <= 0x7ff) {
    out.push_back((char)(0xc0u | (x >> 6)));
    out.push_back((char)(0x80u | (x & 0x3fu)));
  } else if (x <= 0xffff) {
    out.push_back((char)(0xe0u | (x >> 12)));
    out.push_back((char)(0x80u | ((x >> 6) & 0x3fu)));
    out.push_back((char)(0x80u | (x & 0x3fu)));
  } else {
    out.push_back((char)(0xf0u | (x >> 18)));
    out.push_back((char)(0x80u | ((x >> 12) & 0x3fu)));
    out.push_back((char)(0x80u | ((x >> 6) & 0x3fu)));
    out.push_back((char)(0x80u | (x & 0x3fu)));
  }
}

std::string encode_utf8(const std::vector<uint32_t> &text) {
  std::string s;
  for (uint32_t c : text) utf8_append(c, s);
  return s;
}

static bool cont_byte(unsigned char b) { return (b & 0xc0u) == 0x80u; }
static bool codepoint_ok(uint32_t x) { return (x < 0xd800) || (0xdfff < x && x < 0x110000); }

std::vector<uint32_t> decode_utf8(const char *begin, const char *end, bool *invalid) {  // utf8.cpp:111-128
  std::vector<uint32_t> out;
  bool bad = false;
  const unsigned char *p = (const unsigned char *)begin, *e = (const unsigned char *)end;
  while (p < e) {
    size_t avail = (size_t)(e - p);
    uint32_t b0 = p[0], cp = 0;
    size_t len = 1;
    bool ok = false;
    if (b0 < 0x80) { cp = b0; ok = true; }
    else if ((b0 & 0xe0) == 0xc0) {
      if (avail >= 2 && cont_byte(p[1])) { cp = ((b0 & 0x1f) << 6) + (p[1] & 0x3f); if (cp >= 0x80 && codepoint_ok(cp)) { ok = true; len = 2; } }
    } else if ((b0 & 0xf0) == 0xe0) {
      if (avail >= 3 && cont_byte(p[1]) && cont_byte(p[2])) {
        cp = ((b0 & 0x0f) << 12) + ((p[1] & 0x3f) << 6) + (p[2] & 0x3f);
        if (cp >= 0x800 && codepoint_ok(cp)) { ok = true; len = 3; }
      }
    } else if ((b0 & 0xf8) == 0xf0) {
      if (avail >= 4 && cont_byte(p[1]) && cont_byte(p[2]) && cont_byte(p[3])) {
        cp = ((b0 & 0x07) << 18) + ((p[1] & 0x3f) << 12) + ((p[2] & 0x3f) << 6) + (p[3] & 0x3f);
        if (cp >= 0x10000 && codepoint_ok(cp)) { ok = true; len = 4; }
      }
    }
    if (ok) out.push_back(cp); else bad = true;
    p += len;
  }
  if (invalid) *invalid = bad;
  return out;
}

// ------------------------------------------------------------------------------------------------- hash-slot order
// The reference writes the char section of the model file by iterating a ska::flat_hash_map<uint32_t,uint32_t>
// (utils.cpp:57-59), so the line order is the slot order of that robin-hood table (third_party/flat_hash_map.h:
// fibonacci hashing :1274-1300, max load 0.5 :800, insertion :830-873, growth :875-878 / :630-663, copy :361-367).
// To emit byte-identical files we replay exactly that table for the key sequence: insertions into an empty map, then
// one copy construction (bpe.cpp:1289).
namespace {
class RobinHoodReplay {
 public:
  RobinHoodReplay() : dist_(4, -1), key_(4, 0) { dist_[3] = 0; }
  void reserve_like_copy_of(const RobinHoodReplay &o) {
    uint64_t want = (uint64_t)ceil((double)o.n_ / 0.5);
    rehash(std::min(want, o.buckets()));
  }
  void insert(uint32_t k) {
    uint64_t cur = (11400714819323198485ull * (uint64_t)k) >> shift_;
    int d = 0;
    for (; dist_[cur] >= d; ++cur, ++d)
      if (key_[cur] == k) return;
    if (slots_m1_ == 0 || d == max_lookups_ || (double)(n_ + 1) > (double)(slots_m1_ + 1) * (double)0.5f) {
      grow();
      insert(k);
      return;
    }
    if (dist_[cur] < 0) { dist_[cur] = (int8_t)d; key_[cur] = k; n_++; return; }
    std::swap(k, key_[cur]);
    { int8_t t = dist_[cur]; dist_[cur] = (int8_t)d; d = t; }
    const uint64_t first = cur;
    for (++d, ++cur;; ++cur) {
      if (dist_[cur] < 0) { dist_[cur] = (int8_t)d; key_[cur] = k; n_++; return; }
      if (dist_[cur] < d) {
        int8_t t = dist_[cur]; dist_[cur] = (int8_t)d; d = t;
        std::swap(k, key_[cur]);
        ++d;
      } else {
        ++d;
        if (d == max_lookups_) {
          std::swap(k, key_[first]);
          grow();
          insert(k);
          return;
        }
      }
    }
  }
  std::vector<uint32_t> slot_order() const {
    std::vector<uint32_t> out;
    const uint64_t cnt = slots_m1_ + (uint64_t)max_lookups_;
    for (uint64_t i = 0; i < cnt; i++)
      if (dist_[i] >= 0) out.push_back(key_[i]);
    return out;
  }
  uint64_t buckets() const { return slots_m1_ ? slots_m1_ + 1 : 0; }
  uint64_t size() const { return n_; }

 private:
  static int ilog2(uint64_t v) { int r = 0; while (v >>= 1) r++; return r; }
  void grow() { rehash(std::max<uint64_t>(4, 2 * buckets())); }
  void rehash(uint64_t nb) {
    nb = std::max(nb, (uint64_t)ceil((double)n_ / (double)0.5f));
    if (nb == 0) return;
    uint64_t p = 1;
    while (p < nb) p <<= 1;
    nb = std::max<uint64_t>(2, p);
    if (nb == buckets()) return;
    const int new_ml = std::max(4, ilog2(nb));
    std::vector<int8_t> od;
    std::vector<uint32_t> ok;
    od.swap(dist_);
    ok.swap(key_);
    const uint64_t old_cnt = slots_m1_ + (uint64_t)max_lookups_;
    dist_.assign(nb + (uint64_t)new_ml, -1);
    key_.assign(nb + (uint64_t)new_ml, 0);
    dist_.back() = 0;  // end sentinel
    slots_m1_ = nb - 1;
    shift_ = 64 - ilog2(nb);
    max_lookups_ = new_ml;
    n_ = 0;
    for (uint64_t i = 0; i < old_cnt; i++)
      if (od[i] >= 0) insert(ok[i]);
  }
  std::vector<int8_t> dist_;
  std::vector<uint32_t> key_;
  uint64_t slots_m1_ = 0, n_ = 0;
  int shift_ = 63, max_lookups_ = 3;
};
}  // namespace

std::vector<uint32_t> flat_hash_map_order(const std::vector<uint32_t> &keys) {
  RobinHoodReplay a;
  for (uint32_t k : keys) a.insert(k);
  RobinHoodReplay b;
  b.reserve_like_copy_of(a);
  for (uint32_t k : a.slot_order()) b.insert(k);
  return b.slot_order();
}

// ------------------------------------------------------------------------------------------------- alphabet
void compute_alphabet(const std::vector<uint32_t> &cps, const std::vector<unsigned long long> &cnts, unsigned long long data_len,
                      const BpeConfig &cfg, std::vector<std::pair<uint32_t, uint32_t>> &out, uint64_t &n_removed_chars) {
  // bpe.cpp:316-355
  std::vector<std::pair<unsigned long long, uint32_t>> freq;
  for (size_t i = 0; i < cps.size(); i++) freq.emplace_back(cnts[i], cps[i]);
  std::sort(freq.begin(), freq.end());
  uint64_t cur = 0, n_removed = 0;
  for (; cur < freq.size() && (double)(data_len - n_removed - freq[cur].first) > (double)data_len * cfg.character_coverage; cur++)
    n_removed += freq[cur].first;
  fprintf(stderr, "number of unique characters in the training data: %zu\n", freq.size());
  fprintf(stderr, "number of deleted characters: %llu\n", (unsigned long long)cur);
  fprintf(stderr, "number of unique characters left: %llu\n", (unsigned long long)(freq.size() - cur));
  out.clear();
  uint32_t used = (uint32_t)cfg.special_tokens.n_special_tokens();
  out.emplace_back(SPACE_TOKEN, used++);
  for (int64_t i = (int64_t)freq.size() - 1; i >= (int64_t)cur; i--)
    if (!is_space(freq[(size_t)i].second)) out.emplace_back(freq[(size_t)i].second, used++);
  n_removed_chars = cur;
}

// ------------------------------------------------------------------------------------------------- config
Status check_config(BpeConfig &c, int vocab_size) {  // bpe.cpp:1295-1350; messages verbatim
  const SpecialTokens &s = c.special_tokens;
  if (c.character_coverage <= 0 || c.character_coverage > 1)
    return Status(1, "coverage value must be in the range (0, 1]. Current value of coverage = " + std::to_string(c.character_coverage));
  if (s.unk_id < 0 || s.unk_id >= vocab_size)
    return Status(1, "unk_id: must be in the range [0, vocab_size - 1]. Current value of vocab_size = " + std::to_string(vocab_size) +
                         "; unk_id = " + std::to_string(s.unk_id));
  if (s.pad_id < -1 || s.pad_id >= vocab_size)
    return Status(1, "pad_id must be in the range [-1, vocab_size - 1]. Current value of vocab_size = " + std::to_string(vocab_size) +
                         "; pad_id = " + std::to_string(s.pad_id));
  if (s.bos_id < -1 || s.bos_id >= vocab_size)
    return Status(1, "bos_id must be in the range [-1, vocab_size - 1]. Current value of vocab_size = " + std::to_string(vocab_size) +
                         "; bos_id = " + std::to_string(s.bos_id));
  if (s.eos_id < -1 || s.eos_id >= vocab_size)
    return Status(1, "eos_id must be in the range [-1, vocab_size - 1]. Current value of vocab_size = " + std::to_string(vocab_size) +
                         " eos_id = " + std::to_string(s.eos_id));
  std::vector<int> ids;
  if (s.pad_id != -1) ids.push_back(s.pad_id);
  if (s.bos_id != -1) ids.push_back(s.bos_id);
  if (s.eos_id != -1) ids.push_back(s.eos_id);
  ids.push_back(s.unk_id);
  std::sort(ids.begin(), ids.end());
  if (std::adjacent_find(ids.begin(), ids.end()) != ids.end()) return Status(1, "All ids of special tokens must be different.");
  if (c.n_threads == -1) c.n_threads = (int)std::thread::hardware_concurrency();
  c.n_threads = std::min(8, std::max(1, c.n_threads));  // accepted for API compatibility; the GPU path ignores it
  return Status();
}

// ------------------------------------------------------------------------------------------------- model file
// decimal digits of v at p; returns the end.  (The model file is 32 000 lines of three numbers: fprintf took 3.5 ms of a 142 ms training.)
static inline char *put_u32(char *p, uint32_t v) {
  char tmp[10];
  int n = 0;
  do { tmp[n++] = (char)('0' + v % 10u); v /= 10u; } while (v);
  while (n) *p++ = tmp[--n];
  return p;
}
static inline char *put_i32(char *p, int v) {
  if (v < 0) { *p++ = '-'; return put_u32(p, (uint32_t)(-(long long)v)); }
  return put_u32(p, (uint32_t)v);
}

Status BPEState::dump(const std::string &file_name) const {  // utils.cpp:50-66, :10-13
  FILE *f = fopen(file_name.c_str(), "wb");
  if (!f) return Status(1, "Can't open file: " + file_name);
  std::string buf;
  buf.resize(64 + 24 * char2id.size() + 36 * rules.size() + 64);
  char *p = &buf[0];
  p += sprintf(p, "%zu %zu\n", char2id.size(), rules.size());
  for (auto &c : char2id) {
    p = put_u32(p, c.first); *p++ = ' ';
    p = put_u32(p, c.second); *p++ = '\n';
  }
  for (auto &r : rules) {
    p = put_u32(p, r.x); *p++ = ' ';
    p = put_u32(p, r.y); *p++ = ' ';
    p = put_u32(p, r.z); *p++ = '\n';
  }
  p = put_i32(p, special_tokens.unk_id); *p++ = ' ';
  p = put_i32(p, special_tokens.pad_id); *p++ = ' ';
  p = put_i32(p, special_tokens.bos_id); *p++ = ' ';
  p = put_i32(p, special_tokens.eos_id); *p++ = '\n';
  const size_t len = (size_t)(p - &buf[0]);
  const bool ok = fwrite(buf.data(), 1, len, f) == len;
  if (fclose(f) != 0 || !ok) return Status(1, "Can't write file: " + file_name);
  return Status();
}

Status BPEState::load(const std::string &file_name) {  // utils.cpp:68-91
  char2id.clear();
  rules.clear();
  std::ifstream fin(file_name, std::ios::in);
  if (fin.fail()) return Status(1, "Can not open file with model: " + file_name);
  int n = 0, m = 0;
  fin >> n >> m;
  for (int i = 0; i < n; i++) {
    uint32_t cp = 0, id = 0;
    fin >> cp >> id;
    char2id.emplace_back(cp, id);
  }
  for (int i = 0; i < m; i++) {
    BPE_Rule r;
    fin >> r.x >> r.y >> r.z;
    rules.push_back(r);
  }
  fin >> special_tokens.unk_id >> special_tokens.pad_id >> special_tokens.bos_id >> special_tokens.eos_id;
  return Status();
}

}  // namespace yttm
