// capi.cpp -- extern "C" entry points declared in include/yttm_mi355x.h (drop-in boundary) and include/yttm_gpu.h
// (kernel stages).  Thin marshalling only.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/yttm_gpu.h"
#include "../../include/yttm_mi355x.h"
#include "gpu_ctx.h"
#include "host_core.h"

using namespace yttm;

static void put_err(char *err, int errlen, const std::string &m) {
  if (err && errlen > 0) snprintf(err, (size_t)errlen, "%s", m.c_str());
}
static int finish(const Status &s, char *err, int errlen) {
  if (!s.ok()) put_err(err, errlen, s.message);
  return s.code;
}
static BpeConfig make_cfg(double coverage, int n_threads, int pad, int unk, int bos, int eos) {
  BpeConfig c;
  c.character_coverage = coverage;
  c.n_threads = n_threads;
  c.special_tokens.pad_id = pad;
  c.special_tokens.unk_id = unk;
  c.special_tokens.bos_id = bos;
  c.special_tokens.eos_id = eos;
  return c;
}
void yttm_report_to_json(const TrainReport &r, char *buf, int len) {
  if (!buf || len <= 0) return;
  static const char *names[8] = {"char_hist", "segments", "dedup", "build", "pair_count", "merge_apply", "cand_scan", "exchange"};
  std::string s = "{";
  char tmp[1024];
  snprintf(tmp, sizeof tmp, "\"seconds_total\": %.6f, \"seconds_upload\": %.6f, \"seconds_frontend\": %.6f, \"seconds_merge\": %.6f, \"seconds_io\": %.6f, ", r.seconds_total,
           r.seconds_upload, r.seconds_frontend, r.seconds_merge, r.seconds_io);
  s += tmp;
  snprintf(tmp, sizeof tmp, "\"corpus_bytes\": %llu, \"n_unique\": %llu, \"n_tokens\": %llu, \"rounds\": %llu, \"rules\": %llu, \"cand_rescans\": %llu, \"hot_rebuilds\": %llu, \"repacks\": %llu, \"merge_sites\": %llu, ",
           r.corpus_bytes, r.n_unique, r.n_tokens, r.rounds, r.rules, r.cand_rescans, r.hot_rebuilds, r.repacks, r.merge_sites);
  s += tmp;
  snprintf(tmp, sizeof tmp, "\"fused_rounds\": %llu, \"fused_overflows\": %llu, \"exchange_retries\": %llu, \"word_table_retries\": %llu, \"front_end_overlapped\": %llu, \"top_refills\": %llu, \"index_builds\": %llu, \"word_rounds\": %llu, \"word_switch_round\": %llu, \"word_all_rounds\": %llu, \"word_fused_rounds\": %llu, \"rounds_exhausted\": %llu, \"batch_extensions\": %llu, \"batch_splits\": %llu, \"replicated_merge_loop\": %llu, \"front_end_chunks\": %llu, \"peak_device_bytes\": %llu, \"classb_overlapped\": %llu, \"k3_radix\": %llu, ",
           r.fused_rounds, r.fused_overflows, r.exchange_retries, r.word_table_retries, r.front_end_overlapped, r.top_refills, r.index_builds, r.word_rounds, r.word_switch_round, r.word_all_rounds, r.word_fused_rounds, r.rounds_exhausted, r.batch_extensions, r.batch_splits, r.replicated_merge_loop, r.front_end_chunks, r.peak_device_bytes, r.classb_overlapped, r.k3_radix);
  s += tmp;
  snprintf(tmp, sizeof tmp, "\"touched_tiles\": %llu, \"touched_tile_tokens\": %llu, \"touched_words\": %llu, \"touched_word_tokens\": %llu, ",
           r.touched_tiles, r.touched_tile_tokens, r.touched_words, r.touched_word_tokens);
  s += tmp;
  snprintf(tmp, sizeof tmp, "\"split_round\": %llu, \"split_sites\": %llu, \"split_touched_words\": %llu, \"split_touched_word_tokens\": %llu, \"merge_ms_word_rounds\": %.6f, \"merge_launches_word_rounds\": %llu, ",
           r.split_round, r.split_sites, r.split_touched_words, r.split_touched_word_tokens, r.merge_ms_words, r.merge_launches_words);
  s += tmp;
  s += "\"kernels\": {";
  for (int i = 0; i < 8; i++) {
    snprintf(tmp, sizeof tmp, "%s\"%s\": {\"ms\": %.6f, \"launches\": %llu, \"bytes\": %llu}", i ? ", " : "", names[i], r.kt_ms[i], r.kt_launches[i],
             r.kt_bytes[i]);
    s += tmp;
  }
  s += "}}";
  snprintf(buf, (size_t)len, "%s", s.c_str());
}

template <class T>
static T *to_malloc(const std::vector<T> &v) {
  T *p = (T *)malloc((v.size() ? v.size() : 1) * sizeof(T));
  if (!v.empty()) memcpy(p, v.data(), v.size() * sizeof(T));
  return p;
}

extern "C" {

// ------------------------------------------------------------------------------------------------- outer ABI
int yttm_train_bpe(const char *input_path, const char *model_path, int vocab_size, double coverage, int n_threads, int pad_id, int unk_id,
                   int bos_id, int eos_id, char *err, int errlen) {
  return finish(train_bpe(input_path, model_path, vocab_size, make_cfg(coverage, n_threads, pad_id, unk_id, bos_id, eos_id)), err, errlen);
}

int yttm_train_bpe_ex(const char *input_path, const char *model_path, int vocab_size, double coverage, int n_threads, int pad_id, int unk_id,
                      int bos_id, int eos_id, int device, char *report_json, int report_len, char *err, int errlen) {
  TrainReport rep;
  Status s = train_bpe(input_path, model_path, vocab_size, make_cfg(coverage, n_threads, pad_id, unk_id, bos_id, eos_id), device, &rep);
  if (s.ok()) yttm_report_to_json(rep, report_json, report_len);
  return finish(s, err, errlen);
}

int yttm_train_bpe_from_memory(const uint8_t *text, uint64_t n, const char *model_path, int vocab_size, double coverage, int pad_id,
                               int unk_id, int bos_id, int eos_id, int device, char *report_json, int report_len, char *err, int errlen) {
  TrainReport rep;
  Status s = train_bpe_from_memory(text, n, model_path ? model_path : "", vocab_size, make_cfg(coverage, 1, pad_id, unk_id, bos_id, eos_id), device, &rep);
  if (s.ok()) yttm_report_to_json(rep, report_json, report_len);
  return finish(s, err, errlen);
}

int yttm_train_bpe_from_device(const void *d_text, uint64_t n, const char *model_path, int vocab_size, double coverage, int pad_id, int unk_id,
                               int bos_id, int eos_id, int device, int profile, char *report_json, int report_len, char *err, int errlen) {
  TrainReport rep;
  Status s = train_bpe_from_device(d_text, n, model_path ? model_path : "", vocab_size, make_cfg(coverage, 1, pad_id, unk_id, bos_id, eos_id), device,
                                   &rep, nullptr, profile);
  if (s.ok()) yttm_report_to_json(rep, report_json, report_len);
  return finish(s, err, errlen);
}

struct yttm_encoder {
  BaseEncoder *enc;
};

int yttm_encoder_create(const char *model_path, int n_threads, int device, yttm_encoder **out, char *err, int errlen) {
  Status st;
  BaseEncoder *e = new BaseEncoder(model_path, n_threads, &st, device);
  if (!st.ok()) {
    delete e;
    *out = nullptr;
    return finish(st, err, errlen);
  }
  *out = new yttm_encoder{e};
  return 0;
}
void yttm_encoder_destroy(yttm_encoder *h) {
  if (!h) return;
  delete h->enc;
  delete h;
}

int yttm_encode_as_ids(yttm_encoder *h, const uint8_t *bytes, const uint64_t *offsets, uint64_t n_sent, int bos, int eos, int reverse,
                       double dropout_prob, int32_t **ids, uint64_t **out_offsets, char *err, int errlen) {
  unsigned long long *off = nullptr;
  Status s = h->enc->encode_as_ids_malloc(bytes, (const unsigned long long *)offsets, n_sent, bos, eos, reverse, dropout_prob, ids, &off);
  if (!s.ok()) return finish(s, err, errlen);
  *out_offsets = (uint64_t *)off;
  return 0;
}

static void pack_strings(const std::vector<std::string> &v, char **blob, uint64_t **off) {
  size_t total = 0;
  for (auto &s : v) total += s.size();
  char *b = (char *)malloc(total + 1);
  uint64_t *o = (uint64_t *)malloc((v.size() + 1) * sizeof(uint64_t));
  size_t p = 0;
  for (size_t i = 0; i < v.size(); i++) {
    o[i] = p;
    memcpy(b + p, v[i].data(), v[i].size());
    p += v[i].size();
  }
  o[v.size()] = p;
  b[p] = 0;
  *blob = b;
  *off = o;
}

int yttm_encode_as_subwords(yttm_encoder *h, const uint8_t *bytes, const uint64_t *offsets, uint64_t n_sent, int bos, int eos, int reverse,
                            double dropout_prob, char **blob, uint64_t **piece_off, uint64_t *n_pieces, uint64_t **sent_off, char *err,
                            int errlen) {
  std::vector<std::string> pieces;
  std::vector<unsigned long long> so;
  Status s = h->enc->encode_as_subwords(bytes, (const unsigned long long *)offsets, n_sent, bos, eos, reverse, dropout_prob, &pieces, &so);
  if (!s.ok()) return finish(s, err, errlen);
  pack_strings(pieces, blob, piece_off);
  *n_pieces = pieces.size();
  *sent_off = (uint64_t *)to_malloc(so);
  return 0;
}

int yttm_encode_device(yttm_encoder *h, const void *d_bytes, const void *d_offsets, uint64_t n_sent, uint64_t total_bytes,
                       uint64_t max_sentence_bytes, int bos, int eos, int reverse, double dropout_prob, uint64_t *n_ids, double *kernel_ms,
                       char *err, int errlen) {
  unsigned long long n = 0;
  Status s = h->enc->encode_device(d_bytes, d_offsets, n_sent, total_bytes, max_sentence_bytes, bos, eos, reverse, dropout_prob, &n, kernel_ms);
  if (n_ids) *n_ids = n;
  return finish(s, err, errlen);
}
int yttm_encode_fetch(yttm_encoder *h, int32_t *ids, uint64_t *out_offsets, uint64_t n_sent, char *err, int errlen) {
  return finish(h->enc->fetch_device_result(ids, (unsigned long long *)out_offsets, n_sent), err, errlen);
}

int yttm_encoder_set_cache(yttm_encoder *h, int mode, uint64_t min_bytes) {
  h->enc->set_cache(mode, min_bytes);
  return 0;
}
uint64_t yttm_encode_cache_words(yttm_encoder *h) { return h->enc->cache_words(); }

int yttm_id_to_subword(yttm_encoder *h, int id, char **subword, char *err, int errlen) {
  std::string s;
  Status st = h->enc->id_to_subword(id, &s);
  if (!st.ok()) return finish(st, err, errlen);
  *subword = (char *)malloc(s.size() + 1);
  memcpy(*subword, s.c_str(), s.size() + 1);
  return 0;
}
int yttm_subword_to_id(yttm_encoder *h, const char *token) { return h->enc->subword_to_id(token); }

int yttm_decode(yttm_encoder *h, const int32_t *ids, const uint64_t *offsets, uint64_t n_sent, const int32_t *ignore_ids, uint64_t n_ignore,
                char **blob, uint64_t **out_offsets, char *err, int errlen) {
  std::unordered_set<int> ign(ignore_ids, ignore_ids + n_ignore);
  std::vector<std::string> out;
  for (uint64_t i = 0; i < n_sent; i++) {
    std::vector<int> v(ids + offsets[i], ids + offsets[i + 1]);
    std::string s;
    Status st = h->enc->decode(v, &s, &ign);
    if (!st.ok()) return finish(st, err, errlen);
    out.push_back(std::move(s));
  }
  pack_strings(out, blob, out_offsets);
  return 0;
}
int yttm_vocab_size(yttm_encoder *h) { return h->enc->vocab_size(); }
int yttm_vocabulary(yttm_encoder *h, char **blob, uint64_t **offsets, uint64_t *n) {
  std::vector<std::string> v = h->enc->vocabulary();
  pack_strings(v, blob, offsets);
  *n = v.size();
  return 0;
}
int yttm_encode_cli(yttm_encoder *h, const char *output_type, int stream, int bos, int eos, int reverse, double dropout_prob, int in_fd, int out_fd,
                    char *err, int errlen) {
  return finish(h->enc->encode_cli(output_type, stream != 0, bos != 0, eos != 0, reverse != 0, dropout_prob, in_fd, out_fd), err, errlen);
}
int yttm_decode_cli(yttm_encoder *h, const int32_t *ignore_ids, uint64_t n_ignore, int in_fd, int out_fd, char *err, int errlen) {
  std::unordered_set<int> ign;
  for (uint64_t i = 0; i < n_ignore; i++) ign.insert(ignore_ids[i]);
  return finish(h->enc->decode_cli(&ign, in_fd, out_fd), err, errlen);
}
int yttm_vocab_cli(yttm_encoder *h, int verbose, int out_fd, char *err, int errlen) {
  return finish(h->enc->vocab_cli(verbose != 0, out_fd), err, errlen);
}

unsigned long long yttm_ids_fnv1a64(const int32_t *ids, const uint64_t *offsets, uint64_t n_sent) {
  unsigned long long h = 1469598103934665603ull;
  auto mix4 = [&](uint32_t v) {
    for (int k = 0; k < 4; k++) { h ^= (v >> (8 * k)) & 0xffu; h *= 1099511628211ull; }
  };
  for (uint64_t i = 0; i < n_sent; i++) {
    mix4((uint32_t)(offsets[i + 1] - offsets[i]));
    for (uint64_t k = offsets[i]; k < offsets[i + 1]; k++) mix4((uint32_t)ids[k]);
  }
  return h;
}

void yttm_free(void *p) { free(p); }

int yttm_device_info(int device, char *buf, int buflen) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= device) {
    snprintf(buf, (size_t)buflen, "no HIP device %d visible (%s)", device, e != hipSuccess ? hipGetErrorString(e) : "device count too small");
    return 1;
  }
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, device) != hipSuccess) { snprintf(buf, (size_t)buflen, "hipGetDeviceProperties failed"); return 1; }
  snprintf(buf, (size_t)buflen, "%s %s CUs=%d HBM=%.1fGB", p.gcnArchName, p.name, p.multiProcessorCount, (double)p.totalGlobalMem / 1e9);
  return 0;
}

// ------------------------------------------------------------------------------------------------- inner ABI
static thread_local std::string g_last_error;
const char *yttm_gpu_last_error(void) { return g_last_error.c_str(); }

struct yttm_ctx {
  GpuCtx *g;
};

#define GUARD(...)                                   \
  try {                                              \
    __VA_ARGS__;                                     \
    return 0;                                        \
  } catch (const GpuError &e) {                      \
    g_last_error = e.msg;                            \
    return 2;                                        \
  } catch (const std::exception &e) {                \
    g_last_error = e.what();                         \
    return 2;                                        \
  }

// the same with the context's own snapshot of the YTTM_* hooks bound to the calling thread (yttm_config.h CfgBind)
#define GUARD_CTX(...) GUARD({ const CfgBind _bind(c && c->g ? c->g->config_ptr() : nullptr); __VA_ARGS__; })

int yttm_gpu_ctx_create(int device, yttm_ctx **out) {
  *out = nullptr;
  GUARD({ *out = new yttm_ctx{new GpuCtx(device)}; })
}
void yttm_gpu_ctx_destroy(yttm_ctx *c) {
  if (!c) return;
  delete c->g;
  delete c;
}
int yttm_gpu_ctx_set_comm(yttm_ctx *c, void *comm) { GUARD_CTX(c->g->set_comm((Comm *)comm)) }
int yttm_gpu_upload_corpus(yttm_ctx *c, const uint8_t *utf8, uint64_t n) { GUARD_CTX(c->g->upload_corpus(utf8, n)) }
int yttm_gpu_attach_corpus(yttm_ctx *c, const void *p, uint64_t n) { GUARD_CTX(c->g->attach_corpus(p, n)) }

int yttm_gpu_char_hist(yttm_ctx *c, uint32_t *cps, uint64_t *cnts, uint32_t *n_inout, uint64_t *n_codepoints) {
  GUARD_CTX({
    std::vector<uint32_t> a;
    std::vector<unsigned long long> b;
    unsigned long long steps = 0;
    c->g->char_hist(a, b, steps);
    if (a.size() > *n_inout) throw GpuError{"char_hist: output capacity too small"};
    for (size_t i = 0; i < a.size(); i++) { cps[i] = a[i]; cnts[i] = b[i]; }
    *n_inout = (uint32_t)a.size();
    *n_codepoints = steps;
  })
}
int yttm_gpu_build_word_table(yttm_ctx *c, const uint32_t *cp, const uint32_t *id, uint32_t n_alpha, uint32_t space_id, uint32_t n_ids_cap,
                              uint64_t *n_unique, uint64_t *n_tokens) {
  GUARD_CTX({
    c->g->build_word_table(cp, id, n_alpha, space_id, n_ids_cap);
    *n_unique = c->g->n_unique;
    *n_tokens = c->g->n_tokens0;
  })
}
int yttm_gpu_download_word_table(yttm_ctx *c, uint32_t *tok, uint64_t *off, uint32_t *cnt, uint64_t *n_tokens_now) {
  GUARD_CTX({
    std::vector<uint32_t> t, w;
    std::vector<unsigned long long> o;
    c->g->download_word_table(t, o, w);
    memcpy(tok, t.data(), t.size() * 4);
    for (size_t i = 0; i < o.size(); i++) off[i] = o[i];
    memcpy(cnt, w.data(), w.size() * 4);
    *n_tokens_now = t.size();
  })
}
int yttm_gpu_pair_count(yttm_ctx *c, uint64_t *n_pairs) {
  GUARD_CTX({
    c->g->pair_count();
    *n_pairs = c->g->n_keys_host;
  })
}
int yttm_gpu_download_pairs(yttm_ctx *c, uint64_t *pairs, uint64_t *counts, uint64_t *n_inout) {
  GUARD_CTX({
    std::vector<unsigned long long> k, v;
    c->g->download_pairs(k, v);
    if (k.size() > *n_inout) throw GpuError{"download_pairs: output capacity too small"};
    for (size_t i = 0; i < k.size(); i++) { pairs[i] = k[i]; counts[i] = v[i]; }
    *n_inout = k.size();
  })
}
int yttm_gpu_merge_apply(yttm_ctx *c, const uint32_t *xyz, uint32_t k) { GUARD_CTX(c->g->merge_apply(xyz, k, nullptr)) }
int yttm_gpu_pair_query(yttm_ctx *c, const uint64_t *pairs, uint32_t n, uint64_t *counts) {
  GUARD_CTX(c->g->pair_query((const unsigned long long *)pairs, n, (unsigned long long *)counts))
}
int yttm_gpu_candidates(yttm_ctx *c, uint64_t tau_cnt, uint32_t tau_mx, uint64_t *pairs, uint64_t *counts, uint32_t *n_inout) {
  GUARD_CTX({
    std::vector<CandRec> out;
    uint32_t n = c->g->candidates(tau_cnt, tau_mx, out, nullptr);
    uint32_t take = std::min<uint32_t>((uint32_t)out.size(), *n_inout);
    for (uint32_t i = 0; i < take; i++) { pairs[i] = out[i].key; counts[i] = out[i].cnt; }
    *n_inout = n;
  })
}

int yttm_gpu_k4_measure(yttm_ctx *c, int on, uint64_t out[6]) {
  GUARD_CTX({
    c->g->instrument = on != 0;
    if (out) {
      c->g->resolve_timers();
      out[0] = c->g->merge_sites; out[1] = c->g->touched_tiles; out[2] = c->g->touched_tile_tokens;
      out[3] = c->g->touched_words; out[4] = c->g->touched_word_tokens; out[5] = c->g->merge_rounds;
    }
  })
}

void yttm_release_device_memory(void) { yttm::release_device_memory(); }
const char *yttm_config_table(void) { return yttm::config_table_markdown(); }

}  // extern "C"
