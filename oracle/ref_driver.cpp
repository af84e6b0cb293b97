// oracle/ref_driver.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Thin command-line driver around the *unmodified* reference sources
// (/root/reference/youtokentome/cpp/{bpe,utf8,utils}.cpp), compiled where they
// lie by oracle/Makefile into oracle/_ref/yttm_ref_{det,prod}.  It exposes the
// two entry points of the hot path exactly as yttm.pyx binds them:
//   vkcom::train_bpe                      (bpe.h:19,  bpe.cpp:1368)
//   vkcom::BaseEncoder::encode_as_ids     (bpe.h:37,  bpe.cpp:1740)
//   vkcom::BaseEncoder::encode_as_subwords(bpe.h:41,  bpe.cpp:1757)
// so that tests and bench.py's cpu_baseline leg can (a) pin oracle/bpe_oracle.c
// against the real reference and (b) time the reference on the host cores.
//
// Usage:
//   yttm_ref train  <corpus> <model> <vocab> <coverage> <n_threads> <pad> <unk> <bos> <eos>
//   yttm_ref encode <model> <lines.txt> <out.txt|-> <n_threads> <bos> <eos> <reverse> <dropout> [subword]
//   yttm_ref encode_bench <model> <lines.txt> <n_threads> <dropout> [max_lines]
//   yttm_ref encode_hist <model> <lines.txt> <n_threads> <dropout> <max_lines> <out.json>
//   yttm_ref decode <model> <ids.txt> <out.txt>
//   yttm_ref vocab  <model> <out.txt>
// All timing lines go to stdout as one JSON object; reference chatter stays on stderr.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <unordered_set>
#include <vector>

#include "bpe.h"

using clk = std::chrono::steady_clock;
static double secs(clk::time_point a, clk::time_point b) {
  return std::chrono::duration<double>(b - a).count();
}

static std::vector<std::string> read_lines(const char *path, long max_lines = -1) {
  std::vector<std::string> lines;
  std::ifstream fin(path, std::ios::in | std::ios::binary);
  if (!fin) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
  std::string s;
  while ((max_lines < 0 || (long)lines.size() < max_lines) && std::getline(fin, s)) lines.push_back(s);
  return lines;
}

int main(int argc, char **argv) {
  if (argc < 2) { fprintf(stderr, "usage: see header of oracle/ref_driver.cpp\n"); return 2; }
  std::string cmd = argv[1];
  if (cmd == "train") {
    if (argc != 11) return 2;
    vkcom::BpeConfig cfg;
    cfg.character_coverage = atof(argv[5]);
    cfg.n_threads = atoi(argv[6]);
    cfg.special_tokens.pad_id = atoi(argv[7]);
    cfg.special_tokens.unk_id = atoi(argv[8]);
    cfg.special_tokens.bos_id = atoi(argv[9]);
    cfg.special_tokens.eos_id = atoi(argv[10]);
    auto t0 = clk::now();
    vkcom::Status st = vkcom::train_bpe(argv[2], argv[3], atoi(argv[4]), cfg);
    auto t1 = clk::now();
    if (!st.ok()) {
      printf("{\"ok\": false, \"message\": \"%s\"}\n", st.error_message().c_str());
      return 1;
    }
    printf("{\"ok\": true, \"train_seconds\": %.6f}\n", secs(t0, t1));
    return 0;
  }
  if (cmd == "encode") {
    if (argc < 10) return 2;
    vkcom::Status st;
    vkcom::BaseEncoder enc(argv[2], atoi(argv[5]), &st);
    if (!st.ok()) { printf("{\"ok\": false, \"message\": \"%s\"}\n", st.error_message().c_str()); return 1; }
    auto lines = read_lines(argv[3]);
    bool bos = atoi(argv[6]), eos = atoi(argv[7]), rev = atoi(argv[8]);
    double dropout = atof(argv[9]);
    bool subword = argc > 10 && std::string(argv[10]) == "subword";
    FILE *out = std::string(argv[4]) == "-" ? stdout : fopen(argv[4], "wb");
    if (subword) {
      std::vector<std::vector<std::string>> res;
      st = enc.encode_as_subwords(lines, &res, bos, eos, rev, dropout);
      if (!st.ok()) { printf("{\"ok\": false, \"message\": \"%s\"}\n", st.error_message().c_str()); return 1; }
      for (auto &s : res) { for (auto &p : s) { fputs(p.c_str(), out); fputc(' ', out); } fputc('\n', out); }
    } else {
      std::vector<std::vector<int>> res;
      st = enc.encode_as_ids(lines, &res, bos, eos, rev, dropout);
      if (!st.ok()) { printf("{\"ok\": false, \"message\": \"%s\"}\n", st.error_message().c_str()); return 1; }
      for (auto &s : res) { for (int id : s) fprintf(out, "%d ", id); fputc('\n', out); }
    }
    if (out != stdout) fclose(out);
    return 0;
  }
  if (cmd == "encode_bench") {
    if (argc < 6) return 2;
    vkcom::Status st;
    vkcom::BaseEncoder enc(argv[2], atoi(argv[4]), &st);
    if (!st.ok()) { printf("{\"ok\": false, \"message\": \"%s\"}\n", st.error_message().c_str()); return 1; }
    long max_lines = argc > 6 ? atol(argv[6]) : -1;
    auto lines = read_lines(argv[3], max_lines);
    std::vector<std::vector<int>> res;
    auto t0 = clk::now();
    st = enc.encode_as_ids(lines, &res, false, false, false, atof(argv[5]));
    auto t1 = clk::now();
    if (!st.ok()) { printf("{\"ok\": false, \"message\": \"%s\"}\n", st.error_message().c_str()); return 1; }
    uint64_t h = 1469598103934665603ull, n_ids = 0;  // FNV-1a-64 over the int32 id stream, sentence lengths folded in
    for (auto &s : res) {
      uint32_t len = (uint32_t)s.size();
      for (int k = 0; k < 4; k++) { h ^= (len >> (8 * k)) & 0xff; h *= 1099511628211ull; }
      for (int id : s) {
        uint32_t v = (uint32_t)id;
        for (int k = 0; k < 4; k++) { h ^= (v >> (8 * k)) & 0xff; h *= 1099511628211ull; }
      }
      n_ids += s.size();
    }
    printf("{\"ok\": true, \"sentences\": %zu, \"ids\": %llu, \"encode_seconds\": %.6f, \"fnv1a64\": \"%016llx\"}\n",
           lines.size(), (unsigned long long)n_ids, secs(t0, t1), (unsigned long long)h);
    return 0;
  }
  if (cmd == "encode_hist") {  // <model> <lines.txt> <n_threads> <dropout> <max_lines> <out.json>: sentence-length and unigram-id histograms
    if (argc != 8) return 2;
    vkcom::Status st;
    vkcom::BaseEncoder enc(argv[2], atoi(argv[4]), &st);
    if (!st.ok()) { printf("{\"ok\": false, \"message\": \"%s\"}\n", st.error_message().c_str()); return 1; }
    auto lines = read_lines(argv[3], atol(argv[6]));
    std::vector<std::vector<int>> res;
    auto t0 = clk::now();
    st = enc.encode_as_ids(lines, &res, false, false, false, atof(argv[5]));
    auto t1 = clk::now();
    if (!st.ok()) { printf("{\"ok\": false, \"message\": \"%s\"}\n", st.error_message().c_str()); return 1; }
    std::vector<unsigned long long> len_hist, id_hist((size_t)enc.vocab_size(), 0);
    unsigned long long n_ids = 0;
    for (auto &sv : res) {
      if (sv.size() >= len_hist.size()) len_hist.resize(sv.size() + 1, 0);
      len_hist[sv.size()]++;
      for (int id : sv) id_hist[(size_t)id]++;
      n_ids += sv.size();
    }
    FILE *out = fopen(argv[7], "wb");
    if (!out) return 2;
    fprintf(out, "{\"sentences\": %zu, \"ids\": %llu, \"len_hist\": [", lines.size(), n_ids);
    for (size_t i = 0; i < len_hist.size(); i++) fprintf(out, "%s%llu", i ? "," : "", len_hist[i]);
    fprintf(out, "], \"id_hist\": [");
    for (size_t i = 0; i < id_hist.size(); i++) fprintf(out, "%s%llu", i ? "," : "", id_hist[i]);
    fprintf(out, "]}\n");
    fclose(out);
    printf("{\"ok\": true, \"sentences\": %zu, \"ids\": %llu, \"encode_seconds\": %.6f}\n", lines.size(), n_ids, secs(t0, t1));
    return 0;
  }
  if (cmd == "decode") {
    if (argc != 5) return 2;
    vkcom::Status st;
    vkcom::BaseEncoder enc(argv[2], 1, &st);
    if (!st.ok()) return 1;
    auto lines = read_lines(argv[3]);
    std::vector<std::string> res;
    std::unordered_set<int> ignore;
    st = enc.decode(lines, &res, &ignore);
    if (!st.ok()) { printf("{\"ok\": false, \"message\": \"%s\"}\n", st.error_message().c_str()); return 1; }
    FILE *out = fopen(argv[4], "wb");
    for (auto &s : res) { fputs(s.c_str(), out); fputc('\n', out); }
    fclose(out);
    return 0;
  }
  if (cmd == "vocab") {
    if (argc != 4) return 2;
    vkcom::Status st;
    vkcom::BaseEncoder enc(argv[2], 1, &st);
    if (!st.ok()) return 1;
    FILE *out = fopen(argv[3], "wb");
    for (auto &s : enc.vocabulary()) { fputs(s.c_str(), out); fputc('\n', out); }
    fclose(out);
    return 0;
  }
  fprintf(stderr, "unknown command %s\n", cmd.c_str());
  return 2;
}
