// host_cli.cpp -- the streaming loops of the `yttm` command line: BaseEncoder::encode_cli / decode_cli / vocab_cli
// (bpe.h:66-71, bpe.cpp:1896-2028; helpers utils.cpp:103-111 read_lines_from_stdin, utils.h:92-103 write_to_stdout).
//
// encode (batch mode) is a three-stage pipeline instead of the reference's read-all / encode / print-all per batch:
//   reader thread   stdin -> batches of >= 10 MiB of line bytes (the reference's batch_limit), packed as bytes + offsets
//   two workers     each owns one encoder lane (stream + buffers): H2D, K5, D2H for its batch, then formats the ids (or
//                   the subword strings) into one output buffer -- the copies and kernels of one batch overlap the
//                   formatting of the other
//   writer          the calling thread: writes the buffers in batch order, prints the progress line the reference prints
// --stream keeps the reference's behaviour: one line in, one line out, flushed.
// Input and output are file descriptors (the Python module passes 0 and 1) so that tests can drive the loops with pipes.
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <poll.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <sstream>
#include <thread>

#include "host_core.h"
#include "yttm_config.h"

namespace yttm {

namespace {

// bpe.cpp:1976 batch_limit.  YTTM_CLI_BATCH_BYTES (tuning / test hook) overrides it: batching never changes the output.
unsigned long long batch_limit() {
  const unsigned long long n = cfg()->cli_batch_bytes.u;  // (the snapshot the encoder's creation took)
  return n ? n : 10ull * 1024 * 1024;
}

// std::getline over a file descriptor: lines end at '\n'; a last line without one still counts; "" after a final '\n' does not
class LineReader {
 public:
  explicit LineReader(int fd) : fd_(fd), buf_(1u << 20) {}
  // (the batch loop's reader thread: a set flag ends the input -- also while it waits for a slow or interactive stdin)
  void stop_when(const std::atomic<bool> *flag) { stop_ = flag; }
  // appends the next line (without its '\n') to out; false at end of input
  bool next(std::string *out) {
    bool any = false;
    for (;;) {
      if (pos_ == len_) {
        if (eof_) return any;
        if (stop_) {  // wait for input in slices, so that the flag is seen while nothing arrives
          for (;;) {
            if (stop_->load(std::memory_order_relaxed)) { eof_ = true; return any; }
            struct pollfd pf{fd_, POLLIN, 0};
            const int pr = poll(&pf, 1, 100);
            if (pr != 0) break;  // readable, closed, or an error that read() will report
          }
        }
        ssize_t r;
        do r = read(fd_, buf_.data(), buf_.size()); while (r < 0 && errno == EINTR);
        if (r <= 0) { eof_ = true; return any; }
        pos_ = 0;
        len_ = (size_t)r;
      }
      const char *p = buf_.data() + pos_;
      const char *nl = (const char *)memchr(p, '\n', len_ - pos_);
      if (nl) {
        out->append(p, (size_t)(nl - p));
        pos_ = (size_t)(nl - buf_.data()) + 1;
        return true;
      }
      out->append(p, len_ - pos_);
      pos_ = len_;
      any = true;
    }
  }

 private:
  int fd_;
  std::vector<char> buf_;
  size_t pos_ = 0, len_ = 0;
  bool eof_ = false;
  const std::atomic<bool> *stop_ = nullptr;
};

bool write_all(int fd, const char *p, size_t n) {
  while (n) {
    ssize_t w = write(fd, p, n);
    if (w < 0) {
      if (errno == EINTR) continue;
      return false;
    }
    p += w;
    n -= (size_t)w;
  }
  return true;
}

// "<v> " like `std::cout << token << " "` (utils.h:96), at p (room for 12 chars); returns the end.  Two digits per division: the id lines of
// a GB of text are 10^8 numbers.
inline char *put_int(char *p, int v) {
  static const char D2[201] =
      "00010203040506070809101112131415161718192021222324252627282930313233343536373839404142434445464748495051525354555657585960616263646566676869707172737475767778798081828384858687888990919293949596979899";
  char tmp[12];
  int k = 12;
  unsigned int u = v < 0 ? 0u - (unsigned int)v : (unsigned int)v;
  while (u >= 100) {
    const unsigned int r = u % 100;
    u /= 100;
    tmp[--k] = D2[2 * r + 1];
    tmp[--k] = D2[2 * r];
  }
  if (u >= 10) {
    tmp[--k] = D2[2 * u + 1];
    tmp[--k] = D2[2 * u];
  } else {
    tmp[--k] = (char)('0' + u);
  }
  if (v < 0) tmp[--k] = '-';
  memcpy(p, tmp + k, (size_t)(12 - k));
  p += 12 - k;
  *p++ = ' ';
  return p;
}

struct Batch {
  unsigned long long seq = 0;
  std::string bytes;
  std::vector<unsigned long long> off{0};
  unsigned long long processed = 0;  // line bytes (the reference's `processed`)
  std::string out;
  Status st;
};

// encode one packed batch and format it exactly like write_to_stdout (every token followed by one space, '\n' per sentence)
void encode_and_format(const BaseEncoder &enc, Batch &b, bool subword, bool bos, bool eos, bool reverse, double dropout_prob) {
  const unsigned long long n = b.off.size() - 1;
  b.out.clear();
  if (subword) {
    std::vector<std::string> pieces;
    std::vector<unsigned long long> po;
    b.st = enc.encode_as_subwords((const uint8_t *)b.bytes.data(), b.off.data(), n, bos, eos, reverse, dropout_prob, &pieces, &po);
    if (!b.st.ok()) return;
    for (unsigned long long i = 0; i < n; i++) {
      for (unsigned long long k = po[i]; k < po[i + 1]; k++) { b.out += pieces[k]; b.out += ' '; }
      b.out += '\n';
    }
  } else {
    std::vector<int32_t> ids;
    std::vector<unsigned long long> io;
    b.st = enc.encode_as_ids((const uint8_t *)b.bytes.data(), b.off.data(), n, bos, eos, reverse, dropout_prob, &ids, &io);
    if (!b.st.ok()) return;
    b.out.resize(ids.size() * 12 + n);  // (a number and its space: at most 12 chars)
    char *p = &b.out[0];
    for (unsigned long long i = 0; i < n; i++) {
      for (unsigned long long k = io[i]; k < io[i + 1]; k++) p = put_int(p, ids[k]);
      *p++ = '\n';
    }
    b.out.resize((size_t)(p - b.out.data()));
  }
}

}  // namespace

Status BaseEncoder::encode_cli(const std::string &output_type_str, bool stream, bool bos, bool eos, bool reverse, double dropout_prob, int in_fd,
                               int out_fd) const {
  const bool subword = output_type_str != "id";  // the reference asserts "subword" otherwise (bpe.cpp:1949)
  const CfgBind bind(config());  // (the workers below call encode_as_ids / encode_as_subwords, which bind it on their own threads)
  LineReader in(in_fd);
  if (stream) {  // bpe.cpp:1952-1974
    Batch b;
    std::string line;
    for (;;) {
      line.clear();
      if (!in.next(&line)) break;
      b.bytes = line;
      b.off = {0, (unsigned long long)line.size()};
      encode_and_format(*this, b, subword, bos, eos, reverse, dropout_prob);
      if (!b.st.ok()) return b.st;
      if (!write_all(out_fd, b.out.data(), b.out.size())) return Status(1, "write to the output failed");
    }
    return Status();
  }
  const unsigned long long BATCH_LIMIT = batch_limit();
  fprintf(stderr, "n_threads: %d\n", n_threads);  // bpe.cpp:1979
  std::mutex mu;
  std::condition_variable cv;
  std::deque<std::unique_ptr<Batch>> todo, done;  // reader -> workers -> writer
  bool reader_finished = false, abort = false;
  std::atomic<bool> stop_reader{false};  // set together with abort: the reader may be inside read() or its line loop, not at the condition variable
  in.stop_when(&stop_reader);
  unsigned long long n_batches = 0;
  constexpr size_t MAX_AHEAD = 6;  // batches read but not yet written
  size_t in_flight = 0;

  std::thread reader([&]() {
    // read_lines_from_stdin (utils.cpp:103-111): lines until `processed` reaches the limit; the batch loop of encode_cli ends
    // with the first batch that stays below it (bpe.cpp:2011)
    unsigned long long seq = 0;
    for (;;) {
      auto b = std::make_unique<Batch>();
      b->seq = seq++;
      while (b->processed < BATCH_LIMIT) {  // (a line goes straight behind the batch's bytes so far)
        const size_t before = b->bytes.size();
        if (stop_reader.load(std::memory_order_relaxed) || !in.next(&b->bytes)) break;
        b->processed += b->bytes.size() - before;
        b->off.push_back(b->bytes.size());
      }
      const bool last = b->processed < BATCH_LIMIT;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&]() { return in_flight < MAX_AHEAD || abort; });
        if (abort) return;
        in_flight++;
        n_batches = seq;
        todo.push_back(std::move(b));
        if (last) reader_finished = true;
      }
      cv.notify_all();
      if (last) return;
    }
  });
  auto worker = [&]() {
    for (;;) {
      std::unique_ptr<Batch> b;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&]() { return !todo.empty() || (reader_finished && todo.empty()) || abort; });
        if (abort || todo.empty()) return;
        b = std::move(todo.front());
        todo.pop_front();
      }
      encode_and_format(*this, *b, subword, bos, eos, reverse, dropout_prob);
      {
        std::lock_guard<std::mutex> lk(mu);
        done.push_back(std::move(b));
      }
      cv.notify_all();
    }
  };
  // (four workers on an encoder of two lanes: the encode calls take turns, the formatting of their ids -- most of a worker's time -- runs beside them)
  std::thread w1(worker), w2(worker), w3(worker), w4(worker);

  Status result;
  unsigned long long total_progress = 0, next_seq = 0;
  int chars_remove = 0;
  for (;;) {
    std::unique_ptr<Batch> b;
    {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&]() {
        for (auto &d : done)
          if (d->seq == next_seq) return true;
        return reader_finished && next_seq >= n_batches;
      });
      for (auto it = done.begin(); it != done.end(); ++it)
        if ((*it)->seq == next_seq) { b = std::move(*it); done.erase(it); break; }
      if (!b) break;  // everything written
    }
    if (!b->st.ok()) result = b->st;
    else if (!write_all(out_fd, b->out.data(), b->out.size())) result = Status(1, "write to the output failed");
    if (!result.ok()) {
      { std::lock_guard<std::mutex> lk(mu); abort = true; }
      stop_reader.store(true);
      cv.notify_all();
      break;
    }
    total_progress += b->processed;
    for (int i = 0; i < chars_remove; i++) fputc('\b', stderr);  // bpe.cpp:2003-2010
    const std::string msg = "bytes processed: " + std::to_string(total_progress);
    chars_remove = (int)msg.size();
    fputs(msg.c_str(), stderr);
    next_seq++;
    { std::lock_guard<std::mutex> lk(mu); in_flight--; }
    cv.notify_all();
  }
  reader.join();
  w1.join();
  w2.join();
  w3.join();
  w4.join();
  if (result.ok()) fputc('\n', stderr);
  return result;
}

Status BaseEncoder::decode_cli(const std::unordered_set<int> *ignore_ids, int in_fd, int out_fd) const {  // bpe.cpp:2016-2028
  LineReader in(in_fd);
  std::string line, out;
  for (;;) {
    line.clear();
    if (!in.next(&line)) break;
    // decode(const vector<string>&, ...) (bpe.cpp:1863-1882): ints parsed with operator>> until the first failure
    std::stringstream ss;
    ss << line;
    std::vector<int> ids;
    int x;
    while (ss >> x) ids.push_back(x);
    std::string sentence;
    Status st = decode(ids, &sentence, ignore_ids);
    if (!st.ok()) return st;
    out = sentence;
    out += '\n';
    if (!write_all(out_fd, out.data(), out.size())) return Status(1, "write to the output failed");
  }
  return Status();
}

Status BaseEncoder::vocab_cli(bool verbose, int out_fd) const {  // bpe.cpp:1896-1940
  uint32_t n_tokens = 0;
  for (const auto &entry : recipe) n_tokens = std::max(entry.first, n_tokens);
  const SpecialTokens &sp = bpe_state.special_tokens;
  const int max_special = std::max(std::max(sp.pad_id, sp.unk_id), std::max(sp.bos_id, sp.eos_id));  // SpecialTokens::max_id (utils.cpp:21-30)
  if (max_special >= 0) n_tokens = std::max(n_tokens, (uint32_t)max_special);
  n_tokens++;
  std::unordered_map<uint32_t, std::pair<uint32_t, uint32_t>> reversed_rules;
  if (verbose)
    for (const auto &r : bpe_state.rules) reversed_rules[r.z] = {r.x, r.y};
  std::string out;
  for (uint64_t i = 0; i < n_tokens; i++) {
    std::string tz;
    id_to_subword((int)i, &tz);  // (the reference asserts ok; an id of a hole yields "")
    out += std::to_string(i);
    out += '\t';
    out += tz;
    if (verbose) {
      auto it = reversed_rules.find((uint32_t)i);
      if (it != reversed_rules.end()) {
        std::string tx, ty;
        id_to_subword((int)it->second.first, &tx);
        id_to_subword((int)it->second.second, &ty);
        int used = (int)decode_utf8(tz.data(), tz.data() + tz.size()).size() + 1;
        used += (int)decode_utf8(tx.data(), tx.data() + tx.size()).size() + 1 + (int)decode_utf8(ty.data(), ty.data() + ty.size()).size();
        out += "=" + tx + "+" + ty;
        for (int t = 0; t < std::max(2, 50 - used); t++) out += ' ';
        out += std::to_string(it->second.first) + "+" + std::to_string(it->second.second);
      }
    }
    out += '\n';
  }
  if (!write_all(out_fd, out.data(), out.size())) return Status(1, "write to the output failed");
  return Status();
}

}  // namespace yttm
