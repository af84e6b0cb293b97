// gpu_upload.cpp -- the corpus on its way into HBM: plain and staged uploads (pinned chunks, several host threads), the front end under the upload
// (K1 / K2a / K2b on the parts that have landed), the chunked front end for corpora that do not fit, the multi-GPU gather of the shards.
// (Round 5: cut out of gpu_ctx.cpp, code motion only; gpu_ctx_internal.h says what went where.)
#include "gpu_ctx_internal.h"

namespace yttm {

unsigned long long GpuCtx::peak_device_bytes() const { return pool_peak_bytes(); }

void GpuCtx::upload_corpus(const uint8_t *host, unsigned long long n) {
  drop_spec();
  chunked_ = false;
  if (chunk_bytes_for(n)) {  // (too large for the HBM that is free: in chunks, gpu_ctx.cpp front_end_chunked)
    chunk_src_ = [host](void *dst, unsigned long long off, size_t len) {
      memcpy(dst, host + off, len);
      return true;
    };
    chunk_src_n_ = n;
    front_end_chunked(true);
    return;
  }
  if ((n < (32u << 20) && !(cfg_->fe_overlap_min.set && overlap_front_end(n))) || cfg_->plain_upload.set) {  // small, or (tuning hook) the one-copy path for comparison
    HIP_CHECK(hipSetDevice(device_));
    tl_stream = strm();
    tl_device = device_;
    DFREE(d_text_owned_);
    d_text_owned_ = dmalloc<uint8_t>(n + 64);
    if (n) HIP_CHECK(hipMemcpyAsync(d_text_owned_, host, n, hipMemcpyHostToDevice, strm()));
    sync();
    d_text_ = d_text_owned_;
    n_text_ = n;
    corpus_bytes = n;
    return;
  }
  auto from_memory = [&](void *dst, unsigned long long off, size_t len) {
    memcpy(dst, host + off, len);
    return true;
  };
  if (overlap_front_end(n)) upload_overlapped(n, from_memory);
  else upload_staged(n, from_memory);
}
// ---- staged upload: file (or host memory) -> pinned chunks -> HBM -----------------------------------------------------
// fast_read_file_utf8 (bpe.cpp:67-84) reads the file into one std::string; here the bytes only pass through the host.
// A single hipMemcpy from pageable memory (an mmap of the file, a Python bytes object) is staged by the runtime through
// one internal buffer on one thread; instead IO_THREADS workers each own two pinned chunks, fill one (pread from the page
// cache / memcpy) while the other is on its way over PCIe on the worker's own stream.  The pinned chunks are kept for the
// next call (pinning 128 MB costs tens of milliseconds).
namespace {
constexpr size_t IO_CHUNK_MAX = 64u << 20;
constexpr int IO_MAX_THREADS = 32;
struct IoStage {
  std::mutex mu;
  void *pin[2 * IO_MAX_THREADS] = {nullptr};
  size_t pin_bytes[2 * IO_MAX_THREADS] = {0};
  // the workers' copy streams and events are kept as well (creating and destroying a stream and two events per worker and call was
  // a millisecond of every upload, serialised in the runtime); they belong to device `dev`
  hipStream_t cs[IO_MAX_THREADS] = {nullptr};
  hipEvent_t ev[2 * IO_MAX_THREADS] = {nullptr};
  int dev = -1;
  bool busy = false;
} g_io_dir[2];  // [0] towards the device, [1] towards the host: one transfer each way at a time goes through the chunks (the encoder's pipeline)
}  // namespace

void release_io_stage() {
  for (IoStage &g_io : g_io_dir) {
    std::lock_guard<std::mutex> g(g_io.mu);
    if (g_io.busy) continue;
    for (int i = 0; i < 2 * IO_MAX_THREADS; i++) {
      if (g_io.pin[i]) (void)hipHostFree(g_io.pin[i]);
      g_io.pin[i] = nullptr;
      g_io.pin_bytes[i] = 0;
    }
    for (hipEvent_t &e : g_io.ev) {
      if (e) (void)hipEventDestroy(e);
      e = nullptr;
    }
    for (hipStream_t &c : g_io.cs) {
      if (c) (void)hipStreamDestroy(c);
      c = nullptr;
    }
    g_io.dev = -1;
  }
}

void GpuCtx::upload_staged(unsigned long long n, const std::function<bool(void *dst, unsigned long long off, size_t len)> &fill) {
  HIP_CHECK(hipSetDevice(device_));
  tl_stream = strm();
  tl_device = device_;
  DFREE(d_text_owned_);
  d_text_owned_ = dmalloc<uint8_t>(n + 64);
  d_text_ = d_text_owned_;
  n_text_ = n;
  corpus_bytes = n;
  if (!n) return;
  staged_transfer(device_, d_text_owned_, n, true, fill);
}

// n bytes between HBM and the host through the workers' pinned chunks.  to_device: host_side(chunk, off, len) FILLS the pinned chunk with
// bytes [off, off + len) (pread, memcpy) before it goes up; else it DRAINS the chunk that has come down (memcpy to where the caller wants
// the bytes -- several workers at once, which also spreads the page faults of a freshly allocated destination).  Returns when every byte
// has arrived.  Throws GpuError.
size_t staged_chunk_bytes() {
  const std::shared_ptr<const Config> C = cfg();
  const size_t mb = std::min<size_t>(std::max<size_t>((size_t)C->io_chunk_mb.u, 1), IO_CHUNK_MAX >> 20);
  size_t c = mb << 20;
  if (const size_t kb = (size_t)C->io_chunk_kb.u) c = std::min<size_t>(kb << 10, IO_CHUNK_MAX);  // (tests: many chunks of a small batch)
  return c;
}
void staged_transfer(int device, uint8_t *d_ptr, unsigned long long n, bool to_device,
                     const std::function<bool(void *chunk, unsigned long long off, size_t len)> &host_side,
                     const std::function<void(unsigned long long off, size_t len)> &arrived, size_t chunk_bytes) {
  if (!n) return;
  HIP_CHECK(hipSetDevice(device));
  IoStage &g_io = g_io_dir[to_device ? 0 : 1];
  const std::shared_ptr<const Config> C = cfg();
  const size_t IO_CHUNK = chunk_bytes && !C->io_chunk_mb.set && !C->io_chunk_kb.set ? std::min(chunk_bytes, IO_CHUNK_MAX) : staged_chunk_bytes();
  const size_t n_chunks = (size_t)((n + IO_CHUNK - 1) / IO_CHUNK);
  int n_threads = (int)C->io_threads.u;
  // (default 4: one thread preads 40 GB/s out of the page cache on the MI355X box, the link takes 55; eight workers measured SLOWER than three
  // or four -- 34 - 42 ms per GB against 24 -- sixteen much slower: they queue up in the runtime)
  if (n_threads <= 0) n_threads = (int)std::max(1u, std::min(std::thread::hardware_concurrency(), 4u));
  n_threads = (int)std::min<size_t>((size_t)std::min(n_threads, IO_MAX_THREADS), n_chunks);
  bool mine = false;
  {
    std::lock_guard<std::mutex> g(g_io.mu);
    if (!g_io.busy) { g_io.busy = true; mine = true; }
  }
  struct BusyGuard {  // (whatever leaves this function -- any exception -- gives the chunks back)
    IoStage &io;
    bool held;
    ~BusyGuard() {
      if (!held) return;
      std::lock_guard<std::mutex> g(io.mu);
      io.busy = false;
    }
  } busy_guard{g_io, mine};
  if (mine && g_io.dev != device) {  // (the cached streams and events are another device's)
    for (hipEvent_t &e : g_io.ev) {
      if (e) (void)hipEventDestroy(e);
      e = nullptr;
    }
    for (hipStream_t &c : g_io.cs) {
      if (c) (void)hipStreamDestroy(c);
      c = nullptr;
    }
    g_io.dev = device;
  }
  if (!mine) {  // another thread of this process is moving bytes through the shared chunks: plain copies for this one
    std::vector<uint8_t> tmp(IO_CHUNK);
    for (size_t c = 0; c < n_chunks; c++) {
      const unsigned long long off = (unsigned long long)c * IO_CHUNK;
      const size_t len = (size_t)std::min<unsigned long long>(IO_CHUNK, n - off);
      if (to_device) {
        if (!host_side(tmp.data(), off, len)) throw GpuError{"corpus read failed"};
        HIP_CHECK(hipMemcpy(d_ptr + off, tmp.data(), len, hipMemcpyHostToDevice));
        if (arrived) arrived(off, len);
      } else {
        HIP_CHECK(hipMemcpy(tmp.data(), d_ptr + off, len, hipMemcpyDeviceToHost));
        if (!host_side(tmp.data(), off, len)) throw GpuError{"copy to the host failed"};
      }
    }
    return;
  }
  std::atomic<size_t> next{0};
  std::atomic<int> failed{0};
  std::string first_error;
  std::mutex err_mu;
  auto worker = [&](int w) {
    try {
      HIP_CHECK(hipSetDevice(device));
      if (!g_io.cs[w]) HIP_CHECK(hipStreamCreateWithFlags(&g_io.cs[w], hipStreamNonBlocking));
      hipStream_t cs = g_io.cs[w];
      hipEvent_t *ev = &g_io.ev[2 * w];
      bool used[2] = {false, false};
      unsigned long long held_off[2] = {0, 0};  // (down: the bytes a buffer is receiving)
      size_t held_len[2] = {0, 0};
      for (int k = 0; k < 2; k++) {
        if (!ev[k]) HIP_CHECK(hipEventCreate(&ev[k]));
        if (g_io.pin_bytes[2 * w + k] < IO_CHUNK) {  // (pinning is slow: the chunks are kept, and only as large as they are used)
          if (g_io.pin[2 * w + k]) (void)hipHostFree(g_io.pin[2 * w + k]);
          g_io.pin[2 * w + k] = nullptr;
          g_io.pin_bytes[2 * w + k] = 0;
          HIP_CHECK(hipHostMalloc(&g_io.pin[2 * w + k], IO_CHUNK, hipHostMallocDefault));
          g_io.pin_bytes[2 * w + k] = IO_CHUNK;
        }
      }
      auto drain = [&](int k) {
        if (!used[k]) return;
        HIP_CHECK(hipEventSynchronize(ev[k]));
        used[k] = false;
        if (!to_device && !host_side(g_io.pin[2 * w + k], held_off[k], held_len[k])) throw GpuError{"copy to the host failed"};
        if (to_device && arrived) arrived(held_off[k], held_len[k]);
      };
      for (int k = 0;; k ^= 1) {
        const size_t c = next.fetch_add(1);
        if (c >= n_chunks || failed.load()) break;
        const unsigned long long off = (unsigned long long)c * IO_CHUNK;
        const size_t len = (size_t)std::min<unsigned long long>(IO_CHUNK, n - off);
        drain(k);  // up: the buffer's previous copy has left it; down: its bytes have arrived and are handed over
        if (to_device) {
          if (!host_side(g_io.pin[2 * w + k], off, len)) throw GpuError{"corpus read failed"};
          HIP_CHECK(hipMemcpyAsync(d_ptr + off, g_io.pin[2 * w + k], len, hipMemcpyHostToDevice, cs));
        } else {
          HIP_CHECK(hipMemcpyAsync(g_io.pin[2 * w + k], d_ptr + off, len, hipMemcpyDeviceToHost, cs));
        }
        held_off[k] = off;
        held_len[k] = len;
        HIP_CHECK(hipEventRecord(ev[k], cs));
        used[k] = true;
        if (to_device && arrived) drain(k ^ 1);  // (somebody waits for the bytes: say that the previous chunk has landed now, not a fill later)
      }
      drain(0);
      drain(1);
      HIP_CHECK(hipStreamSynchronize(cs));
    } catch (const GpuError &e) {
      failed.store(1);
      std::lock_guard<std::mutex> g(err_mu);
      if (first_error.empty()) first_error = e.msg;
    } catch (const std::exception &e) {  // (bad_alloc in a callback, ...: reported like a GPU error, never std::terminate in a worker)
      failed.store(1);
      std::lock_guard<std::mutex> g(err_mu);
      if (first_error.empty()) first_error = std::string("staged transfer: ") + e.what();
    }
    // A worker that gave up may have copies queued on its stream that still read / write its pinned chunks and d_ptr: they must be over
    // before the caller frees the device buffer or the next transfer reuses the chunks.
    if (failed.load() && g_io.cs[w]) (void)hipStreamSynchronize(g_io.cs[w]);
  };
  struct Joiner {  // (a throwing emplace_back / worker(0) must not destroy joinable threads)
    std::vector<std::thread> th;
    ~Joiner() {
      for (auto &t : th)
        if (t.joinable()) t.join();
    }
  } joiner;
  try {
    for (int w = 1; w < n_threads; w++) joiner.th.emplace_back(worker, w);
  } catch (const std::exception &e) {  // (no more threads to be had: the ones that started, and this one, do the work)
    (void)e;
  }
  worker(0);
  for (auto &t : joiner.th) t.join();
  if (failed.load()) throw GpuError{first_error};
}

void GpuCtx::drop_spec() {
  if (spec_.ht) DFREE(spec_.ht);
  spec_ = FrontSpec();
}

// The front end under the upload (single GPU).  A GB of file needs 18 ms on the link, and the device idles through them; K1, K2a and K2b of
// the same GB are 10 ms of work that needs nothing but the bytes: K1 and K2a by construction, K2b -- the dedup -- if every word is compared
// by its CODE POINTS instead of its token ids, which is the same partition of the segments into words whenever the alphabet keeps every
// char of the text (coverage 1, the default): the ids are then an injective renaming of the chars.  So the text is worked on in parts as they
// land -- the workers of staged_transfer report the chunks, a part is ready when every byte up to one scan chunk behind its end is there --
// and build_word_table() takes the finished word table if the alphabet turns out to keep everything, else runs its own K2a / K2b as before.
// (`fill` brings bytes [off, off + len) of the source -- a file's byte range, host memory -- into a pinned chunk.)
// A part's last segment may run on into bytes that have not arrived: it is inserted with the next part that has a segment of its own.
void GpuCtx::upload_overlapped(unsigned long long n, const std::function<bool(void *dst, unsigned long long off, size_t len)> &fill) {
  HIP_CHECK(hipSetDevice(device_));
  tl_stream = strm();
  tl_device = device_;
  drop_spec();
  DFREE(d_text_owned_);
  d_text_owned_ = dmalloc<uint8_t>(n + 64);
  d_text_ = d_text_owned_;
  n_text_ = n;
  corpus_bytes = n;
  // K1's variant from four samples of the SOURCE (char_hist samples the text in HBM, which is not there yet)
  bool wide_chars = false;
  if (n >= (1u << 16)) {
    unsigned int wide = 0;
    uint8_t smp[4096];
    for (int i = 0; i < 4; i++) {
      if (!fill(smp, (n / 4) * (unsigned long long)i, sizeof smp)) throw GpuError{"corpus read failed"};
      for (size_t j = 0; j < sizeof smp; j++) wide += smp[j] >= 0xE0u;
    }
    wide_chars = wide * 100u > 4u * 4096u;
  } else {
    wide_chars = true;
  }
  if (cfg_->k1_wide.set) wide_chars = cfg_->k1_wide.i != 0;
  if (!d_hist_) d_hist_ = dmalloc<unsigned long long>(N_CODEPOINTS);
  HIP_CHECK(hipMemsetAsync(d_hist_, 0, (size_t)N_CODEPOINTS * 8, strm()));
  HIP_CHECK(hipMemsetAsync(d_counters_, 0, 64 * 8, strm()));
  const unsigned long long nch = fe_chunks(n);
  DFREE(d_chunk_segs_);
  d_chunk_segs_ = dmalloc<uint32_t>(nch + 1);
  // the speculative map: a char's id is its code point
  uint32_t *d_cpmap_spec = dmalloc<uint32_t>(N_CODEPOINTS);
  {
    static std::vector<uint32_t> ident;
    static std::once_flag once;
    std::call_once(once, [] {
      ident.resize(N_CODEPOINTS);
      for (uint32_t c = 0; c < N_CODEPOINTS; c++) ident[c] = c;
      const uint32_t spaces[] = {9, 10, 11, 12, 13, 32, 9601};
      for (uint32_t sp : spaces) ident[sp] = CP_SPACE;
    });
    HIP_CHECK(hipMemcpyAsync(d_cpmap_spec, ident.data(), (size_t)N_CODEPOINTS * 4, hipMemcpyHostToDevice, strm()));
  }
  unsigned long long *d_chunk_off = dmalloc<unsigned long long>(nch + 1);
  // ---- the upload, on a thread of its own; what has landed, in order
  const size_t io_chunk = staged_chunk_bytes();
  const size_t n_io = (size_t)((n + io_chunk - 1) / io_chunk);
  std::vector<uint8_t> landed(n_io, 0);
  std::mutex mu;
  std::condition_variable cv;
  size_t next_io = 0;  // chunks [0, next_io) have landed
  bool finished = false, up_failed = false;
  std::string up_error;  // (read after the join)
  const auto t_start = std::chrono::steady_clock::now();
  auto ms_now = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count(); };
  double ms_link = 0;
  //  a failure on this thread -- a front-end kernel, an allocation -- must not wait for the rest of the corpus to be read and
  // copied: the wrapper around `fill` gives up once `stop` is set, which ends staged_transfer; and the thread is joined on every way out.
  // (`fill` is called from several threads at once -- the samples above, then the upload's workers: gpu_ctx.h says so.)
  std::atomic<bool> stop{false};
  const std::function<bool(void *, unsigned long long, size_t)> fill_or_stop = [&](void *dst, unsigned long long off, size_t len) {
    return !stop.load(std::memory_order_relaxed) && fill(dst, off, len);
  };
  std::thread up([&] {
    try {
      staged_transfer(device_, d_text_owned_, n, true, fill_or_stop, [&](unsigned long long off, size_t) {
        std::lock_guard<std::mutex> g(mu);
        landed[(size_t)(off / io_chunk)] = 1;
        bool moved = false;
        while (next_io < n_io && landed[next_io]) { next_io++; moved = true; }
        if (moved) cv.notify_all();
      });
    } catch (const GpuError &e) {
      std::lock_guard<std::mutex> g(mu);
      up_error = e.msg;
      up_failed = true;
    }
    ms_link = ms_now();
    std::lock_guard<std::mutex> g(mu);
    finished = true;
    cv.notify_all();
  });
  struct UpJoin {
    std::thread &t;
    std::atomic<bool> &stop;
    ~UpJoin() {
      if (!t.joinable()) return;
      stop.store(true);
      t.join();
    }
  } up_join{up, stop};
  auto wait_for = [&](unsigned long long bytes) {  // until [0, bytes) has landed (or the upload is over); false: it failed
    std::unique_lock<std::mutex> g(mu);
    cv.wait(g, [&] { return finished || std::min<unsigned long long>(n, (unsigned long long)next_io * io_chunk) >= bytes; });
    return !up_failed;
  };
  // ---- the parts
  const unsigned long long FC = fe_chunk_bytes();
  unsigned long long part = cfg_->fe_part_kb.u << 10;  // (32 MB: the last part is 0.8 ms of work behind the last byte; tests: a few KB)
  part = std::max(FC, part / FC * FC);
  bool spec_on = !cfg_->fe_no_spec.set;
  const unsigned int k2b_blocks = (unsigned int)cfg_->fe_k2b_blocks.u;  // (tuning hook)
  unsigned long long *d_seg = nullptr, seg_cap = 0, base = 0;
  unsigned int *d_status = (unsigned int *)(d_counters_ + 24);
  unsigned long long pending = 0;  // the last segment so far: not inserted yet
  bool have_pending = false;
  std::string fail;
  try {
    for (unsigned long long b0 = 0; b0 < n; b0 += part) {
      const unsigned long long b1 = std::min(n, b0 + part);
      if (!wait_for(std::min(n, b1 + FC))) break;
      const unsigned long long c_lo = b0 / FC, c_hi = fe_chunks(b1);
      t_begin(KT_CHAR_HIST);
      launch_char_hist(d_text_, n, d_hist_, d_counters_, wide_chars, d_chunk_segs_, strm(), c_lo, c_hi);
      t_end(KT_CHAR_HIST, b1 - b0);
      if (!spec_on) continue;
      // the part's segments: where they go (relative to the part's first), how many
      unsigned long long *scan_tmp = dmalloc<unsigned long long>(scan_scratch_blocks(c_hi - c_lo));
      t_begin(KT_SEGS);
      launch_exclusive_scan(d_chunk_segs_ + c_lo, c_hi - c_lo, d_chunk_off + c_lo, scan_tmp, d_counters_ + 16, strm());
      unsigned long long n_p = 0;
      HIP_CHECK(hipMemcpyAsync(&n_p, d_counters_ + 16, 8, hipMemcpyDeviceToHost, strm()));
      sync();
      DFREE(scan_tmp);
      if (b0 == 0) {  // sizes from the first part's density of segments (build_word_table's rules, on an estimate)
        const double est = (double)n_p * ((double)n / (double)(b1 - b0)) * 1.1 + 1024.0;
        seg_cap = (unsigned long long)(est * 1.1);
        d_seg = dmalloc<unsigned long long>(seg_cap);
        spec_.long_segments = n_p == 0 || (b1 - b0) / std::max<unsigned long long>(n_p, 1) >= 16;
        const unsigned long long ns = (unsigned long long)est;
        spec_.ht_cap = !spec_.long_segments && !cfg_->word_table_full.set ? pow2_at_least(std::max<unsigned long long>(ns / 4, 1ull << 16))
                                                                                : pow2_at_least(ns + ns / 2 + 1024);
        spec_.ht = dmalloc<unsigned long long>(3 * spec_.ht_cap);
        launch_word_table_clear(spec_.ht, spec_.ht_cap, strm());
        HIP_CHECK(hipMemsetAsync(d_status, 0, 32, strm()));
      }
      if (base + n_p > seg_cap) {  // denser than the first part promised: no room for the segment starts -- the usual way then
        t_end(KT_SEGS, 0);
        spec_on = false;
        continue;
      }
      launch_seg_write(d_text_, n, d_seg + base, d_chunk_off, strm(), c_lo, c_hi);
      t_end(KT_SEGS, (b1 - b0) + 8 * n_p);
      if (n_p) {
        // every segment that starts in this part but its last -- that one may run on into bytes that have not landed -- and the last of the
        // parts before, which ended in front of this part's first segment
        const unsigned long long from = have_pending ? pending : base, to = base + n_p - 1;
        if (to > from) {
          t_begin(KT_DEDUP);
          launch_insert_words(d_text_, n, d_cpmap_spec, d_seg + from, to - from, spec_.ht, spec_.ht_cap - 1, d_status, strm(), k2b_blocks);
          t_end(KT_DEDUP, (b1 - b0) + 8 * n_p);
        }
        pending = to;
        have_pending = true;
      }
      base += n_p;
    }
  } catch (const GpuError &e) {
    fail = e.msg;
    stop.store(true);  // (the upload ends with its chunk in flight instead of with the file's last byte)
  }
  up.join();
  if (fail.empty() && !up_error.empty()) fail = up_error;
  if (fail.empty()) {
    try {
      if (spec_on && d_seg) {
        if (have_pending) {
          t_begin(KT_DEDUP);
          launch_insert_words(d_text_, n, d_cpmap_spec, d_seg + pending, 1, spec_.ht, spec_.ht_cap - 1, d_status, strm());
          t_end(KT_DEDUP, 0);
        }
        HIP_CHECK(hipMemcpyAsync(spec_.h_status, d_status, 32, hipMemcpyDeviceToHost, strm()));
      }
      sync();
    } catch (const GpuError &e) {
      fail = e.msg;
    }
  }
  DFREE(d_seg);
  DFREE(d_chunk_off);
  DFREE(d_cpmap_spec);
  if (!fail.empty()) {
    drop_spec();
    throw GpuError{fail};
  }
  if (cfg_->trace.set)
    fprintf(stderr, "[yttm] front end under the upload: the last byte landed after %.2f ms, the last part was done after %.2f ms (%llu segments, parts of %llu MB, word table %s)\n",
            ms_link, ms_now(), base, part >> 20, spec_on && spec_.ht ? "made" : "left to build_word_table");
  spec_.hist_done = true;
  spec_.n_segs = base;
  spec_.words_done = spec_on && spec_.ht != nullptr;
  if (!spec_.words_done && spec_.ht) DFREE(spec_.ht);
}

// (the front end under the upload: a text worth the trouble.  Round 5: on every rank of a multi-GPU run too -- K1, K2a and K2b of a rank's byte
// range need nothing from the other ranks; what the ranks exchange -- the char histogram, then the pair counts -- comes after, as before.  The
// rank's word table is taken if the COMMON alphabet keeps every char of the whole text, a sufficient condition that every rank evaluates
// alike; a rank that cannot take it redoes its own dedup, no collective depends on it.)
bool GpuCtx::overlap_front_end(unsigned long long n) const {
  return n >= cfg_->fe_overlap_min.u && !cfg_->fe_no_overlap.set;
}

void GpuCtx::upload_corpus_fd(int fd, unsigned long long lo, unsigned long long n) {
  auto from_file = [&](void *dst, unsigned long long off, size_t len) {
    size_t got = 0;
    while (got < len) {
      const ssize_t r = pread(fd, (char *)dst + got, len - got, (off_t)(lo + off + got));
      if (r <= 0) return false;
      got += (size_t)r;
    }
    return true;
  };
  chunked_ = false;
  if (chunk_bytes_for(n)) {  // (too large for the HBM that is free: in chunks, front_end_chunked below; the descriptor stays open until the training is over)
    drop_spec();
    chunk_src_ = [fd, lo](void *dst, unsigned long long off, size_t len) {
      size_t got = 0;
      while (got < len) {
        const ssize_t r = pread(fd, (char *)dst + got, len - got, (off_t)(lo + off + got));
        if (r <= 0) return false;
        got += (size_t)r;
      }
      return true;
    };
    chunk_src_n_ = n;
    front_end_chunked(true);
    return;
  }
  if (overlap_front_end(n)) {
    upload_overlapped(n, from_file);
    return;
  }
  drop_spec();
  upload_staged(n, [&](void *dst, unsigned long long off, size_t len) {
    size_t got = 0;
    while (got < len) {
      const ssize_t r = pread(fd, (char *)dst + got, len - got, (off_t)(lo + off + got));
      if (r <= 0) return false;
      got += (size_t)r;
    }
    return true;
  });
}

// ---- corpora larger than the HBM left for them (round 5; VERDICT r4 "missing" #2) --------------------------------------------------------
// The reference's limit is host memory (fast_read_file_utf8, bpe.cpp:67-84); here the whole text, its segment starts (8 bytes per word) and
// the word table had to sit in HBM together -- about six bytes per byte of text.  Nothing after the dedup needs the text, only the distinct
// words: so a text that does not fit crosses the device in CHUNKS cut at white space (like the reference's per-thread split, bpe.cpp:864-873).
// One buffer, [chunk region C bytes | 64 spaces | lexicon]: K1 and K2a see a chunk as they see a whole text; K2b inserts its words into the ONE
// word table, `sub` segments per launch with the table grown (rehashed) ahead of a launch that could fill it beyond half; then k2b_relocate
// copies the bytes of every word first seen in this chunk to the lexicon and points its slot there, and the next chunk overwrites the region.
// At the end the "text" the rest of the trainer reads words from -- compaction, token fill -- is the lexicon: the same offsets into the same
// buffer.  first_pass: K1 runs and words are compared by code points (upload_overlapped's speculation: right whenever the alphabet keeps every
// char); else -- coverage dropped chars -- the source is read a second time with the real char map.  Peak HBM is 2 C (the region and the next
// chunk's landing buffer) + the lexicon + the table + 8 bytes per segment of one chunk, whatever the size of the file.
unsigned long long GpuCtx::chunk_bytes_for(unsigned long long n) const {
  unsigned long long c = cfg_->fe_chunk_kb.u ? cfg_->fe_chunk_kb.u << 10 : cfg_->fe_chunk_mb.u << 20;
  if (!c) {
    const unsigned long long free_b = free_device_bytes();
    if (6 * n + (1ull << 30) <= free_b / 4 * 3) return 0;  // text + segment starts + word table + tiles fit at once
    c = std::min<unsigned long long>(std::max<unsigned long long>(free_b / 24, 256ull << 20), 4ull << 30);
  }
  c = std::max<unsigned long long>(c / 4096 * 4096, 4096);
  return n > c ? c : 0;
}

void GpuCtx::front_end_chunked(bool first_pass) {
  HIP_CHECK(hipSetDevice(device_));
  tl_stream = strm();
  tl_device = device_;
  const unsigned long long n = chunk_src_n_;
  const auto &fill = chunk_src_;
  unsigned long long C = (!first_pass && chunk_cap_) ? chunk_cap_ : chunk_bytes_for(n);  // (a second pass: the first one's size -- less memory is free now)
  if (!C) C = std::max<unsigned long long>(n / 4096 * 4096 + 4096, 4096);
  drop_spec();
  DFREE(d_text_owned_);
  chunked_ = true;
  chunk_cap_ = C;
  corpus_bytes = n;
  // ---- where the chunks end: behind the last ASCII white space at or before start + C
  std::vector<unsigned long long> cuts{0};
  {
    std::vector<uint8_t> win(1u << 16);
    while (cuts.back() < n) {
      const unsigned long long b0 = cuts.back();
      unsigned long long b1 = std::min(n, b0 + C);
      if (b1 < n) {
        unsigned long long hi = b1, found = ~0ull;
        while (hi > b0 && found == ~0ull) {
          const unsigned long long lo = hi - std::min<unsigned long long>(hi - b0, win.size());
          if (!fill(win.data(), lo, (size_t)(hi - lo))) throw GpuError{"corpus read failed"};
          for (unsigned long long k = hi - lo; k-- > 0;) {
            const uint8_t b = win[(size_t)k];
            if (b == 32 || (b >= 9 && b <= 13)) { found = lo + k + 1; break; }
          }
          hi = lo;
        }
        if (found == ~0ull || found <= b0) throw GpuError{"a word longer than the front end's chunk (" + std::to_string(C) + " bytes): raise YTTM_FE_CHUNK_MB"};
        b1 = found;
      }
      cuts.push_back(b1);
    }
  }
  const size_t n_chunks = cuts.size() - 1;
  front_end_chunks = n_chunks;
  // ---- buffers
  const unsigned long long GAP = 64, LEX0 = C + GAP;
  lex_cap_ = std::max<unsigned long long>(C / 4, 4096);
  lex_used_ = 0;
  uint8_t *B = dmalloc<uint8_t>(LEX0 + lex_cap_ + 2 * GAP);
  d_text_owned_ = B;
  d_text_ = B;
  const unsigned long long sub = std::min<unsigned long long>(std::max<unsigned long long>(C / 16, 256), 4ull << 20);  // segments per K2b launch
  unsigned long long cap = pow2_at_least(std::max<unsigned long long>(4 * sub, 1024));
  // every block this function holds in a local: freed when it is left by a throw as well (a caller that catches the error and retries -- the C
  // ABI's guarded() -- must not lose up to a chunk of HBM per attempt); `landing` outlives the uploader's future (Landed, declared below)
  unsigned long long *ht = nullptr, *nht = nullptr, *d_seg = nullptr, *d_chunk_off = nullptr, *scan_tmp = nullptr, *scratch_hist = nullptr;
  uint32_t *d_map_spec = nullptr;
  uint8_t *landing = nullptr;
  DevScope scope;
  scope.hold(ht); scope.hold(nht); scope.hold(d_seg); scope.hold(d_chunk_off); scope.hold(scan_tmp); scope.hold(scratch_hist); scope.hold(d_map_spec); scope.hold(landing);
  ht = dmalloc<unsigned long long>(3 * cap);
  launch_word_table_clear(ht, cap, strm());
  unsigned int *d_status = (unsigned int *)(d_counters_ + 24);
  HIP_CHECK(hipMemsetAsync(d_status, 0, 32, strm()));
  unsigned long long *d_cur = d_counters_ + 56;  // [0] the lexicon's end (an offset into B), [1] bytes a relocation will need
  {
    const unsigned long long init[2] = {LEX0, 0};
    HIP_CHECK(hipMemcpyAsync(d_cur, init, 16, hipMemcpyHostToDevice, strm()));
  }
  // K1's variant from four samples of the source, as upload_overlapped does
  bool wide_chars = true;
  if (n >= (1u << 16)) {
    unsigned int wide = 0;
    uint8_t smp[4096];
    for (int i = 0; i < 4; i++) {
      if (!fill(smp, (n / 4) * (unsigned long long)i, sizeof smp)) throw GpuError{"corpus read failed"};
      for (size_t j = 0; j < sizeof smp; j++) wide += smp[j] >= 0xE0u;
    }
    wide_chars = wide * 100u > 4u * 4096u;
  }
  if (cfg_->k1_wide.set) wide_chars = cfg_->k1_wide.i != 0;
  unsigned long long *hist = nullptr, *counters = nullptr;
  if (first_pass) {
    if (!d_hist_) d_hist_ = dmalloc<unsigned long long>(N_CODEPOINTS);
    HIP_CHECK(hipMemsetAsync(d_hist_, 0, (size_t)N_CODEPOINTS * 8, strm()));
    HIP_CHECK(hipMemsetAsync(d_counters_, 0, 24 * 8, strm()));
    hist = d_hist_;
    counters = d_counters_;
  } else {  // (the second pass needs K1 only for the chunks' segment counts: its histogram and counters go to a scratch copy)
    scratch_hist = dmalloc<unsigned long long>(N_CODEPOINTS + 8);
    HIP_CHECK(hipMemsetAsync(scratch_hist, 0, ((size_t)N_CODEPOINTS + 8) * 8, strm()));
    hist = scratch_hist;
    counters = scratch_hist + N_CODEPOINTS;
  }
  // the char map words are compared by: code points (first pass) or the alphabet's ids
  const uint32_t *d_map = d_cpmap_;
  if (first_pass) {
    std::vector<uint32_t> ident(N_CODEPOINTS);
    for (uint32_t c = 0; c < N_CODEPOINTS; c++) ident[c] = c;
    for (uint32_t sp : {9u, 10u, 11u, 12u, 13u, 32u, 9601u}) ident[sp] = CP_SPACE;
    d_map_spec = dmalloc<uint32_t>(N_CODEPOINTS);
    HIP_CHECK(hipMemcpyAsync(d_map_spec, ident.data(), (size_t)N_CODEPOINTS * 4, hipMemcpyHostToDevice, strm()));
    sync();  // (ident goes out of scope)
    d_map = d_map_spec;
  }
  const unsigned long long nch_max = fe_chunks(C) + 2;
  DFREE(d_chunk_segs_);
  d_chunk_segs_ = dmalloc<uint32_t>(nch_max);
  d_chunk_off = dmalloc<unsigned long long>(nch_max);
  scan_tmp = dmalloc<unsigned long long>(scan_scratch_blocks(nch_max));
  unsigned long long segs_total = 0, n_unique_host = 0;
  unsigned int h_status[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const auto t0 = std::chrono::steady_clock::now();
  // The link and the kernels at once: chunk k + 1 crosses the link into a landing buffer of its own while the kernels work on chunk k in
  // the region; a device-to-device copy (C bytes at HBM's rate: 0.3 ms per 512 MB) then moves it over.  Two regions the kernels alternate
  // between would save that copy and cost every offset in the word table a region base; the copy is 1 % of a chunk's upload.
  landing = n_chunks > 1 && !cfg_->fe_chunk_serial.set ? dmalloc<uint8_t>(C) : nullptr;
  auto upload = [&](size_t ck, uint8_t *dst) {
    const unsigned long long b0 = cuts[ck], len = cuts[ck + 1] - b0;
    if (len) staged_transfer(device_, dst, len, true, [&fill, b0](void *d, unsigned long long off, size_t l) { return fill(d, b0 + off, l); });
  };
  std::future<void> in_flight;
  struct Landed {  // (whatever leaves this loop: no upload still writing into a buffer that is being freed)
    std::future<void> &f;
    ~Landed() {
      if (f.valid()) f.wait();
    }
  } landed{in_flight};
  if (landing) in_flight = std::async(std::launch::async, upload, (size_t)0, landing);
  for (size_t ck = 0; ck < n_chunks; ck++) {
    const unsigned long long len = cuts[ck + 1] - cuts[ck];
    // ---- the chunk, then spaces behind it (its last segment ends there if the text does not end with white space)
    if (landing) {
      in_flight.get();  // (rethrows the uploader's error)
      if (len) HIP_CHECK(hipMemcpyAsync(B, landing, (size_t)len, hipMemcpyDeviceToDevice, strm()));
      sync();
      if (ck + 1 < n_chunks) in_flight = std::async(std::launch::async, upload, ck + 1, landing);
    } else {
      upload(ck, B);
    }
    if (!len) continue;
    HIP_CHECK(hipMemsetAsync(B + len, 32, GAP, strm()));
    const unsigned long long nch = fe_chunks(len);
    t_begin(KT_CHAR_HIST);
    launch_char_hist(B, len, hist, counters, wide_chars, d_chunk_segs_, strm());
    t_end(KT_CHAR_HIST, first_pass ? len : 0);
    t_begin(KT_SEGS);
    launch_exclusive_scan(d_chunk_segs_, nch, d_chunk_off, scan_tmp, d_counters_ + 16, strm());
    unsigned long long n_p = 0;
    HIP_CHECK(hipMemcpyAsync(&n_p, d_counters_ + 16, 8, hipMemcpyDeviceToHost, strm()));
    sync();
    d_seg = dmalloc<unsigned long long>(std::max<unsigned long long>(n_p, 1));
    launch_seg_write(B, len, d_seg, d_chunk_off, strm());
    t_end(KT_SEGS, len + 8 * n_p);
    segs_total += n_p;
    // ---- its words into the table, `sub` segments per launch; the table is grown ahead of a launch that could fill it beyond half
    const unsigned long long extent = LEX0 + lex_cap_ + GAP;  // (every offset a kernel may read from: the chunk, the gap, the lexicon)
    for (unsigned long long s0 = 0; s0 < n_p; s0 += sub) {
      const unsigned long long cnt = std::min(sub, n_p - s0);
      if (2 * (n_unique_host + cnt) > cap) {
        unsigned long long ncap = cap;
        while (2 * (n_unique_host + cnt) > ncap / 2) ncap <<= 1;  // (a quarter full at most after this launch: growth is rare)
        nht = dmalloc<unsigned long long>(3 * ncap);
        launch_word_table_clear(nht, ncap, strm());
        launch_word_table_rehash(B, extent, d_map, ht, cap, nht, ncap, strm());
        sync();
        DFREE(ht);
        ht = nht;
        nht = nullptr;
        cap = ncap;
        word_table_retries++;
      }
      t_begin(KT_DEDUP);
      launch_insert_words(B, extent, d_map, d_seg + s0, cnt, ht, cap - 1, d_status, strm());
      t_end(KT_DEDUP, (len * cnt) / std::max<unsigned long long>(n_p, 1) + 8 * cnt);
      HIP_CHECK(hipMemcpyAsync(h_status, d_status, 32, hipMemcpyDeviceToHost, strm()));
      sync();
      if (h_status[6]) throw GpuError{"word table overflow (chunked front end)"};
      n_unique_host = h_status[0];
    }
    DFREE(d_seg);
    // ---- the chunk's new words move to the lexicon: first how many bytes, the lexicon grown if they do not fit, then the move
    HIP_CHECK(hipMemsetAsync(d_cur + 1, 0, 8, strm()));
    launch_words_relocate(B, C, len + 1, ht, cap, d_cur + 1, /*move=*/false, strm());
    unsigned long long need = 0;
    HIP_CHECK(hipMemcpyAsync(&need, d_cur + 1, 8, hipMemcpyDeviceToHost, strm()));
    sync();
    if (lex_used_ + need > lex_cap_) {
      unsigned long long ncap = lex_cap_;
      while (lex_used_ + need > ncap) ncap <<= 1;
      uint8_t *NB = dmalloc<uint8_t>(LEX0 + ncap + 2 * GAP);
      HIP_CHECK(hipMemcpyAsync(NB, B, (size_t)(LEX0 + lex_used_), hipMemcpyDeviceToDevice, strm()));  // (the chunk too: its new words are still read from it)
      sync();
      DFREE(d_text_owned_);
      B = NB;
      d_text_owned_ = B;
      d_text_ = B;
      lex_cap_ = ncap;
    }
    if (need) launch_words_relocate(B, C, len + 1, ht, cap, d_cur, /*move=*/true, strm());
    lex_used_ += need;
    sync();  // (the next chunk's upload runs on the workers' streams: the region must not be overwritten under the move)
  }
  HIP_CHECK(hipMemsetAsync(B + LEX0 + lex_used_, 32, 2 * GAP, strm()));
  HIP_CHECK(hipMemcpyAsync(h_status, d_status, 32, hipMemcpyDeviceToHost, strm()));
  sync();
  DFREE(landing);
  DFREE(d_chunk_off);
  DFREE(scan_tmp);
  DFREE(d_map_spec);
  DFREE(scratch_hist);
  n_text_ = LEX0 + lex_used_ + GAP;  // what build_word_table reads words from: offsets into B, the lexicon behind the (now idle) chunk region
  if (cfg_->trace.set)
    fprintf(stderr, "[yttm] chunked front end (%s pass): %zu chunks of <= %llu MB in %.1f ms, %llu segments, %u distinct words in %llu slots, lexicon %llu bytes\n",
            first_pass ? "first" : "second", n_chunks, C >> 20, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), segs_total,
            h_status[0], cap, lex_used_);
  spec_.hist_done = first_pass;
  spec_.words_done = true;
  spec_.n_segs = segs_total;
  spec_.ht = ht;
  ht = nullptr;  // (handed on: no longer this scope's to free)
  spec_.ht_cap = cap;
  spec_.long_segments = true;  // (the table's fill is what the growth rule above made it: no second guess in build_word_table)
  memcpy(spec_.h_status, h_status, sizeof h_status);
}

// multi-GPU, small word tables (host_trainer.cpp learn_bpe): every rank ends up with the WHOLE corpus -- the ranks' byte ranges in rank
// order are the file -- and goes on alone.  Returns the ranks' summed dedup token count when called with gather = false (the decision).
unsigned long long GpuCtx::allreduce_scalar(unsigned long long v) {
  HIP_CHECK(hipMemcpyAsync(d_counters_ + 40, &v, 8, hipMemcpyHostToDevice, strm()));
  comm_->allreduce_sum_u64(d_counters_ + 40, 1, strm());
  unsigned long long out = 0;
  HIP_CHECK(hipMemcpyAsync(&out, d_counters_ + 40, 8, hipMemcpyDeviceToHost, strm()));
  sync();
  return out;
}
unsigned long long GpuCtx::free_device_bytes() const {
  if (cfg_->test_free_bytes.set) return cfg_->test_free_bytes.u;
  size_t fr = 0, tot = 0;
  if (hipMemGetInfo(&fr, &tot) != hipSuccess) return 0;
  return (unsigned long long)fr + pool_cached_bytes();
}
void GpuCtx::gather_full_corpus() {
  drop_spec();
  HIP_CHECK(hipSetDevice(device_));
  tl_stream = strm();
  tl_device = device_;
  chain_event_ = nullptr;
  const int W = comm_->world, R = comm_->rank;
  std::vector<unsigned long long> sizes((size_t)W, 0);
  unsigned long long *d_sz = dmalloc<unsigned long long>((size_t)W);
  sizes[(size_t)R] = n_text_;
  HIP_CHECK(hipMemcpyAsync(d_sz, sizes.data(), 8 * (size_t)W, hipMemcpyHostToDevice, strm()));
  comm_->allreduce_sum_u64(d_sz, (size_t)W, strm());
  HIP_CHECK(hipMemcpyAsync(sizes.data(), d_sz, 8 * (size_t)W, hipMemcpyDeviceToHost, strm()));
  sync();
  DFREE(d_sz);
  unsigned long long maxb = 8, total = 0;
  for (unsigned long long v : sizes) { maxb = std::max(maxb, (v + 7) & ~7ull); total += v; }
  uint8_t *d_send = dmalloc<uint8_t>(maxb), *d_recv = dmalloc<uint8_t>(maxb * (unsigned long long)W);
  HIP_CHECK(hipMemsetAsync(d_send, 32, maxb, strm()));
  if (n_text_) HIP_CHECK(hipMemcpyAsync(d_send, d_text_, n_text_, hipMemcpyDeviceToDevice, strm()));
  comm_->allgather_blocks(d_send, d_recv, maxb, strm());
  uint8_t *d_full = dmalloc<uint8_t>(total + 64);
  unsigned long long off = 0;
  for (int r = 0; r < W; r++) {
    if (sizes[(size_t)r]) HIP_CHECK(hipMemcpyAsync(d_full + off, d_recv + maxb * (unsigned long long)r, sizes[(size_t)r], hipMemcpyDeviceToDevice, strm()));
    off += sizes[(size_t)r];
  }
  sync();
  DFREE(d_send);
  DFREE(d_recv);
  DFREE(d_text_owned_);
  d_text_owned_ = d_full;
  d_text_ = d_full;
  n_text_ = total;
  corpus_bytes = total;
}

void GpuCtx::attach_corpus(const void *dev, unsigned long long n) {
  drop_spec();
  chunked_ = false;
  HIP_CHECK(hipSetDevice(device_));
  tl_stream = strm();
  tl_device = device_;
  DFREE(d_text_owned_);
  if (((uintptr_t)dev & 15u) != 0) throw GpuError{"attach_corpus: device pointer must be 16-byte aligned"};
  d_text_ = (const uint8_t *)dev;
  n_text_ = n;
  corpus_bytes = n;
}

}  // namespace yttm
