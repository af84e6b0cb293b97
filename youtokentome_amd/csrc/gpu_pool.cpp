// gpu_pool.cpp -- device memory pool of the trainer's contexts, and what a finished context leaves behind for the next one (its stream, the pinned
// staging buffer).
// (Round 5: cut out of gpu_ctx.cpp, code motion only; gpu_ctx_internal.h says what went where.)
#include "gpu_ctx_internal.h"

namespace yttm {

thread_local hipStream_t tl_stream = nullptr;
thread_local int tl_device = 0;

// ---- device memory pool --------------------------------------------------------------------------------------------
// A training allocates ~10 GB in a dozen large pieces and frees them again; hipMalloc/hipFree of that size cost several
// milliseconds (hipFree also synchronises the device) and, measured, an occasional 100 ms hiccup.  Freed blocks are kept
// and handed out again to requests of (nearly) the same size.  A block freed by a context that is still running may
// only be reused on that context's stream (same-stream order makes that safe); when the context is destroyed -- after
// a stream synchronisation -- its blocks become free for everyone.  YTTM_NO_POOL=1 turns the pool off,
// yttm_release_device_memory() (capi.cpp) returns the cached blocks to the driver.
namespace {
struct PoolBlock {
  void *p;
  size_t bytes;
  hipStream_t owner;  // nullptr: quiescent
  int device;
};
struct DevPool {
  std::mutex mu;
  std::multimap<size_t, PoolBlock> free_blocks;
  std::unordered_map<void *, size_t> live;
  size_t cached = 0;
  size_t in_use = 0, peak = 0;  // bytes handed out / their high-water mark (GpuCtx::peak_device_bytes; only kept while the pool is on)
};
DevPool g_pool;
void *g_pin_cached = nullptr;  // one pinned staging buffer (PIN_BYTES) kept between contexts
std::vector<std::pair<int, hipStream_t>> g_streams_cached;  // streams of finished contexts, by device (creating and destroying one costs ~2 ms of a training)
constexpr size_t POOL_MAX_CACHED = 96ull << 30;  // (a third of the HBM: the segment starts of 4.4e9 one-letter words alone are 35 GB)
}  // namespace

bool pool_enabled() {
  static const bool on = !(cfg()->no_pool.set && cfg()->no_pool.raw.c_str()[0] == '1');  // (one verdict per process: blocks cached under one policy are not freed under the other)
  return on;
}
void *pool_alloc(size_t bytes) {
  if (bytes == 0) bytes = 1;
  bytes = (bytes + 255) & ~(size_t)255;
  if (pool_enabled()) {
    std::lock_guard<std::mutex> g(g_pool.mu);
    for (auto it = g_pool.free_blocks.lower_bound(bytes); it != g_pool.free_blocks.end() && it->first <= bytes + bytes / 4 + 65536; ++it) {
      const PoolBlock &b = it->second;
      if (b.device != tl_device || (b.owner != nullptr && b.owner != tl_stream)) continue;
      void *p = b.p;
      g_pool.live[p] = b.bytes;
      g_pool.cached -= b.bytes;
      g_pool.in_use += b.bytes;
      g_pool.peak = std::max(g_pool.peak, g_pool.in_use);
      g_pool.free_blocks.erase(it);
      return p;
    }
  }
  void *p = nullptr;
  hipError_t e = hipMalloc(&p, bytes);
  if (e != hipSuccess && pool_enabled()) {  // out of memory with blocks cached: give them back and retry
    {
      std::lock_guard<std::mutex> g(g_pool.mu);
      for (auto &kv : g_pool.free_blocks) (void)hipFree(kv.second.p);
      g_pool.free_blocks.clear();
      g_pool.cached = 0;
    }
    e = hipMalloc(&p, bytes);
  }
  HIP_CHECK(e);
  if (pool_enabled()) {
    std::lock_guard<std::mutex> g(g_pool.mu);
    g_pool.live[p] = bytes;
    g_pool.in_use += bytes;
    g_pool.peak = std::max(g_pool.peak, g_pool.in_use);
  }
  return p;
}
void pool_free(void *p) {
  if (!p) return;
  if (pool_enabled()) {
    std::lock_guard<std::mutex> g(g_pool.mu);
    auto it = g_pool.live.find(p);
    if (it != g_pool.live.end()) {
      const size_t bytes = it->second;
      g_pool.live.erase(it);
      g_pool.in_use -= std::min(g_pool.in_use, bytes);
      if (g_pool.cached + bytes <= POOL_MAX_CACHED) {
        g_pool.free_blocks.emplace(bytes, PoolBlock{p, bytes, tl_stream, tl_device});
        g_pool.cached += bytes;
        return;
      }
    }
  }
  (void)hipFree(p);
}
void pool_quiesce(hipStream_t st) {  // the stream was synchronised: its blocks may now go to anybody
  std::lock_guard<std::mutex> g(g_pool.mu);
  for (auto &kv : g_pool.free_blocks)
    if (kv.second.owner == st) kv.second.owner = nullptr;
}

std::mutex &pool_mutex() { return g_pool.mu; }
unsigned long long pool_cached_bytes() {
  std::lock_guard<std::mutex> g(g_pool.mu);
  return (unsigned long long)g_pool.cached;
}
unsigned long long pool_peak_bytes() {
  std::lock_guard<std::mutex> g(g_pool.mu);
  return (unsigned long long)g_pool.peak;
}
void pool_reset_peak() {
  std::lock_guard<std::mutex> g(g_pool.mu);
  g_pool.peak = g_pool.in_use;
}
hipStream_t pool_take_stream(int device) {
  if (!pool_enabled()) return nullptr;
  std::lock_guard<std::mutex> g(g_pool.mu);
  for (size_t i = 0; i < g_streams_cached.size(); i++)
    if (g_streams_cached[i].first == device) {
      hipStream_t st = g_streams_cached[i].second;
      g_streams_cached.erase(g_streams_cached.begin() + (long)i);
      return st;
    }
  return nullptr;
}
bool pool_give_stream(int device, hipStream_t st) {  // (the caller has synchronised it: nothing is pending on it)
  if (!pool_enabled()) return false;
  std::lock_guard<std::mutex> g(g_pool.mu);
  if (g_streams_cached.size() >= 8) return false;
  g_streams_cached.emplace_back(device, st);
  return true;
}
void *pool_take_pin() {
  std::lock_guard<std::mutex> g(g_pool.mu);
  void *p = g_pin_cached;
  g_pin_cached = nullptr;
  return p;
}
bool pool_give_pin(void *p) {
  std::lock_guard<std::mutex> g(g_pool.mu);
  if (g_pin_cached || !pool_enabled()) return false;
  g_pin_cached = p;
  return true;
}

void release_device_memory() {
  release_io_stage();
  std::lock_guard<std::mutex> g(g_pool.mu);
  if (g_pin_cached) {
    (void)hipHostFree(g_pin_cached);
    g_pin_cached = nullptr;
  }
  for (auto &ds : g_streams_cached) (void)hipStreamDestroy(ds.second);
  g_streams_cached.clear();
  for (auto it = g_pool.free_blocks.begin(); it != g_pool.free_blocks.end();) {
    if (it->second.owner == nullptr) {
      (void)hipFree(it->second.p);
      g_pool.cached -= it->second.bytes;
      it = g_pool.free_blocks.erase(it);
    } else {
      ++it;
    }
  }
}

}  // namespace yttm
