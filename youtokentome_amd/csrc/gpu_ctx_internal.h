// gpu_ctx_internal.h -- what the translation units of GpuCtx share (round 5: gpu_ctx.cpp, 2 600 lines, was cut along its concerns: gpu_pool.cpp the
// device-memory pool, gpu_upload.cpp corpus -> HBM incl. the overlapped and the chunked front end, gpu_frontend.cpp K1 / K2 / tiles,
// gpu_pairs.cpp pair table + candidate lists, gpu_exchange.cpp the multi-GPU delta exchange, gpu_words.cpp word mode + pair index,
// gpu_ctx.cpp what is left: construction, timers, the merge round).  Code motion only.
#pragma once
#include "gpu_ctx.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <future>
#include <map>
#include <mutex>
#include <thread>
#include <unordered_map>

namespace yttm {

inline unsigned long long pow2_at_least(unsigned long long v) {
  unsigned long long c = 1;
  while (c < v) c <<= 1;
  return c;
}

// ---- gpu_pool.cpp: device memory pool, cached streams / pinned staging of finished contexts
extern thread_local hipStream_t tl_stream;  // the stream / device of the context this thread works for (a freed block may be reused on its own stream at once)
extern thread_local int tl_device;
bool pool_enabled();
void *pool_alloc(size_t bytes);
void pool_free(void *p);
void pool_quiesce(hipStream_t st);
std::mutex &pool_mutex();                 // also guards the event pool of the kernel-family timers (gpu_ctx.cpp)
unsigned long long pool_cached_bytes();   // bytes the pool holds for reuse
unsigned long long pool_peak_bytes();     // high-water mark of the bytes handed out ...
void pool_reset_peak();                   // ... since this call
hipStream_t pool_take_stream(int device);           // a stream a finished context left behind (nullptr: none)
bool pool_give_stream(int device, hipStream_t st);  // false: not kept (the caller destroys it)
void *pool_take_pin();                               // the pinned staging buffer of a finished context (nullptr: none)
bool pool_give_pin(void *p);
void release_io_stage();  // gpu_upload.cpp: the upload workers' pinned chunks, streams and events

template <class T>
static T *dmalloc(size_t n) {
  return (T *)pool_alloc((n ? n : 1) * sizeof(T));
}
#define DFREE(p)            \
  do {                      \
    if (p) pool_free((void *)(p)); \
    p = nullptr;            \
  } while (0)

// Pool blocks a function holds in local raw pointers: whatever is still non-null when the scope is left -- by return or by a throw -- goes back
// to the pool (DFREE nulls what it frees; a block handed on to a member is nulled by hand).  Declare it BEFORE anything whose destructor must
// run while the blocks are still alive (destruction is in reverse order).
struct DevScope {
  std::vector<void **> held;
  template <class T> void hold(T *&p) { held.push_back((void **)&p); }
  ~DevScope() {
    for (void **pp : held)
      if (*pp) { pool_free(*pp); *pp = nullptr; }
  }
};


constexpr unsigned int CAND_CAP = 1u << 20;
constexpr unsigned int HOT_CAP = 1u << 18;  // hot-list slots (entries appended between rebuilds included)
// a rebuild picks the threshold that lists about HOT_TARGET pairs; fewer live entries than HOT_MIN: lower the threshold.
// YTTM_HOT_TARGET / YTTM_HOT_MIN / YTTM_HOT_CAP override them (the test-suite shrinks them to exercise rebuilds on tiny corpora).
constexpr unsigned int TOP_CAP = 1u << 15;  // top-list slots
constexpr unsigned int RULES_CAP = 1u << 14;  // hash slots for the per-round rule table (batch <= RULES_CAP/2)
constexpr size_t PIN_BYTES = (size_t)CAND_CAP * sizeof(CandRec) + (size_t)RULES_CAP * sizeof(RuleSlot) + (1u << 20);

}  // namespace yttm
