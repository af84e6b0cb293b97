// tests/hipsim/include/hip/hip_runtime.h -- TEST INFRASTRUCTURE ONLY.
//
// A tiny single-process emulator of the HIP execution model, used by the CPU test-suite to run the UNMODIFIED
// product sources (youtokentome_amd/csrc/*.hip, *.cpp) without a GPU: the kernels are compiled by g++ with this
// header shadowing <hip/hip_runtime.h>; every workgroup runs as a set of cooperative fibers (one per thread);
// __syncthreads and the wave64 cross-lane ops (__ballot, __shfl*) are rendezvous points.  It checks LOGIC (indexing,
// barriers, hash tables, scans, merge semantics); it says nothing about performance and is never part of the product.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <functional>

#define __global__
#define __device__
#define __host__
#define __shared__ static
#define __launch_bounds__(...)
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
#define __HIP_MEMORY_SCOPE_WORKGROUP 3

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint4 { uint32_t x, y, z, w; };
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }

typedef int hipError_t;
typedef struct hipsimStream *hipStream_t;
typedef struct hipsimEvent *hipEvent_t;
enum { hipSuccess = 0, hipErrorNotReady = 600, hipErrorUnknown = 999 };
enum { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum { hipStreamNonBlocking = 1, hipHostMallocDefault = 0 };
struct hipDeviceProp_t {
  char name[256];
  char gcnArchName[256];
  int multiProcessorCount;
  size_t totalGlobalMem;
};

namespace hipsim {
struct Idx { unsigned x, y, z; };
extern Idx g_threadIdx, g_blockIdx;
extern dim3 g_blockDim, g_gridDim;
void launch(dim3 grid, dim3 block, const std::function<void()> &body);
void sync_block();
unsigned long long wave_ballot(bool pred);
unsigned long long wave_shfl(unsigned long long v, int src_lane);
void wave_barrier();
}  // namespace hipsim

#define threadIdx (hipsim::g_threadIdx)
#define blockIdx (hipsim::g_blockIdx)
#define blockDim (hipsim::g_blockDim)
#define gridDim (hipsim::g_gridDim)

static inline const char *hipGetErrorString(hipError_t) { return "hipsim error"; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) {
  memset(p, 0, sizeof(*p));
  strcpy(p->name, "hipsim (CPU emulation, tests only)");
  strcpy(p->gcnArchName, "hipsim");
  p->multiProcessorCount = 1;
  return hipSuccess;
}
static inline hipError_t hipMemGetInfo(size_t *fr, size_t *tot) { *fr = (size_t)64 << 30; *tot = (size_t)288 << 30; return hipSuccess; }  // (the emulator's "HBM" is host memory)
static inline hipError_t hipMalloc(void **p, size_t n) { *p = calloc(n ? n : 1, 1); return *p ? hipSuccess : hipErrorUnknown; }
static inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = calloc(n ? n : 1, 1); return hipSuccess; }
static inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, int) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, int, hipStream_t) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = (hipStream_t)calloc(1, 8); return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = (hipEvent_t)calloc(1, 8); return hipSuccess; }
#define hipEventDisableTiming 2u
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = (hipEvent_t)calloc(1, 8); return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }  // (launches run at once: nothing to wait for)
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }

template <class K, class... A>
static inline void hipLaunchKernelGGL(K kernel, dim3 grid, dim3 block, size_t, hipStream_t, A... args) {
  hipsim::launch(grid, block, [=]() { kernel(args...); });
}

// ---- device builtins -------------------------------------------------------------------------------------------------
static inline void __syncthreads() { hipsim::sync_block(); }
static inline void __threadfence() {}
static inline void __threadfence_system() {}
static inline unsigned long long __ballot(int pred) { return hipsim::wave_ballot(pred != 0); }
template <class T>
static inline T __shfl(T v, int src) {
  unsigned long long u = 0;
  memcpy(&u, &v, sizeof(T));
  u = hipsim::wave_shfl(u, src);
  T r;
  memcpy(&r, &u, sizeof(T));
  return r;
}
template <class T>
static inline T __shfl_up(T v, int d) {
  int lane = (int)(threadIdx.x & 63u);
  return __shfl(v, lane - d >= 0 ? lane - d : lane);
}
template <class T>
static inline T __shfl_down(T v, int d) {
  int lane = (int)(threadIdx.x & 63u);
  return __shfl(v, lane + d < 64 ? lane + d : lane);
}
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline long long clock64() { return 0; }
static inline long long wall_clock64() { return 0; }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
static inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }

template <class T>
static inline T atomicAdd(T *p, T v) { T o = *p; *p = o + v; return o; }
template <class T>
static inline T atomicOr(T *p, T v) { T o = *p; *p = o | v; return o; }
template <class T>
static inline T atomicAnd(T *p, T v) { T o = *p; *p = o & v; return o; }
template <class T>
static inline T atomicMin(T *p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <class T>
static inline T atomicMax(T *p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T>
static inline T atomicCAS(T *p, T cmp, T v) { T o = *p; if (o == cmp) *p = v; return o; }
template <class T>
static inline T atomicExch(T *p, T v) { T o = *p; *p = v; return o; }
#define __hip_atomic_load(p, order, scope) (*(p))
#define __hip_atomic_store(p, v, order, scope) (*(p) = (v))
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_wave_barrier() hipsim::wave_barrier()
// wave-uniform values: readfirstlane is the identity here (the value is the same in every lane by contract); the mask
// builtins index the uniform mask with the lane id
#define __builtin_amdgcn_readfirstlane(v) (v)
#define __builtin_amdgcn_readlane(v, l) __shfl((int)(v), (int)(l))
static inline bool __builtin_amdgcn_inverse_ballot_w64(unsigned long long m) { return (m >> (threadIdx.x & 63u)) & 1ull; }
static inline unsigned __builtin_amdgcn_mbcnt_lo(unsigned lo, unsigned acc) {
  const unsigned l = threadIdx.x & 63u;
  return acc + (unsigned)__builtin_popcount(l >= 32 ? lo : (lo & ((1u << l) - 1u)));
}
static inline unsigned __builtin_amdgcn_mbcnt_hi(unsigned hi, unsigned acc) {
  const unsigned l = threadIdx.x & 63u;
  return acc + (l > 32 ? (unsigned)__builtin_popcount(hi & ((1u << (l - 32)) - 1u)) : 0u);
}
