// Micro-benchmark (tuning aid): throughput of 64-bit global atomics on MI355X as the pair-table flush issues them.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__device__ inline unsigned long long mix(unsigned long long x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
// mode 0: no-return add, 1: returning add, 2: load key + no-return add (dependent), 3: load + returning add, 4: plain load only
template <int MODE>
__global__ void k(unsigned long long *tab, unsigned long long mask, int per_thread, unsigned long long salt, unsigned long long *sink, int distinct) {
  unsigned long long acc = 0;
  const unsigned long long gid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (int j = 0; j < per_thread; j++) {
    unsigned long long key = distinct ? gid * 131 + j : (unsigned long long)(threadIdx.x + j * 256);  // !distinct: every block hits the same 256*per_thread slots
    unsigned long long i = mix(key + salt) & mask;
    if (MODE == 0) atomicAdd(&tab[2 * i + 1], 1ull);
    if (MODE == 1) acc += atomicAdd(&tab[2 * i + 1], 1ull);
    if (MODE == 2) { unsigned long long kk = __hip_atomic_load(&tab[2 * i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (kk != 12345) atomicAdd(&tab[2 * i + 1], 1ull); }
    if (MODE == 3) { unsigned long long kk = __hip_atomic_load(&tab[2 * i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (kk != 12345) acc += atomicAdd(&tab[2 * i + 1], 1ull); }
    if (MODE == 4) acc += tab[2 * i];
  }
  if (acc == 0x1234567) *sink = acc;
}
template <int MODE>
void run(const char *name, unsigned long long *tab, unsigned long long mask, int blocks, int per_thread, int distinct, unsigned long long *sink) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int rep = 0; rep < 3; rep++) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, tab, mask, per_thread, (unsigned long long)rep * 977, sink, distinct);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (rep == 2) printf("%-28s blocks %5d x256 x%d %s: %8.1f us  -> %.2f Gops/s\n", name, blocks, per_thread, distinct ? "distinct" : "shared  ", ms * 1e3, blocks * 256.0 * per_thread / ms / 1e6);
  }
}
int main() {
  unsigned long long slots = 1ull << 26; unsigned long long *tab, *sink;
  CK(hipMalloc(&tab, slots * 16)); CK(hipMemset(tab, 0, slots * 16)); CK(hipMalloc(&sink, 8));
  for (int distinct = 1; distinct >= 0; distinct--)
    for (int blocks : {64, 256, 1024}) {
      run<0>("add (no return)", tab, slots - 1, blocks, 1, distinct, sink);
      run<1>("add (return)", tab, slots - 1, blocks, 1, distinct, sink);
      run<2>("load key + add", tab, slots - 1, blocks, 1, distinct, sink);
      run<3>("load key + add(return)", tab, slots - 1, blocks, 1, distinct, sink);
      run<4>("plain load", tab, slots - 1, blocks, 1, distinct, sink);
    }
  // small table (L2-resident)
  run<0>("add no-ret, 1 MB table", tab, (1ull << 16) - 1, 1024, 1, 1, sink);
  run<1>("add ret, 1 MB table", tab, (1ull << 16) - 1, 1024, 1, 1, sink);
  return 0;
}
