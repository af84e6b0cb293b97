// Micro-benchmark (tuning aid): latency of DEPENDENT random loads (pointer chase) over tables of different sizes -- what one "trip" of a
// latency-bound late merge round costs.  Per lane an independent chain (64 chains per wave in flight), plain vs device-scope loads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__device__ __host__ inline unsigned long long mix(unsigned long long x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
__global__ void init(unsigned long long *tab, unsigned long long n) {
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) tab[i] = mix(i * 0x9e3779b97f4a7c15ull + 1);
}
template <int MODE>
__global__ void chase(const unsigned long long *tab, unsigned long long mask, int hops, unsigned long long *out) {
  unsigned long long i = mix((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x + 12345) & mask;
  const unsigned long long t0 = clock64(), w0 = wall_clock64();
  for (int h = 0; h < hops; h++) {
    unsigned long long v = MODE ? __hip_atomic_load(&tab[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : tab[i];
    i = v & mask;
  }
  const unsigned long long t1 = clock64(), w1 = wall_clock64();
  if ((threadIdx.x & 63) == 0) { atomicAdd(&out[0], t1 - t0); atomicAdd(&out[1], 1ull); atomicAdd(&out[2], w1 - w0); }
  if (i == 0x123456789) out[3] = i;
}
int main() {
  unsigned long long *tab, *out; const unsigned long long max_n = 1ull << 30;  // 8 GB
  CK(hipMalloc(&tab, max_n * 8)); CK(hipMalloc(&out, 64));
  hipLaunchKernelGGL(init, dim3(4096), dim3(256), 0, 0, tab, max_n); CK(hipDeviceSynchronize());
  const int hops = 64;
  for (unsigned long long n : {1ull << 20, 1ull << 25, 1ull << 27, 1ull << 30})
    for (int blocks : {8, 256, 2048})
      for (int mode = 0; mode < 2; mode++) {
        unsigned long long h[4]; float ms = 0; hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        for (int rep = 0; rep < 3; rep++) {
          CK(hipMemset(out, 0, 64)); CK(hipEventRecord(a));
          if (mode) hipLaunchKernelGGL(chase<1>, dim3(blocks), dim3(256), 0, 0, tab, n - 1, hops, out); else hipLaunchKernelGGL(chase<0>, dim3(blocks), dim3(256), 0, 0, tab, n - 1, hops, out);
          CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
        }
        CK(hipMemcpy(h, out, 32, hipMemcpyDeviceToHost));
        printf("table %5llu MB  blocks %4d  %s : %7.0f cycles/hop  %6.0f ns/hop (100 MHz clock)  kernel %8.1f us\n", n * 8 >> 20, blocks, mode ? "agent-scope" : "plain      ",
               (double)h[0] / h[1] / hops, (double)h[2] / h[1] / hops * 10.0, ms * 1e3);
      }
  return 0;
}
