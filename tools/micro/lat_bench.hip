// Micro-benchmark (tuning aid): latency of ONE random 8-byte load per thread into tables of different sizes, issued right
// at kernel start or after a stretch of LDS/ALU-only work, with few or many workgroups.  Answers: what does a cold access
// to the pair table cost late in training, when each launch touches only a few thousand slots?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__device__ inline unsigned long long mix(unsigned long long x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
__global__ void k(const unsigned long long *tab, unsigned long long mask, int spin, unsigned long long salt, unsigned long long *out, int active_lanes) {
  __shared__ unsigned long long sh[256];
  unsigned long long acc = threadIdx.x;
  sh[threadIdx.x] = acc;
  for (int j = 0; j < spin; j++) { acc = mix(acc + sh[(threadIdx.x + j) & 255]); }
  const unsigned long long gid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long i = mix(gid * 7919 + salt) & mask;
  unsigned long long t0 = clock64(), v = 0;
  if ((int)(threadIdx.x & 63) < active_lanes) v = __hip_atomic_load(&tab[2 * i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  acc += v;
  unsigned long long t1 = clock64();
  if ((threadIdx.x & 63) == 0) { atomicAdd(&out[0], t1 - t0 + (acc == 0x12345 ? 1 : 0)); atomicAdd(&out[1], 1ull); atomicMax(&out[2], t1 - t0); }
}
int main() {
  unsigned long long *tab, *out; const unsigned long long max_slots = 1ull << 28;  // 4 GB
  CK(hipMalloc(&tab, max_slots * 16)); CK(hipMemset(tab, 0, max_slots * 16)); CK(hipMalloc(&out, 64));
  for (unsigned long long slots : {1ull << 20, 1ull << 26, 1ull << 28})
    for (int blocks : {16, 1024})
      for (int spin : {0, 4000})
        for (int lanes : {64, 1}) {
          unsigned long long h[3];
          for (int rep = 0; rep < 4; rep++) {
            CK(hipMemset(out, 0, 64));
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, tab, slots - 1, spin, (unsigned long long)rep * 104729 + slots, out, lanes);
            CK(hipMemcpy(h, out, 24, hipMemcpyDeviceToHost));
          }
          printf("table %5llu MB  blocks %4d  spin %4d  lanes %2d : mean %7.0f cycles  max %7llu\n", slots * 16 >> 20, blocks, spin, lanes, (double)h[0] / h[1], h[2]);
        }
  return 0;
}
