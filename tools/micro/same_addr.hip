// Micro-benchmark (tuning aid): what does "one atomicAdd per workgroup on a shared counter" cost a launch?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ void k(unsigned long long *ctr, int n_addr, int per_block, int stride_words) {
  if (threadIdx.x < (unsigned)per_block) atomicAdd(&ctr[(threadIdx.x % n_addr) * stride_words], 1ull);
}
int main() {
  unsigned long long *ctr; CK(hipMalloc(&ctr, 1 << 20)); CK(hipMemset(ctr, 0, 1 << 20));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int blocks : {1, 256, 1024, 4096})
    for (int per_block : {0, 1, 4, 16})
      for (int stride : {1, 32}) {
        float best = 1e9;
        for (int rep = 0; rep < 5; rep++) {
          CK(hipEventRecord(a));
          hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, ctr, per_block ? per_block : 1, per_block, stride);
          CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
          float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
        }
        printf("blocks %5d  atomics/block %2d on %2d counters (stride %3d B): %7.1f us\n", blocks, per_block, per_block ? per_block : 0, stride * 8, best * 1e3);
      }
  return 0;
}
