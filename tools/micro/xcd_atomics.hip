// Micro-benchmark (tuning aid): 64-bit atomic adds into XCD-PRIVATE tables with workgroup-scope atomics (executed in the XCD's own L2:
// every workgroup that touches table x runs on XCD x, read from HW_REG_XCC_ID) against device-scope atomics into one shared table.
// Also checks that nothing is lost: the sum over the private tables must be the number of adds.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__device__ inline unsigned long long mix(unsigned long long x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
__device__ inline unsigned int xcc_id() {
  unsigned int v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xfu;
}
// zipf != 0: keys drawn with a heavy skew (the low bits of a product of two uniforms), else uniform
template <int LOCAL>
__global__ void k(unsigned long long *tab, unsigned long long slots_per_table, int per_thread, unsigned long long salt, int zipf, unsigned int *xcc_hist) {
  const unsigned int x = xcc_id();
  if (threadIdx.x == 0) atomicAdd(&xcc_hist[x & 7u], 1u);
  unsigned long long *t = LOCAL ? tab + (unsigned long long)(x & 7u) * slots_per_table : tab;
  const unsigned long long gid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (int j = 0; j < per_thread; j++) {
    unsigned long long r = mix(gid * 131 + j + salt);
    unsigned long long i;
    if (zipf) {
      const double u = (double)(r >> 11) * (1.0 / 9007199254740992.0);
      i = (unsigned long long)((double)slots_per_table * u * u * u * u) % slots_per_table;
    } else {
      i = r % slots_per_table;
    }
    if (LOCAL) __hip_atomic_fetch_add(&t[i], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_fetch_add(&t[i], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
__global__ void k_sum(const unsigned long long *tab, unsigned long long n, unsigned long long *out) {
  unsigned long long s = 0;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) s += tab[i];
  atomicAdd(out, s);
}
int main() {
  const unsigned long long max_slots = 1ull << 27;  // 8 tables x 2^24 slots x 8 B = 1 GB
  unsigned long long *tab, *out; unsigned int *hist;
  CK(hipMalloc(&tab, max_slots * 8)); CK(hipMalloc(&out, 8)); CK(hipMalloc(&hist, 32));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int blocks = 2048, per_thread = 64;
  const double n_ops = (double)blocks * 256 * per_thread;
  for (int zipf = 0; zipf < 2; zipf++)
    for (unsigned long long spt : {1ull << 14, 1ull << 17, 1ull << 20, 1ull << 22, 1ull << 24}) {  // slots per table: 128 KB .. 128 MB
      for (int local = 0; local < 2; local++) {
        CK(hipMemset(tab, 0, max_slots * 8)); CK(hipMemset(out, 0, 8)); CK(hipMemset(hist, 0, 32));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        if (local) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, tab, spt, per_thread, 7ull, zipf, hist);
        else hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, tab, spt, per_thread, 7ull, zipf, hist);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        hipLaunchKernelGGL(k_sum, dim3(1024), dim3(256), 0, 0, tab, local ? 8 * spt : spt, out);
        unsigned long long total = 0; unsigned int h[8];
        CK(hipMemcpy(&total, out, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(h, hist, 32, hipMemcpyDeviceToHost));
        printf("%s keys, %8llu slots/table (%6.1f MB) %s: %8.1f us -> %6.2f Gops/s  sum %s  blocks per XCD %u %u %u %u %u %u %u %u\n", zipf ? "skewed " : "uniform", spt, spt * 8 / 1e6,
               local ? "XCD-private, L2 scope" : "one table, device scope", ms * 1e3, n_ops / ms / 1e6, total == (unsigned long long)n_ops ? "ok" : "LOST UPDATES", h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
      }
    }
  return 0;
}
