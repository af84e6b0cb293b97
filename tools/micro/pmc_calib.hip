// Micro-benchmark (tuning aid): what rocprofv3's FETCH_SIZE / WRITE_SIZE count for the access patterns of the BPE kernels, on known byte counts.
// /opt/skills/guides/MI355X_MICROARCH.md (HBM) calibrates one pattern -- a wide coalesced streaming read is reported at exactly half its bytes --
// and says to calibrate every other pattern before trusting an absolute.  K1 / K2 / the tile rounds of K4 stream; word mode gathers words, probes
// hash tables and issues random 64-bit atomics: those are the patterns below.  One launch per kernel and pass:
//   rocprofv3 --pmc FETCH_SIZE -- ./pmc_calib ; rocprofv3 --pmc WRITE_SIZE -- ./pmc_calib      (tools/gpu/r5_n.sh prints counter / known bytes)
// Every buffer is far beyond the 256 MB Infinity Cache and is written by a different kernel than the one measured.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__device__ inline unsigned long long mix(unsigned long long x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }

__global__ void calib_fill(uint4 *p, unsigned long long n16) {
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (unsigned long long)gridDim.x * blockDim.x) p[i] = make_uint4((unsigned)i, 1, 2, 3);
}
// 16 bytes per lane, coalesced, every byte once
__global__ void calib_stream16(const uint4 *p, unsigned long long n16, unsigned long long *sink) {
  unsigned acc = 0;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (unsigned long long)gridDim.x * blockDim.x) { uint4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) *sink = acc;
}
// 4 bytes per lane, coalesced
__global__ void calib_stream4(const unsigned *p, unsigned long long n4, unsigned long long *sink) {
  unsigned acc = 0;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (unsigned long long)gridDim.x * blockDim.x) acc += p[i];
  if (acc == 0x12345678u) *sink = acc;
}
// 1 byte per lane, coalesced (the byte-serial decode of K2e)
__global__ void calib_stream1(const unsigned char *p, unsigned long long n, unsigned long long *sink) {
  unsigned acc = 0;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) acc += p[i];
  if (acc == 0x12345678u) *sink = acc;
}
// one random aligned 8-byte load per lane (a hash probe, an index entry); BYTES: 8, 16 (uint4, a table slot), 64 (four uint4 in a row: a word's tokens), 128 (a whole L2 line)
template <int BYTES>
__global__ void calib_gather(const uint4 *p, unsigned long long mask16, int per_thread, unsigned long long *sink) {
  unsigned acc = 0;
  const unsigned long long gid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (int j = 0; j < per_thread; j++) {
    const unsigned long long i = mix(gid * 131 + j + 7) & mask16 & ~3ull;  // (64-byte aligned)
    if (BYTES == 8) acc += (unsigned)reinterpret_cast<const unsigned long long *>(p)[2 * i];
    if (BYTES == 16) { uint4 v = p[i]; acc += v.x ^ v.w; }
    if (BYTES == 64) for (int q = 0; q < 4; q++) { uint4 v = p[i + q]; acc += v.x ^ v.w; }
    if (BYTES == 128) for (int q = 0; q < 8; q++) { uint4 v = p[(i & ~7ull) + q]; acc += v.x ^ v.w; }  // (both halves of a 128-byte line)
  }
  if (acc == 0x12345678u) *sink = acc;
}
// one random 64-bit atomic add per lane, no return (the pair table's count updates)
__global__ void calib_atomic8(unsigned long long *p, unsigned long long mask8, int per_thread) {
  const unsigned long long gid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (int j = 0; j < per_thread; j++) atomicAdd(&p[mix(gid * 131 + j + 11) & mask8], 1ull);
}
// one random 8-byte store per lane
__global__ void calib_scatter8(unsigned long long *p, unsigned long long mask8, int per_thread) {
  const unsigned long long gid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (int j = 0; j < per_thread; j++) p[mix(gid * 131 + j + 13) & mask8] = gid;
}
// 16 bytes per lane, coalesced store
__global__ void calib_store16(uint4 *p, unsigned long long n16) {
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (unsigned long long)gridDim.x * blockDim.x) p[i] = make_uint4(7, (unsigned)i, 2, 3);
}

int main() {
  const unsigned long long BYTES = 4ull << 30, n16 = BYTES / 16;
  uint4 *buf;
  unsigned long long *sink;
  CK(hipMalloc(&buf, BYTES));
  CK(hipMalloc(&sink, 8));
  const int G = 8192, T = 256, PER = 8;             // gathers / atomics / scatters: 8192 * 256 * 8 = 2^24 accesses
  const unsigned long long N_ACC = (unsigned long long)G * T * PER;
  hipLaunchKernelGGL(calib_fill, dim3(G), dim3(T), 0, 0, buf, n16);
  CK(hipDeviceSynchronize());
  hipLaunchKernelGGL(calib_stream16, dim3(G), dim3(T), 0, 0, buf, n16, sink);
  hipLaunchKernelGGL(calib_stream4, dim3(G), dim3(T), 0, 0, (const unsigned *)buf, BYTES / 4, sink);
  hipLaunchKernelGGL(calib_stream1, dim3(G), dim3(T), 0, 0, (const unsigned char *)buf, BYTES / 4, sink);  // (1 GB of it)
  hipLaunchKernelGGL(calib_gather<8>, dim3(G), dim3(T), 0, 0, buf, n16 - 1, PER, sink);
  hipLaunchKernelGGL(calib_gather<16>, dim3(G), dim3(T), 0, 0, buf, n16 - 1, PER, sink);
  hipLaunchKernelGGL(calib_gather<64>, dim3(G), dim3(T), 0, 0, buf, n16 - 1, PER, sink);
  hipLaunchKernelGGL(calib_gather<128>, dim3(G), dim3(T), 0, 0, buf, n16 - 1, PER, sink);
  hipLaunchKernelGGL(calib_atomic8, dim3(G), dim3(T), 0, 0, (unsigned long long *)buf, BYTES / 8 - 1, PER);
  hipLaunchKernelGGL(calib_scatter8, dim3(G), dim3(T), 0, 0, (unsigned long long *)buf, BYTES / 8 - 1, PER);
  hipLaunchKernelGGL(calib_store16, dim3(G), dim3(T), 0, 0, buf, n16);
  CK(hipDeviceSynchronize());
  printf("known: stream16 %llu B, stream4 %llu B, stream1 %llu B, gathers / atomics / scatters %llu accesses each, store16 %llu B\n", BYTES, BYTES, BYTES / 4, N_ACC, BYTES);
  return 0;
}
