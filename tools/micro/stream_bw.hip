// Read bandwidth of the tile-streaming pattern the K3/K4 kernels use: persistent waves, each reading a 2 KB tile per step
// (one tile ahead), as 8 dword loads or 2 dwordx4 loads per lane, and summing.  hipcc --offload-arch=gfx950 -O3 stream_bw.hip -o stream_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE, int DEPTH>
__global__ __launch_bounds__(256) void k_read(const uint32_t *__restrict__ src, unsigned int n_tiles, unsigned long long *out) {
  const int lane = threadIdx.x & 63;
  const unsigned int n_waves = gridDim.x * 4u;
  unsigned int t = blockIdx.x * 4u + (threadIdx.x >> 6);
  uint32_t acc = 0;
  if (MODE == 0) {  // 8 dword loads per lane per tile, DEPTH tiles in flight
    uint32_t r[DEPTH][8];
#pragma unroll
    for (int d = 0; d < DEPTH; d++)
      if (t + d * n_waves < n_tiles)
#pragma unroll
        for (int c = 0; c < 8; c++) r[d][c] = src[(size_t)(t + d * n_waves) * 512 + 64 * c + lane];
    for (; t < n_tiles; t += n_waves) {
#pragma unroll
      for (int c = 0; c < 8; c++) acc += r[0][c];
#pragma unroll
      for (int d = 0; d + 1 < DEPTH; d++)
#pragma unroll
        for (int c = 0; c < 8; c++) r[d][c] = r[d + 1][c];
      if (t + DEPTH * n_waves < n_tiles)
#pragma unroll
        for (int c = 0; c < 8; c++) r[DEPTH - 1][c] = src[(size_t)(t + DEPTH * n_waves) * 512 + 64 * c + lane];
    }
  } else {  // 2 dwordx4 loads per lane per tile
    uint4 r[DEPTH][2];
    const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
#pragma unroll
    for (int d = 0; d < DEPTH; d++)
      if (t + d * n_waves < n_tiles)
#pragma unroll
        for (int c = 0; c < 2; c++) r[d][c] = s4[(size_t)(t + d * n_waves) * 128 + 64 * c + lane];
    for (; t < n_tiles; t += n_waves) {
#pragma unroll
      for (int c = 0; c < 2; c++) acc += r[0][c].x + r[0][c].y + r[0][c].z + r[0][c].w;
#pragma unroll
      for (int d = 0; d + 1 < DEPTH; d++)
#pragma unroll
        for (int c = 0; c < 2; c++) r[d][c] = r[d + 1][c];
      if (t + DEPTH * n_waves < n_tiles)
#pragma unroll
        for (int c = 0; c < 2; c++) r[DEPTH - 1][c] = s4[(size_t)(t + DEPTH * n_waves) * 128 + 64 * c + lane];
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <int MODE, int DEPTH>
void run(const char *name, const uint32_t *d, unsigned int n_tiles, unsigned long long *out, int bpc) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const unsigned int grid = 256u * bpc;
  hipLaunchKernelGGL((k_read<MODE, DEPTH>), dim3(grid), dim3(256), 0, 0, d, n_tiles, out);
  CK(hipEventRecord(a));
  for (int i = 0; i < 10; i++) hipLaunchKernelGGL((k_read<MODE, DEPTH>), dim3(grid), dim3(256), 0, 0, d, n_tiles, out);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  ms /= 10;
  printf("%-28s bpc %2d: %.3f ms  %.2f TB/s\n", name, bpc, ms, (double)n_tiles * 2048 / ms / 1e9);
}

int main() {
  const unsigned int n_tiles = 540000;
  uint32_t *d; unsigned long long *out;
  CK(hipMalloc(&d, (size_t)n_tiles * 2048));
  CK(hipMalloc(&out, 8));
  CK(hipMemset(d, 1, (size_t)n_tiles * 2048));
  for (int bpc : {4, 8}) {
    run<0, 1>("dword x8, 1 tile ahead", d, n_tiles, out, bpc);
    run<0, 2>("dword x8, 2 tiles ahead", d, n_tiles, out, bpc);
    run<1, 1>("dwordx4 x2, 1 tile ahead", d, n_tiles, out, bpc);
    run<1, 2>("dwordx4 x2, 2 tiles ahead", d, n_tiles, out, bpc);
    run<1, 4>("dwordx4 x2, 4 tiles ahead", d, n_tiles, out, bpc);
  }
  return 0;
}
