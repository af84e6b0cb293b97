// Micro-benchmark (tuning aid): two kernels of one merge round side by side on two streams against one after the other on one.
// The round: kernel B (GB workgroups, busy for ~TB us) and kernel A (GA workgroups, busy for ~TA us); A's last workgroup (ticket) waits for
// B's flag -- raised by B's last workgroup -- and publishes the round's number in the host's pinned mailbox; the host polls, then starts the
// next round.  Measured per round: the host's wall time, the host time inside the launch calls, and on the device clock how long after A's
// first workgroup B's first workgroup ran (side by side) -- what the protocol of k_words + class-B tiles would see (profiles/r5_round_timeline.txt).
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/two_streams.hip -o tools/micro/two_streams ; run: tools/micro/two_streams [rounds] [TA_us] [TB_us]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
using clk = std::chrono::steady_clock;
static double us_since(clk::time_point t) { return std::chrono::duration<double, std::micro>(clk::now() - t).count(); }

struct Sync {
  unsigned int ticket_a, ticket_b, flag_b, pad;
  unsigned long long t_first_a, t_first_b;  // wall_clock64 (100 MHz) of the kernels' first workgroups, this round
};

__device__ inline void busy(unsigned long long ticks) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
}

__global__ void k_b(Sync *s, unsigned int seq, unsigned long long ticks) {
  if (blockIdx.x == 0 && threadIdx.x == 0) s->t_first_b = wall_clock64();
  busy(ticks);
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(&s->ticket_b, 1u) == gridDim.x - 1) {
      s->ticket_b = 0;
      __hip_atomic_store(&s->flag_b, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

__global__ void k_a(Sync *s, unsigned int seq, unsigned long long ticks, volatile unsigned int *mailbox, int wait_b) {
  if (blockIdx.x == 0 && threadIdx.x == 0) s->t_first_a = wall_clock64();
  busy(ticks);
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(&s->ticket_a, 1u) == gridDim.x - 1) {
      s->ticket_a = 0;
      if (wait_b)
        while (__hip_atomic_load(&s->flag_b, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != seq) __builtin_amdgcn_s_sleep(4);
      unsigned int *mb = (unsigned int *)mailbox;
      mb[1] = (unsigned int)(s->t_first_b - s->t_first_a);  // ticks B's first workgroup ran after A's (side by side only)
      __threadfence_system();
      __hip_atomic_store(mb, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

int main(int argc, char **argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 2000;
  const double ta = argc > 2 ? atof(argv[2]) : 100.0, tb = argc > 3 ? atof(argv[3]) : 30.0;
  const int GA = 300, GB = 600;
  Sync *s;
  CK(hipMalloc(&s, sizeof(Sync)));
  CK(hipMemset(s, 0, sizeof(Sync)));
  unsigned int *mailbox;
  CK(hipHostMalloc(&mailbox, 64, hipHostMallocDefault));
  mailbox[0] = 0;
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  const unsigned long long ka = (unsigned long long)(ta * 100), kb = (unsigned long long)(tb * 100);
  for (int mode = 0; mode < 3; mode++) {  // 0: B then A on one stream; 1: B on the second stream, A waits for its flag; 2: the same, A launched FIRST
    double wall = 0, calls = 0, skew = 0;
    unsigned int seq = (unsigned int)mode * 1000000u;
    for (int r = -50; r < rounds; r++) {
      seq++;
      const auto t0 = clk::now();
      if (mode == 0) {
        hipLaunchKernelGGL(k_b, dim3(GB), dim3(64), 0, sa, s, seq, kb);
        hipLaunchKernelGGL(k_a, dim3(GA), dim3(512), 0, sa, s, seq, ka, (volatile unsigned int *)mailbox, 0);
      } else if (mode == 1) {
        hipLaunchKernelGGL(k_b, dim3(GB), dim3(64), 0, sb, s, seq, kb);
        hipLaunchKernelGGL(k_a, dim3(GA), dim3(512), 0, sa, s, seq, ka, (volatile unsigned int *)mailbox, 1);
      } else {
        hipLaunchKernelGGL(k_a, dim3(GA), dim3(512), 0, sa, s, seq, ka, (volatile unsigned int *)mailbox, 1);
        hipLaunchKernelGGL(k_b, dim3(GB), dim3(64), 0, sb, s, seq, kb);
      }
      const double c = us_since(t0);
      while (__atomic_load_n(mailbox, __ATOMIC_ACQUIRE) != seq) {}
      if (r >= 0) {
        wall += us_since(t0);
        calls += c;
        skew += (double)(int)mailbox[1] * 0.01;
      }
    }
    CK(hipDeviceSynchronize());
    printf("%-62s per round %7.1f us wall, %5.1f us in the two launch calls%s\n",
           mode == 0 ? "B then A on one stream" : mode == 1 ? "B on a second stream beside A, launched first" : "B on a second stream beside A, launched second", wall / rounds, calls / rounds,
           mode == 0 ? "" : "");
    if (mode) printf("    B's first workgroup ran %.1f us after A's (device clock)\n", skew / rounds);
  }
  printf("(A: %d workgroups of 512 threads busy %.0f us; B: %d workgroups of 64 threads busy %.0f us; %d rounds)\n", GA, ta, GB, tb, rounds);
  return 0;
}
