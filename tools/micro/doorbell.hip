// Micro-benchmark (tuning aid): what a merge round's host turn costs by protocol -- a kernel launch per round against a kernel that is
// already on the device and waits for the host's batch ("doorbell").  Every variant is a ping-pong of N rounds: the host hands over a
// sequence number, G workgroups see it, the LAST of them (ticket) publishes it in the host's pinned mailbox, the host polls the mailbox.
//   launch      : one hipLaunchKernelGGL per round (the number rides in the kernel arguments; ARGB bytes of arguments)
//   gate        : the NEXT round's kernel is launched before the host knows its batch; its workgroup 0 polls a doorbell in pinned host
//                 memory (bounded: it gives up after TIMEOUT us), relays the batch to a flag in device memory, the others poll that
//   resident    : one kernel for all N rounds, the same relay, no kernel boundary between rounds
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/doorbell.hip -o tools/micro/doorbell ; run: tools/micro/doorbell [rounds] [grid]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
using clk = std::chrono::steady_clock;
static double us_since(clk::time_point t) { return std::chrono::duration<double, std::micro>(clk::now() - t).count(); }

struct Big { unsigned int w[512]; };  // 2 KB of kernel arguments, like k_words<FUSED>'s

__device__ inline void publish(volatile unsigned int *mailbox, unsigned int *ticket, unsigned int seq, unsigned int G) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    const unsigned int t = atomicAdd(ticket, 1u);
    if (t == G - 1) {
      *ticket = 0;
      __threadfence();
      __hip_atomic_store((unsigned int *)mailbox, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

__global__ void k_launch(unsigned int seq, volatile unsigned int *mailbox, unsigned int *ticket) { publish(mailbox, ticket, seq, gridDim.x); }
__global__ void k_launch_big(Big b, volatile unsigned int *mailbox, unsigned int *ticket) { publish(mailbox, ticket, b.w[0], gridDim.x); }

// the batch: 64 words in pinned host memory, [0] = sequence number (written last by the host)
__device__ inline unsigned int wait_batch(const unsigned int *door /* pinned host */, unsigned int *relay /* device */, unsigned int *lds_batch, unsigned int want,
                                          unsigned long long timeout_ticks, unsigned int batch_words) {
  __shared__ unsigned int got;
  if (threadIdx.x == 0) {
    unsigned int v = 0;
    if (blockIdx.x == 0) {
      const unsigned long long t0 = wall_clock64();
      for (;;) {
        v = __hip_atomic_load(door, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (v == want) break;
        if (wall_clock64() - t0 > timeout_ticks) { v = 0xffffffffu; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      if (v == want)
        for (unsigned int i = 1; i < batch_words; i++) __hip_atomic_store(&relay[i], door[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&relay[0], v == want ? want : 0xffffffffu - want, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      for (;;) {  // (bounded by workgroup 0's verdict)
        v = __hip_atomic_load(&relay[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        if (v == want || v == 0xffffffffu - want) break;
        __builtin_amdgcn_s_sleep(1);
      }
      if (v != want) v = 0xffffffffu;
    }
    got = v;
  }
  __syncthreads();
  if (got == want)
    for (unsigned int i = threadIdx.x; i < batch_words; i += blockDim.x) lds_batch[i] = __hip_atomic_load(&relay[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  return got;
}

__global__ void k_gate(const unsigned int *door, unsigned int *relay, volatile unsigned int *mailbox, unsigned int *ticket, unsigned int want, unsigned long long timeout_ticks,
                       unsigned int batch_words) {
  __shared__ unsigned int batch[256];
  const unsigned int got = wait_batch(door, relay, batch, want, timeout_ticks, batch_words);
  publish(mailbox, ticket, got == want ? batch[batch_words - 1] : 0xdead0000u | (want & 0xffffu), gridDim.x);
}

__global__ void k_resident(const unsigned int *door, unsigned int *relay, volatile unsigned int *mailbox, unsigned int *ticket, unsigned int first, unsigned int rounds,
                           unsigned long long timeout_ticks, unsigned int batch_words) {
  __shared__ unsigned int batch[256];
  for (unsigned int r = 0; r < rounds; r++) {
    const unsigned int want = first + r;
    const unsigned int got = wait_batch(door, relay, batch, want, timeout_ticks, batch_words);
    if (got != want) {
      publish(mailbox, ticket, 0xdead0000u | (want & 0xffffu), gridDim.x);
      return;
    }
    publish(mailbox, ticket, batch[batch_words - 1], gridDim.x);
  }
}

static void wait_mailbox(volatile unsigned int *mb, unsigned int v) {
  const auto t0 = clk::now();
  while (*mb != v) {
    if ((*mb & 0xffff0000u) == 0xdead0000u) { printf("  (the device gave up: %08x)\n", *mb); exit(2); }
    if (us_since(t0) > 5e6) { printf("  mailbox timeout waiting for %u (has %u)\n", v, *mb); exit(3); }
  }
}

int main(int argc, char **argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 2000;
  unsigned int *mb_h, *door_h, *relay, *ticket;
  CK(hipHostMalloc((void **)&mb_h, 4096, hipHostMallocDefault));
  CK(hipHostMalloc((void **)&door_h, 4096, hipHostMallocDefault));
  CK(hipMalloc((void **)&relay, 4096));
  CK(hipMalloc((void **)&ticket, 64));
  CK(hipMemset(relay, 0, 4096));
  CK(hipMemset(ticket, 0, 64));
  memset(mb_h, 0, 4096);
  memset(door_h, 0, 4096);
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  volatile unsigned int *mb = mb_h;
  const unsigned long long timeout_ticks = 100ull * 20000;  // 20 ms at 100 MHz
  unsigned int seq = 1;
  const int grids[] = {1, 64, 256, 512};
  for (int gi = 0; gi < 4; gi++) {
    if (argc > 2 && atoi(argv[2]) != grids[gi]) continue;
    const unsigned int G = (unsigned int)grids[gi];
    // ---- launch per round
    for (int big = 0; big < 2; big++) {
      for (int rep = 0; rep < 2; rep++) {
        double api = 0;
        const auto t0 = clk::now();
        for (int i = 0; i < N; i++, seq++) {
          const auto ta = clk::now();
          if (big) { Big b; b.w[0] = seq; hipLaunchKernelGGL(k_launch_big, dim3(G), dim3(512), 0, st, b, mb, ticket); }
          else hipLaunchKernelGGL(k_launch, dim3(G), dim3(512), 0, st, seq, mb, ticket);
          api += us_since(ta);
          wait_mailbox(mb, seq);
        }
        if (rep) printf("grid %3u  launch per round (%4d B args): %6.2f us per round (the launch call itself %5.2f)\n", G, big ? 2048 : 4, us_since(t0) / N, api / N);
      }
    }
    CK(hipStreamSynchronize(st));
    // ---- gate: round i+1's kernel is enqueued while round i runs; the host rings the doorbell
    for (unsigned int bw : {16u, 256u}) {
      for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_gate, dim3(G), dim3(512), 0, st, door_h, relay, mb, ticket, seq, timeout_ticks, bw);
        const auto t0 = clk::now();
        double api = 0;
        for (int i = 0; i < N; i++, seq++) {
          const auto ta = clk::now();
          hipLaunchKernelGGL(k_gate, dim3(G), dim3(512), 0, st, door_h, relay, mb, ticket, seq + 1, timeout_ticks, bw);  // (the next round's, ahead of its batch)
          api += us_since(ta);
          door_h[bw - 1] = seq;  // the "batch"
          __atomic_store_n(&door_h[0], seq, __ATOMIC_RELEASE);
          wait_mailbox(mb, seq);
        }
        // the last pre-launched kernel: ring it too
        door_h[bw - 1] = seq;
        __atomic_store_n(&door_h[0], seq, __ATOMIC_RELEASE);
        wait_mailbox(mb, seq);
        seq++;
        if (rep) printf("grid %3u  gate, %4u B batch: %6.2f us per round (pre-launch call %5.2f, off the critical path)\n", G, bw * 4, us_since(t0) / N, api / N);
        CK(hipStreamSynchronize(st));
      }
    }
    // ---- resident
    for (unsigned int bw : {16u, 256u}) {
      for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_resident, dim3(G), dim3(512), 0, st, door_h, relay, mb, ticket, seq, (unsigned int)N, timeout_ticks, bw);
        const auto t0 = clk::now();
        for (int i = 0; i < N; i++, seq++) {
          door_h[bw - 1] = seq;
          __atomic_store_n(&door_h[0], seq, __ATOMIC_RELEASE);
          wait_mailbox(mb, seq);
        }
        if (rep) printf("grid %3u  resident, %4u B batch: %6.2f us per round\n", G, bw * 4, us_since(t0) / N);
        CK(hipStreamSynchronize(st));
      }
    }
  }
  // ---- a doorbell nobody rings: the gate gives up by itself (bounded spin)
  {
    const auto t0 = clk::now();
    hipLaunchKernelGGL(k_gate, dim3(256), dim3(512), 0, st, door_h, relay, mb, ticket, seq, 100ull * 2000 /* 2 ms */, 16u);
    CK(hipStreamSynchronize(st));
    printf("gate without a ring: gave up after %.0f us, mailbox %08x\n", us_since(t0), *mb);
  }
  return 0;
}
