// Micro-benchmark (tuning aid), round 6: the doorbell variants round 5's tools/micro/doorbell.hip did NOT measure.
// Round 5's gate had ONE thread of workgroup 0 read the batch from pinned host memory a dword at a time (each a PCIe read of ~0.3 us:
// 82 us per KB) and the other workgroups poll a relay flag with ACQUIRE loads.  Here:
//   wide   : the batch is read by one wave, 16 bytes per lane, all loads in flight (1 KB = one instruction); the relay flag is polled with
//            RELAXED loads and one acquire fence at the end
//   direct : every workgroup polls the host's doorbell itself and reads the batch itself (no relay; G x batch bytes cross the link)
//   bar    : the host WRITES batch + flag into DEVICE memory (a fine-grained allocation mapped for the CPU through the large BAR);
//            every workgroup polls local memory
// Every variant is a ping-pong of N rounds through a pre-launched gate kernel (the next round's kernel is enqueued while this one runs),
// G workgroups, the last one (ticket) publishes in the host's pinned mailbox.  `launch` (a launch per round) is measured beside them.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/doorbell2.hip -o tools/micro/doorbell2 -lhsa-runtime64 ; run: tools/micro/doorbell2 [rounds]
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <setjmp.h>
#include <signal.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
using clk = std::chrono::steady_clock;
static double us_since(clk::time_point t) { return std::chrono::duration<double, std::micro>(clk::now() - t).count(); }

constexpr unsigned int BATCH_WORDS = 256;  // 1 KB: [0] = sequence number (written last), [255] = the payload that is echoed

__device__ inline void publish(volatile unsigned int *mailbox, unsigned int *ticket, unsigned int seq, unsigned int G) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    const unsigned int t = atomicAdd(ticket, 1u);
    if (t == G - 1) {
      __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store((unsigned int *)mailbox, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}
__global__ void k_launch(unsigned int seq, volatile unsigned int *mailbox, unsigned int *ticket) { publish(mailbox, ticket, seq, gridDim.x); }

// mode 0 wide relay | 1 direct | 2 bar (door is device memory)
__global__ __launch_bounds__(512) void k_gate(const unsigned int *door, unsigned int *relay, volatile unsigned int *mailbox, unsigned int *ticket, unsigned int want,
                                              unsigned long long timeout_ticks, int mode) {
  __shared__ unsigned int batch[BATCH_WORDS];
  __shared__ unsigned int got;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave == 0) {
    unsigned int v = 0;
    const unsigned long long t0 = wall_clock64();
    const bool from_door = mode != 0 || blockIdx.x == 0;
    const unsigned int *flag = from_door ? door : relay;
    for (;;) {
      if (lane == 0) v = from_door ? __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      v = (unsigned int)__shfl((int)v, 0);
      if (v == want || v == 0xffffffffu - want) break;
      if (from_door && wall_clock64() - t0 > timeout_ticks) { v = 0xffffffffu - want; break; }
      __builtin_amdgcn_s_sleep(1);
    }
    if (v == want) {
      if (from_door) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // system scope
      else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      // the batch: 16 bytes per lane, one load each, all in flight
      const uint4 q = from_door ? reinterpret_cast<const uint4 *>(door)[lane] : reinterpret_cast<const uint4 *>(relay)[lane];
      reinterpret_cast<uint4 *>(batch)[lane] = q;
      if (mode == 0 && blockIdx.x == 0) {  // relay it (the flag word last)
        if (lane) reinterpret_cast<uint4 *>(relay)[lane] = q;
        else { relay[1] = q.y; relay[2] = q.z; relay[3] = q.w; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (lane == 0) __hip_atomic_store(&relay[0], want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else if (mode == 0 && blockIdx.x == 0 && lane == 0) {
      __hip_atomic_store(&relay[0], 0xffffffffu - want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane == 0) got = v;
  }
  __syncthreads();
  publish(mailbox, ticket, got == want ? batch[BATCH_WORDS - 1] : 0xdead0000u | (want & 0xffffu), gridDim.x);
}

static void wait_mailbox(volatile unsigned int *mb, unsigned int v) {
  const auto t0 = clk::now();
  while (*mb != v) {
    if ((*mb & 0xffff0000u) == 0xdead0000u) { printf("  (the device gave up: %08x)\n", *mb); exit(2); }
    if (us_since(t0) > 5e6) { printf("  mailbox timeout waiting for %u (has %u)\n", v, *mb); exit(3); }
  }
}

// ---- device memory the CPU can write: three ways, each probed in a child process (a fault must not take the benchmark down)
struct HsaFind { hsa_agent_t cpu, gpu; bool have_cpu, have_gpu; hsa_amd_memory_pool_t pool; bool have_pool; int want_flags; };
static hsa_status_t agent_cb(hsa_agent_t a, void *d) {
  HsaFind *f = (HsaFind *)d;
  hsa_device_type_t t;
  hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
  if (t == HSA_DEVICE_TYPE_CPU && !f->have_cpu) { f->cpu = a; f->have_cpu = true; }
  if (t == HSA_DEVICE_TYPE_GPU && !f->have_gpu) { f->gpu = a; f->have_gpu = true; }
  return HSA_STATUS_SUCCESS;
}
static hsa_status_t pool_cb(hsa_amd_memory_pool_t p, void *d) {
  HsaFind *f = (HsaFind *)d;
  hsa_amd_segment_t seg;
  hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
  if (seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
  uint32_t fl = 0;
  hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &fl);
  bool alloc = false;
  hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc);
  printf("  gpu pool: flags %u alloc %d\n", fl, (int)alloc);
  if (alloc && (fl & (uint32_t)f->want_flags) && !f->have_pool) { f->pool = p; f->have_pool = true; }
  return HSA_STATUS_SUCCESS;
}
static unsigned int *bar_alloc_hsa(int want_flags) {
  if (hsa_init() != HSA_STATUS_SUCCESS) return nullptr;
  HsaFind f{};
  f.want_flags = want_flags;
  hsa_iterate_agents(agent_cb, &f);
  if (!f.have_cpu || !f.have_gpu) return nullptr;
  hsa_amd_agent_iterate_memory_pools(f.gpu, pool_cb, &f);
  if (!f.have_pool) return nullptr;
  void *p = nullptr;
  if (hsa_amd_memory_pool_allocate(f.pool, 4096, 0, &p) != HSA_STATUS_SUCCESS) return nullptr;
  hsa_amd_memory_pool_access_t acc;
  hsa_status_t s = hsa_amd_agent_memory_pool_get_info(f.cpu, f.pool, HSA_AMD_AGENT_MEMORY_POOL_INFO_ACCESS, &acc);
  printf("  cpu access to that pool: status %d access %d (0 never, 1 allowed by default, 2 disallowed by default)\n", (int)s, (int)acc);
  hsa_agent_t both[2] = {f.cpu, f.gpu};
  s = hsa_amd_agents_allow_access(2, both, nullptr, p);
  printf("  hsa_amd_agents_allow_access(cpu, gpu): %d\n", (int)s);
  if (s != HSA_STATUS_SUCCESS) return nullptr;
  return (unsigned int *)p;
}
static sigjmp_buf g_jmp;
static void on_fault(int) { siglongjmp(g_jmp, 1); }
static bool cpu_can_write(unsigned int *p) {  // in this process (KFD mappings are not inherited by a fork): a SIGSEGV / SIGBUS is caught and is the answer
  fflush(stdout);
  struct sigaction sa, old_segv, old_bus;
  memset(&sa, 0, sizeof sa);
  sa.sa_handler = on_fault;
  sigemptyset(&sa.sa_mask);
  sigaction(SIGSEGV, &sa, &old_segv);
  sigaction(SIGBUS, &sa, &old_bus);
  bool ok = false;
  if (sigsetjmp(g_jmp, 1) == 0) {
    volatile unsigned int *q = p;
    q[0] = 0x1234u;
    q[1] = 0x5678u;
    ok = q[0] == 0x1234u;
  }
  sigaction(SIGSEGV, &old_segv, nullptr);
  sigaction(SIGBUS, &old_bus, nullptr);
  return ok;
}

int main(int argc, char **argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 2000;
  unsigned int *mb_h, *door_h, *relay, *ticket;
  CK(hipSetDevice(0));
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

  // ---- a CPU-writable pointer into device memory?
  unsigned int *bar = nullptr;
  const char *bar_how = "none";
  {
    unsigned int *p = nullptr;
    if (hipExtMallocWithFlags((void **)&p, 4096, hipDeviceMallocFinegrained) == hipSuccess && p) {
      CK(hipMemset(p, 0, 4096));
      CK(hipDeviceSynchronize());
      const bool ok = cpu_can_write(p);
      printf("hipExtMallocWithFlags(hipDeviceMallocFinegrained): CPU write %s\n", ok ? "WORKS" : "faults");
      if (ok) { bar = p; bar_how = "hipExtMallocWithFlags(hipDeviceMallocFinegrained)"; }
    } else printf("hipExtMallocWithFlags(hipDeviceMallocFinegrained) failed\n");
    (void)hipGetLastError();
    if (!bar && hipExtMallocWithFlags((void **)&p, 4096, hipDeviceMallocUncached) == hipSuccess && p) {
      CK(hipMemset(p, 0, 4096));
      CK(hipDeviceSynchronize());
      const bool ok = cpu_can_write(p);
      printf("hipExtMallocWithFlags(hipDeviceMallocUncached): CPU write %s\n", ok ? "WORKS" : "faults");
      if (ok) { bar = p; bar_how = "hipExtMallocWithFlags(hipDeviceMallocUncached)"; }
    }
    (void)hipGetLastError();
    if (!bar) {
      for (int fl : {(int)HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_FINE_GRAINED, (int)HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_EXTENDED_SCOPE_FINE_GRAINED, (int)HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED}) {
        printf("HSA pool with flag %d:\n", fl);
        p = bar_alloc_hsa(fl);
        if (!p) continue;
        const bool ok = cpu_can_write(p);
        printf("  CPU write %s\n", ok ? "WORKS" : "faults");
        if (ok) { bar = p; bar_how = "hsa_amd_memory_pool_allocate + hsa_amd_agents_allow_access(cpu)"; break; }
      }
    }
    printf("CPU-writable device memory: %s\n", bar_how);
    if (bar) {
      for (int i = 0; i < 1024; i++) ((volatile unsigned int *)bar)[i] = 0;
      // what a host store costs: 1 KB + flag, with a fence between
      const auto t0 = clk::now();
      for (int r = 0; r < 1000; r++) {
        for (unsigned int i = 1; i < BATCH_WORDS; i++) ((volatile unsigned int *)bar)[i] = (unsigned int)r;
        __atomic_thread_fence(__ATOMIC_RELEASE);
        ((volatile unsigned int *)bar)[0] = 0;
        __atomic_thread_fence(__ATOMIC_SEQ_CST);
      }
      printf("host side of a BAR ring (1 KB of dword stores + flag): %.2f us\n", us_since(t0) / 1000);
    }
  }

  unsigned int seq = 1;
  const int grids[] = {1, 64, 256, 512, 1024};
  for (int gi = 0; gi < 5; gi++) {
    const unsigned int G = (unsigned int)grids[gi];
    for (int rep = 0; rep < 2; rep++) {
      const auto t0 = clk::now();
      for (int i = 0; i < N; i++, seq++) {
        hipLaunchKernelGGL(k_launch, dim3(G), dim3(512), 0, st, seq, mb, ticket);
        wait_mailbox(mb, seq);
      }
      if (rep) printf("grid %4u  launch per round: %6.2f us per round\n", G, us_since(t0) / N);
    }
    CK(hipStreamSynchronize(st));
    for (int mode = 0; mode < 3; mode++) {
      if (mode == 2 && !bar) continue;
      unsigned int *door = mode == 2 ? bar : door_h;
      static const char *names[3] = {"gate, wide relay ", "gate, direct     ", "gate, BAR doorbell"};
      for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_gate, dim3(G), dim3(512), 0, st, (const unsigned int *)door, relay, mb, ticket, seq, timeout_ticks, mode);
        const auto t0 = clk::now();
        for (int i = 0; i <= N; i++, seq++) {
          if (i < N) hipLaunchKernelGGL(k_gate, dim3(G), dim3(512), 0, st, (const unsigned int *)door, relay, mb, ticket, seq + 1, timeout_ticks, mode);  // (the next round's, ahead of its batch)
          for (unsigned int w = 1; w < BATCH_WORDS; w++) ((volatile unsigned int *)door)[w] = seq;  // the "batch"
          __atomic_thread_fence(__ATOMIC_RELEASE);
          ((volatile unsigned int *)door)[0] = seq;
          if (mode == 2) __atomic_thread_fence(__ATOMIC_SEQ_CST);  // (push the write-combining buffer out)
          wait_mailbox(mb, seq);
        }
        if (rep) printf("grid %4u  %s, 1 KB batch: %6.2f us per round\n", G, names[mode], us_since(t0) / (N + 1));
        CK(hipStreamSynchronize(st));
      }
    }
  }
  {
    const auto t0 = clk::now();
    hipLaunchKernelGGL(k_gate, dim3(256), dim3(512), 0, st, (const unsigned int *)door_h, relay, mb, ticket, seq, 100ull * 2000 /* 2 ms */, 0);
    CK(hipStreamSynchronize(st));
    printf("gate without a ring: gave up after %.0f us, mailbox %08x\n", us_since(t0), *mb);
  }
  return 0;
}
