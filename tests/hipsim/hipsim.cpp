// tests/hipsim/hipsim.cpp -- fiber scheduler of the HIP emulator (TEST INFRASTRUCTURE ONLY; see the header).
#include <hip/hip_runtime.h>
#include <stdio.h>

#include <mutex>
#include <vector>

namespace hipsim {
Idx g_threadIdx{0, 0, 0}, g_blockIdx{0, 0, 0};
dim3 g_blockDim(1, 1, 1), g_gridDim(1, 1, 1);

enum State { RUNNABLE, WAIT_BLOCK, WAIT_WAVE, DONE };
enum WaveOp { OP_BALLOT, OP_SHFL, OP_BARRIER };
// Minimal x86-64 System V context switch (callee-saved registers + stack pointer); ~100x cheaper than swapcontext,
// which matters because every emulated GPU thread is a fiber.
extern "C" void hipsim_swap(void **save_sp, void *new_sp);
asm(R"(
.text
.globl hipsim_swap
.type hipsim_swap,@function
hipsim_swap:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size hipsim_swap,.-hipsim_swap
)");

struct Fiber {
  void *sp = nullptr;
  char *stack = nullptr;
  State state = DONE;
  unsigned tid = 0;
  WaveOp op = OP_BARRIER;
  bool pred = false;
  unsigned long long val = 0, result = 0;
  int src = 0;
};
static std::vector<Fiber> g_fibers;
static void *g_sched_sp = nullptr;
static Fiber *g_cur = nullptr;
static const std::function<void()> *g_body = nullptr;
static const size_t STACK = 128 * 1024;

static void yield_to_sched() { hipsim_swap(&g_cur->sp, g_sched_sp); }
static void fiber_main() {
  (*g_body)();
  g_cur->state = DONE;
  for (;;) yield_to_sched();
}

void sync_block() {
  if (!g_cur) return;
  g_cur->state = WAIT_BLOCK;
  yield_to_sched();
}
static unsigned long long wave_op(WaveOp op, bool pred, unsigned long long v, int src) {
  g_cur->op = op;
  g_cur->pred = pred;
  g_cur->val = v;
  g_cur->src = src;
  g_cur->state = WAIT_WAVE;
  yield_to_sched();
  return g_cur->result;
}
unsigned long long wave_ballot(bool pred) { return wave_op(OP_BALLOT, pred, 0, 0); }
unsigned long long wave_shfl(unsigned long long v, int src) { return wave_op(OP_SHFL, false, v, src); }
void wave_barrier() { (void)wave_op(OP_BARRIER, false, 0, 0); }

static void run_block(unsigned nthreads) {
  if (g_fibers.size() < nthreads) g_fibers.resize(nthreads);
  for (unsigned t = 0; t < nthreads; t++) {
    Fiber &f = g_fibers[t];
    if (!f.stack) f.stack = (char *)malloc(STACK);
    {
      uintptr_t top = ((uintptr_t)f.stack + STACK) & ~(uintptr_t)15;
      void **sp = (void **)top;
      *--sp = nullptr;               // alignment slot: fiber_main starts with rsp = 8 (mod 16), as after a call
      *--sp = (void *)fiber_main;    // return address popped by hipsim_swap's ret
      for (int r = 0; r < 6; r++) *--sp = nullptr;  // rbp rbx r12 r13 r14 r15
      f.sp = (void *)sp;
    }
    f.state = RUNNABLE;
    f.tid = t;
  }
  unsigned done = 0;
  while (done < nthreads) {
    bool progressed = false;
    for (unsigned t = 0; t < nthreads; t++) {
      Fiber &f = g_fibers[t];
      if (f.state != RUNNABLE) continue;
      g_cur = &f;
      g_threadIdx.x = f.tid;
      hipsim_swap(&g_sched_sp, f.sp);
      progressed = true;
      if (f.state == DONE) done++;
    }
    g_cur = nullptr;
    // wave rendezvous
    const unsigned nwaves = (nthreads + 63) / 64;
    for (unsigned w = 0; w < nwaves; w++) {
      unsigned lo = w * 64, hi = lo + 64 > nthreads ? nthreads : lo + 64;
      unsigned waiting = 0, alive = 0;
      for (unsigned t = lo; t < hi; t++) {
        if (g_fibers[t].state != DONE) alive++;
        if (g_fibers[t].state == WAIT_WAVE) waiting++;
      }
      if (!alive || waiting != alive) continue;
      // every live lane of the wave is at a cross-lane op: they must all be the same op (convergent code)
      WaveOp op = OP_BARRIER;
      bool first = true;
      unsigned long long mask = 0;
      for (unsigned t = lo; t < hi; t++) {
        Fiber &f = g_fibers[t];
        if (f.state != WAIT_WAVE) continue;
        if (first) { op = f.op; first = false; }
        else if (f.op != op) { fprintf(stderr, "hipsim: divergent wave ops in wave %u (lane %u)\n", w, t - lo); abort(); }
        if (f.pred) mask |= 1ull << (t - lo);
      }
      for (unsigned t = lo; t < hi; t++) {
        Fiber &f = g_fibers[t];
        if (f.state != WAIT_WAVE) continue;
        if (op == OP_BALLOT) f.result = mask;
        else if (op == OP_SHFL) {
          unsigned s = lo + (unsigned)(f.src & 63);
          f.result = (s < hi && g_fibers[s].state == WAIT_WAVE) ? g_fibers[s].val : f.val;
        }
      }
      for (unsigned t = lo; t < hi; t++)
        if (g_fibers[t].state == WAIT_WAVE) g_fibers[t].state = RUNNABLE;
      progressed = true;
    }
    // block barrier
    unsigned at_barrier = 0, alive = 0;
    for (unsigned t = 0; t < nthreads; t++) {
      if (g_fibers[t].state != DONE) alive++;
      if (g_fibers[t].state == WAIT_BLOCK) at_barrier++;
    }
    if (alive && at_barrier == alive) {
      for (unsigned t = 0; t < nthreads; t++)
        if (g_fibers[t].state == WAIT_BLOCK) g_fibers[t].state = RUNNABLE;
      progressed = true;
    }
    if (!progressed && done < nthreads) {
      fprintf(stderr, "hipsim: deadlock in block %u (alive=%u at_barrier=%u)\n", g_blockIdx.x, alive, at_barrier);
      abort();
    }
  }
}

void launch(dim3 grid, dim3 block, const std::function<void()> &body) {
  static std::mutex launch_mu;  // one emulated kernel at a time (host threads of the product may launch concurrently)
  std::lock_guard<std::mutex> lock(launch_mu);
  g_body = &body;
  g_gridDim = grid;
  g_blockDim = block;
  for (unsigned b = 0; b < grid.x; b++) {
    g_blockIdx.x = b;
    run_block(block.x);
  }
  g_body = nullptr;
}
}  // namespace hipsim
