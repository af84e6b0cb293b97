// comm_rccl.cpp -- multi-GPU exchange over RCCL/xGMI (one process per GPU) + the C entry points to create it.
//
// What is exchanged: once, the dense char histogram (all-reduce, uint64 sum); after K3 and after every
// K4 round, the ranks' sparse (pair, delta) records (all-gather-v built from grouped ncclSend/ncclRecv -- xGMI is
// point to point, every rank talks to its 7 peers directly), which each rank folds into its replica of the global
// pair table.  This is the RCCL form of the reference's main thread summing per-thread maps (bpe.cpp:1099-1108,
// :1245-1251).  Messages are small (KBs..MBs): latency-bound, so one grouped launch per round.
#include <rccl/rccl.h>
#include <string.h>

#include <vector>

#include "../../include/yttm_mi355x.h"
#include "comm_plan.h"
#include "gpu_ctx.h"
#include "host_core.h"

namespace yttm {

#define NCCL_CHECK(expr)                                                                              \
  do {                                                                                                \
    ncclResult_t _r = (expr);                                                                         \
    if (_r != ncclSuccess) throw GpuError{std::string(#expr) + ": " + ncclGetErrorString(_r)};        \
  } while (0)

struct RcclComm : Comm {
  ncclComm_t comm = nullptr;
  int device = 0;
  unsigned long long *d_counts = nullptr;  // [world]
  std::vector<unsigned long long> h_counts;
  ~RcclComm() override {
    if (d_counts) (void)hipFree(d_counts);
    if (comm) (void)ncclCommDestroy(comm);
  }
  void allreduce_sum_u64(unsigned long long *dev, size_t n, hipStream_t st) override {
    NCCL_CHECK(ncclAllReduce(dev, dev, n, ncclUint64, ncclSum, comm, st));
    HIP_CHECK(hipStreamSynchronize(st));
  }
  void allgather_blocks(const void *send, void *recv, size_t bytes_per_rank, hipStream_t st) override {
    NCCL_CHECK(ncclAllGather(send, recv, bytes_per_rank / 8, ncclUint64, comm, st));
  }
  bool allgather_recs(const DeltaRec *send, unsigned long long n_local, DeltaRec *recv, size_t cap, hipStream_t st, unsigned long long *need_all,
                      size_t *n_remote) override {
    unsigned long long mine = n_local;
    HIP_CHECK(hipMemcpyAsync(d_counts + rank, &mine, 8, hipMemcpyHostToDevice, st));
    NCCL_CHECK(ncclAllGather(d_counts + rank, d_counts, 1, ncclUint64, comm, st));
    HIP_CHECK(hipMemcpyAsync(h_counts.data(), d_counts, 8 * (size_t)world, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    const RecsPlan plan = plan_allgather_recs(h_counts.data(), world, rank, cap);  // (comm_plan.h: shared with the host transport the gloo tests drive)
    *need_all = plan.all;
    if (plan.lost) return false;
    *n_remote = plan.n_remote;
    if (!plan.fits) return false;  // (the same verdict on every rank: `all` and the agreed capacity are)
    NCCL_CHECK(ncclGroupStart());
    for (int r = 0; r < world; r++) {
      if (r == rank) continue;
      if (n_local) NCCL_CHECK(ncclSend(send, n_local * 2, ncclUint64, r, comm, st));
      if (plan.recv_cnt[(size_t)r]) NCCL_CHECK(ncclRecv(recv + plan.recv_off[(size_t)r], plan.recv_cnt[(size_t)r] * 2, ncclUint64, r, comm, st));
    }
    NCCL_CHECK(ncclGroupEnd());
    return true;
  }
};

}  // namespace yttm

using namespace yttm;

extern "C" {

int yttm_comm_rccl_unique_id(uint8_t out[128]) {
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) return 1;
  memcpy(out, &id, 128);
  return 0;
}

int yttm_comm_rccl_create(const uint8_t id_bytes[128], int rank, int world, int device, yttm_comm **out) {
  *out = nullptr;
  try {
    HIP_CHECK(hipSetDevice(device));
    ncclUniqueId id;
    memcpy(&id, id_bytes, 128);
    RcclComm *c = new RcclComm();
    c->rank = rank;
    c->world = world;
    c->device = device;
    c->h_counts.resize((size_t)world);
    NCCL_CHECK(ncclCommInitRank(&c->comm, world, id, rank));
    void *p = nullptr;
    HIP_CHECK(hipMalloc(&p, 8 * (size_t)world));
    c->d_counts = (unsigned long long *)p;
    *out = (yttm_comm *)static_cast<Comm *>(c);
    return 0;
  } catch (const GpuError &e) {
    fprintf(stderr, "yttm_comm_rccl_create: %s\n", e.msg.c_str());
    return 2;
  }
}

}  // extern "C"
