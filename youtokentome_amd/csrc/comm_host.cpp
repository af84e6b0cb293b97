// comm_host.cpp -- host-callback exchange (any transport the application has: torch.distributed/gloo, MPI, ...), and
// the communicator-taking training entry points.  The callback variant stages through host memory; it exists so that
// the N>1 control flow can run (and is tested) where no RCCL/xGMI is available.
#include <string.h>

#include <vector>

#include "../../include/yttm_mi355x.h"
#include "comm_plan.h"
#include "gpu_ctx.h"
#include "host_core.h"

namespace yttm {

struct CallbackComm : Comm {
  yttm_allreduce_u64_fn allreduce = nullptr;
  yttm_allgather_bytes_fn allgather = nullptr;
  void *user = nullptr;
  std::vector<unsigned long long> h_buf;
  std::vector<unsigned char> h_send, h_recv;
  void allreduce_sum_u64(unsigned long long *dev, size_t n, hipStream_t st) override {
    h_buf.resize(n);
    HIP_CHECK(hipMemcpyAsync(h_buf.data(), dev, n * 8, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    if (allreduce(user, h_buf.data(), n) != 0) throw GpuError{"allreduce callback failed"};
    HIP_CHECK(hipMemcpyAsync(dev, h_buf.data(), n * 8, hipMemcpyHostToDevice, st));
    HIP_CHECK(hipStreamSynchronize(st));
  }
  // the others' byte strings back to back, in rank order, through the application's callback
  bool gather_others(const void *send, size_t nbytes, size_t cap_bytes, unsigned long long *got) {
    h_recv.resize(cap_bytes + 1);
    if (allgather(user, send, nbytes, h_recv.data(), cap_bytes, got) != 0) throw GpuError{"allgather callback failed"};
    return *got <= cap_bytes;
  }
  bool allgather_recs(const DeltaRec *send, unsigned long long n_local, DeltaRec *recv, size_t cap, hipStream_t st, unsigned long long *need_all,
                      size_t *n_remote) override {
    // counts first (8 bytes per rank), so that every rank takes the same branch; the callback hands back the OTHERS' strings in rank order:
    // with this rank's own count put in its place they are the all-gathered count vector the RCCL transport has (comm_rccl.cpp)
    unsigned long long got = 0;
    gather_others(&n_local, 8, 8 * (size_t)world, &got);
    if (got != 8 * (size_t)(world - 1)) throw GpuError{"allgather callback returned the wrong number of counts"};
    std::vector<unsigned long long> counts((size_t)world, 0);
    for (int r = 0, k = 0; r < world; r++) {
      if (r == rank) counts[(size_t)r] = n_local;
      else memcpy(&counts[(size_t)r], h_recv.data() + 8 * (size_t)k++, 8);
    }
    const RecsPlan plan = plan_allgather_recs(counts.data(), world, rank, cap);  // (comm_plan.h: the verdicts and offsets of the RCCL path)
    *need_all = plan.all;
    if (plan.lost) return false;
    *n_remote = plan.n_remote;
    if (!plan.fits) return false;
    h_send.resize((size_t)n_local * sizeof(DeltaRec) + 1);
    if (n_local) HIP_CHECK(hipMemcpyAsync(h_send.data(), send, (size_t)n_local * sizeof(DeltaRec), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    if (!gather_others(h_send.data(), (size_t)n_local * sizeof(DeltaRec), cap * sizeof(DeltaRec), &got)) throw GpuError{"allgather callback overflow"};
    if (got != plan.n_remote * sizeof(DeltaRec)) throw GpuError{"allgather callback returned the wrong number of records"};
    // every peer's run to where the plan puts it (one copy per peer, like the RCCL path's one ncclRecv per peer)
    size_t src = 0;
    for (int r = 0; r < world; r++) {
      const size_t c = plan.recv_cnt[(size_t)r];
      if (r == rank || !c) continue;
      HIP_CHECK(hipMemcpyAsync(recv + plan.recv_off[(size_t)r], h_recv.data() + src, c * sizeof(DeltaRec), hipMemcpyHostToDevice, st));
      src += c * sizeof(DeltaRec);
    }
    HIP_CHECK(hipStreamSynchronize(st));
    return true;
  }
  void allgather_blocks(const void *send, void *recv, size_t bytes_per_rank, hipStream_t st) override {
    h_send.resize(bytes_per_rank);
    HIP_CHECK(hipMemcpyAsync(h_send.data(), send, bytes_per_rank, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    unsigned long long got = 0;
    if (!gather_others(h_send.data(), bytes_per_rank, bytes_per_rank * (size_t)world, &got) || got != bytes_per_rank * (size_t)(world - 1))
      throw GpuError{"allgather callback returned the wrong size"};
    size_t src = 0;
    for (int r = 0; r < world; r++) {
      if (r == rank) {
        HIP_CHECK(hipMemcpyAsync((char *)recv + (size_t)r * bytes_per_rank, send, bytes_per_rank, hipMemcpyDeviceToDevice, st));
      } else {
        HIP_CHECK(hipMemcpyAsync((char *)recv + (size_t)r * bytes_per_rank, h_recv.data() + src, bytes_per_rank, hipMemcpyHostToDevice, st));
        src += bytes_per_rank;
      }
    }
    HIP_CHECK(hipStreamSynchronize(st));
  }
};

}  // namespace yttm

using namespace yttm;

static void put_err(char *err, int errlen, const std::string &m) {
  if (err && errlen > 0) snprintf(err, (size_t)errlen, "%s", m.c_str());
}
static BpeConfig make_cfg(double coverage, int pad, int unk, int bos, int eos) {
  BpeConfig c;
  c.character_coverage = coverage;
  c.n_threads = 1;
  c.special_tokens.pad_id = pad;
  c.special_tokens.unk_id = unk;
  c.special_tokens.bos_id = bos;
  c.special_tokens.eos_id = eos;
  return c;
}
void yttm_report_to_json(const TrainReport &r, char *buf, int len);  // capi.cpp

extern "C" {

int yttm_comm_callback_create(int rank, int world, yttm_allreduce_u64_fn allreduce, yttm_allgather_bytes_fn allgather, void *user,
                              yttm_comm **out) {
  CallbackComm *c = new CallbackComm();
  c->rank = rank;
  c->world = world;
  c->allreduce = allreduce;
  c->allgather = allgather;
  c->user = user;
  *out = (yttm_comm *)static_cast<Comm *>(c);
  return 0;
}

void yttm_comm_destroy(yttm_comm *c) { delete (Comm *)c; }

int yttm_train_bpe_from_device_comm(const void *d_text, uint64_t n, const char *model_path, int vocab_size, double coverage, int pad_id,
                                    int unk_id, int bos_id, int eos_id, int device, int profile, yttm_comm *comm, char *report_json,
                                    int report_len, char *err, int errlen) {
  TrainReport rep;
  Status s = train_bpe_from_device(d_text, n, model_path ? model_path : "", vocab_size, make_cfg(coverage, pad_id, unk_id, bos_id, eos_id), device,
                                   &rep, (Comm *)comm, profile);
  if (s.ok()) yttm_report_to_json(rep, report_json, report_len);
  else put_err(err, errlen, s.message);
  return s.code;
}

int yttm_train_bpe_comm(const char *input_path, const char *model_path, int vocab_size, double coverage, int n_threads, int pad_id, int unk_id, int bos_id,
                        int eos_id, int device, int profile, yttm_comm *comm, char *report_json, int report_len, char *err, int errlen) {
  TrainReport rep;
  BpeConfig cfg = make_cfg(coverage, pad_id, unk_id, bos_id, eos_id);
  cfg.n_threads = n_threads;
  Status s = train_bpe(input_path, model_path ? model_path : "", vocab_size, cfg, device, &rep, (Comm *)comm, profile);
  if (s.ok()) yttm_report_to_json(rep, report_json, report_len);
  else put_err(err, errlen, s.message);
  return s.code;
}

int yttm_train_bpe_from_memory_comm(const uint8_t *text, uint64_t n, const char *model_path, int vocab_size, double coverage, int pad_id,
                                    int unk_id, int bos_id, int eos_id, int device, yttm_comm *comm, char *report_json, int report_len,
                                    char *err, int errlen) {
  TrainReport rep;
  Status s = train_bpe_from_memory(text, n, model_path ? model_path : "", vocab_size, make_cfg(coverage, pad_id, unk_id, bos_id, eos_id), device,
                                   &rep, (Comm *)comm);
  if (s.ok()) yttm_report_to_json(rep, report_json, report_len);
  else put_err(err, errlen, s.message);
  return s.code;
}

}  // extern "C"
