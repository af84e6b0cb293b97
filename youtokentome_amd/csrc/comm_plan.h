// comm_plan.h -- the transport-independent half of Comm::allgather_recs (gpu_ctx.h): from the ranks' record counts to the common verdict and
// to where each peer's records land in the receive buffer.  ONE implementation for the RCCL transport (comm_rccl.cpp: grouped ncclSend /
// ncclRecv, which have only ever run on hardware with one rank) and the host-callback transport (comm_host.cpp: what the gloo tests drive at
// world sizes 1 .. 8) -- the offsets and verdicts the RCCL path uses are the ones those tests exercise.
// Reference analogue: the main thread summing the per-thread maps, bpe.cpp:1099-1108, :1245-1251.
#pragma once
#include <stddef.h>

#include <vector>

namespace yttm {

struct RecsPlan {
  bool lost = false;                  // some rank reported ~0: its send buffer overflowed (every rank sees it: the counts are all-gathered)
  bool fits = false;                  // the records of ALL ranks fit the agreed capacity (the same verdict on every rank)
  unsigned long long all = 0;         // records of all ranks (~0 when lost)
  size_t n_remote = 0;                // records of the OTHER ranks = what this rank receives
  std::vector<size_t> recv_off;       // [world] record offset of rank r's run in the receive buffer (peers back to back in rank order; own: unused)
  std::vector<size_t> recv_cnt;       // [world] records to receive from rank r (own: 0)
};

// counts[r] = records rank r contributes (~0ull: overflow), for r in [0, world); cap = capacity of the receive buffer in records.
inline RecsPlan plan_allgather_recs(const unsigned long long *counts, int world, int rank, size_t cap) {
  RecsPlan p;
  p.recv_off.assign((size_t)world, 0);
  p.recv_cnt.assign((size_t)world, 0);
  for (int r = 0; r < world; r++)
    if (counts[r] == ~0ull) p.lost = true;
  if (p.lost) {
    p.all = ~0ull;
    return p;
  }
  size_t off = 0;
  for (int r = 0; r < world; r++) {
    p.all += counts[r];
    if (r == rank) continue;
    p.recv_off[(size_t)r] = off;
    p.recv_cnt[(size_t)r] = (size_t)counts[r];
    off += (size_t)counts[r];
  }
  p.n_remote = off;
  p.fits = p.all <= cap;
  return p;
}

}  // namespace yttm
