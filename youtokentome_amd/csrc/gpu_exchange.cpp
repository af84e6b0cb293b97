// gpu_exchange.cpp -- multi-GPU: the per-round exchange of pair-count deltas -- send blocks, the all-gather, the fold, block sizing.
// (Round 5: cut out of gpu_ctx.cpp, code motion only; gpu_ctx_internal.h says what went where.)
#include "gpu_ctx_internal.h"

namespace yttm {

void GpuCtx::grow_recv(unsigned long long need) {
  if (need <= recv_cap_) return;
  DFREE(d_recv_);
  recv_cap_ = need + need / 4 + 1024;
  d_recv_ = dmalloc<DeltaRec>(recv_cap_);
}

// Set-up exchange (after K3, once per training): every rank's records, however many.  The counts travel first, so buffers
// grow before anything is received and the verdicts (fits / overflow) are the same on every rank: nobody is left waiting in a
// collective the others never posted.
void GpuCtx::exchange_deltas() {
  if (!multi()) return;
  chain_event_ = nullptr;
  launch_fold_stats(d_stats_, pt_.n_keys, strm());  // the apply kernels leave their slot counts in per-workgroup rows
  DeltaRec *cur = d_send2_[xch_parity_];  // (K3's updates went straight into the block's records: dt_add)
  unsigned long long n_local = 0;
  unsigned int nk_local = 0;  // keys in the table after this rank's own updates (one round trip for both numbers)
  HIP_CHECK(hipMemcpyAsync(&n_local, cur, 8, hipMemcpyDeviceToHost, strm()));
  HIP_CHECK(hipMemcpyAsync(&nk_local, pt_.n_keys, 4, hipMemcpyDeviceToHost, strm()));
  sync();
  n_keys_host = nk_local;
  const unsigned long long mine = n_local > send_cap_ ? ~0ull : n_local;
  size_t n_remote = 0;
  for (int attempt = 0;; attempt++) {
    unsigned long long need_all = 0;
    if (comm_->allgather_recs(cur + XHDR, mine, d_recv_, (size_t)recv_cap_, strm(), &need_all, &n_remote)) break;
    if (need_all == ~0ull) throw GpuError{"delta exchange buffer overflow (on some rank)"};
    if (attempt) throw GpuError{"delta receive buffer could not be sized"};
    grow_recv(need_all);
  }
  finish_block((unsigned int)std::min<unsigned long long>(n_local, 1u << 20));
  ensure_table_capacity(n_keys_host + n_remote);
  launch_pt_apply(pt_, d_recv_, n_remote, strm());  // (no candidate list exists yet: nothing is listed)
  unsigned int nk = 0;
  HIP_CHECK(hipMemcpyAsync(&nk, pt_.n_keys, 4, hipMemcpyDeviceToHost, strm()));
  sync();
  n_keys_host = nk;
}

// Per-round exchange, all stream-ordered, no host round trip:
//   ncclAllGather of the first blk_ units (header + records) of every rank's send block -- the apply kernels left it complete: dt_add -- ->
//   k_pt_apply_blocks (phase 1: the other ranks' deltas into the replica, thresholds off) -> k_fold_list (phase 2: the lists, by the final
//   counts; then -- `scan` -- the round's candidate scan straight into the host's mailbox) -> k_dt_clean (during the host's turn).
// ONE collective per round.  What does not fit a block is reported through the mailbox (xstat), and candidates() repeats the exchange
// for exactly those ranks with larger blocks.
// behind an exchange: the table's slots of the block just sent are freed, the other block is made ready, and the next round's updates go
// there (k_dt_clean: off the critical path -- it runs while the host picks the next batch)
void GpuCtx::finish_block(unsigned int n_hint) {
  db_.send = d_send2_[xch_parity_];
  launch_dt_clean(db_, d_send2_[xch_parity_ ^ 1u], n_hint, d_stats_, cls_[0].n_tiles, d_maybe_n_ + 1, strm());
  xch_last_ = d_send2_[xch_parity_];
  xch_parity_ ^= 1u;
  db_.send = d_send2_[xch_parity_];
}
PairTable GpuCtx::pt_nolist() const {
  PairTable p = pt_;
  p.maybe = d_maybe_;  // (the adds note the slots that may have crossed a threshold: k_fold_list looks at those, by their final counts)
  p.maybe_n = d_maybe_n_;
  p.maybe_cap = maybe_cap_;
  p.maybe_hot = pt_.hot_tau;
  p.maybe_top = pt_.top_tau;
  p.hot_tau = ~0ull;
  p.top_tau = ~0ull;
  return p;
}

void GpuCtx::exchange_round(unsigned long long only_mask, const ScanArgs *scan) {
  chain_event_ = nullptr;
  const DeltaRec *block = only_mask ? xch_last_ : d_send2_[xch_parity_];  // (a repeat gathers the same block again, wider)
  grow_recv(blk_ * (unsigned long long)comm_->world);
  comm_->allgather_blocks(block, d_recv_, (size_t)blk_ * sizeof(DeltaRec), strm());
  const bool alone = comm_->world == 1;  // (no other rank's block: phase 1 has nothing to add, the fold kernel reads the header itself)
  if (!alone) launch_pt_apply_blocks(pt_nolist(), d_recv_, blk_, comm_->world, comm_->rank, only_mask, d_xstat_, d_stats_, strm());
  PairTable fpt = pt_;  // (the real thresholds, and the notes to go through)
  fpt.maybe = d_maybe_;
  fpt.maybe_n = d_maybe_n_;
  fpt.maybe_cap = maybe_cap_;
  launch_fold_list(fpt, d_recv_, blk_, comm_->world, only_mask, scan, d_stats_, pending_zero_ && !zero_ba_.k ? d_rules_ : nullptr, zero_cap_ - 1, zero_self_key_,
                   pending_zero_ && zero_ba_.k ? &zero_ba_ : nullptr, d_xstat_, alone, strm());
  if (scan) pending_zero_ = false;  // (the scan zeroes the finished batch's pairs)
  if (!only_mask) finish_block((unsigned int)std::min<unsigned long long>(blk_ / 2, 1u << 18));  // (about as many records as the block was sized for)
}

// multi-GPU: the round's delta table (pair -> this rank's summed count change) and the send block made from it, for `cap` slots; a
// table more than half full counts as overflow.  Called at set-up and, from merge_apply, when a round's bound on the distinct pairs it
// can touch does not fit -- between rounds the table is empty (k_dt_pack frees what a round claimed).
void GpuCtx::alloc_delta_table(unsigned long long cap) {
  DFREE(db_.keys); DFREE(db_.touched); DFREE(d_send2_[0]); DFREE(d_send2_[1]);
  db_.keys = dmalloc<DtSlot>(cap);
  db_.mask = cap - 1;
  launch_dt_init(db_.keys, cap, strm());
  send_cap_ = cap / 2;
  db_.send_cap = send_cap_;
  db_.touched = dmalloc<uint32_t>(send_cap_);
  // the two send blocks { header, records }: all zeros but the capacity in the header -- the peers check a block's count against it
  for (int b = 0; b < 2; b++) {
    d_send2_[b] = dmalloc<DeltaRec>(send_cap_ + XHDR);
    HIP_CHECK(hipMemsetAsync(d_send2_[b], 0, (send_cap_ + XHDR) * sizeof(DeltaRec), strm()));
    const long long capv = (long long)send_cap_;
    HIP_CHECK(hipMemcpyAsync(&d_send2_[b][0].delta, &capv, 8, hipMemcpyHostToDevice, strm()));
  }
  xch_parity_ = 0;
  xch_last_ = d_send2_[0];
  db_.send = d_send2_[0];
  sync();
}

// multi-GPU: what the fold kernel of this round's exchange reported (ranks whose block was too small, the largest record count,
// "a rank lost records").  Repeats the exchange for the skipped ranks with blocks that fit, and sizes the next round's blocks --
// from numbers that are the same on every rank.  True if the table changed (a scan made before that is stale).
bool GpuCtx::settle_exchange(unsigned long long xmask, unsigned long long xmax, unsigned long long fatal) {
  if (fatal) throw GpuError{"delta exchange buffer overflow (on some rank)"};
  unsigned long long want = blk_min_;  // (after a repeat: twice what the busiest rank sent this round)
  while (want < 2 * xmax + 2 * XHDR) want <<= 1;
  if (xmax || xmask) {
    // records per merge site of the round that was just exchanged (its batch's summed pair counts = its sites, over all ranks): what the
    // next rounds' blocks are sized from (merge_apply) -- rounds differ by a factor of four in their batches, much less in this rate
    xrate_[1] = xrate_[0];
    xrate_[0] = xch_sites_ ? (double)xmax / (double)xch_sites_ : 5.0;
  }
  if (!xmask) return false;
  blk_ = blk_min_;
  while (blk_ < xmax + XHDR) blk_ <<= 1;
  exchange_round(xmask, nullptr);
  {  // that fold's own report (same blocks, so nothing new): consumed here
    HIP_CHECK(hipMemsetAsync(d_xstat_, 0, 32, strm()));
  }
  blk_ = std::max(blk_, want);
  exchange_retries++;
  // the scan that came too early also zeroed the finished batch's pairs; deltas that arrived after that (k_giant.hip retracts
  // every old adjacency of a re-counted tile, the merged pairs included) must be zeroed again
  pending_zero_ = zero_valid_;
  return true;
}

}  // namespace yttm
