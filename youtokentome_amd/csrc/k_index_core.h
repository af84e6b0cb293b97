// k_index_core.h -- the pair index as the kernels see it (k_index.hip builds it, k_words.hip looks the batch's rules up in it).
#pragma once
#include "yttm_device.h"
#include "yttm_kernels.h"

namespace yttm {

// ------------------------------------------------------------------------------------------------- pair index (K4 worklists)
// Late in training a batch touches a few percent of the tiles, and which ones cannot be told from the tokens a tile holds (random
// text: every tile holds both halves of nearly every late rule).  So the pairs that can still be merged -- the hot list -- get an
// inverted index, pair -> tiles that hold it (the reference's pair2pos, bpe.cpp:438/:626/:694, at tile granularity): built with two
// streaming passes (count, fill) when the hot list is rebuilt, exact for pairs of tokens that existed then (a merge only creates
// adjacencies of its NEW token, and tokens never change tiles until a repack, which invalidates the index).  A round whose rules
// are all in the index gathers their posting lists into the worklist of the apply kernel instead of streaming every tile.
struct PairIndex {
  unsigned long long *key;  // [mask + 1] open addressing, PT_EMPTY = free
  uint32_t *cnt;            // [(mask + 1) * IDX_SHARDS] postings per key and shard (count pass), then the fill cursors
  unsigned long long *off;  // [(mask + 1) * IDX_SHARDS + 2] start of a (key, shard)'s postings (launch_exclusive_scan of the counts; the last = their total)
  uint32_t *bloom;          // [ENC_BLOOM_WORDS] blocked Bloom filter of the keys (staged into LDS by the streaming passes)
  uint32_t *post;           // tile ids
  unsigned int mask;
};
__device__ inline uint32_t idx_find(const PairIndex &ix, unsigned long long key, uint32_t h) {
  uint32_t s = h & ix.mask;
  for (;;) {
    const unsigned long long k = ix.key[s];
    if (k == key) return s;
    if (k == PT_EMPTY) return 0xffffffffu;
    s = (s + 1) & ix.mask;
  }
}
}  // namespace yttm
