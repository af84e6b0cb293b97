// k_wcache.hip -- N4: word-level cache for the batch encoder (the reference has no counterpart).
//
// The reference encodes every word of every sentence from scratch (bpe.cpp:1497-1632: a word's merges depend on nothing but
// the word).  In text most word occurrences repeat a few thousand distinct words, so a batch is encoded in three steps:
//   1. k5w_insert   every word occurrence of the batch -> its slot in a batch-local hash table of distinct words (words of up to
//                   7 bytes ARE their key; longer ones are keyed by hash tag + length + position of the first occurrence and
//                   compared as bytes), the slot remembered per occurrence;
//   2. the distinct words, compacted out of the table, go through K5 (k_encode.hip) as if each were a sentence -- the same
//                   tokenizer, the same merge rounds, so the ids are K5's ids by construction;
//   3. k5w_count / k5w_scatter   per sentence: the words' ids, looked up through the table, are laid end to end.
// BPE-dropout draws per occurrence and never comes here.  A "word" is what enc_tokenize makes of the bytes: a maximal run of
// valid non-space chars; invalid bytes inside it are part of its key (two spellings of one word then take two slots, which is
// only less sharing).
#include "yttm_device.h"
#include "yttm_kernels.h"

namespace yttm {

constexpr unsigned long long WC_LONG = 1ull << 63;       // key of a word of 8 .. WC_MAX_LEN bytes: WC_LONG | tag:7 | len:16 | pos:40
constexpr unsigned long long WC_POS_MASK = (1ull << 40) - 1ull;
constexpr uint32_t WC_MAX_LEN = 0xffffu;                 // longer words are not cached: every occurrence is encoded (list `extra`)
constexpr uint32_t WC_EXTRA = 0x80000000u;               // occ value: index into `extra` instead of a table slot
constexpr uint32_t WC_NONE = 0xffffffffu;                // occ value: no word starts here
constexpr int WC_MAX_PROBES = 512;
constexpr unsigned int WC_CBLK = BLOCK * 8;              // table slots per workgroup of the compaction kernels
constexpr int WC_CLASSES = 8;                            // classes of word length the list of distinct words is laid out in

// bytes [pos, pos + 8) of the text, for any alignment of pos (the text itself is 8-byte aligned; the caller has checked that the 16 bytes from
// pos & ~7 on lie inside it)
__device__ inline unsigned long long wc_load8(const uint8_t *__restrict__ text, unsigned long long pos) {
  const unsigned long long *base = reinterpret_cast<const unsigned long long *>(text + (pos & ~7ull));
  const unsigned long long lo = base[0], hi = base[1];
  const unsigned int sh = (unsigned int)(pos & 7ull) * 8u;
  return sh ? (lo >> sh) | (hi << (64u - sh)) : lo;
}
// the first min(len, 8) bytes of the word in the low bytes of the result, the rest zero
__device__ inline unsigned long long wc_head(const uint8_t *__restrict__ text, unsigned long long total, unsigned long long pos, uint32_t len) {
  unsigned long long w = 0;
  if ((pos & ~7ull) + 16 <= total) {
    w = wc_load8(text, pos);
  } else {  // the last bytes of the buffer: one at a time
    for (uint32_t k = 0; k < 8 && pos + k < total; k++) w |= (unsigned long long)text[pos + k] << (8 * k);
  }
  return len >= 8 ? w : w & ((1ull << (8 * len)) - 1ull);
}
__device__ inline unsigned long long wc_hash_long(const uint8_t *__restrict__ text, unsigned long long total, unsigned long long pos, uint32_t len) {
  unsigned long long h = 0x9e3779b97f4a7c15ull ^ len;
  for (uint32_t done = 0; done < len; done += 8) h = (h ^ wc_head(text, total, pos + done, len - done)) * 0xff51afd7ed558ccdull + (h >> 29);
  return mix64(h);
}
__device__ inline bool wc_bytes_equal(const uint8_t *__restrict__ text, unsigned long long total, unsigned long long a, unsigned long long b, uint32_t len) {
  if (a == b) return true;
  for (uint32_t done = 0; done < len; done += 8)
    if (wc_head(text, total, a + done, len - done) != wc_head(text, total, b + done, len - done)) return false;
  return true;
}

// One 64-byte step of the walk over a sentence's words: lane = byte b0 + lane; position nbytes acts as a space behind the text.
// Returns true in the lane that CLOSES a word (the space behind it), with the word's bytes [*ws, *we).  Same classification as
// enc_tokenize (k_encode.hip): chars by the reference's left-to-right decode, invalid bytes dropped, spaces by cpmap.
struct WordWalk {
  bool carry_space = true;             // class of the last valid char so far (the start of a sentence acts like a space)
  unsigned long long carry_start = 0;  // first byte of the word that is open at the end of the previous step
};
__device__ inline bool wc_walk_step(const EncModel &m, const uint8_t *__restrict__ s, unsigned long long nbytes, unsigned long long b0, uint32_t byte /* s[b0 + lane], 0 behind the end */,
                                    WordWalk &st, unsigned long long *ws, unsigned long long *we) {
  const int lane = lane_id();
  const unsigned long long lt = lanemask_lt();
  const unsigned long long i = b0 + (unsigned long long)lane;
  bool valid = false, space = false;
  if (__ballot(byte >= 0x80u) == 0ull) {  // 64 ASCII bytes (the usual step): every byte is a char, the spaces are utils.cpp:99-101's
    valid = i <= nbytes;
    space = i == nbytes || byte == 32u || (byte - 9u) < 5u;
  } else if (i < nbytes) {
    if (u8_is_start(s, i, nbytes)) {
      uint32_t len;
      const uint32_t cp = u8_decode_at(s, i, nbytes, &len);
      if (cp != INVALID_CP) {
        valid = true;
        space = m.cpmap[cp] == CP_SPACE;
      }
    }
  } else if (i == nbytes) {
    valid = space = true;
  }
  const unsigned long long V = __ballot(valid), S = __ballot(space);
  bool prev_space = st.carry_space;
  const unsigned long long pv = V & lt;
  if (pv) prev_space = (S >> (63 - __clzll((long long)pv))) & 1ull;
  const bool wstart = valid && !space && prev_space;
  const bool closing = valid && space && !prev_space;
  const unsigned long long WSM = __ballot(wstart);
  unsigned long long start = st.carry_start;
  const unsigned long long wlt = WSM & lt;
  if (wlt) start = b0 + (unsigned long long)(63 - __clzll((long long)wlt));
  *ws = start;
  *we = i;
  if (V) st.carry_space = (S >> (63 - __clzll((long long)V))) & 1ull;
  if (WSM) st.carry_start = b0 + (unsigned long long)(63 - __clzll((long long)WSM));
  return closing;
}

// ---- 1. every word occurrence -> table slot ------------------------------------------------------------------------------------
// one word: look it up / insert it, remember its slot at occ[oidx]
__device__ inline void wc_insert_word(const uint8_t *__restrict__ text, unsigned long long total, const WordCache &wc, unsigned long long pos,
                                      unsigned long long len64, unsigned long long oidx) {
  if (len64 > WC_MAX_LEN) {
    const unsigned int e = atomicAdd(wc.extra_n, 1u);
    if (e < wc.extra_cap) {
      wc.extra[2 * e] = pos;
      wc.extra[2 * e + 1] = pos + len64;
    }
    wc.occ[oidx] = WC_EXTRA | e;
    return;
  }
  const uint32_t len = (uint32_t)len64;
  unsigned long long key, h;
  if (len < 8) {
    key = ((unsigned long long)len << 56) | wc_head(text, total, pos, len);
    h = mix64(key);
  } else {
    h = wc_hash_long(text, total, pos, len);
    key = WC_LONG | ((h >> 57) << 56) | ((unsigned long long)len << 40) | pos;
  }
  unsigned long long i = h & wc.mask;
  uint32_t found = WC_NONE;
  for (int probes = 0; probes < WC_MAX_PROBES; probes++) {
    unsigned long long cur = ld_agent(&wc.slot[i]);
    if (cur == PT_EMPTY) {
      cur = atomicCAS(&wc.slot[i], PT_EMPTY, key);
      if (cur == PT_EMPTY) {  // this occurrence is the word's first: it lends the word its bytes
        wc.pos[i] = pos;
        found = (uint32_t)i;
        break;
      }
    }
    if (len < 8 ? cur == key : ((cur ^ key) >> 40) == 0ull && wc_bytes_equal(text, total, cur & WC_POS_MASK, pos, len)) {
      found = (uint32_t)i;
      break;
    }
    i = (i + 1) & wc.mask;
  }
  if (found == WC_NONE) atomicOr(wc.status, 1u);  // table too full: the host doubles it and starts over
  wc.occ[oidx] = found;
}

// A wave walks its sentences 64 bytes at a time; the words it closes (about ten per step) are queued in LDS and inserted 64 at a
// time, one per lane -- the table probe is a chain of dependent loads from HBM, and it is paid once per 64 words instead of once
// per step with a handful of lanes busy.
constexpr int WC_QUEUE = 128;
__global__ __launch_bounds__(BLOCK) void k5w_insert(EncModel m, const uint8_t *__restrict__ text, unsigned long long total,
                                                    const unsigned long long *__restrict__ offsets, unsigned long long n_sent, WordCache wc) {
  __shared__ unsigned long long q_pos[NWAVES][WC_QUEUE], q_occ[NWAVES][WC_QUEUE];
  __shared__ uint32_t q_len[NWAVES][WC_QUEUE];
  const int wave = (int)(threadIdx.x >> 6), lane = lane_id();
  const unsigned long long lt = lanemask_lt();
  const unsigned long long gw = (unsigned long long)blockIdx.x * NWAVES + wave;
  const unsigned long long n_waves = (unsigned long long)gridDim.x * NWAVES;
  unsigned long long *qp = q_pos[wave], *qo = q_occ[wave];
  uint32_t *ql = q_len[wave];
  int queued = 0;
  // (the bytes of the next step -- of this sentence or the first of the wave's next one -- are on their way while this one is
  // worked on: with ASCII text the byte is the step's only load)
  unsigned long long sidx = gw;
  if (sidx >= n_sent) return;
  unsigned long long b_lo = offsets[sidx], nbytes = offsets[sidx + 1] - b_lo, b0 = 0;
  unsigned long long nx_lo = 0, nx_n = 0;  // the wave's next sentence
  if (sidx + n_waves < n_sent) { nx_lo = offsets[sidx + n_waves]; nx_n = offsets[sidx + n_waves + 1] - nx_lo; }
  uint32_t byte = (unsigned long long)lane < nbytes ? text[b_lo + lane] : 0u;
  WordWalk st;
  for (;;) {
    // where the next step is
    const bool same = b0 + 64 <= nbytes;
    const unsigned long long n_sidx = same ? sidx : sidx + n_waves;
    const bool more = n_sidx < n_sent;
    const unsigned long long n_lo = same ? b_lo : nx_lo, n_n = same ? nbytes : nx_n, n_b0 = same ? b0 + 64 : 0;
    uint32_t n_byte = 0;
    if (more && n_b0 + (unsigned long long)lane < n_n) n_byte = text[n_lo + n_b0 + lane];
    unsigned long long nn_lo = nx_lo, nn_n = nx_n;
    if (!same && more && n_sidx + n_waves < n_sent) { nn_lo = offsets[n_sidx + n_waves]; nn_n = offsets[n_sidx + n_waves + 1] - nn_lo; }
    // this step
    unsigned long long ws, we;
    const bool closing = wc_walk_step(m, text + b_lo, nbytes, b0, byte, st, &ws, &we);
    const unsigned long long CM = __ballot(closing);
    if (closing) {
      const int k = queued + (int)__popcll(CM & lt);
      qp[k] = b_lo + ws;
      ql[k] = (uint32_t)(we - ws > 0xfffffffeull ? 0xffffffffull : we - ws);
      qo[k] = (b_lo + ws + sidx) >> 1;
    }
    queued += (int)__popcll(CM);
    wave_sync();
    if (queued >= 64) {
      wc_insert_word(text, total, wc, qp[lane], ql[lane], qo[lane]);
      wave_sync();
      const int rest = queued - 64;  // (< 64)
      unsigned long long p = 0, o = 0;
      uint32_t l = 0;
      if (lane < rest) { p = qp[64 + lane]; l = ql[64 + lane]; o = qo[64 + lane]; }
      wave_sync();
      if (lane < rest) { qp[lane] = p; ql[lane] = l; qo[lane] = o; }
      queued = rest;
      wave_sync();
    }
    if (!more) break;
    if (!same) {
      st = WordWalk();
      nx_lo = nn_lo;
      nx_n = nn_n;
    }
    sidx = n_sidx;
    b_lo = n_lo;
    nbytes = n_n;
    b0 = n_b0;
    byte = n_byte;
  }
  if (lane < queued) wc_insert_word(text, total, wc, qp[lane], ql[lane], qo[lane]);
}

// ---- 2. the table's words as a list ------------------------------------------------------------------------------------------
// The list is in classes of word length, short words first: K5 packs consecutive items into a wavefront's arrays and a pack takes as
// many merge rounds as its longest word -- words of a kind side by side keep the lanes of a pack busy for the same number of rounds.
// (The order inside a class is whatever the atomics make it; nothing depends on it: a word's ids are found through its table slot.)
__device__ inline int wc_len_class(unsigned long long key, int classes) {
  if (classes <= 1) return 0;
  const uint32_t len = (uint32_t)((key & WC_LONG) ? (key >> 40) & 0xffffull : key >> 56);
  return len <= 4 ? 0 : len <= 7 ? 1 : len <= 10 ? 2 : len <= 13 ? 3 : len <= 17 ? 4 : len <= 22 ? 5 : len <= 30 ? 6 : 7;
}
// blk_cnt[c * gridDim.x + block] = words of class c in the block's slots
__global__ __launch_bounds__(BLOCK) void k5w_count_slots(WordCache wc, uint32_t *__restrict__ blk_cnt, int classes) {
  __shared__ unsigned int acc[WC_CLASSES];
  if (threadIdx.x < WC_CLASSES) acc[threadIdx.x] = 0;
  __syncthreads();
  const unsigned long long base = (unsigned long long)blockIdx.x * WC_CBLK;
  for (unsigned int k = 0; k < 8; k++) {
    const unsigned long long i = base + k * BLOCK + threadIdx.x;
    const unsigned long long key = i <= wc.mask ? wc.slot[i] : PT_EMPTY;
    const bool used = key != PT_EMPTY;
    const int c = wc_len_class(key, classes);
    for (int q = 0; q < classes; q++) {
      const unsigned long long M = __ballot(used && c == q);
      if (M && lane_id() == 0) atomicAdd(&acc[q], (unsigned int)__popcll(M));
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < classes) blk_cnt[(unsigned long long)threadIdx.x * gridDim.x + blockIdx.x] = acc[threadIdx.x];
}
// word u of the list: bytes [ustart[u], uend[u]), table slot uslot[u]; the uncached words follow the table's
__global__ __launch_bounds__(BLOCK) void k5w_list(WordCache wc, const unsigned long long *__restrict__ blk_off, unsigned long long n_table,
                                                  unsigned long long *__restrict__ ustart, unsigned long long *__restrict__ uend,
                                                  uint32_t *__restrict__ uslot, int classes) {
  __shared__ unsigned int next[WC_CLASSES];  // words of the class this block has listed so far
  if (threadIdx.x < WC_CLASSES) next[threadIdx.x] = 0;
  __syncthreads();
  const unsigned long long base = (unsigned long long)blockIdx.x * WC_CBLK;
  const unsigned long long lt = lanemask_lt();
  for (unsigned int k = 0; k < 8; k++) {
    const unsigned long long i = base + k * BLOCK + threadIdx.x;
    const unsigned long long key = i <= wc.mask ? wc.slot[i] : PT_EMPTY;
    const bool used = key != PT_EMPTY;
    const int c = wc_len_class(key, classes);
    unsigned int r = 0;
    for (int q = 0; q < classes; q++) {  // a wave takes its places of class q with one LDS atomic
      const unsigned long long M = __ballot(used && c == q);
      if (!M) continue;
      unsigned int first = 0;
      if (lane_id() == 0) first = atomicAdd(&next[q], (unsigned int)__popcll(M));
      first = (unsigned int)__shfl((int)first, 0);
      if (used && c == q) r = first + (unsigned int)__popcll(M & lt);
    }
    if (used) {
      const unsigned long long u = blk_off[(unsigned long long)c * gridDim.x + blockIdx.x] + r, p = wc.pos[i];
      const unsigned long long len = (key & WC_LONG) ? (key >> 40) & 0xffffull : key >> 56;
      ustart[u] = p;
      uend[u] = p + len;
      uslot[u] = (uint32_t)i;
    }
  }
  if (blockIdx.x == 0) {
    const unsigned int n_extra = *wc.extra_n;
    for (unsigned int e = threadIdx.x; e < n_extra; e += BLOCK) {
      ustart[n_table + e] = wc.extra[2 * e];
      uend[n_table + e] = wc.extra[2 * e + 1];
    }
  }
}
// after K5 has encoded the list: the table slot (the `extra` entry) of a word now says where its ids are -- K5 left those of word u at
// scratch + 2 ustart[u] (k_encode.hip SentView), counts[u] of them: offset << 20 | count (a cached word has at most 65 536 ids)
__global__ __launch_bounds__(BLOCK) void k5w_publish(WordCache wc, unsigned long long n_table, unsigned long long n_words, const uint32_t *__restrict__ uslot,
                                                     const unsigned long long *__restrict__ ustart, const uint32_t *__restrict__ ucounts) {
  const unsigned long long u = (unsigned long long)blockIdx.x * BLOCK + threadIdx.x;
  if (u >= n_words) return;
  const unsigned long long o = 2 * ustart[u], n = ucounts[u];
  if (u < n_table) {
    wc.slot[uslot[u]] = (o << 20) | n;
  } else {
    wc.extra[2 * (u - n_table)] = o;
    wc.extra[2 * (u - n_table) + 1] = n;
  }
}

// ---- 3. per sentence: the words' ids end to end ------------------------------------------------------------------------------
__device__ inline void wc_result(const WordCache &wc, uint32_t o, unsigned long long *off, uint32_t *n) {
  if (o & WC_EXTRA) {
    *off = wc.extra[2 * (o & ~WC_EXTRA)];
    *n = (uint32_t)wc.extra[2 * (o & ~WC_EXTRA) + 1];
  } else {
    const unsigned long long r = wc.slot[o];
    *off = r >> 20;
    *n = (uint32_t)(r & 0xfffffull);
  }
}
// A sentence's word occurrences are the entries of occ between its first and last possible index that are not WC_NONE (the host
// clears the array before k5w_insert): the second and third walk over the text are scans of that array, 64 entries = 128 bytes of
// text per step, no UTF-8.
__device__ inline void wc_occ_range(const unsigned long long *__restrict__ offsets, unsigned long long sidx, unsigned long long *lo, unsigned long long *hi) {
  const unsigned long long b_lo = offsets[sidx], b_hi = offsets[sidx + 1];
  *lo = (b_lo + sidx) >> 1;
  *hi = b_hi > b_lo ? ((b_hi - 1 + sidx) >> 1) + 1 : *lo;  // (exclusive)
}
__global__ __launch_bounds__(BLOCK) void k5w_count(const unsigned long long *__restrict__ offsets, unsigned long long n_sent, WordCache wc,
                                                   int n_fixed /* bos + eos */, uint32_t *__restrict__ counts) {
  const unsigned long long gw = (unsigned long long)blockIdx.x * NWAVES + (threadIdx.x >> 6);
  const unsigned long long n_waves = (unsigned long long)gridDim.x * NWAVES;
  for (unsigned long long sidx = gw; sidx < n_sent; sidx += n_waves) {
    unsigned long long lo, hi;
    wc_occ_range(offsets, sidx, &lo, &hi);
    unsigned long long mine = 0;
    for (unsigned long long i = lo + (unsigned long long)lane_id(); i < hi; i += 64) {
      const uint32_t o = wc.occ[i];
      if (o != WC_NONE) {
        unsigned long long off;
        uint32_t n;
        wc_result(wc, o, &off, &n);
        mine += n;
      }
    }
    const unsigned long long tot = wave_sum_u64(mine);
    if (lane_id() == 0) counts[sidx] = (uint32_t)tot + (uint32_t)n_fixed;
  }
}
__global__ __launch_bounds__(BLOCK) void k5w_scatter(EncModel m, const unsigned long long *__restrict__ offsets, unsigned long long n_sent, WordCache wc,
                                                     const int32_t *__restrict__ uids /* K5's scratch */, int bos, int eos, int reverse,
                                                     const unsigned long long *__restrict__ out_off, int32_t *__restrict__ ids_out) {
  const unsigned long long gw = (unsigned long long)blockIdx.x * NWAVES + (threadIdx.x >> 6);
  const unsigned long long n_waves = (unsigned long long)gridDim.x * NWAVES;
  const int lane = lane_id();
  for (unsigned long long sidx = gw; sidx < n_sent; sidx += n_waves) {
    unsigned long long lo, hi;
    wc_occ_range(offsets, sidx, &lo, &hi);
    const unsigned long long o0 = out_off[sidx], n_ids = out_off[sidx + 1] - o0;
    int32_t *out = ids_out + o0;
    if (lane == 0) {
      if (bos) out[reverse ? n_ids - 1 : 0] = m.bos_id;
      if (eos) out[reverse ? 0 : n_ids - 1] = m.eos_id;
    }
    unsigned long long q = bos ? 1 : 0;  // ids of the sentence laid down so far
    for (unsigned long long i0 = lo; i0 < hi; i0 += 64) {
      const unsigned long long i = i0 + (unsigned long long)lane;
      unsigned long long off = 0;
      uint32_t n = 0;
      if (i < hi) {
        const uint32_t o = wc.occ[i];
        if (o != WC_NONE) wc_result(wc, o, &off, &n);
      }
      const uint32_t inc = wave_incl_scan(n);
      const unsigned long long mine = q + inc - n;
      for (uint32_t k = 0; k < n; k++) out[reverse ? n_ids - 1 - (mine + k) : mine + k] = uids[off + k];
      q += (unsigned long long)__shfl(inc, 63);
    }
  }
}

// ---- launchers -----------------------------------------------------------------------------------------------------------------
static inline unsigned int wave_grid(unsigned long long n_items, unsigned int max_blocks) {
  unsigned long long b = (n_items + NWAVES - 1) / NWAVES;
  if (b > max_blocks) b = max_blocks;
  return b ? (unsigned int)b : 1u;
}
static inline unsigned int wcache_blocks(const WordCache &wc) { return (unsigned int)((wc.mask + WC_CBLK) / WC_CBLK); }
// YTTM_K5_CLASSES=1: the list in table order (measurements)
static int wcache_classes() {
  const char *e = getenv("YTTM_K5_CLASSES");
  const int c = e ? atoi(e) : WC_CLASSES;
  return c <= 1 ? 1 : WC_CLASSES;
}
unsigned long long wcache_count_cells(const WordCache &wc) { return (unsigned long long)wcache_blocks(wc) * (unsigned long long)wcache_classes(); }
void launch_wcache_insert(const EncModel &m, const uint8_t *text, unsigned long long total, const unsigned long long *offsets, unsigned long long n_sent,
                          const WordCache &wc, hipStream_t st) {
  hipLaunchKernelGGL(k5w_insert, dim3(wave_grid(n_sent, 256 * 16)), dim3(BLOCK), 0, st, m, text, total, offsets, n_sent, wc);
}
void launch_wcache_count_slots(const WordCache &wc, uint32_t *blk_cnt, hipStream_t st) {
  hipLaunchKernelGGL(k5w_count_slots, dim3(wcache_blocks(wc)), dim3(BLOCK), 0, st, wc, blk_cnt, wcache_classes());
}
void launch_wcache_list(const WordCache &wc, const unsigned long long *blk_off, unsigned long long n_table, unsigned long long *ustart,
                        unsigned long long *uend, uint32_t *uslot, hipStream_t st) {
  hipLaunchKernelGGL(k5w_list, dim3(wcache_blocks(wc)), dim3(BLOCK), 0, st, wc, blk_off, n_table, ustart, uend, uslot, wcache_classes());
}
void launch_wcache_publish(const WordCache &wc, unsigned long long n_table, unsigned long long n_words, const uint32_t *uslot,
                           const unsigned long long *ustart, const uint32_t *ucounts, hipStream_t st) {
  if (!n_words) return;
  hipLaunchKernelGGL(k5w_publish, dim3((unsigned int)((n_words + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, st, wc, n_table, n_words, uslot, ustart, ucounts);
}
void launch_wcache_count(const unsigned long long *offsets, unsigned long long n_sent, const WordCache &wc, int n_fixed, uint32_t *counts, hipStream_t st) {
  hipLaunchKernelGGL(k5w_count, dim3(wave_grid(n_sent, 256 * 16)), dim3(BLOCK), 0, st, offsets, n_sent, wc, n_fixed, counts);
}
void launch_wcache_scatter(const EncModel &m, const unsigned long long *offsets, unsigned long long n_sent, const WordCache &wc, const int32_t *uids, int bos,
                           int eos, int reverse, const unsigned long long *out_off, int32_t *ids_out, hipStream_t st) {
  hipLaunchKernelGGL(k5w_scatter, dim3(wave_grid(n_sent, 256 * 16)), dim3(BLOCK), 0, st, m, offsets, n_sent, wc, uids, bos, eos, reverse, out_off, ids_out);
}

}  // namespace yttm
