// k_wcache.hip -- N4: word-level cache for the batch encoder (the reference has no counterpart).
//
// The reference encodes every word of every sentence from scratch (bpe.cpp:1497-1632: a word's merges depend on nothing but
// the word).  In text most word occurrences repeat a few thousand distinct words, so a batch is encoded in three steps:
//   1. k5w_insert   every word occurrence of the batch -> its slot in a batch-local hash table of distinct words (words of up to
//                   7 bytes ARE their key; longer ones are keyed by hash tag + length + position of the first occurrence and
//                   compared as bytes), the slot remembered per occurrence;
//   2. the distinct words, compacted out of the table, go through K5 (k_encode.hip k5_words) as if each were a sentence -- the same
//                   tokenizer, the same merges in the same order, so the ids are K5's ids by construction; a word's table slot then
//                   says where its ids are and how many;
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
constexpr int WC_SHORT_PROBES = 24;                      // slots a short word tries in the table's short region before it goes on in the whole table
constexpr int WC_SBLK = 64;                              // sentences a wavefront of the flat walks (insert, count, scatter) takes at a time, at most
constexpr int WC_CLASSES = 4;                            // classes of word length the list of distinct words is laid out in
constexpr unsigned int WC_CBLK = BLOCK * 8;              // table slots per workgroup of the compaction kernels

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
  // Words of up to 7 bytes -- few distinct ones, most of the occurrences -- start their probes in the table's first short_mask + 1 slots:
  // a few MB that stay in L2 / the Infinity Cache instead of a random line of a table of hundreds of MB per occurrence.
  unsigned long long i = h & (len < 8 ? wc.short_mask : wc.mask);
  uint32_t found = WC_NONE;
  for (int probes = 0; probes < WC_MAX_PROBES; probes++) {
    // (a plain load: a slot is written once, so whatever key it shows is final; an EMPTY may be stale, and the CAS settles that)
    unsigned long long cur = wc.slot[i];
    if (cur == PT_EMPTY) {
      cur = atomicCAS(&wc.slot[i], PT_EMPTY, key);
      if (cur == PT_EMPTY) {  // this occurrence is the word's first: it lends the word its bytes
        if (len < 8) wc.pos[i] = pos;  // (a long word's key says where: one random store less for each of them)
        found = (uint32_t)i;
        break;
      }
    }
    if (len < 8 ? cur == key : ((cur ^ key) >> 40) == 0ull && wc_bytes_equal(text, total, cur & WC_POS_MASK, pos, len)) {
      found = (uint32_t)i;
      break;
    }
    // a crowded short region (more distinct short words than it was sized for): after WC_SHORT_PROBES slots of it, on through the whole table
    if (len < 8 && probes == WC_SHORT_PROBES - 1 && wc.short_mask != wc.mask) i = (h >> 7) & wc.mask;
    else i = (i + 1) & wc.mask;
  }
  if (found == WC_NONE) atomicOr(wc.status, len < 8 ? 3u : 1u);  // table too full: the host starts over with a larger one (bit 1: it was a short word that found no slot)
  wc.occ[oidx] = found;
}

// A wavefront takes sblk consecutive sentences -- one run of bytes, the sentences lie back to back -- and walks the run whatever the
// sentences' lengths (a sentence of 129 bytes walked alone is three 64-byte steps, the third with one lane busy).  Same classification as
// enc_tokenize (k_encode.hip): chars by the reference's left-to-right decode inside their sentence, invalid bytes dropped, spaces by cpmap;
// a sentence's first byte has the sentence start in front of it, which acts like a space: it closes the word the previous sentence left
// open and lets a new one begin.  Steps of 512 bytes, EIGHT per lane, while the bytes are ASCII: white space, word starts and word ends are
// bit masks of a lane's eight bytes (the trainer's k_scan_bytes does the same); any byte beyond ASCII sends those 512 bytes through the
// walk of one byte per lane with the exact decode.  The words a step closes (about eighty) are queued in LDS and inserted 64 at a time,
// one per lane -- the table probe is a chain of dependent loads, paid once per 64 words.
constexpr int WC_QUEUE = 64 + 4 * 64 + 64;  // what a flush leaves + what a step can close (a word and its space are two bytes)
struct WalkState {
  bool carry_space = true;             // class of the last valid char of the current sentence so far (its start acts like a space)
  unsigned long long carry_start = 0;  // where the open word began
  int queued = 0;
};
struct WordQueue {
  unsigned long long *pos, *occ;
  uint32_t *len;
};
// the sentence (of the block: bnd[0 .. ns]) of byte p: the last one that begins at or before it (empty sentences pile up on one byte)
__device__ inline int wc_sentence_of(const unsigned long long *bnd, int ns, unsigned long long p) {
  int a = 0, b = ns;  // bnd[a] <= p; sentences a .. b - 1 are candidates
  while (b - a > 1) {
    const int mid = (a + b) >> 1;
    if (bnd[mid] <= p) a = mid; else b = mid;
  }
  return a;
}
__device__ inline void wc_queue_push(const WordQueue &q, int at, const unsigned long long *bnd, int ns, unsigned long long s0, unsigned long long ws,
                                     unsigned long long we) {
  q.pos[at] = ws;
  q.len[at] = (uint32_t)(we - ws > 0xfffffffeull ? 0xffffffffull : we - ws);
  q.occ[at] = (ws + s0 + (unsigned long long)wc_sentence_of(bnd, ns, ws)) >> 1;
}
// 64 words at a time from the queue's head; what is left (< 64) moves to the front
__device__ inline void wc_queue_drain(const uint8_t *__restrict__ text, unsigned long long total, const WordCache &wc, const WordQueue &q, int &queued) {
  const int lane = lane_id();
  int head = 0;
  for (; queued - head >= 64; head += 64) wc_insert_word(text, total, wc, q.pos[head + lane], q.len[head + lane], q.occ[head + lane]);
  if (head == 0) return;
  const int rest = queued - head;  // (< 64)
  unsigned long long pp = 0, oo = 0;
  uint32_t ll = 0;
  wave_sync();
  if (lane < rest) { pp = q.pos[head + lane]; ll = q.len[head + lane]; oo = q.occ[head + lane]; }
  wave_sync();
  if (lane < rest) { q.pos[lane] = pp; q.len[lane] = ll; q.occ[lane] = oo; }
  queued = rest;
  wave_sync();
}
// one byte per lane: bytes [p0, p0 + 64) of the block [P0, P1), any UTF-8
__device__ inline void wc_walk64(const EncModel &m, const uint8_t *__restrict__ text, const unsigned long long *bnd, int ns, unsigned long long s0,
                                 unsigned long long P0, unsigned long long P1, unsigned long long p0, WalkState &st, const WordQueue &q) {
  const int lane = lane_id();
  const unsigned long long lt = lanemask_lt();
  const unsigned long long p = p0 + (unsigned long long)lane;
  const bool in = p >= P0 && p < P1;
  bool valid = false, space = false, first = p == P1;  // (the position behind the block's last byte: the last sentence's end)
  if (in) {
    const int k = wc_sentence_of(bnd, ns, p);
    const unsigned long long s_lo = bnd[k], nbytes = bnd[k + 1] - s_lo;
    first = s_lo == p;
    if (u8_is_start(text + s_lo, p - s_lo, nbytes)) {
      uint32_t len;
      const uint32_t cp = u8_decode_at(text + s_lo, p - s_lo, nbytes, &len);
      if (cp != INVALID_CP) {
        valid = true;
        space = m.cpmap[cp] == CP_SPACE;
      }
    }
  }
  const unsigned long long V = __ballot(valid), S = __ballot(space), F = __ballot(first);
  // the state in front of this lane: the last valid char below it, unless a sentence began since
  const unsigned long long pv = V & lt;
  const int j = pv ? 63 - __clzll((long long)pv) : -1;
  const unsigned long long since = F & lt & ~(j >= 0 ? (2ull << j) - 1ull : 0ull);
  const bool pre_space = since ? true : (j >= 0 ? (bool)((S >> j) & 1ull) : st.carry_space);
  const bool prev_space = first ? true : pre_space;  // ... and in front of this lane's own char
  const bool wstart = valid && !space && prev_space;
  const bool closing = (first && !pre_space) || (valid && space && !prev_space);  // by the sentence's end, or by a space
  const unsigned long long WSM = __ballot(wstart), CM = __ballot(closing);
  const unsigned long long wlt = WSM & lt;
  if (closing) wc_queue_push(q, st.queued + (int)__popcll(CM & lt), bnd, ns, s0, wlt ? p0 + (unsigned long long)(63 - __clzll((long long)wlt)) : st.carry_start, p);
  st.queued += (int)__popcll(CM);
  if (V) {
    const int jl = 63 - __clzll((long long)V);
    st.carry_space = (F & ~((2ull << jl) - 1ull)) ? true : (bool)((S >> jl) & 1ull);
  } else if (F) {
    st.carry_space = true;
  }
  if (WSM) st.carry_start = p0 + (unsigned long long)(63 - __clzll((long long)WSM));
  wave_sync();
}
__global__ __launch_bounds__(BLOCK) void k5w_insert(EncModel m, const uint8_t *__restrict__ text, unsigned long long total,
                                                    const unsigned long long *__restrict__ offsets, unsigned long long n_sent, WordCache wc, int sblk) {
  __shared__ unsigned long long q_pos[NWAVES][WC_QUEUE], q_occ[NWAVES][WC_QUEUE];
  __shared__ uint32_t q_len[NWAVES][WC_QUEUE];
  __shared__ unsigned long long bnd_all[NWAVES][WC_SBLK + 1];
  const int wave = uni((int)(threadIdx.x >> 6)), lane = lane_id();  // (uni: the wave's number is the same in its lanes -- what follows from it stays in scalar registers)
  const unsigned long long lt = lanemask_lt();
  const unsigned long long gw = (unsigned long long)blockIdx.x * NWAVES + wave;
  const unsigned long long n_waves = (unsigned long long)gridDim.x * NWAVES;
  unsigned long long *bnd = bnd_all[wave];
  const WordQueue q{q_pos[wave], q_occ[wave], q_len[wave]};
  WalkState st;
  const unsigned long long n_blk = (n_sent + sblk - 1) / sblk;
  for (unsigned long long blk = gw; blk < n_blk; blk += n_waves) {
    const unsigned long long s0 = blk * (unsigned long long)sblk;
    const int ns = (int)(n_sent - s0 < (unsigned long long)sblk ? n_sent - s0 : (unsigned long long)sblk);
    wave_sync();
    if (lane < ns) bnd[lane] = offsets[s0 + lane];  // sentence k of the block = bytes [bnd[k], bnd[k + 1])
    if (lane == 0) bnd[ns] = offsets[s0 + ns];
    wave_sync();
    const unsigned long long P0 = uni64(bnd[0]), P1 = uni64(bnd[ns]);
    st.carry_space = true;
    st.carry_start = 0;
    int kn = 0;                      // the first boundary the walk has not come to yet,
    unsigned long long next_b = P0;  // and where it is
    for (unsigned long long base = P0 & ~7ull; base <= P1; base += 512) {  // (the position behind the last byte is walked too: the last sentence's end)
      const unsigned long long q0 = base + 8ull * (unsigned long long)lane;
      // the lane's eight bytes (one aligned load while it lies inside the text), what is outside the block counts as white space
      unsigned long long w = 0x2020202020202020ull;
      if (q0 + 8 <= total) {
        w = *reinterpret_cast<const unsigned long long *>(text + q0);
      } else {
        for (int j = 0; j < 8; j++)
          if (q0 + j < total) w = (w & ~(0xffull << (8 * j))) | ((unsigned long long)text[q0 + j] << (8 * j));
      }
      uint32_t inr = 0xffu;  // the bytes of the block
      if (q0 < P0) inr &= q0 + 8 <= P0 ? 0u : 0xffu << (uint32_t)(P0 - q0);
      if (q0 + 8 > P1) inr &= q0 >= P1 ? 0u : 0xffu >> (uint32_t)(q0 + 8 - P1);
      if (inr != 0xffu) {
        unsigned long long keep = 0;  // (0xff in the bytes that count)
        for (int j = 0; j < 8; j++) keep |= ((inr >> j) & 1u) ? 0xffull << (8 * j) : 0ull;
        w = (w & keep) | (0x2020202020202020ull & ~keep);
      }
      const uint32_t end_bit = P1 >= q0 && P1 < q0 + 8 ? 1u << (uint32_t)(P1 - q0) : 0u;  // (white space like everything behind the block: it closes the last word)
      // sentence starts among the lane's bytes (the walk's boundary cursor: uniform)
      uint32_t fb = 0;
      const unsigned long long lim = base + 512 < P1 ? base + 512 : P1;
      while (kn <= ns && next_b < lim) {
        if (next_b >= q0 && next_b < q0 + 8) fb |= 1u << (uint32_t)(next_b - q0);
        kn++;
        next_b = kn <= ns ? uni64(bnd[kn]) : 0ull;
      }
      if (__ballot((w & 0x8080808080808080ull) != 0ull) != 0ull) {  // beyond ASCII somewhere in these 512 bytes: one byte per lane
        for (int sub = 0; sub < 8; sub++) {
          const unsigned long long p0 = base + 64ull * (unsigned long long)sub;
          if (p0 > P1) break;
          if (p0 + 64 > P0) wc_walk64(m, text, bnd, ns, s0, P0, P1, p0, st, q);
          wc_queue_drain(text, total, wc, q, st.queued);
        }
        continue;
      }
      // white space: 0x20 or 9 .. 13 (utils.cpp:99-101), a flag per byte -> eight bits
      uint32_t sp = 0;
      {
        const unsigned long long t = w ^ 0x2020202020202020ull;
        const unsigned long long eq = ~(((t & 0x7f7f7f7f7f7f7f7full) + 0x7f7f7f7f7f7f7f7full) | t) & 0x8080808080808080ull;  // byte == 0x20
        const unsigned long long d = (w | 0x8080808080808080ull) - 0x0909090909090909ull;                                    // bit 7 iff byte >= 9
        const unsigned long long lt5 = ~((d & 0x7f7f7f7f7f7f7f7full) + 0x7b7b7b7b7b7b7b7bull) & 0x8080808080808080ull;         // (byte - 9) mod 128 < 5
        const unsigned long long f = eq | (d & lt5);
        sp = (uint32_t)((((f >> 7) & 0x0101010101010101ull) * 0x0102040810204080ull) >> 56);
      }
      // the state in front of every byte: the byte before it (all are chars here); in front of the lane's first, the lane below's last
      uint32_t below = (uint32_t)__shfl_up((int)sp, 1);
      const uint32_t in0 = lane == 0 ? (st.carry_space ? 1u : 0u) : (below >> 7) & 1u;
      const uint32_t psp = ((sp << 1) | in0) & 0xffu;
      const uint32_t ws_bits = ~sp & (psp | fb) & inr;                            // a word begins: not white space, behind white space or a sentence start
      const uint32_t cl_bits = ((fb & ~psp) | (sp & ~(psp | fb))) & (inr | end_bit) & 0xffu;  // a word ends: by the sentence's end, or by a space
      // where the word that is open in front of this lane began
      const unsigned long long LW = __ballot(ws_bits != 0u);
      const unsigned long long mylast = q0 + (unsigned long long)(31 - __clz((int)(ws_bits | 1u)));
      const unsigned long long lw_lt = LW & lt;
      const int src = lw_lt ? 63 - __clzll((long long)lw_lt) : 0;
      const unsigned long long got = __shfl(mylast, src);
      const unsigned long long open_start = lw_lt ? got : st.carry_start;
      // the lane's closings, in order, to their places in the queue
      const uint32_t nc = (uint32_t)__popc(cl_bits);
      const uint32_t inc = wave_incl_scan(nc);
      int at = st.queued + (int)(inc - nc);
      for (uint32_t c = cl_bits; c; c &= c - 1u) {
        const uint32_t j = (uint32_t)__ffs((int)c) - 1u;
        const uint32_t wb = ws_bits & ((1u << j) - 1u);  // word starts of this lane in front of the closing byte
        const unsigned long long ws = wb ? q0 + (unsigned long long)(31 - __clz((int)wb)) : open_start;
        wc_queue_push(q, at++, bnd, ns, s0, ws, q0 + j);
      }
      st.queued += (int)__shfl((int)inc, 63);
      // carry: the last byte of the step (inside the block: steps but the last are whole), the last word start
      st.carry_space = ((uint32_t)__shfl((int)sp, 63) >> 7) & 1u;
      if (LW) st.carry_start = __shfl(mylast, 63 - __clzll((long long)LW));
      wave_sync();
      wc_queue_drain(text, total, wc, q, st.queued);
    }
  }
  wave_sync();
  if (lane < st.queued) wc_insert_word(text, total, wc, q.pos[lane], q.len[lane], q.occ[lane]);
}

// ---- 2. the table's words as a list ------------------------------------------------------------------------------------------
// The list is in classes of word length, short words first: K5 packs consecutive items into a wavefront's arrays and a pack takes as
// many merge rounds as its longest word -- words of a kind side by side keep the lanes of a pack busy for the same number of rounds.
// (The order inside a class is whatever the atomics make it; nothing depends on it: a word's ids are found through its table slot.)
__device__ inline int wc_len_class(unsigned long long key, int classes) {
  if (classes <= 1) return 0;
  const uint32_t len = (uint32_t)((key & WC_LONG) ? (key >> 40) & 0xffffull : key >> 56);
  return len <= 9 ? 0 : len <= 14 ? 1 : len <= 20 ? 2 : 3;
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
      const unsigned long long u = blk_off[(unsigned long long)c * gridDim.x + blockIdx.x] + r, p = (key & WC_LONG) ? key & WC_POS_MASK : wc.pos[i];
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
// A sentence's word occurrences are the entries of occ from its first possible index on that are not WC_NONE (the host clears the array
// before k5w_insert): the second and third walk over the text are scans of that array, 64 entries = 128 bytes of text per step, no UTF-8.
// Sentence s owns the entries [wc_occ_lo(s), wc_occ_lo(s + 1)): ((byte + s) >> 1 grows from sentence to sentence, and what lies between
// the last word start a sentence can have and the next sentence's first index is never written.)  A wavefront takes sblk <= WC_SBLK
// consecutive sentences (wc_sblk: fewer when the batch has few sentences, so that every wavefront still gets some) and walks their entries as ONE range, whatever the sentences' lengths -- a sentence of 129 bytes is 65 entries, two steps of
// which the second has one lane busy; 64 of them are 65 steps, all full.  The sentence of an entry: the range's boundaries are in LDS, and
// the entries of a step are in order, so a lane counts the boundaries at or below its entry as the wave passes them.
__device__ inline unsigned long long wc_occ_lo(const unsigned long long *__restrict__ offsets, unsigned long long sidx) {
  return (offsets[sidx] + sidx) >> 1;
}
struct SentBlock {
  unsigned long long s0;  // first sentence
  int ns;                 // sentences
  unsigned long long e_lo, e_hi;  // their entries
};
// the block's boundaries -> bnd[0 .. ns]
__device__ inline SentBlock wc_block_begin(const unsigned long long *__restrict__ offsets, unsigned long long n_sent, unsigned long long blk, int sblk,
                                           unsigned long long *bnd) {
  SentBlock B;
  B.s0 = blk * (unsigned long long)sblk;
  B.ns = (int)(n_sent - B.s0 < (unsigned long long)sblk ? n_sent - B.s0 : (unsigned long long)sblk);
  const int lane = lane_id();
  if (lane < B.ns) bnd[lane] = wc_occ_lo(offsets, B.s0 + lane);
  if (lane == 0) bnd[B.ns] = wc_occ_lo(offsets, B.s0 + B.ns);
  wave_sync();
  B.e_lo = bnd[0];
  B.e_hi = bnd[B.ns];
  return B;
}
// sentence (within the block) of entry e of the step [e0, e0 + 64); kn = first boundary not yet passed (uniform, carried from step to step),
// k0 = sentence of the entries below bnd[kn]
__device__ inline int wc_block_sentence(const unsigned long long *bnd, int ns, unsigned long long e0, unsigned long long e, int &kn) {
  int k = kn - 1;
  while (kn <= ns && bnd[kn] < e0 + 64) {
    k += e >= bnd[kn] ? 1 : 0;
    kn++;
  }
  return k;
}
__global__ __launch_bounds__(BLOCK) void k5w_count(const unsigned long long *__restrict__ offsets, unsigned long long n_sent, WordCache wc,
                                                   int n_fixed /* bos + eos */, uint32_t *__restrict__ counts, int sblk) {
  __shared__ unsigned long long bnd_all[NWAVES][WC_SBLK + 1];
  __shared__ uint32_t acc_all[NWAVES][WC_SBLK];
  const int wave = uni((int)(threadIdx.x >> 6)), lane = lane_id();  // (uni: the wave's number is the same in its lanes -- what follows from it stays in scalar registers)
  unsigned long long *bnd = bnd_all[wave];
  uint32_t *acc = acc_all[wave];
  const unsigned long long gw = (unsigned long long)blockIdx.x * NWAVES + wave;
  const unsigned long long n_waves = (unsigned long long)gridDim.x * NWAVES;
  const unsigned long long n_blk = (n_sent + sblk - 1) / sblk;
  for (unsigned long long blk = gw; blk < n_blk; blk += n_waves) {
    acc[lane] = 0;
    const SentBlock B = wc_block_begin(offsets, n_sent, blk, sblk, bnd);
    int kn = 1;
    // (the next step's entries are on their way while this step's table look-ups are: each step is a chain of two dependent loads)
    uint32_t o_next = B.e_lo + (unsigned long long)lane < B.e_hi ? wc.occ[B.e_lo + lane] : WC_NONE;
    for (unsigned long long e0 = B.e_lo; e0 < B.e_hi; e0 += 64) {
      const unsigned long long e = e0 + (unsigned long long)lane;
      const uint32_t o = o_next;
      o_next = e + 64 < B.e_hi ? wc.occ[e + 64] : WC_NONE;
      const int k = wc_block_sentence(bnd, B.ns, e0, e, kn);
      if (o != WC_NONE) {
        unsigned long long off;
        uint32_t n;
        wc_result(wc, o, &off, &n);
        atomicAdd(&acc[k], n);
      }
    }
    wave_sync();
    if (lane < B.ns) counts[B.s0 + lane] = acc[lane] + (uint32_t)n_fixed;
    wave_sync();
  }
}
__global__ __launch_bounds__(BLOCK) void k5w_scatter(EncModel m, const unsigned long long *__restrict__ offsets, unsigned long long n_sent, WordCache wc,
                                                     const int32_t *__restrict__ uids /* K5's scratch */, int bos, int eos, int reverse,
                                                     const unsigned long long *__restrict__ out_off, int32_t *__restrict__ ids_out, int sblk) {
  __shared__ unsigned long long bnd_all[NWAVES][WC_SBLK + 1], oo_all[NWAVES][WC_SBLK + 1];
  const int wave = uni((int)(threadIdx.x >> 6)), lane = lane_id();  // (uni: the wave's number is the same in its lanes -- what follows from it stays in scalar registers)
  unsigned long long *bnd = bnd_all[wave], *oo = oo_all[wave];
  const unsigned long long gw = (unsigned long long)blockIdx.x * NWAVES + wave;
  const unsigned long long n_waves = (unsigned long long)gridDim.x * NWAVES;
  const unsigned long long n_blk = (n_sent + sblk - 1) / sblk;
  const unsigned long long fixed = (unsigned long long)((bos ? 1 : 0) + (eos ? 1 : 0));
  for (unsigned long long blk = gw; blk < n_blk; blk += n_waves) {
    const SentBlock B = wc_block_begin(offsets, n_sent, blk, sblk, bnd);
    if (lane < B.ns) oo[lane] = out_off[B.s0 + lane];
    if (lane == 0) oo[B.ns] = out_off[B.s0 + B.ns];
    wave_sync();
    if (lane < B.ns) {
      const unsigned long long o0 = oo[lane], n_ids = oo[lane + 1] - o0;
      if (bos) ids_out[o0 + (reverse ? n_ids - 1 : 0)] = m.bos_id;
      if (eos) ids_out[o0 + (reverse ? 0 : n_ids - 1)] = m.eos_id;
    }
    // The block's ids without bos / eos are one run: sentence k's start at oo[k] - oo[0] - k * fixed of it.
    unsigned long long q = 0;  // ids of the block laid down so far
    int kn = 1;
    // A step is a chain of three dependent loads -- the entry, the word's slot, its ids -- so the entries run two steps ahead and the
    // slots one.
    auto entry = [&](unsigned long long e) -> uint32_t { return e < B.e_hi ? wc.occ[e] : WC_NONE; };
    uint32_t o_a = entry(B.e_lo + lane), o_b = entry(B.e_lo + 64 + lane);
    unsigned long long off_a = 0;
    uint32_t n_a = 0;
    if (o_a != WC_NONE) wc_result(wc, o_a, &off_a, &n_a);
    for (unsigned long long e0 = B.e_lo; e0 < B.e_hi; e0 += 64) {
      const unsigned long long e = e0 + (unsigned long long)lane;
      const unsigned long long off = off_a;
      const uint32_t n = n_a;
      // the next step's slot, the entry of the one after
      off_a = 0;
      n_a = 0;
      if (o_b != WC_NONE) wc_result(wc, o_b, &off_a, &n_a);
      o_b = entry(e + 128);
      const int k = wc_block_sentence(bnd, B.ns, e0, e, kn);
      const uint32_t inc = wave_incl_scan(n);
      if (n) {
        const unsigned long long o0 = oo[k], n_ids = oo[k + 1] - o0;
        const unsigned long long mine = q + inc - n - (o0 - oo[0] - (unsigned long long)k * fixed) + (bos ? 1 : 0);  // place in the sentence
        int32_t *out = ids_out + o0;
        for (uint32_t j = 0; j < n; j++) out[reverse ? n_ids - 1 - (mine + j) : mine + j] = uids[off + j];
      }
      q += (unsigned long long)__shfl(inc, 63);
    }
    wave_sync();
  }
}

// ---- launchers -----------------------------------------------------------------------------------------------------------------
// sentences a wavefront of the flat walks takes at a time: up to WC_SBLK, fewer while that leaves wavefronts without any
static inline int wc_sblk(unsigned long long n_sent) {
  const std::shared_ptr<const Config> C = cfg();
  if (C->wc_sblk.set) {  // (tests: small batches with many sentences per wavefront)
    const int v = (int)C->wc_sblk.i;
    return v < 1 ? 1 : v > WC_SBLK ? WC_SBLK : v;
  }
  const unsigned long long per = n_sent / (256ull * 16 * NWAVES);
  return per < 1 ? 1 : per > (unsigned long long)WC_SBLK ? WC_SBLK : (int)per;
}
static inline unsigned int wave_grid(unsigned long long n_items, unsigned int max_blocks) {
  unsigned long long b = (n_items + NWAVES - 1) / NWAVES;
  if (b > max_blocks) b = max_blocks;
  return b ? (unsigned int)b : 1u;
}
static inline unsigned int wcache_blocks(const WordCache &wc) { return (unsigned int)((wc.mask + WC_CBLK) / WC_CBLK); }
static int wcache_classes() {
  const int c = (int)cfg()->k5_classes.i;
  return c <= 1 ? 1 : WC_CLASSES;
}
unsigned long long wcache_count_blocks(const WordCache &wc) { return (unsigned long long)wcache_blocks(wc) * (unsigned long long)wcache_classes(); }
void launch_wcache_insert(const EncModel &m, const uint8_t *text, unsigned long long total, const unsigned long long *offsets, unsigned long long n_sent,
                          const WordCache &wc, hipStream_t st) {
  const int sblk = wc_sblk(n_sent);
  hipLaunchKernelGGL(k5w_insert, dim3(wave_grid((n_sent + sblk - 1) / sblk, 256 * 16)), dim3(BLOCK), 0, st, m, text, total, offsets, n_sent, wc, sblk);
}
void launch_wcache_count_slots(const WordCache &wc, uint32_t *blk_cnt, hipStream_t st) {
  hipLaunchKernelGGL(k5w_count_slots, dim3(wcache_blocks(wc)), dim3(BLOCK), 0, st, wc, blk_cnt, wcache_classes());
}
void launch_wcache_list(const WordCache &wc, const unsigned long long *blk_off, unsigned long long n_table, unsigned long long *ustart,
                        unsigned long long *uend, uint32_t *uslot, hipStream_t st) {
  hipLaunchKernelGGL(k5w_list, dim3(wcache_blocks(wc)), dim3(BLOCK), 0, st, wc, blk_off, n_table, ustart, uend, uslot, wcache_classes());
}
void launch_wcache_count(const unsigned long long *offsets, unsigned long long n_sent, const WordCache &wc, int n_fixed, uint32_t *counts, hipStream_t st) {
  const int sblk = wc_sblk(n_sent);
  hipLaunchKernelGGL(k5w_count, dim3(wave_grid((n_sent + sblk - 1) / sblk, 256 * 16)), dim3(BLOCK), 0, st, offsets, n_sent, wc, n_fixed, counts, sblk);
}
void launch_wcache_scatter(const EncModel &m, const unsigned long long *offsets, unsigned long long n_sent, const WordCache &wc, const int32_t *uids, int bos,
                           int eos, int reverse, const unsigned long long *out_off, int32_t *ids_out, hipStream_t st) {
  const int sblk = wc_sblk(n_sent);
  hipLaunchKernelGGL(k5w_scatter, dim3(wave_grid((n_sent + sblk - 1) / sblk, 256 * 16)), dim3(BLOCK), 0, st, m, offsets, n_sent, wc, uids, bos, eos, reverse, out_off,
                     ids_out, sblk);
}

}  // namespace yttm
