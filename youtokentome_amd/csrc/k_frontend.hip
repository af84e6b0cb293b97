// k_frontend.hip -- K1 (char histogram + UTF-8 scan) and K2 (word split + hash dedup + tile build) for gfx950.
//
// Replaces the O(N) front end of the reference trainer:
//   K1  compute_char_count            bpe.cpp:839-857   (+ UTF8Iterator utf8.h:21-64, chars_to_utf8 utf8.cpp:37-74)
//   K2  remove_rare_chars             bpe.cpp:357-380
//       compute_word_count            bpe.cpp:388-418   (+ merge of per-thread maps :1029-1044)
//       build_linked_list (layout)    bpe.cpp:436-478   -> flat token tiles instead of linked lists
// All of it is HBM-bound byte/integer work: coalesced 16 B/lane loads staged through LDS, LDS-private histograms,
// one global atomic per wave/block for cursors.  No MFMA.
#include <algorithm>
#include "yttm_device.h"
#include "yttm_kernels.h"

namespace yttm {

constexpr int FE_BYTES_PER_THREAD = 16;
constexpr int FE_CHUNK = BLOCK * FE_BYTES_PER_THREAD;  // 4096 bytes per block iteration
constexpr int LH_BINS = 2048;                          // LDS-private histogram covers code points < 0x800

// byte k of the 24-byte register window W[6] (k is a compile-time constant after unrolling)
#define WB(k) ((W[(k) >> 2] >> (8 * ((k)&3))) & 0xffu)

// The 24-byte window W of a lane (4 bytes before its 16, 4 behind) as text of ASCII and "simple" three-byte chars: true iff every byte is
// ASCII, a continuation byte, or a lead byte E1 / E3 .. EC / EE / EF, and for every position p in 2 .. 23: byte p is a continuation iff a
// lead stands at p - 1 or p - 2 (so every lead that matters to bytes 4 .. 19 has its two continuations, and every continuation among them
// belongs to a lead).  *cont: bit p = byte p is a continuation; *sp: bit p = byte p is ASCII white space (0x20, 9 .. 13).
__device__ inline bool simple3_window(const uint32_t (&W)[6], uint32_t *cont, uint32_t *sp) {
  uint32_t Cm = 0, Lm = 0, Sm = 0, bad = 0;
#pragma unroll
  for (int w = 0; w < 6; w++) {
    const uint32_t x = W[w];
    const uint32_t hi = x & 0x80808080u;                  // bytes >= 0x80
    const uint32_t c = hi & ~(x << 1);                    // 10xxxxxx (the shift moves bit 6 onto bit 7 of its own byte)
    const uint32_t l = hi & (x << 1);                     // 11xxxxxx
    const uint32_t e = l & (x << 2) & ~(x << 3);          // 1110xxxx
    bad |= l & ~e;                                        // a lead byte of a two- or four-byte char, or F8 .. FF
    // E0 (overlong forms), E2 (U+2581, the white-space marker, lives there), ED (surrogates): exact zero-byte tests of x ^ the byte
    uint32_t t = x ^ 0xE0E0E0E0u;
    bad |= ~(((t & 0x7f7f7f7fu) + 0x7f7f7f7fu) | t) & 0x80808080u;
    t = x ^ 0xE2E2E2E2u;
    bad |= ~(((t & 0x7f7f7f7fu) + 0x7f7f7f7fu) | t) & 0x80808080u;
    t = x ^ 0xEDEDEDEDu;
    bad |= ~(((t & 0x7f7f7f7fu) + 0x7f7f7f7fu) | t) & 0x80808080u;
    // ASCII white space (k_scan_bytes' ASCII path has the derivation; here bytes >= 0x80 exist and are masked out)
    t = x ^ 0x20202020u;
    const uint32_t eq = ~(((t & 0x7f7f7f7fu) + 0x7f7f7f7fu) | t) & 0x80808080u;
    const uint32_t d = (x | 0x80808080u) - 0x09090909u;
    const uint32_t lt5 = ~((d & 0x7f7f7f7fu) + 0x7b7b7b7bu) & 0x80808080u;
    const uint32_t f = (eq | (d & lt5)) & ~hi & 0x80808080u;
    Cm |= ((((c >> 7) & 0x01010101u) * 0x01020408u) >> 24) << (4 * w);
    Lm |= ((((e >> 7) & 0x01010101u) * 0x01020408u) >> 24) << (4 * w);
    Sm |= ((((f >> 7) & 0x01010101u) * 0x01020408u) >> 24) << (4 * w);
  }
  *cont = Cm;
  *sp = Sm;
  return bad == 0u && ((Cm ^ ((Lm << 1) | (Lm << 2))) & 0x00fffffcu) == 0u;
}

// MODE 0: histogram + number of decode steps + number of segment starts.
// MODE 1: append the byte offsets of segment starts (a segment = maximal run of non-space chars) to seg_pos.
// HK (MODE 0): slots of an LDS hash code point -> count for the chars beyond the direct bins.  A text of three-byte chars (CJK: a few
// thousand distinct ones, Zipfian) otherwise sends one 64-bit global atomic per char to a few thousand addresses -- measured 416 ms per GB
// against 1.9 ms for ASCII text.  The launcher picks the variant from a sample of the text (the hash costs occupancy ASCII text does not need).
template <int MODE, int HK = 0>
__global__ __launch_bounds__(BLOCK) void k_scan_bytes(const uint8_t *__restrict__ text, unsigned long long n,
                                                      unsigned long long *__restrict__ hist,
                                                      unsigned long long *__restrict__ counters /* [0]=steps [1]=segs */,
                                                      unsigned long long *__restrict__ seg_pos,
                                                      const unsigned long long *__restrict__ chunk_off /* MODE 1: segments before the chunk */,
                                                      uint32_t *__restrict__ chunk_segs /* MODE 0: segments of the chunk (out) */,
                                                      unsigned long long chunk_lo, unsigned long long chunk_hi /* the chunks of this launch: the
                                                      whole text, or the part of it that has arrived (gpu_ctx.cpp upload_corpus_fd) */) {
  __shared__ __attribute__((aligned(16))) uint8_t stage[FE_CHUNK + 16];  // [0..3] halo before, [4..4+FE_CHUNK) main, then halo after
  __shared__ unsigned int lh[MODE == 0 ? LH_BINS : 1];
  // ASCII chars (nearly all of most texts, a handful of distinct values) are counted in LC_COPIES lane-indexed copies of the first 128 bins,
  // LC_STRIDE words apart (odd: lanes with the same char and different copies fall into different banks): one LDS atomic per byte into FOUR hot
  // bins had the lanes of a wave queue up sixteen deep (1 GB of 'abcd ' in 1.93 ms = 0.52 TB/s)
  constexpr int LC_COPIES = 16, LC_STRIDE = 129;
  __shared__ unsigned int lc[MODE == 0 ? LC_COPIES * LC_STRIDE : 1];
  __shared__ unsigned int hkey[HK ? HK : 1], hval[HK ? HK : 1];
  __shared__ uint32_t scan_lds[NWAVES];
  __shared__ uint32_t seg_part[NWAVES];  // MODE 0: the waves' segment counts of the chunk just scanned
  const int tid = (int)threadIdx.x;
  if (MODE == 0) {
    for (int b = tid; b < LH_BINS; b += BLOCK) lh[b] = 0;
    for (int b = tid; b < LC_COPIES * LC_STRIDE; b += BLOCK) lc[b] = 0;
    for (int b = tid; b < HK; b += BLOCK) { hkey[b] = 0; hval[b] = 0; }
  }
  unsigned long long my_steps = 0, my_segs = 0;
  // one char beyond the direct bins (cp >= 0x800): the workgroup's LDS hash first, the global histogram for what finds no slot there
  auto count_wide = [&](uint32_t cp) {
    bool done = false;
    if (HK) {  // (cp >= 0x800: never 0, the empty key)
      unsigned int h = (cp * 0x9E3779B1u) & (unsigned int)(HK - 1);
      h ^= (cp * 0x9E3779B1u) >> 19;
      h &= (unsigned int)(HK - 1);
      for (int probe = 0; probe < 4 && !done; probe++) {
        unsigned int kcur = __hip_atomic_load(&hkey[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (kcur == 0u) {
          kcur = atomicCAS(&hkey[h], 0u, cp);
          if (kcur == 0u) kcur = cp;
        }
        if (kcur == cp) {
          atomicAdd(&hval[h], 1u);
          done = true;
        }
        h = (h + 1) & (unsigned int)(HK - 1);
      }
    }
    if (!done) atomicAdd(&hist[cp], 1ull);
  };
  const unsigned long long first_chunk = chunk_lo + blockIdx.x;
  for (unsigned long long chunk = first_chunk; chunk < chunk_hi; chunk += gridDim.x) {
    const unsigned long long c0 = chunk * FE_CHUNK;
    __syncthreads();  // previous iteration done with `stage`
    if (MODE == 0 && tid == 0 && chunk != first_chunk) {  // the previous chunk's segment count (its waves' parts are in: the barrier above)
      uint32_t t = 0;
      for (int w = 0; w < NWAVES; w++) t += seg_part[w];
      chunk_segs[chunk - gridDim.x] = t;
    }
    {
      // coalesced 16 B/lane load of the chunk (text base is 256 B aligned: hipMalloc), zero fill past the end
      unsigned long long g = c0 + (unsigned long long)tid * 16;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (g + 16 <= n) {
        v = *reinterpret_cast<const uint4 *>(text + g);
      } else if (g < n) {
        uint32_t t[4] = {0, 0, 0, 0};
        for (int j = 0; j < 16 && g + j < n; j++) t[j >> 2] |= (uint32_t)text[g + j] << (8 * (j & 3));
        v = make_uint4(t[0], t[1], t[2], t[3]);
      }
      *reinterpret_cast<uint4 *>(&stage[4 + tid * 16]) = v;
      if (tid == 0) {
        uint32_t h = 0;
        if (c0 >= 4) h = *reinterpret_cast<const uint32_t *>(text + c0 - 4);
        *reinterpret_cast<uint32_t *>(&stage[0]) = h;
      }
      if (tid == 1) {
        uint32_t h = 0;
        unsigned long long e = c0 + FE_CHUNK;
        for (int j = 0; j < 4; j++)
          if (e + j < n) h |= (uint32_t)text[e + j] << (8 * j);
        *reinterpret_cast<uint32_t *>(&stage[4 + FE_CHUNK]) = h;
      }
    }
    __syncthreads();
    const unsigned long long i0 = c0 + (unsigned long long)tid * 16;
    uint32_t W[6];
    {
      W[0] = *reinterpret_cast<const uint32_t *>(&stage[tid * 16]);
      uint4 m = *reinterpret_cast<const uint4 *>(&stage[4 + tid * 16]);
      W[1] = m.x; W[2] = m.y; W[3] = m.z; W[4] = m.w;
      W[5] = *reinterpret_cast<const uint32_t *>(&stage[4 + tid * 16 + 16]);
    }
    uint32_t seg_mask = 0;  // bit j: byte i0+j starts a segment
    uint32_t n_starts = 0;
    uint32_t s3_cont = 0, s3_sp = 0;
    const bool all_ascii = ((W[0] | W[1] | W[2] | W[3] | W[4] | W[5]) & 0x80808080u) == 0;
    if (all_ascii && i0 + 20 <= n) {
      // ASCII text, four bytes per instruction: every byte is a char (16 decode steps); a byte is white space iff it is 0x20 or in
      // 9 .. 13 (utils.cpp:99-101) -- flags in bit 7 of each byte, no carries between bytes (all bytes are below 0x80) -- and starts a
      // segment iff it is not white space and its left neighbour is.  (The byte-by-byte loop below -- any UTF-8, the text's last bytes
      // -- costs ~35 instructions per byte: both scans ran at 0.5 TB/s on ASCII text.)
      uint32_t sp16 = 0;
#pragma unroll
      for (int w = 0; w < 4; w++) {
        const uint32_t x = W[1 + w];
        const uint32_t t = x ^ 0x20202020u;
        const uint32_t eq = ~(((t & 0x7f7f7f7fu) + 0x7f7f7f7fu) | t) & 0x80808080u;  // byte == 0x20
        const uint32_t d = (x | 0x80808080u) - 0x09090909u;                            // byte + 119: bit 7 iff byte >= 9
        const uint32_t lt5 = ~((d & 0x7f7f7f7fu) + 0x7b7b7b7bu) & 0x80808080u;        // (byte - 9) mod 128 < 5
        const uint32_t f = eq | (d & lt5 & 0x80808080u);
        sp16 |= ((((f >> 7) & 0x01010101u) * 0x01020408u) >> 24) << (4 * w);            // bits 7, 15, 23, 31 -> four adjacent bits
      }
      const uint32_t p1 = WB(3);
      const uint32_t prev0 = (i0 == 0 || p1 == 32u || (p1 >= 9u && p1 <= 13u)) ? 1u : 0u;
      seg_mask = ~sp16 & ((sp16 << 1) | prev0) & 0xffffu;
      n_starts = 16;
      if (MODE == 0) {
        unsigned int *mine = &lc[(tid & (LC_COPIES - 1)) * LC_STRIDE];
#pragma unroll
        for (int k = 0; k < 16; k++)
          if (!((sp16 >> k) & 1u)) atomicAdd(&mine[WB(4 + k)], 1u);
      }
    } else if (i0 + 20 <= n && simple3_window(W, &s3_cont, &s3_sp)) {
      // Text of three-byte chars (CJK: E4 .. E9 leads) with ASCII in between, round 5.  The byte-by-byte path below decodes at EVERY position
      // -- a continuation byte looks back for its lead and decodes that -- ~35 instructions per byte: 22 ms per GB of such text against 0.8
      // for ASCII.  Here the lane's 24-byte window is classified four bytes per instruction (simple3_window): every byte is ASCII, a
      // continuation, or a lead E1 / E3 .. EC / EE / EF (with two continuations, such a lead IS a valid char: no overlong forms, no
      // surrogates, never U+2581), and the leads and continuations fit together.  Then a char starts at every byte that is no continuation,
      // white space is ASCII white space, and only the leads' code points have to be put together.  Any other byte in the window (two- and
      // four-byte leads, E0 / E2 / ED, stray continuations) sends the lane to the exact path -- same counts, same starts.
      const uint32_t own_start = (~s3_cont >> 4) & 0xffffu;  // bit j: byte i0 + j starts a char
      const uint32_t sp16 = (s3_sp >> 4) & 0xffffu;          // bit j: byte i0 + j is (ASCII) white space
      const uint32_t prev0 = (i0 == 0 || ((s3_sp >> 3) & 1u)) ? 1u : 0u;
      seg_mask = own_start & ~sp16 & ((sp16 << 1) | prev0);
      n_starts = (uint32_t)__popc(own_start);
      if (MODE == 0) {
        unsigned int *mine = &lc[(tid & (LC_COPIES - 1)) * LC_STRIDE];
#pragma unroll
        for (int k = 0; k < 16; k++) {
          if (!((own_start >> k) & 1u) || ((sp16 >> k) & 1u)) continue;
          const uint32_t b = WB(4 + k);
          if (b < 0x80u) atomicAdd(&mine[b], 1u);
          else count_wide(((b & 0x0fu) << 12) | ((WB(5 + k) & 0x3fu) << 6) | (WB(6 + k) & 0x3fu));
        }
      }
    } else if (i0 < n) {
#pragma unroll
      for (int k = 4; k < 20; k++) {
        const unsigned long long gi = i0 + (unsigned long long)(k - 4);
        if (gi < n) {
          const uint32_t b = WB(k);
          bool start = true;
          uint32_t cp = b;
          if (!all_ascii) {
            if (u8_cont(b)) {
#pragma unroll
              for (int d = 1; d <= 3; d++) {
                if ((unsigned long long)d > gi) break;
                const uint32_t c = WB(k - d);
                if (u8_cont(c)) continue;
                uint32_t len;
                uint32_t q = u8_decode(c, WB(k - d + 1), WB(k - d + 2), WB(k - d + 3), n - (gi - d), &len);
                start = !(q != INVALID_CP && len > (uint32_t)d);
                break;
              }
              cp = INVALID_CP;  // a continuation byte that starts a char is an invalid char
            } else if (b >= 0x80u) {
              uint32_t len;
              cp = u8_decode(b, WB(k + 1), WB(k + 2), WB(k + 3), n - gi, &len);
            }
          }
          if (start) {
            n_starts++;
            const bool space = cp != INVALID_CP && cp_is_space(cp);
            if (MODE == 0) {
              if (cp != INVALID_CP && !space) {
                if (cp < 128u) {
                  atomicAdd(&lc[(tid & (LC_COPIES - 1)) * LC_STRIDE + (int)cp], 1u);
                } else if (cp < (uint32_t)LH_BINS) {
                  atomicAdd(&lh[cp], 1u);
                } else {
                  count_wide(cp);
                }
              }
            }
            if (!space) {
              // previous char is a space (or start of text)?  ASCII space byte, or the 3 bytes E2 96 81 ("▁").
              bool prev_space = gi == 0;
              if (!prev_space) {
                const uint32_t p1 = WB(k - 1);
                prev_space = (p1 == 32u || (p1 >= 9u && p1 <= 13u)) ||
                             (gi >= 3 && p1 == 0x81u && WB(k - 2) == 0x96u && WB(k - 3) == 0xe2u);
              }
              if (prev_space) seg_mask |= 1u << (k - 4);
            }
          }
        }
      }
    }
    my_steps += n_starts;
    const uint32_t nseg = (uint32_t)__popc(seg_mask);
    my_segs += nseg;
    if (MODE == 0) {
      // Per chunk, for the pass that writes the segment starts: where a chunk's segments go is then known without a cursor -- one returning
      // atomic per chunk on ONE address (244 k of them at 1 GB, ~12 ns each) was 2.9 of that pass's 3.5 ms.  (The count rides on the
      // barrier the next iteration starts with.)
      uint32_t ws = nseg;
      for (int o = 32; o > 0; o >>= 1) ws += __shfl_down(ws, o);
      if (lane_id() == 0) seg_part[tid >> 6] = ws;
    }
    if (MODE == 1) {
      uint32_t total;
      uint32_t off = block_excl_scan(nseg, scan_lds, &total);
      unsigned long long o = chunk_off[chunk] + off;
      uint32_t m = seg_mask;
      while (m) {
        int j = __ffs((int)m) - 1;
        m &= m - 1;
        seg_pos[o++] = i0 + (unsigned long long)j;
      }
    }
  }
  if (MODE == 0) {
    __syncthreads();
    if (tid == 0 && chunk_hi > first_chunk) {  // the last chunk this block scanned
      uint32_t t = 0;
      for (int w = 0; w < NWAVES; w++) t += seg_part[w];
      const unsigned long long last = first_chunk + ((chunk_hi - 1 - first_chunk) / gridDim.x) * (unsigned long long)gridDim.x;
      chunk_segs[last] = t;
    }
    unsigned long long s = wave_sum_u64(my_steps);
    unsigned long long g = wave_sum_u64(my_segs);
    if (lane_id() == 0) {
      if (s) atomicAdd(&counters[0], s);
      if (g) atomicAdd(&counters[1], g);
    }
    __syncthreads();
    for (int b = tid; b < LH_BINS; b += BLOCK) {
      unsigned int v = lh[b];
      if (b < 128)
        for (int c = 0; c < LC_COPIES; c++) v += lc[c * LC_STRIDE + b];
      if (v) atomicAdd(&hist[b], (unsigned long long)v);
    }
    for (int b = tid; b < HK; b += BLOCK) {
      const unsigned int kcp = hkey[b], v = hval[b];
      if (kcp && v) atomicAdd(&hist[kcp], (unsigned long long)v);
    }
  }
}

// Compact the non-zero bins of the dense code-point histogram into (cp,count) lists.
__global__ __launch_bounds__(BLOCK) void k_hist_compact(const unsigned long long *__restrict__ hist, uint32_t *__restrict__ cps,
                                                        unsigned long long *__restrict__ cnts, unsigned int *__restrict__ n_out,
                                                        unsigned int cap) {
  unsigned int cp = blockIdx.x * BLOCK + threadIdx.x;
  if (cp >= N_CODEPOINTS) return;
  unsigned long long c = hist[cp];
  if (c) {
    unsigned int o = atomicAdd(n_out, 1u);
    if (o < cap) { cps[o] = cp; cnts[o] = c; }
  }
}

// ------------------------------------------------------------------------------------------------- K2: word dedup
// One thread per segment.  The word is the segment's kept chars (cpmap != DROP, valid UTF-8); words are identified by
// their token-id sequence.  Hash table slot = (tag:24 | representative byte offset:40) claimed with one 64-bit CAS, so
// a matching tag is always verified against the representative's bytes: the dedup is exact, never probabilistic.

__device__ inline unsigned long long word_hash_step(unsigned long long h, uint32_t id) {
  return (h ^ (unsigned long long)id) * 0x100000001b3ull + 0x9e3779b97f4a7c15ull;
}

// walks a segment starting at byte `pos`; returns number of kept chars and the hash
__device__ inline uint32_t seg_scan(const uint8_t *__restrict__ text, unsigned long long n, const uint32_t *__restrict__ cpmap,
                                    unsigned long long pos, unsigned long long *hash_out) {
  unsigned long long h = 0xcbf29ce484222325ull;
  uint32_t L = 0;
  unsigned long long i = pos;
  while (i < n) {
    uint32_t len;
    uint32_t cp = u8_decode_at(text, i, n, &len);
    if (cp != INVALID_CP) {
      uint32_t id = cpmap[cp];
      if (id == CP_SPACE) break;
      if (id != CP_DROP) { h = word_hash_step(h, id); L++; }
    }
    i += len;
  }
  *hash_out = mix64(h ^ ((unsigned long long)L << 48));
  return L;
}

// exact comparison of the kept-token sequences of the segments at a and b
__device__ inline bool seg_equal(const uint8_t *__restrict__ text, unsigned long long n, const uint32_t *__restrict__ cpmap,
                                 unsigned long long a, unsigned long long b) {
  if (a == b) return true;
  for (;;) {
    uint32_t ia = CP_SPACE, ib = CP_SPACE;
    while (a < n) {
      uint32_t len;
      uint32_t cp = u8_decode_at(text, a, n, &len);
      a += len;
      if (cp == INVALID_CP) continue;
      uint32_t id = cpmap[cp];
      if (id == CP_DROP) continue;
      ia = id;
      break;
    }
    while (b < n) {
      uint32_t len;
      uint32_t cp = u8_decode_at(text, b, n, &len);
      b += len;
      if (cp == INVALID_CP) continue;
      uint32_t id = cpmap[cp];
      if (id == CP_DROP) continue;
      ib = id;
      break;
    }
    if (ia != ib) return false;
    if (ia == CP_SPACE) return true;  // both ended (space or end of text)
  }
}

// ---- the common case without byte-serial loads -------------------------------------------------------------------------------
// seg_scan / seg_equal walk a segment char by char: per char a byte load and a cpmap load, each waiting for the one before.  Most
// words are a few ASCII letters, none dropped: then the word's tokens ARE its bytes ("pure").  The 16 bytes at the segment start
// come with three aligned 8-byte loads, the ASCII half of cpmap sits in LDS, and two pure words of equal length are compared as
// bytes -- one wide load of the representative instead of a walk.  Anything else (a byte >= 0x80, a dropped char, a word longer
// than what the wide loads cover, the last bytes of the text) continues in / falls back to the exact walk.
// A word of at most seven ASCII chars IS its key in the word table: bit 63 | chars:3 << 56 | the chars (below) -- equal keys, equal words, no
// look at a representative's bytes (one of the three random HBM accesses of a probe; four words in five of random 'abcd ' text, two in
// three of English-like text).  Where such a word's bytes are comes from the slot's entry in a third array, written by whoever claims it.
constexpr unsigned long long WH_SHORT = 1ull << 63;
__device__ inline void load16(const uint8_t *__restrict__ text, unsigned long long pos, unsigned long long &w0, unsigned long long &w1) {
  const unsigned long long *base = reinterpret_cast<const unsigned long long *>(text + (pos & ~7ull));
  const unsigned long long lo = base[0], mid = base[1], hi = base[2];
  const unsigned int sh = (unsigned int)(pos & 7ull) * 8u;
  w0 = sh ? (lo >> sh) | (mid << (64u - sh)) : lo;
  w1 = sh ? (mid >> sh) | (hi << (64u - sh)) : mid;
}
// like seg_scan; *pure = the word is L single-byte chars, consecutive from pos, none dropped (so its bytes identify it)
__device__ inline uint32_t seg_scan_fast(const uint8_t *__restrict__ text, unsigned long long n, const uint32_t *__restrict__ cpmap,
                                         const uint32_t *cp_ascii /* LDS: cpmap[0..128) */, unsigned long long pos, unsigned long long *hash_out,
                                         bool *pure, unsigned long long *short_key /* the word's key if it is at most 7 ASCII chars, else 0 */) {
  unsigned long long h = 0xcbf29ce484222325ull;
  uint32_t L = 0;
  unsigned long long i = pos;
  bool is_pure = true, done = false;
  unsigned long long packed = 0;  // the first seven kept chars, when they are ASCII (a char and its id determine each other)
  bool ascii_only = true;
  while (!done && i + 24 <= n) {  // 16 bytes at a time while they are ASCII
    unsigned long long w0, w1;
    load16(text, i, w0, w1);
    int k = 0;
    for (; k < 16; k++) {
      const uint32_t b = (uint32_t)((k < 8 ? w0 >> (8 * k) : w1 >> (8 * (k - 8))) & 0xffull);
      if (b >= 0x80u) break;
      const uint32_t id = cp_ascii[b];
      if (id == CP_SPACE) { done = true; break; }
      if (id == CP_DROP) { is_pure = false; continue; }
      h = word_hash_step(h, id);
      if (L < 7) packed |= (unsigned long long)b << (8 * L);
      L++;
    }
    i += (unsigned long long)k;
    if (k < 16) break;  // a space (done) or a byte that needs the exact decode
  }
  if (!done) {  // the exact walk from where the fast one stopped
    while (i < n) {
      uint32_t len;
      const uint32_t cp = u8_decode_at(text, i, n, &len);
      if (cp != INVALID_CP) {
        const uint32_t id = cpmap[cp];
        if (id == CP_SPACE) break;
        if (id != CP_DROP) {
          h = word_hash_step(h, id);
          if (cp >= 0x80u) ascii_only = false;
          else if (L < 7) packed |= (unsigned long long)cp << (8 * L);
          L++;
        } else {
          is_pure = false;
        }
        if (len != 1) is_pure = false;
      } else {
        is_pure = false;
      }
      i += len;
    }
  }
  *hash_out = mix64(h ^ ((unsigned long long)L << 48));
  *pure = is_pure;
  *short_key = (ascii_only && L >= 1 && L <= 7) ? (WH_SHORT | ((unsigned long long)L << 56) | packed) : 0ull;
  return L;
}
// two PURE words of L chars each: equal iff their L bytes are
__device__ inline bool bytes_equal(const uint8_t *__restrict__ text, unsigned long long n, unsigned long long a, unsigned long long b, uint32_t L) {
  if (a == b) return true;
  uint32_t done = 0;
  while (done < L) {
    if (a + done + 24 > n || b + done + 24 > n) {  // the last bytes of the text: one at a time
      for (; done < L; done++)
        if (text[a + done] != text[b + done]) return false;
      return true;
    }
    unsigned long long a0, a1, b0, b1;
    load16(text, a + done, a0, a1);
    load16(text, b + done, b0, b1);
    const uint32_t left = L - done;
    if (left >= 16) {
      if (a0 != b0 || a1 != b1) return false;
    } else if (left > 8) {
      const unsigned long long m = ~0ull >> (8 * (16 - left));
      if (a0 != b0 || ((a1 ^ b1) & m)) return false;
    } else {
      const unsigned long long m = ~0ull >> (8 * (8 - left));
      if ((a0 ^ b0) & m) return false;
    }
    done += 16;
  }
  return true;
}

constexpr unsigned long long WH_POS_MASK = (1ull << 40) - 1;
constexpr int WL_SLOTS = 512;  // per-workgroup LDS combiner for the frequent words.  Dedup ms at 1 GB, abcd / Zipf -- round 1's kernel: 32 slots 38/51, 128: 23/19,
                               // 256: 24/20, 512: 25/22, 2048: 37/31; with the 16-byte scan of round 2: 128: 13.4/10.4, 256: 9.9/6.5, 512: 9.9/5.8, 1024: 9.8/5.9,
                               // 2048: 14.4/8.5 (LDS then limits the workgroups per CU); 512 slots with 8 probes instead of 4: 10.9/6.6
// Word table in HBM: keys in ht[0 .. cap), counts in ht[cap .. 2 cap).  (One 16-byte slot per word was measured 2.3x slower:
// the atomics on a frequent word's count then serialise with every other workgroup's read of its key -- same cache line.)
//   key = 0:1 | pure:1 | tag:6 | min(tokens, 0xffff):16 | byte offset of the representative segment:40   (PT_EMPTY = free),
//   or WH_SHORT | chars:3 | 7 ASCII chars (above); positions of the short words' representatives in ht[2 cap .. 3 cap)
// A probe compares tag and length first (23 bits) and then, always, the representative itself -- as bytes when both words are
// "pure" (seg_scan_fast), token by token otherwise: the dedup is exact.  (The pure bit is not part of the comparison: the same
// word can occur pure and, say, with an invalid byte in it.)
constexpr uint32_t WH_LEN_CAP = 0xffffu;
constexpr unsigned long long WH_PURE = 1ull << 62, WH_CMP_MASK = ~WH_PURE;
__device__ inline unsigned long long wh_key(unsigned long long h, uint32_t len_tokens, unsigned long long pos, bool pure) {
  const unsigned long long l16 = len_tokens < WH_LEN_CAP ? len_tokens : WH_LEN_CAP;
  return (pure ? WH_PURE : 0ull) | ((h >> 58) << 56) | (l16 << 40) | pos;
}
// is the word at `pos` (key `mine`, L chars) the word whose key `cur` sits in a slot?
__device__ inline bool wh_same(const uint8_t *__restrict__ text, unsigned long long n, const uint32_t *__restrict__ cpmap, unsigned long long cur,
                               unsigned long long mine, unsigned long long pos, uint32_t L) {
  if (((cur ^ mine) & WH_CMP_MASK) >> 40) return false;  // tag or length differ (or `cur` is a short word's key)
  if ((cur & mine & WH_PURE) && L + 1 < WH_LEN_CAP) return bytes_equal(text, n, cur & WH_POS_MASK, pos, L);
  return seg_equal(text, n, cpmap, cur & WH_POS_MASK, pos);
}
constexpr int WH_MAX_PROBES = 4096;  // longer than this: the table was sized too small for this corpus (status[6]; the host retries)

// insert-or-add `count` occurrences of the word whose representative segment starts at `pos` into the HBM table; `mine` = its key
// (wh_key, or the short word's own key)
__device__ inline void word_insert_global(const uint8_t *__restrict__ text, unsigned long long n, const uint32_t *__restrict__ cpmap,
                                          unsigned long long h, unsigned long long mine, unsigned long long pos, uint32_t len_tokens, unsigned long long count,
                                          unsigned long long *__restrict__ ht, unsigned long long ht_mask, unsigned int *__restrict__ status) {
  const bool is_short = (mine & WH_SHORT) != 0;
  unsigned long long i = (h >> 8) & ht_mask;
  for (int probes = 0; probes < WH_MAX_PROBES; probes++) {
    unsigned long long cur = ld_agent(&ht[i]);
    if (cur == PT_EMPTY) {
      cur = atomicCAS(&ht[i], PT_EMPTY, mine);
      if (cur == PT_EMPTY) {
        if (is_short) ht[2 * (ht_mask + 1) + i] = pos;  // (read by the compaction pass, a kernel later)
        atomicAdd(&ht[ht_mask + 1 + i], count);
        atomicAdd(&status[0], 1u);  // `status` is the workgroup's LDS copy (k2b_insert_words): 1.6e7 bumps of one HBM counter cost ~2 ns each
        if (len_tokens > (uint32_t)TILE_NOM_A) atomicAdd(&status[2], 1u);
        if (len_tokens > (uint32_t)TILE_NOM_B) {  // class C (very long words): count and longest
          atomicAdd(&status[4], 1u);
          atomicMax(&status[5], len_tokens);
        }
        return;
      }
    }
    if (is_short ? cur == mine : wh_same(text, n, cpmap, cur, mine, pos, len_tokens - 1u)) {
      atomicAdd(&ht[ht_mask + 1 + i], count);
      return;
    }
    i = (i + 1) & ht_mask;
  }
  atomicOr(&status[6], 1u);
}

// One thread per segment.  Frequent (short) words would otherwise hammer a handful of HBM addresses with atomics, so
// each workgroup first combines its segments in an LDS table (same exact scheme: a (tag | representative offset) word
// claimed by one CAS, tag matches verified against the representative's bytes) and flushes one (word, count) per
// distinct word at the end; what does not fit goes straight to the HBM table.
__global__ __launch_bounds__(BLOCK) void k2b_insert_words(const uint8_t *__restrict__ text, unsigned long long n,
                                                          const uint32_t *__restrict__ cpmap,
                                                          const unsigned long long *__restrict__ seg_pos, unsigned long long n_segs,
                                                          unsigned long long *__restrict__ ht, unsigned long long ht_mask,
                                                          unsigned int *__restrict__ status /* [0]=n_unique [1]=flags [2]=n_unique of classes B+C [3]=longest class-A word [4]=n_unique of class C [5]=longest word [6]=table too small */) {
  __shared__ uint32_t cp_ascii[128];               // cpmap[0..128): the fast scan's table
  __shared__ unsigned long long l_key[WL_SLOTS];   // a word-table key (wh_key) or PT_EMPTY
  __shared__ unsigned long long l_hash[WL_SLOTS];  // full hash of the word (for the flush)
  __shared__ unsigned long long l_pos[WL_SLOTS];   // its representative (a short word's key does not say)
  __shared__ unsigned int l_cnt[WL_SLOTS];
  __shared__ unsigned int l_len[WL_SLOTS];
  __shared__ unsigned int l_maxlen;  // longest class-A word seen by this workgroup (one global atomicMax at the end)
  __shared__ unsigned int l_status[8];  // this workgroup's share of status[] (added once at the end)
  for (int i = (int)threadIdx.x; i < WL_SLOTS; i += BLOCK) { l_key[i] = PT_EMPTY; l_cnt[i] = 0; }
  if (threadIdx.x < 128) cp_ascii[threadIdx.x] = cpmap[threadIdx.x];
  if (threadIdx.x < 8) l_status[threadIdx.x] = 0;
  if (threadIdx.x == 0) l_maxlen = 0;
  __syncthreads();
  unsigned long long s = (unsigned long long)blockIdx.x * BLOCK + threadIdx.x;
  const unsigned long long stride = (unsigned long long)gridDim.x * BLOCK;
  for (; s < n_segs; s += stride) {
    const unsigned long long pos = seg_pos[s];
    unsigned long long h;
    bool pure;
    unsigned long long skey;
    const uint32_t L = seg_scan_fast(text, n, cpmap, cp_ascii, pos, &h, &pure, &skey);
    if (L == 0) continue;  // segment made only of dropped chars: no word (bpe.cpp:357-380 deletes them first)
    const unsigned long long mine = skey ? skey : wh_key(h, L + 1, pos, pure);
    if (L + 1 <= (uint32_t)TILE_NOM_A && L + 1 > l_maxlen) atomicMax(&l_maxlen, L + 1);
    bool done = false;
    if (L <= 24) {  // very long words are not frequent enough to be worth an LDS slot (Zipf text: the top words reach 12+ chars)
      unsigned int j = (unsigned int)(h >> 8) & (WL_SLOTS - 1);
      for (int probe = 0; probe < 4 && !done; probe++) {
        unsigned long long cur = __hip_atomic_load(&l_key[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // ds_read, not a flat load
        if (cur == PT_EMPTY) {
          cur = atomicCAS(&l_key[j], PT_EMPTY, mine);
          if (cur == PT_EMPTY) {
            l_hash[j] = h;
            l_pos[j] = pos;
            l_len[j] = L + 1;
            atomicAdd(&l_cnt[j], 1u);
            done = true;
            break;
          }
        }
        if (skey ? cur == mine : wh_same(text, n, cpmap, cur, mine, pos, L)) {
          atomicAdd(&l_cnt[j], 1u);
          done = true;
          break;
        }
        j = (j + 1) & (WL_SLOTS - 1);
      }
    }
    if (!done) word_insert_global(text, n, cpmap, h, mine, pos, L + 1, 1ull, ht, ht_mask, l_status);
  }
  __syncthreads();
  for (int i = (int)threadIdx.x; i < WL_SLOTS; i += BLOCK) {
    const unsigned long long k = l_key[i];
    if (k != PT_EMPTY) word_insert_global(text, n, cpmap, l_hash[i], k, l_pos[i], l_len[i], (unsigned long long)l_cnt[i], ht, ht_mask, l_status);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (l_maxlen) atomicMax(&status[3], l_maxlen);
    if (l_status[0]) atomicAdd(&status[0], l_status[0]);
    if (l_status[2]) atomicAdd(&status[2], l_status[2]);
    if (l_status[4]) atomicAdd(&status[4], l_status[4]);
    if (l_status[5]) atomicMax(&status[5], l_status[5]);
    if (l_status[6]) atomicOr(&status[6], 1u);
  }
}

// Compact occupied hash slots into the unique-word arrays of the two tile classes (short words: block-aggregated
// append; long words are rare: one atomic each).  Order is not significant.
__global__ __launch_bounds__(BLOCK) void k2c_compact_words(const uint8_t *__restrict__ text, unsigned long long n, const uint32_t *__restrict__ cpmap,
                                                           const unsigned long long *__restrict__ ht, unsigned long long n_slots,
                                                           unsigned long long *__restrict__ posA, uint32_t *__restrict__ cntA,
                                                           uint32_t *__restrict__ lenA, unsigned long long *__restrict__ posB,
                                                           uint32_t *__restrict__ cntB, uint32_t *__restrict__ lenB,
                                                           unsigned long long *__restrict__ posC, uint32_t *__restrict__ cntC,
                                                           uint32_t *__restrict__ lenC, unsigned int *__restrict__ cursor /* [0]=A [1]=B [2]=C [3]=heavy */,
                                                           unsigned int *__restrict__ status, unsigned long long wmax,
                                                           unsigned long long *__restrict__ heavy /* [HEAVY_CAP][3]: pos, len, count still to place */) {
  // A workgroup takes CH consecutive slots, counts its class-A words, reserves their places with ONE atomic (a
  // cursor bumped once per 256 slots serialised at ~11 ns per atomic: 12 ms) and writes them in a second pass over the same,
  // now cached, slots.  The table is sized by an estimate of the number of distinct words, not by the number of occurrences.
  constexpr unsigned long long CH = 64 * BLOCK;
  __shared__ uint32_t scan_lds[NWAVES];
  __shared__ unsigned int blk_base;
  const unsigned long long *keys = ht, *cnts = ht + n_slots, *spos = ht + 2 * n_slots;
  for (unsigned long long c0 = (unsigned long long)blockIdx.x * CH; c0 < n_slots; c0 += (unsigned long long)gridDim.x * CH) {
    uint32_t mine = 0;
    for (int j = 0; j < 64; j++) {
      const unsigned long long i = c0 + (unsigned long long)j * BLOCK + threadIdx.x;
      if (i < n_slots) {
        const unsigned long long k = keys[i];
        if (k != PT_EMPTY && ((k & WH_SHORT) || ((uint32_t)(k >> 40) & 0xffffu) <= (uint32_t)TILE_NOM_A)) mine++;
      }
    }
    uint32_t total;
    const uint32_t off = block_excl_scan(mine, scan_lds, &total);
    if (threadIdx.x == 0) blk_base = total ? atomicAdd(&cursor[0], total) : 0u;
    __syncthreads();
    unsigned int o_a = blk_base + off;
    for (int j = 0; j < 64; j++) {
      const unsigned long long i = c0 + (unsigned long long)j * BLOCK + threadIdx.x;
      if (i >= n_slots) break;
      const unsigned long long k = keys[i];
      if (k == PT_EMPTY) continue;
      const unsigned long long c = cnts[i];
      const bool is_short = (k & WH_SHORT) != 0;
      const unsigned long long wpos = is_short ? spos[i] : (k & WH_POS_MASK);
      uint32_t len = is_short ? (uint32_t)((k >> 56) & 7ull) + 1u : (uint32_t)(k >> 40) & 0xffffu;
      if (!is_short && len >= WH_LEN_CAP) {  // (a word of 65535 tokens or more: its exact length from the representative)
        unsigned long long hh;
        len = seg_scan(text, n, cpmap, wpos, &hh) + 1u;
      }
      // Word weights are uint32 in the tiles; the reference counts in uint64 (bpe.cpp:382-385).  A word seen more than `wmax` (2^32 - 1)
      // times is kept as SEVERAL equal words whose weights add up to its count -- every pair count is a sum over words, so nothing the
      // merge loop computes can tell: the first copy takes wmax here, what is left goes to a short list the host turns into more copies.
      uint32_t c32 = (uint32_t)c;
      if (c > wmax) {
        c32 = (uint32_t)wmax;
        const unsigned int o = atomicAdd(&cursor[3], 1u);
        if (o < (unsigned int)HEAVY_CAP) {
          heavy[3 * o] = wpos;
          heavy[3 * o + 1] = len;
          heavy[3 * o + 2] = c - wmax;
        } else {
          atomicOr(&status[1], 2u);  // (more such words than a corpus that fits HBM can hold)
        }
      }
      if (len > (uint32_t)TILE_NOM_B) {
        const unsigned int o = atomicAdd(&cursor[2], 1u);
        posC[o] = wpos; cntC[o] = c32; lenC[o] = len;
      } else if (len > (uint32_t)TILE_NOM_A) {
        const unsigned int o = atomicAdd(&cursor[1], 1u);
        posB[o] = wpos; cntB[o] = c32; lenB[o] = len;
      } else {
        posA[o_a] = wpos; cntA[o_a] = c32; lenA[o_a] = len;
        o_a++;
      }
    }
    __syncthreads();
  }
}

// ---- chunked front end (round 5): corpora larger than the HBM left for them ------------------------------------------------------------
// The text crosses the device one CHUNK at a time: the buffer is [chunk region, `chunk_end` bytes | lexicon]; K1 / K2a / K2b work on the chunk
// as they do on a whole text, and a word first seen in this chunk has its representative there.  Before the region is overwritten by the next
// chunk, k2b_relocate copies the bytes of every such word (its segment, then one space) to the lexicon's end and points the word's slot at
// the copy: the table only ever refers to bytes that stay.  Positions are offsets into the ONE buffer either way, so nothing else changes.
// move == 0: only add up the bytes that will be needed (the host grows the lexicon first if they do not fit); move == 1: copy and re-point.
__device__ inline unsigned long long segment_bytes(const uint8_t *__restrict__ text, unsigned long long pos, unsigned long long end) {
  unsigned long long i = pos;
  while (i < end) {
    const uint32_t b = text[i];
    if (b == 32u || (b >= 9u && b <= 13u)) break;
    if (b == 0xe2u && i + 2 < end && text[i + 1] == 0x96u && text[i + 2] == 0x81u) break;  // U+2581 is white space too (utils.cpp:99-101)
    i++;
  }
  return i - pos;
}
__global__ __launch_bounds__(BLOCK) void k2b_relocate(uint8_t *__restrict__ text, unsigned long long chunk_end, unsigned long long chunk_len,
                                                      unsigned long long *__restrict__ ht, unsigned long long n_slots,
                                                      unsigned long long *__restrict__ cursor, int move) {
  unsigned long long *spos = ht + 2 * n_slots;
  const unsigned long long per_pass = (unsigned long long)gridDim.x * BLOCK;
  const unsigned long long passes = (n_slots + per_pass - 1) / per_pass;
  for (unsigned long long ps = 0; ps < passes; ps++) {  // (every lane takes part in every pass: the wave-wide sums below)
    const unsigned long long i = ps * per_pass + (unsigned long long)blockIdx.x * BLOCK + threadIdx.x;
    unsigned long long key = PT_EMPTY, pos = ~0ull;
    if (i < n_slots) {
      key = ht[i];
      if (key != PT_EMPTY) pos = (key & WH_SHORT) ? spos[i] : (key & WH_POS_MASK);
    }
    const bool mine = key != PT_EMPTY && pos < chunk_end;
    const uint32_t need = mine ? (uint32_t)segment_bytes(text, pos, chunk_len) + 1u : 0u;
    const uint32_t incl = wave_incl_scan(need);
    const uint32_t total = (uint32_t)__shfl((int)incl, 63);
    if (total == 0u) continue;
    unsigned long long base = 0;
    if (lane_id() == 63) base = atomicAdd(cursor, (unsigned long long)total);
    base = __shfl(base, 63);
    if (!move || !mine) continue;
    const unsigned long long dst = base + (unsigned long long)(incl - need);
    for (uint32_t k = 0; k + 1 < need; k++) text[dst + k] = text[pos + k];
    text[dst + need - 1] = 32u;
    if (key & WH_SHORT) spos[i] = dst;
    else ht[i] = (key & ~WH_POS_MASK) | dst;
  }
}
// the word table into a larger one (the chunked front end cannot size it once from the whole text): every word's hash again from its bytes
// (they all lie in the lexicon by now), an empty slot by linear probing -- the words are distinct, nothing is compared
__global__ __launch_bounds__(BLOCK) void k2b_rehash(const uint8_t *__restrict__ text, unsigned long long n, const uint32_t *__restrict__ cpmap,
                                                    const unsigned long long *__restrict__ old_ht, unsigned long long old_slots,
                                                    unsigned long long *__restrict__ new_ht, unsigned long long new_mask) {
  __shared__ uint32_t cp_ascii[128];
  if (threadIdx.x < 128) cp_ascii[threadIdx.x] = cpmap[threadIdx.x];
  __syncthreads();
  unsigned long long i = (unsigned long long)blockIdx.x * BLOCK + threadIdx.x;
  const unsigned long long stride = (unsigned long long)gridDim.x * BLOCK;
  for (; i < old_slots; i += stride) {
    const unsigned long long key = old_ht[i];
    if (key == PT_EMPTY) continue;
    const unsigned long long pos = (key & WH_SHORT) ? old_ht[2 * old_slots + i] : (key & WH_POS_MASK);
    unsigned long long h, skey;
    bool pure;
    (void)seg_scan_fast(text, n, cpmap, cp_ascii, pos, &h, &pure, &skey);
    unsigned long long j = (h >> 8) & new_mask;
    while (atomicCAS(&new_ht[j], PT_EMPTY, key) != PT_EMPTY) j = (j + 1) & new_mask;
    new_ht[new_mask + 1 + j] = old_ht[old_slots + i];
    new_ht[2 * (new_mask + 1) + j] = old_ht[2 * old_slots + i];
  }
}

// keys = PT_EMPTY, counts = 0
__global__ __launch_bounds__(BLOCK) void k_wh_clear(unsigned long long *__restrict__ ht, unsigned long long n_slots) {
  unsigned long long i = (unsigned long long)blockIdx.x * BLOCK + threadIdx.x;
  const unsigned long long stride = (unsigned long long)gridDim.x * BLOCK;
  for (; i < 2 * n_slots; i += stride) ht[i] = i < n_slots ? PT_EMPTY : 0ull;
}

// ---- generic exclusive scan of uint32 -> uint64 (3 kernels: block sums, scan of sums, add) ------------------------
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = BLOCK * SCAN_ITEMS;

__global__ __launch_bounds__(BLOCK) void k_scan_block_sums(const uint32_t *__restrict__ in, unsigned long long n,
                                                           unsigned long long *__restrict__ block_sums) {
  __shared__ unsigned long long ws[NWAVES];
  unsigned long long base = (unsigned long long)blockIdx.x * SCAN_TILE;
  unsigned long long s = 0;
  for (int j = 0; j < SCAN_ITEMS; j++) {
    unsigned long long i = base + (unsigned long long)j * BLOCK + threadIdx.x;
    if (i < n) s += in[i];
  }
  s = wave_sum_u64(s);
  if (lane_id() == 0) ws[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
    for (int k = 0; k < NWAVES; k++) t += ws[k];
    block_sums[blockIdx.x] = t;
  }
}

// single-block serial-by-chunks exclusive scan of the block sums (n_blocks <= a few hundred thousand)
__global__ __launch_bounds__(BLOCK) void k_scan_sums(unsigned long long *__restrict__ block_sums, unsigned long long n_blocks,
                                                     unsigned long long *__restrict__ total_out) {
  __shared__ unsigned long long ws[NWAVES];
  __shared__ unsigned long long carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (unsigned long long b0 = 0; b0 < n_blocks; b0 += BLOCK) {
    unsigned long long i = b0 + threadIdx.x;
    unsigned long long v = i < n_blocks ? block_sums[i] : 0;
    // inclusive scan in the wave
    unsigned long long inc = v;
    for (int o = 1; o < 64; o <<= 1) {
      unsigned long long t = __shfl_up(inc, o);
      if (lane_id() >= o) inc += t;
    }
    if (lane_id() == 63) ws[threadIdx.x >> 6] = inc;
    __syncthreads();
    unsigned long long base = carry;
    for (int k = 0; k < (int)(threadIdx.x >> 6); k++) base += ws[k];
    if (i < n_blocks) block_sums[i] = base + inc - v;
    __syncthreads();
    if (threadIdx.x == BLOCK - 1) carry = base + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total_out = carry;
}

__global__ __launch_bounds__(BLOCK) void k_scan_apply(const uint32_t *__restrict__ in, unsigned long long n,
                                                      const unsigned long long *__restrict__ block_sums,
                                                      unsigned long long *__restrict__ out) {
  __shared__ uint32_t scan_lds[NWAVES];
  unsigned long long base = (unsigned long long)blockIdx.x * SCAN_TILE + (unsigned long long)threadIdx.x * SCAN_ITEMS;
  uint32_t v[SCAN_ITEMS];
  uint32_t s = 0;
  for (int j = 0; j < SCAN_ITEMS; j++) {
    unsigned long long i = base + j;
    v[j] = i < n ? in[i] : 0;
    s += v[j];
  }
  // per-thread sums are < 2^32 because word lengths are <= MAX_WORD_TOKENS
  uint32_t total;
  uint32_t off = block_excl_scan(s, scan_lds, &total);
  unsigned long long run = block_sums[blockIdx.x] + off;
  for (int j = 0; j < SCAN_ITEMS; j++) {
    unsigned long long i = base + j;
    if (i < n) out[i] = run;
    run += v[j];
  }
}

// Write the token tiles: word u = [space_id|TOK_WS, id(c1), id(c2), ...]  (bpe.cpp:406-411) at its place in the slot
// of its tile.  Word u belongs to tile uw_off[u] / nom; the first word of each tile records the tile start.
__global__ __launch_bounds__(BLOCK) void k2f_tiles(const unsigned long long *__restrict__ uw_off, unsigned int n_words, unsigned int nom,
                                                   unsigned long long *__restrict__ tile_start, uint32_t *__restrict__ tile_word0) {
  unsigned int u = blockIdx.x * BLOCK + threadIdx.x;
  if (u >= n_words) return;
  unsigned long long t = uw_off[u] / nom;
  if (u == 0 || uw_off[u - 1] / nom != t) {
    tile_start[t] = uw_off[u];
    tile_word0[t] = u;
  }
}

__global__ __launch_bounds__(BLOCK) void k2g_tile_len(const unsigned long long *__restrict__ tile_start, unsigned int n_tiles,
                                                      unsigned long long total_tokens, uint32_t *__restrict__ tile_len) {
  unsigned int t = blockIdx.x * BLOCK + threadIdx.x;
  if (t >= n_tiles) return;
  unsigned long long e = (t + 1 < n_tiles) ? tile_start[t + 1] : total_tokens;
  tile_len[t] = (uint32_t)(e - tile_start[t]);
}

// Tokens of every unique word into its tile slot.  A wavefront takes 64 consecutive words; their slots are consecutive in
// memory (apart from the unused tails of tiles), so the lanes decode into an LDS window and the wave stores the window with
// coalesced 256-byte writes: a thread storing its word token by token produced 8.4 GB of HBM write traffic for 1 GB of tokens
// (partial sectors evicted from L2 between the stores).  Words that do not fit the window (classes B and C) are written
// directly.
// FILL_CAP: tokens per wavefront window: enough for 64 ordinary words and the odd tile gap.  Where words are long -- CJK-shaped text,
// clauses of ~40 chars: 64 of them and the tile gaps between span ~4600 slots -- one word behind the window used to send the WHOLE wave the
// direct way (round 4), one 4-byte store per lane and word position: 16 ms of token fill per GB.  Round 5: the lanes whose word lies inside
// the window use it, the others write directly, four tokens per store where the slot's alignment allows.  (A window of 4096 was measured:
// 64 KB of LDS per workgroup halve the waves per CU, and the byte-serial decode is latency-bound: 16.1 -> 21.4 ms.)
template <int FILL_CAP>
__global__ __launch_bounds__(BLOCK) void k2e_fill_tokens(const uint8_t *__restrict__ text, unsigned long long n,
                                                         const uint32_t *__restrict__ cpmap, uint32_t space_id,
                                                         const unsigned long long *__restrict__ uw_pos,
                                                         const unsigned long long *__restrict__ uw_off, unsigned int n_words,
                                                         unsigned int nom, unsigned int slot,
                                                         const unsigned long long *__restrict__ tile_start, uint32_t *__restrict__ tok) {
  __shared__ uint32_t stage_all[NWAVES][FILL_CAP];
  __shared__ uint32_t cp_ascii[128];  // cpmap[0..128): ASCII words are decoded 16 bytes per load (load16), not a byte and a map entry at a time
  if (threadIdx.x < 128) cp_ascii[threadIdx.x] = cpmap[threadIdx.x];
  __syncthreads();
  const int lane = lane_id();
  uint32_t *stage = stage_all[threadIdx.x >> 6];
  const unsigned int u = blockIdx.x * BLOCK + threadIdx.x;
  const bool have = u < n_words;  // (the lanes of a wave that have a word are a prefix)
  unsigned long long i0 = 0, o = 0;
  if (have) {
    i0 = uw_pos[u];
    const unsigned long long t = uw_off[u] / nom;
    o = t * slot + (uw_off[u] - tile_start[t]);
  }
  const unsigned long long o_base = __shfl(o, 0);
  for (int k = lane; k < FILL_CAP / 4; k += 64) reinterpret_cast<uint4 *>(stage)[k] = make_uint4(0, 0, 0, 0);
  wave_sync();
  // decode into the window; a word that does not fit sends the whole wave the direct way
  const unsigned long long rel = o - o_base;
  bool overflow = have && rel >= (unsigned long long)FILL_CAP;
  uint32_t cnt = 0;
  if (have && !overflow) {
    const uint32_t r0 = (uint32_t)rel;
    stage[r0] = space_id | TOK_WS;
    cnt = 1;
    unsigned long long i = i0;
    bool done = false;
    while (!done && !overflow && i + 24 <= n) {  // 16 bytes at a time while they are ASCII (cf. seg_scan_fast)
      unsigned long long w0, w1;
      load16(text, i, w0, w1);
      int k = 0;
      for (; k < 16; k++) {
        const uint32_t b = (uint32_t)((k < 8 ? w0 >> (8 * k) : w1 >> (8 * (k - 8))) & 0xffull);
        if (b >= 0x80u) break;
        const uint32_t id = cp_ascii[b];
        if (id == CP_SPACE) { done = true; break; }
        if (id == CP_DROP) continue;
        if (r0 + cnt >= (uint32_t)FILL_CAP) { overflow = true; break; }
        stage[r0 + cnt++] = id;
      }
      i += (unsigned long long)k;
      if (k < 16) break;  // a space (done), the window's end, or a byte that needs the exact decode
    }
    while (!done && !overflow && i < n) {
      uint32_t len;
      const uint32_t cp = u8_decode_at(text, i, n, &len);
      if (cp != INVALID_CP) {
        const uint32_t id = cpmap[cp];
        if (id == CP_SPACE) break;
        if (id != CP_DROP) {
          if (r0 + cnt >= (uint32_t)FILL_CAP) {
            overflow = true;
            break;
          }
          stage[r0 + cnt++] = id;
        }
      }
      i += len;
    }
  }
  wave_sync();
  {
    // the window ends behind the last word that lies in it (a word that ran out of window mid-way left a correct prefix there: harmless,
    // the direct path below writes the same tokens again)
    uint32_t end = have && !overflow ? (uint32_t)rel + cnt : 0u;
    for (int d = 32; d > 0; d >>= 1) {
      const uint32_t other = __shfl_down(end, d);
      end = other > end ? other : end;
    }
    end = __shfl(end, 0);
    for (uint32_t k = (uint32_t)lane; k < end; k += 64) tok[o_base + k] = stage[k];  // (zeros in the gaps: tile tails are zero)
  }
  if (!have || !overflow) return;
  unsigned long long i = i0;
  uint32_t b0 = 0, b1 = 0, b2 = 0;  // tokens on their way out: a 16-byte store once the position is 16-byte aligned and four are there
  int nb = 0;                       // (three named registers, not an array: a dynamically indexed one would live in scratch)
  auto put = [&](uint32_t v) {
    if ((o & 3ull) != 0ull && nb == 0) {  // (up to the first aligned position: one at a time)
      tok[o++] = v;
      return;
    }
    if (nb == 0) b0 = v;
    else if (nb == 1) b1 = v;
    else if (nb == 2) b2 = v;
    if (++nb == 4) {
      *reinterpret_cast<uint4 *>(tok + o) = make_uint4(b0, b1, b2, v);
      o += 4;
      nb = 0;
    }
  };
  put(space_id | TOK_WS);
  while (i < n) {
    uint32_t len;
    uint32_t cp = u8_decode_at(text, i, n, &len);
    if (cp != INVALID_CP) {
      uint32_t id = cpmap[cp];
      if (id == CP_SPACE) break;
      if (id != CP_DROP) put(id);
    }
    i += len;
  }
  if (nb > 0) tok[o] = b0;
  if (nb > 1) tok[o + 1] = b1;
  if (nb > 2) tok[o + 2] = b2;
}

// ------------------------------------------------------------------------------------------------- launchers
static inline unsigned int grid_for(unsigned long long items, unsigned int per_block, unsigned int max_blocks) {
  unsigned long long b = (items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > max_blocks) b = max_blocks;
  return (unsigned int)b;
}

void launch_char_hist(const uint8_t *text, unsigned long long n, unsigned long long *hist, unsigned long long *counters, bool wide_chars,
                      uint32_t *chunk_segs, hipStream_t st, unsigned long long chunk_lo, unsigned long long chunk_hi) {
  if (chunk_hi > fe_chunks(n)) chunk_hi = fe_chunks(n);
  if (chunk_lo >= chunk_hi) return;
  const unsigned long long bytes = (chunk_hi - chunk_lo) * FE_CHUNK;
  // (a part of the text: at least 32 chunks per workgroup -- every workgroup ends with a flush of its LDS histograms, ~2 000 atomics)
  const unsigned int cap = chunk_lo == 0 && chunk_hi == fe_chunks(n) ? 256u * 8 : (unsigned int)std::max<unsigned long long>(64, std::min<unsigned long long>(256 * 8, (chunk_hi - chunk_lo) / 32));
  if (wide_chars) {  // (76 KB of LDS per workgroup: two per CU)
    unsigned int g = grid_for(bytes, FE_CHUNK, std::min(cap, 256u * 2));
    hipLaunchKernelGGL((k_scan_bytes<0, 8192>), dim3(g), dim3(BLOCK), 0, st, text, n, hist, counters, (unsigned long long *)nullptr,
                       (const unsigned long long *)nullptr, chunk_segs, chunk_lo, chunk_hi);
    return;
  }
  unsigned int g = grid_for(bytes, FE_CHUNK, cap);
  hipLaunchKernelGGL((k_scan_bytes<0, 0>), dim3(g), dim3(BLOCK), 0, st, text, n, hist, counters, (unsigned long long *)nullptr,
                     (const unsigned long long *)nullptr, chunk_segs, chunk_lo, chunk_hi);
}
unsigned long long fe_chunks(unsigned long long n) { return (n + FE_CHUNK - 1) / FE_CHUNK; }
unsigned long long fe_chunk_bytes() { return FE_CHUNK; }
void launch_seg_write(const uint8_t *text, unsigned long long n, unsigned long long *seg_pos, const unsigned long long *chunk_off,
                      hipStream_t st, unsigned long long chunk_lo, unsigned long long chunk_hi) {
  if (chunk_hi > fe_chunks(n)) chunk_hi = fe_chunks(n);
  if (chunk_lo >= chunk_hi) return;
  unsigned int g = grid_for((chunk_hi - chunk_lo) * FE_CHUNK, FE_CHUNK, 256 * 8);
  hipLaunchKernelGGL((k_scan_bytes<1, 0>), dim3(g), dim3(BLOCK), 0, st, text, n, (unsigned long long *)nullptr,
                     (unsigned long long *)nullptr, seg_pos, chunk_off, (uint32_t *)nullptr, chunk_lo, chunk_hi);
}
void launch_hist_compact(const unsigned long long *hist, uint32_t *cps, unsigned long long *cnts, unsigned int *n_out,
                         unsigned int cap, hipStream_t st) {
  hipLaunchKernelGGL(k_hist_compact, dim3((N_CODEPOINTS + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st, hist, cps, cnts, n_out, cap);
}
void launch_word_table_clear(unsigned long long *ht, unsigned long long n_slots, hipStream_t st) {
  unsigned int g = grid_for(n_slots, BLOCK, 256 * 16);
  hipLaunchKernelGGL(k_wh_clear, dim3(g), dim3(BLOCK), 0, st, ht, n_slots);
}
void launch_insert_words(const uint8_t *text, unsigned long long n, const uint32_t *cpmap, const unsigned long long *seg_pos,
                         unsigned long long n_segs, unsigned long long *ht, unsigned long long ht_mask, unsigned int *status, hipStream_t st,
                         unsigned int max_blocks) {
  unsigned int g = grid_for(n_segs, BLOCK, max_blocks);  // (256 * 8: dedup 11.6 ms at 1 GB instead of 9.9; 256 * 64: 9.6)
  hipLaunchKernelGGL(k2b_insert_words, dim3(g), dim3(BLOCK), 0, st, text, n, cpmap, seg_pos, n_segs, ht, ht_mask, status);
}
void launch_words_relocate(uint8_t *text, unsigned long long chunk_end, unsigned long long chunk_len, unsigned long long *ht, unsigned long long n_slots,
                           unsigned long long *cursor, bool move, hipStream_t st) {
  const unsigned long long blocks = std::min<unsigned long long>((n_slots + BLOCK - 1) / BLOCK, 256ull * 16);
  hipLaunchKernelGGL(k2b_relocate, dim3((unsigned int)std::max<unsigned long long>(blocks, 1)), dim3(BLOCK), 0, st, text, chunk_end, chunk_len, ht, n_slots, cursor, move ? 1 : 0);
}
void launch_word_table_rehash(const uint8_t *text, unsigned long long n, const uint32_t *cpmap, const unsigned long long *old_ht, unsigned long long old_slots,
                              unsigned long long *new_ht, unsigned long long new_slots, hipStream_t st) {
  const unsigned long long blocks = std::min<unsigned long long>((old_slots + BLOCK - 1) / BLOCK, 256ull * 16);
  hipLaunchKernelGGL(k2b_rehash, dim3((unsigned int)std::max<unsigned long long>(blocks, 1)), dim3(BLOCK), 0, st, text, n, cpmap, old_ht, old_slots, new_ht, new_slots - 1);
}
void launch_compact_words(const uint8_t *text, unsigned long long n, const uint32_t *cpmap, const unsigned long long *ht, unsigned long long n_slots,
                          unsigned long long *posA, uint32_t *cntA, uint32_t *lenA, unsigned long long *posB, uint32_t *cntB, uint32_t *lenB,
                          unsigned long long *posC, uint32_t *cntC, uint32_t *lenC, unsigned int *cursor, unsigned int *status, unsigned long long wmax,
                          unsigned long long *heavy, hipStream_t st) {
  unsigned int g = grid_for(n_slots, BLOCK, 256 * 16);
  hipLaunchKernelGGL(k2c_compact_words, dim3(g), dim3(BLOCK), 0, st, text, n, cpmap, ht, n_slots, posA, cntA, lenA, posB, cntB, lenB, posC, cntC, lenC,
                     cursor, status, wmax, heavy);
}
void launch_exclusive_scan(const uint32_t *in, unsigned long long n, unsigned long long *out, unsigned long long *block_sums,
                           unsigned long long *total_out, hipStream_t st) {
  unsigned long long nb = (n + SCAN_TILE - 1) / SCAN_TILE;
  if (nb == 0) nb = 1;
  hipLaunchKernelGGL(k_scan_block_sums, dim3((unsigned int)nb), dim3(BLOCK), 0, st, in, n, block_sums);
  hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(BLOCK), 0, st, block_sums, nb, total_out);
  hipLaunchKernelGGL(k_scan_apply, dim3((unsigned int)nb), dim3(BLOCK), 0, st, in, n, block_sums, out);
}
unsigned long long scan_scratch_blocks(unsigned long long n) {
  unsigned long long nb = (n + SCAN_TILE - 1) / SCAN_TILE;
  return nb ? nb : 1;
}
void launch_fill_tokens(const uint8_t *text, unsigned long long n, const uint32_t *cpmap, uint32_t space_id,
                        const unsigned long long *uw_pos, const unsigned long long *uw_off, unsigned int n_words, unsigned int nom,
                        unsigned int slot, const unsigned long long *tile_start, uint32_t *tok, hipStream_t st, unsigned long long total_tokens) {
  if (!n_words) return;
  (void)total_tokens;
  hipLaunchKernelGGL(k2e_fill_tokens<2048>, dim3((n_words + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st, text, n, cpmap, space_id, uw_pos,
                     uw_off, n_words, nom, slot, tile_start, tok);
}
void launch_tiles(const unsigned long long *uw_off, unsigned int n_words, unsigned int nom, unsigned long long *tile_start,
                  uint32_t *tile_word0, hipStream_t st) {
  if (!n_words) return;
  hipLaunchKernelGGL(k2f_tiles, dim3((n_words + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st, uw_off, n_words, nom, tile_start, tile_word0);
}
void launch_tile_len(const unsigned long long *tile_start, unsigned int n_tiles, unsigned long long total_tokens,
                     uint32_t *tile_len, hipStream_t st) {
  if (!n_tiles) return;
  hipLaunchKernelGGL(k2g_tile_len, dim3((n_tiles + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st, tile_start, n_tiles, total_tokens,
                     tile_len);
}

}  // namespace yttm
