// yttm_device.h -- shared device-side definitions for the MI355X (gfx950) BPE kernels.
//
// Data layout in HBM:
//   * token stream `tok[T]`   : uint32 per token; bit31 (TOK_WS) marks the first token of a word (the one that carries
//                               the leading "▁", bpe.cpp:407/1514), bits 0..30 = compact token id.  Words are stored back to
//                               back; a pair (tok[p],tok[p+1]) is an adjacency iff tok[p+1] has no TOK_WS.
//   * tiles                   : the stream is cut at word boundaries into tiles of ~TILE_TOK tokens; one workgroup owns a
//                               tile, stages it in LDS, and compacts it in place as merges remove tokens.
//   * word weights `wcnt[U]`  : uint32 frequency per unique word (bpe.cpp:382-385 WordCount::cnt).
//   * pair table              : open-addressing hash map (x<<32|y) -> uint64 count in HBM (replaces pair2cnt, bpe.cpp:891).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace yttm {

constexpr uint32_t TOK_WS = 0x80000000u;
constexpr uint32_t TOK_MASK = 0x7fffffffu;

// code point -> class map values (cpmap[0x110000])
constexpr uint32_t CP_DROP = 0xffffffffu;   // removed by coverage / not in the alphabet (train) -- deleted from the text
constexpr uint32_t CP_SPACE = 0xfffffffeu;  // utils.cpp:99-101 is_space
constexpr uint32_t CP_UNK = 0xfffffffdu;    // encode: char not in the model (bpe.cpp:1517)
constexpr uint32_t N_CODEPOINTS = 0x110000u;
constexpr uint32_t INVALID_CP = 0x0fffffffu;  // utf8.h:9

// Token tiles.  A tile holds whole words and lives in a fixed slot of SLOT tokens (tile t = tok[t*SLOT ..)), of which
// the first tile_len[t] are live; merges compact a tile in place.  One WAVEFRONT owns a tile (no workgroup barriers in
// the hot loops).  Two classes so that LDS stays small for the common case:
//   class A: words of <= 256 tokens, nominal 256 tokens per tile (more when the longest word is shorter), slot 512
//            (4 waves per workgroup).  Measured at 1 GB: slot 1024 -> 392 ms, 512 -> 383 ms, 256 -> 401 ms per step.
//   class B: words of 257..2048 tokens, nominal 2048 per tile, slot 4096 (1 wave per workgroup; rare)
constexpr int TILE_NOM_A = 256, TILE_SLOT_A = 512;
constexpr int TILE_NOM_B = 2048, TILE_SLOT_B = 4096;
constexpr int MAX_WORD_TOKENS = TILE_NOM_B;  // longest word the tile kernels accept (incl. the leading space token)
constexpr int BLOCK = 256;      // 4 wavefronts of 64 lanes
constexpr int NWAVES = BLOCK / 64;

constexpr unsigned long long PT_EMPTY = ~0ull;

// Two flag bits of a count mark the candidate lists a slot is on.  The HOT list (L1) holds every slot whose count reached
// hot_tau since the list was built -- tens of thousands of slots instead of the whole table; the TOP list (L2) holds those of
// them that reached top_tau >= hot_tau -- about a thousand, few enough for ONE workgroup to read at the end of a merge round
// (k_merge_shared.h scan_top).  The scan of a round reads L2; L2 is refilled from L1 when it runs dry, L1 from the table.
constexpr unsigned long long PT_HOT = 1ull << 63;
constexpr unsigned long long PT_TOP = 1ull << 62;
constexpr unsigned long long PT_FLAGS = PT_HOT | PT_TOP;
constexpr unsigned long long PT_CNT = PT_TOP - 1;

struct PairTable {
  unsigned long long *slots;  // [2 * capacity]: slot i = { key (PT_EMPTY = free), count (bits 0..61) | flags } -- one 16-byte
                              // record, so the probe of a key and the update of its count touch the same HBM sector
  unsigned long long mask;    // capacity - 1 (capacity is a power of two)
  __host__ __device__ unsigned long long *key_p(unsigned long long i) const { return slots + 2 * i; }
  __host__ __device__ unsigned long long *cnt_p(unsigned long long i) const { return slots + 2 * i + 1; }
  unsigned int *n_keys;      // number of occupied slots
  unsigned long long hot_tau;  // a count reaching this puts its slot on the hot list (~0ull: list off)
  uint32_t *hot_slots;         // [hot_cap]
  unsigned int *hot_n;         // appended entries (may exceed hot_cap: then the list is rebuilt)
  unsigned int hot_cap;
  unsigned int top_cap;
  unsigned long long top_tau;  // ... and this, on the top list (~0ull: list off)
  uint32_t *top_slots;         // [top_cap]
  unsigned int *top_n;
  // Multi-GPU: the apply kernels and the fold of the other ranks' deltas run with the two thresholds above OFF (~0) -- what joins a list is
  // decided behind the exchange, by the FINAL counts, which are the same on every rank (k_fold_list).  So that this needs no pass over
  // every delta record, an add that takes a count across a threshold, on a slot not yet on that list, notes the slot here: a rank-local
  // superset of the slots that can have ended the round on the other side (a slot not on a list starts the round below the list's
  // threshold: to end at or above it, some add of the sequence this rank applies must cross it), re-examined -- by final count -- in
  // the fold kernel.  maybe_n == nullptr: not kept.
  uint32_t *maybe;
  unsigned int *maybe_n;
  unsigned int maybe_cap;
  unsigned long long maybe_hot, maybe_top;  // the thresholds the notes are taken by
};

struct TileSet {
  uint32_t *tok;               // [n_tiles * SLOT]
  uint32_t *tile_len;          // [n_tiles] live tokens (compacted prefix of the slot)
  const uint32_t *tile_word0;  // [n_tiles] index of the first word of the tile
  const uint32_t *wcnt;        // [U] word frequencies, in tile order
  uint32_t n_tiles;
};

// ---- word mode (k_words.hip).  Once a merge round touches few of the words, class-A words are no longer processed tile by
// tile: every word keeps the slot it had in the tile array at that moment (wmeta: first token, live length) and shrinks IN PLACE --
// the tokens a merge frees become TOK_HOLE (word-start bit AND bit 30: no kernel takes a hole for a token, an adjacency or a word
// start) -- and a round visits only the words that hold a merge site.
constexpr uint32_t TOK_HOLE = 0xC0000000u;
constexpr uint32_t NBR_NONE = 0xffffffffu;  // "no neighbour inside the word" in an instance record
__host__ __device__ inline bool tok_is_ws(uint32_t t) { return (t >> 30) == 2u; }
struct WordSet {
  uint32_t *tok;               // the class-A token array (tile slots as they were at the switch)
  unsigned long long *wmeta;   // [n_words] index of the word's first token << 16 | live tokens
  const uint32_t *wcnt;        // [n_words] word frequencies
  uint32_t n_words;
};
// Where the sites of a rule (x,y) are.  Pairs of tokens that existed when the pair index was last built: its postings (word ids).
// Pairs with a younger token: every instance of a token z created since then has ONE record {word, left neighbour, right neighbour
// after its round} in list(z), a contiguous run of the record arrays allotted when z's rule was gathered; an adjacency (a,b) is found
// in the list of its YOUNGER token (the larger id: when that instance was made the other one was already its neighbour -- a neighbour
// made later would itself be the younger one).  Stale records (an instance merged away since) cost a look at the word, nothing else.
struct TokLists {
  unsigned long long *base;    // [n_ids] start of list(z)
  uint32_t *cap, *fill;        // [n_ids] records allotted / written
  uint32_t *rec_word, *rec_l, *rec_r;  // [log_cap]
  unsigned long long *cursor;  // records allotted so far
  unsigned long long log_cap;
  unsigned int *broken;        // in the host's pinned memory; != 0: records were dropped (log full) or a round took every word -- the host
                               // rebuilds the index (which covers every token that exists then) before the next round
};

// one slot of the per-round rule table uploaded by the host (key = x<<32|y)
struct RuleSlot {
  unsigned long long key;
  uint32_t z;
  uint32_t pad;
};

// exchange record (multi-GPU): signed count delta for one pair
struct DeltaRec {
  unsigned long long key;
  long long delta;
};

// A rank's send block of a round, in 16-byte units: XHDR header units, then the records.  Header: [0] = {records, capacity of the rank's
// send buffer}, [1] = {merge sites so far, tokens streamed so far} on that rank, [2] = {its class-A tiles, -}: every rank sees every
// header after the all-gather, so what is decided from them (block overflow, the switch to word mode) is decided alike everywhere.
constexpr unsigned int XHDR = 4;
// Multi-GPU: what a rank's merge round changes, summed by pair before it travels.  Every count update of a round is also added
// (dt_add) to a (pair, delta) RECORD of the round's send block: a rank-local open-addressing table maps the pair to its record -- the
// thread that claims a table slot takes the next record (the block's own header word counts them), writes the pair there and
// publishes the record number in the slot; everybody else adds to that record.  When the round's last apply kernel is done the block
// is what the ranks all-gather: nothing is packed, copied or counted in between (a table of (pair, sum) that a kernel of its own
// turned into the block cost 6 us per round on the critical path; a record per update instead -- the first version -- put every update
// of a round through one global cursor: 0.9 ms per round at 1 GB, and ten times the bytes on the links).  Two blocks alternate: behind
// the exchange, off the critical path (k_dt_clean), the table's slots are freed and the block of the round before is zeroed for the
// round to come, while the block just sent stays as it is for a possible repeat of the exchange with wider blocks.
constexpr uint32_t DT_NOIDX = 0xffffffffu;  // the slot's claimant has not published the record number yet
constexpr uint32_t DT_LOST = 0xfffffffeu;   // ... found the block full: the update is dropped, the count in the header says so, every rank stops
struct DtSlot {
  unsigned long long key;  // PT_EMPTY = free
  uint32_t idx;            // record number in the send block
  uint32_t pad;
};
struct DeltaBuf {
  DtSlot *keys;              // [mask + 1]; nullptr = single-GPU mode (nothing is kept)
  uint32_t *touched;         // [send_cap] slot of record j (for the clean-up)
  DeltaRec *send;            // this round's send block: send[0].key = records claimed so far
  unsigned long long send_cap;
  unsigned long long mask;
};


__host__ __device__ inline unsigned long long mix64(unsigned long long x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdull;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ull;
  x ^= x >> 33;
  return x;
}

// Hash of a pair for the small per-round tables (the batch's rule hash, the LDS delta aggregator): a 64-bit mix costs ~16
// VALU instructions, and the apply kernel hashes four keys per 64-token chunk; these tables tolerate a weaker hash.
__host__ __device__ inline uint32_t pair_hash32(unsigned long long key) {
  uint32_t h = (uint32_t)(key >> 32) * 0x9E3779B1u;
  h ^= (uint32_t)key * 0x85EBCA77u;
  h ^= h >> 15;
  return h;
}
__host__ __device__ inline unsigned long long pair_key(uint32_t x, uint32_t y) {
  return ((unsigned long long)x << 32) | (unsigned long long)y;
}

// ---- racy reads go through agent-scope atomic loads (per-CU L1 is not coherent; MI355X_MICROARCH.md) -------------
__device__ inline unsigned long long ld_agent(const unsigned long long *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// pair table: insert-or-add.  Keys are never removed, so a non-empty slot observed once stays valid.
// `new_keys` (optional): a workgroup-local counter of freshly claimed slots; the caller adds it to pt.n_keys once per
// workgroup -- thousands of waves bumping ONE global counter serialise at the L2 atomic unit and put a floor of
// ~0.1 ms under every launch.
__device__ inline void pt_add(const PairTable &pt, unsigned long long key, long long delta, unsigned int *new_keys = nullptr) {
  unsigned long long i = mix64(key) & pt.mask;
  for (;;) {
    unsigned long long k = ld_agent(pt.key_p(i));
    if (k == PT_EMPTY) {
      k = atomicCAS(pt.key_p(i), PT_EMPTY, key);
      if (k == PT_EMPTY) {
        atomicAdd(new_keys ? new_keys : pt.n_keys, 1u);
        k = key;
      }
    }
    if (k == key) {
      if (delta > 0 && pt.hot_tau != ~0ull) {
        // only an increase can cross a threshold; the adder that observes the crossing (exactly one: the adds on a slot are
        // serialised) sets the list's flag, and whoever sets it first appends the slot.  (Write-through stores: the workgroup
        // that scans a list at the end of THIS launch -- k_merge_shared.h scan_top -- may sit on another XCD, whose L2 does not see
        // plain stores before a cache write-back.)
        const unsigned long long old = atomicAdd(pt.cnt_p(i), (unsigned long long)delta);
        const unsigned long long now = (old & PT_CNT) + (unsigned long long)delta;
        unsigned long long want = 0;
        if (!(old & PT_HOT) && now >= pt.hot_tau) want |= PT_HOT;
        if (!(old & PT_TOP) && now >= pt.top_tau) want |= PT_TOP;
        if (want) {
          const unsigned long long fresh = want & ~atomicOr(pt.cnt_p(i), want);
          if (fresh & PT_HOT) {
            const unsigned int j = atomicAdd(pt.hot_n, 1u);
            if (j < pt.hot_cap) __hip_atomic_store(&pt.hot_slots[j], (uint32_t)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          if (fresh & PT_TOP) {
            const unsigned int j = atomicAdd(pt.top_n, 1u);
            if (j < pt.top_cap) __hip_atomic_store(&pt.top_slots[j], (uint32_t)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      } else if (delta > 0 && pt.maybe_n && pt.maybe_hot != ~0ull) {
        const unsigned long long old = atomicAdd(pt.cnt_p(i), (unsigned long long)delta);
        const unsigned long long now = (old & PT_CNT) + (unsigned long long)delta;
        const unsigned long long was = old & PT_CNT;
        // (the add that CROSSES a threshold, not every add above it: the flags are not set during the round, and a new pair of the first
        // rounds takes a hundred thousand adds on its way to ten million)
        if ((!(old & PT_HOT) && was < pt.maybe_hot && now >= pt.maybe_hot) || (!(old & PT_TOP) && was < pt.maybe_top && now >= pt.maybe_top)) {
          const unsigned int j = atomicAdd(pt.maybe_n, 1u);
          if (j < pt.maybe_cap) __hip_atomic_store(&pt.maybe[j], (uint32_t)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      } else {
        atomicAdd(pt.cnt_p(i), (unsigned long long)delta);
      }
      return;
    }
    i = (i + 1) & pt.mask;
  }
}

__device__ inline unsigned long long pt_get(const PairTable &pt, unsigned long long key) {
  unsigned long long i = mix64(key) & pt.mask;
  for (;;) {
    unsigned long long k = *pt.key_p(i);
    if (k == PT_EMPTY) return 0;
    if (k == key) return *pt.cnt_p(i) & PT_CNT;
    i = (i + 1) & pt.mask;
  }
}

__device__ inline void dt_add(const DeltaBuf &db, unsigned long long key, long long delta) {
  if (!db.keys) return;
  unsigned long long i = mix64(key * 0x9e3779b97f4a7c15ull) & db.mask;
  bool found = false;
  for (unsigned long long probes = 0; probes <= db.mask; probes++) {
    unsigned long long k = ld_agent(&db.keys[i].key);
    if (k == PT_EMPTY) {
      k = atomicCAS(&db.keys[i].key, PT_EMPTY, key);
      if (k == PT_EMPTY) {  // mine: the next record of the block (the claimant publishes its number HERE, inside the loop, before any lane of
                            // its wave gets to wait for one below)
        const unsigned long long j = atomicAdd(&db.send[0].key, 1ull);
        if (j < db.send_cap) {
          db.send[XHDR + j].key = key;  // (read after the kernel: by the all-gather and the clean-up)
          db.touched[j] = (uint32_t)i;
        }
        __hip_atomic_store(&db.keys[i].idx, j < db.send_cap ? (uint32_t)j : DT_LOST, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        k = key;
      }
    }
    if (k == key) {
      found = true;
      break;
    }
    i = (i + 1) & db.mask;
  }
  if (!found) {
    atomicAdd(&db.send[0].key, db.send_cap + 2);  // table full (the count then exceeds the capacity: every rank stops the training)
    return;
  }
  uint32_t j;
  do {
    j = __hip_atomic_load(&db.keys[i].idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } while (j == DT_NOIDX);  // (a few hundred cycles at most: between the claimant's CAS and its store)
  if (j != DT_LOST) atomicAdd(reinterpret_cast<unsigned long long *>(&db.send[XHDR + j].delta), (unsigned long long)delta);
}

// ---- UTF-8 (utf8.cpp:14-74), device version ----------------------------------------------------------------------
__host__ __device__ inline bool u8_cont(uint32_t b) { return (b & 0xc0u) == 0x80u; }
__host__ __device__ inline bool cp_ok(uint32_t x) { return (x < 0xd800u) || (0xdfffu < x && x < 0x110000u); }

// Decodes the char whose first byte is b0 at position i; `avail` = bytes available from i (>=1); b1..b3 are the
// following bytes (only read when available).  Returns code point or INVALID_CP; *len = bytes consumed (1 if invalid).
__host__ __device__ inline uint32_t u8_decode(uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3, unsigned long long avail,
                                              uint32_t *len) {
  *len = 1;
  if ((b0 & 0x80u) == 0) return b0;
  if ((b0 & 0xe0u) == 0xc0u) {
    if (avail >= 2 && u8_cont(b1)) {
      uint32_t cp = ((b0 & 0x1fu) << 6) + (b1 & 0x3fu);
      if (cp >= 0x80u && cp_ok(cp)) { *len = 2; return cp; }
    }
  } else if ((b0 & 0xf0u) == 0xe0u) {
    if (avail >= 3 && u8_cont(b1) && u8_cont(b2)) {
      uint32_t cp = ((b0 & 0x0fu) << 12) + ((b1 & 0x3fu) << 6) + (b2 & 0x3fu);
      if (cp >= 0x800u && cp_ok(cp)) { *len = 3; return cp; }
    }
  } else if ((b0 & 0xf8u) == 0xf0u) {
    if (avail >= 4 && u8_cont(b1) && u8_cont(b2) && u8_cont(b3)) {
      uint32_t cp = ((b0 & 0x07u) << 18) + ((b1 & 0x3fu) << 12) + ((b2 & 0x3fu) << 6) + (b3 & 0x3fu);
      if (cp >= 0x10000u && cp_ok(cp)) { *len = 4; return cp; }
    }
  }
  return INVALID_CP;
}

__host__ __device__ inline bool cp_is_space(uint32_t ch) {  // utils.cpp:99-101; isspace() in the C locale
  return (ch < 256u && (ch == 32u || (ch >= 9u && ch <= 13u))) || ch == 9601u;
}

// Sequential decode at byte offset i of text[0..n): used by the one-thread-per-segment front-end kernels.
__device__ inline uint32_t u8_decode_at(const uint8_t *text, unsigned long long i, unsigned long long n, uint32_t *len) {
  uint32_t b0 = text[i];
  if (b0 < 0x80u) { *len = 1; return b0; }
  unsigned long long avail = n - i;
  uint32_t b1 = avail > 1 ? text[i + 1] : 0, b2 = avail > 2 ? text[i + 2] : 0, b3 = avail > 3 ? text[i + 3] : 0;
  return u8_decode(b0, b1, b2, b3, avail, len);
}

// Is byte i the first byte of a char in the reference's left-to-right decode (utf8.cpp:111-128)?  Every
// non-continuation byte is; a continuation byte is unless a VALID multi-byte char starting <= 3 bytes earlier covers it.
__device__ inline bool u8_is_start(const uint8_t *text, unsigned long long i, unsigned long long n) {
  uint32_t b = text[i];
  if (!u8_cont(b)) return true;
  for (uint32_t d = 1; d <= 3 && d <= i; d++) {
    uint32_t c = text[i - d];
    if (u8_cont(c)) continue;
    // nearest non-continuation byte at distance d
    uint32_t len;
    uint32_t cp = u8_decode_at(text, i - d, n, &len);
    return !(cp != INVALID_CP && len > d);
  }
  return true;
}

__device__ inline int lane_id() { return (int)(threadIdx.x & 63u); }
// LDS hand-off between the lanes of ONE wavefront: DS ops of a wave execute in order, so only the compiler must be
// kept from reordering (fence) and the lanes kept converged (wave_barrier).
__device__ inline void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
// Makes a per-lane value opaque to the optimiser at this point (no code): keeps it from hoisting what is computed from the
// value out of a loop and then spilling it for lack of registers.
#if defined(__HIP_DEVICE_COMPILE__)
#define YTTM_OPAQUE_V(x) asm volatile("" : "+v"(x))
#else
#define YTTM_OPAQUE_V(x) ((void)0)
#endif
__device__ inline unsigned long long lanemask_lt() { return (1ull << lane_id()) - 1ull; }
// Wave-uniform values.  The compiler cannot know that a value read from LDS at a uniform address, or handed out by a
// shuffle, is the same in every lane: it keeps it in vector registers and turns every loop and branch on it into exec-mask
// code.  uni() moves it to scalar registers (v_readfirstlane); 64-bit masks then shift and count on the scalar unit,
// lane_bit() selects a lane's bit of a uniform mask with one v_cndmask and lanes_below() counts the bits below the lane
// with v_mbcnt.  MUST only be used on values that are uniform across the wave.
__device__ inline uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ inline int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ inline unsigned long long uni64(unsigned long long v) {
  const uint32_t lo = uni((uint32_t)v), hi = uni((uint32_t)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}
// ballot of a condition the compiler already holds as a lane mask (__ballot takes an int and makes the compiler turn the mask
// into 0/1 values and compare them again)
__device__ inline unsigned long long ballot_b(bool p) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_ballot_w64(p);
#else
  return __ballot(p ? 1 : 0);
#endif
}
// the value the lane to the right (lane + 1) holds; lane 63 gets an unspecified one.  A DPP wave shift: no LDS crossbar trip.
__device__ inline uint32_t from_lane_right(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
#else
  return __shfl_down(v, 1);
#endif
}
// lane 0's value, for every lane (v_readlane: the scalar unit holds it)
__device__ inline uint32_t from_lane0(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 0);
#else
  return __shfl(v, 0);
#endif
}
__device__ inline bool lane_bit(unsigned long long uniform_mask) { return __builtin_amdgcn_inverse_ballot_w64(uniform_mask); }
__device__ inline uint32_t lanes_below(unsigned long long uniform_mask) {  // popcount(mask & lanemask_lt())
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(uniform_mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)uniform_mask, 0u));
}

// ---- wave / block scans (wave = 64 lanes; all lanes of the block must call) ---------------------------------------
__device__ inline uint32_t wave_incl_scan(uint32_t v) {
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t t = __shfl_up(v, o);
    if (lane_id() >= o) v += t;
  }
  return v;
}
__device__ inline unsigned long long wave_sum_u64(unsigned long long v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  return v;  // valid in lane 0
}
// exclusive scan over the BLOCK threads; lds = NWAVES words of scratch; *total = block sum
__device__ inline uint32_t block_excl_scan(uint32_t v, uint32_t *lds, uint32_t *total) {
  uint32_t inc = wave_incl_scan(v);
  int w = (int)(threadIdx.x >> 6);
  if (lane_id() == 63) lds[w] = inc;
  __syncthreads();
  uint32_t base = 0, tot = 0;
  for (int k = 0; k < NWAVES; k++) {
    uint32_t s = lds[k];
    if (k < w) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

}  // namespace yttm
