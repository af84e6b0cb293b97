// k_encode.hip -- K5: batch BPE encode, one sentence per wavefront (gfx950, wave64).
//
// Replaces BaseEncoder::encode_sentence (bpe.cpp:1455-1632) + encode_parallel (bpe.cpp:1697-1738) for dropout_prob == 0.
// The reference pops (rule id, position) events from a per-word priority queue; that is equivalent to
// rounds of "per word: find the smallest rule id among adjacent pairs, apply all its occurrences left to right".  A
// wavefront runs those rounds for ALL words of its sentence at once: lanes = token positions, word-segmented minimum
// through LDS atomics, x==x runs resolved by parity from the run start, in-place compaction with wave ballots.
// Working arrays live in LDS (3 x 4 B per token); sentences too long for LDS use an HBM scratch with the same code.
#include <type_traits>

#include "yttm_device.h"
#include "yttm_kernels.h"

namespace yttm {

constexpr int ENC_WCAP = ENC_LDS_TOKENS;     // tokens per wave held in LDS (3 arrays x 2 KB)
constexpr int ENC_WAVES = ENC_WAVES_PER_BLOCK;
constexpr int ENC_THREADS = ENC_WAVES * 64;
constexpr uint32_t ENC_DIRTY = 0xfffffffeu;  // pair (p,p+1) must be looked up again
constexpr uint32_t ENC_SITE = 0x80000000u;   // pair (p,p+1) is merged in this round
constexpr uint32_t ENC_SENT = 0x40000000u;   // bit 30 of a working token: first token of a sentence (several sentences share a wave)
constexpr uint32_t ENC_IDM = 0x3fffffffu;    // id bits of a working token (bit 31 = TOK_WS)
constexpr uint32_t ENC_UNKP = 0x3ffffff0u;   // placeholder token for a run of unknown chars (bpe.cpp:1517-1527)
constexpr uint32_t ENC_INF = 0xffffffffu;

struct LdsArr {
  uint32_t *p;
  __device__ uint32_t get(int i) const { return p[i]; }
  __device__ void set(int i, uint32_t v) const { p[i] = v; }
  __device__ void amin(int i, uint32_t v) const { atomicMin(&p[i], v); }
};
struct GlbArr {  // HBM scratch for long sentences: bypass the non-coherent L1 (agent-scope atomics)
  uint32_t *p;
  __device__ uint32_t get(int i) const { return __hip_atomic_load(&p[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  __device__ void set(int i, uint32_t v) const { __hip_atomic_store(&p[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  __device__ void amin(int i, uint32_t v) const { atomicMin(&p[i], v); }
};

__device__ inline RuleSlot enc_load_slot(const EncModel &m, uint32_t h) {  // one global_load_dwordx4
  const uint4 v = *reinterpret_cast<const uint4 *>(&m.rules[h]);
  RuleSlot r;
  r.key = ((unsigned long long)v.y << 32) | v.x;
  r.z = v.z;
  r.pad = v.w;
  return r;
}
// priority (= rule index, smaller merges first) of the pair (a,b), or ENC_INF.  The Bloom filter in LDS answers most
// "no such rule" cases without leaving the CU; a positive costs one 16-byte load from the rule hash (L2-resident).
__device__ inline uint32_t enc_pair_prio(const EncModel &m, const uint32_t *bloom, uint32_t a, uint32_t b) {
  const uint32_t h = enc_hash(a, b);
  const uint32_t bits = enc_bloom_bits(h);
  if ((bloom[enc_bloom_word(h)] & bits) != bits) return ENC_INF;
  const unsigned long long key = pair_key(a, b);
  uint32_t sl = h & m.rule_mask;
  for (;;) {
    const RuleSlot r = enc_load_slot(m, sl);
    if (r.key == key) return r.pad;
    if (r.key == PT_EMPTY) return ENC_INF;
    sl = (sl + 1) & m.rule_mask;
  }
}

// priorities of two pairs, both rule-hash loads in flight together (w0 / w1: which of the two are asked for; bloom == nullptr: no filter in
// front of the rule hash -- the dropout kernel, whose LDS holds event queues instead)
__device__ inline void enc_pair_prio2(const EncModel &m, const uint32_t *bloom, bool w0, uint32_t a0, uint32_t b0, bool w1, uint32_t a1, uint32_t b1,
                                      uint32_t *p0, uint32_t *p1) {
  const uint32_t h0 = enc_hash(a0, b0), h1 = enc_hash(a1, b1);
  const uint32_t g0 = enc_bloom_bits(h0), g1 = enc_bloom_bits(h1);
  w0 = w0 && a0 != ENC_UNKP && b0 != ENC_UNKP && (!bloom || (bloom[enc_bloom_word(h0)] & g0) == g0);
  w1 = w1 && a1 != ENC_UNKP && b1 != ENC_UNKP && (!bloom || (bloom[enc_bloom_word(h1)] & g1) == g1);
  uint32_t s0 = h0 & m.rule_mask, s1 = h1 & m.rule_mask;
  RuleSlot r0{PT_EMPTY, 0u, 0u}, r1{PT_EMPTY, 0u, 0u};
  if (w0) r0 = enc_load_slot(m, s0);
  if (w1) r1 = enc_load_slot(m, s1);
  const unsigned long long k0 = pair_key(a0, b0), k1 = pair_key(a1, b1);
  *p0 = ENC_INF;
  *p1 = ENC_INF;
  while (r0.key != PT_EMPTY) {
    if (r0.key == k0) { *p0 = r0.pad; break; }
    s0 = (s0 + 1) & m.rule_mask;
    r0 = enc_load_slot(m, s0);
  }
  while (r1.key != PT_EMPTY) {
    if (r1.key == k1) { *p1 = r1.pad; break; }
    s1 = (s1 + 1) & m.rule_mask;
    r1 = enc_load_slot(m, s1);
  }
}

// id of the token that rule r creates.  Trained models number merged tokens consecutively in rule order, skipping the
// (<= 4) special ids (bpe.cpp:814-837), so z is r + z_base + #{breakpoints <= r}; a hand-made model without that
// structure reads the table instead.
__device__ inline uint32_t enc_rule_z(const EncModel &m, uint32_t r) {
  if (!m.z_affine) return m.rule_z[r];
  return m.z_base + r + (r >= m.z_bp[0]) + (r >= m.z_bp[1]) + (r >= m.z_bp[2]) + (r >= m.z_bp[3]);
}

// ---- BPE-dropout (bpe.cpp:1417-1453 DropoutQueue + :1560-1589) --------------------------------------------------------
// Exact per-word process of the reference: events (rule index, position) in priority order; every pop walks the queue,
// each event is skipped with probability p, the first one not skipped is taken (stale events included: they consume the
// pop), all-skipped ends the word.  One word per lane (the process is inherently sequential within a word); the RNG is a
// counter-based hash of (seed, sentence, word, draw) -- a per-lane stream cannot reproduce the reference's single global
// mt19937 order, so parity is a distribution match (BASELINE.json configs[4]).
struct DropoutArgs {
  unsigned long long thr;   // skip iff hash < thr  (thr = p * 2^64); always_skip for p == 1
  unsigned long long seed;
  int enabled, always_skip;
  int heap_from;            // words of at least this many tokens keep their events in a binary heap instead of a sorted array
  int lds_queues;           // packs of up to ENC_DROP_WCAP tokens keep their event queues in LDS (EvLds)
  int pack_links;           // working arrays in LDS: both links of a position in one word, the third array holds the pairs' rules (dropout_merge)
  int pack_sent;            // at most this many sentences share a pack (0: as many as fit; tests)
  uint32_t *wsl;            // [cap] word start positions
  unsigned long long *ev;   // [3*cap] sorted event queues, word w owns [3*ws, 3*we)
};

// A word's stream of draws is keyed by the encoder's salt, the word's sentence and its number there (dropout_merge: skip()).
__device__ inline unsigned long long drop_key(const DropoutArgs &d, unsigned long long sidx, uint32_t word) {
  return d.seed + sidx * 0x9e3779b97f4a7c15ull + ((unsigned long long)word << 34);
}

// Where a word's event queue lives.  Sentences that fit the wavefront's LDS arrays (the common case) keep it in LDS, an event packed into
// 32 bits (rule index << 9 | position: positions are below 512 there; EncModel: fewer than 2^23 rules) -- the queue in HBM made every
// insertion, shift and pop a chain of dependent global round trips (10^7 sentences of 128 chars: 381 ms against 62 ms without dropout).
// Long sentences (HBM working arrays) keep 64-bit events in the per-wave HBM scratch.  Same order of pops either way.
struct EvGlb {
  typedef unsigned long long raw_t;  // rule index << 32 | position
  unsigned long long *p;
  static __device__ raw_t pack(uint32_t rule, uint32_t pos) { return ((unsigned long long)rule << 32) | (unsigned long long)pos; }
  static __device__ uint32_t rule(raw_t v) { return (uint32_t)(v >> 32); }
  static __device__ uint32_t pos(raw_t v) { return (uint32_t)v; }
  __device__ raw_t get(int i) const { return p[i]; }
  __device__ void set(int i, raw_t v) const { p[i] = v; }
  __device__ EvGlb at(size_t off) const { return EvGlb{p + off}; }
};
struct EvLds {
  typedef uint32_t raw_t;  // rule index << 9 | position: the same order as the pair (rule, position), compared without unpacking
  uint32_t *p;
  static __device__ raw_t pack(uint32_t rule, uint32_t pos) { return (rule << 9) | pos; }
  static __device__ uint32_t rule(raw_t v) { return v >> 9; }
  static __device__ uint32_t pos(raw_t v) { return v & 511u; }
  __device__ raw_t get(int i) const { return p[i]; }
  __device__ void set(int i, raw_t v) const { p[i] = v; }
  __device__ EvLds at(size_t off) const { return EvLds{p + off}; }
};
// dropout: tokens of a pack.  A wave's share of the workgroup's 80 KB is 2 560 words: three working arrays, the event queues (3 events per
// token) and the word starts (a word has at least two tokens) + the next-word counter = 6.5 tokens' worth -> 392 tokens: THREE sentences of
// 128 chars (at most 130 tokens each) where round 4's 256 held two of the shorter ones at best.
constexpr int ENC_DROP_WCAP = 392;
constexpr int ENC_DROP_WORDS = ENC_DROP_WCAP / 2;              // word-start entries; entry [ENC_DROP_WORDS] is the counter
constexpr int ENC_DROP_WAVE_WORDS = 6 * ENC_DROP_WCAP + ENC_DROP_WORDS + 4;  // LDS words per wave

template <class Q>
__device__ inline void ev_insert(const Q &ev, int &ne, typename Q::raw_t key) {
  int j = ne++;
  while (j > 0) {
    const typename Q::raw_t prev = ev.get(j - 1);
    if (prev <= key) break;
    ev.set(j, prev);
    j--;
  }
  ev.set(j, key);
}

// The same queue as a binary min-heap, for long words: the sorted array costs O(queue) per insertion -- a single word of
// 80 000 chars (a base64 blob in the input) kept one lane busy for minutes.  Pops come in the same ascending order, so the draws
// and the result are those of the array (DropoutQueue itself is a std::priority_queue plus the skipped events, bpe.cpp:1417-1453).
template <class Q>
__device__ inline void heap_push(const Q &ev, int &nh, typename Q::raw_t key) {
  int i = nh++;
  while (i > 0) {
    const int p = (i - 1) >> 1;
    const typename Q::raw_t pv = ev.get(p);
    if (pv <= key) break;
    ev.set(i, pv);
    i = p;
  }
  ev.set(i, key);
}
template <class Q>
__device__ inline typename Q::raw_t heap_pop(const Q &ev, int &nh) {
  const typename Q::raw_t top = ev.get(0), last = ev.get(--nh);
  int i = 0;
  for (;;) {
    int c = 2 * i + 1;
    if (c >= nh) break;
    typename Q::raw_t cv = ev.get(c);
    if (c + 1 < nh) {
      const typename Q::raw_t c1 = ev.get(c + 1);
      if (c1 < cv) { c++; cv = c1; }
    }
    if (cv >= last) break;
    ev.set(i, cv);
    i = c;
  }
  if (nh > 0) ev.set(i, last);
  return top;
}

template <class A, class WS, class Q>
__device__ int dropout_merge(const EncModel &m, A wt, A wr /*next*/, A wm /*prev*/, int n, const DropoutArgs &d, unsigned long long sidx,
                             WS wsl /* [words] word start positions */, Q evq /* [3 * tokens] the words' event queues, word w owns [3 ws, 3 we) */,
                             unsigned long long my_sid = 0 /* a pack: lane j holds the index of its j-th non-empty sentence */) {
  const int lane = lane_id();
  const unsigned long long lt = lanemask_lt();
  constexpr uint32_t DEAD = 0xffffffffu, NIL = 0xffffffffu;
  // Round 5: a merge cost three dependent global round trips per lane -- rule_xy[rule] to see whether the popped event is stale, rule_z[rule]
  // for the new token, then the two new pairs' rule-hash slots -- and a wave waits for its longest word.  With the working arrays in LDS
  // (positions below 512) both links of a position share one word (next | prev << 16) and the third array holds the RULE of the pair that
  // starts at each position: an event (rule r at p) is live iff that is still r -- the same test as "the tokens are the rule's x and y"
  // (bpe.cpp:1569-1572: a pair has one rule, a rule one pair) -- and z comes from the rule's number (enc_rule_z).  One trip per merge is left.
  const bool PACK = std::is_same<A, LdsArr>::value && d.pack_links;  // (YTTM_DROPOUT_NO_PACK: the three-trip scheme, for the differential test)
  // A pack of several sentences (LDS word starts): a word's draws are keyed by ITS sentence and its number within that sentence, never by the
  // pack -- the ids do not depend on how a batch was cut into packs (and equal those of the one-sentence-at-a-time HBM path).  A word-start
  // entry is position | number in the sentence << 9 | (sentence - the pack's first) << 18: all three below 512 (ENC_DROP_WCAP), and the words
  // are handed to the lanes by a counter behind the entries -- a lane that is done takes the next word, the wave waits for its longest LANE.
  constexpr bool WPACK = std::is_same<WS, LdsArr>::value;
  constexpr uint32_t NIL16 = 0xffffu;
  // word starts
  int nw = 0;
  {
    int sent_carry = -1, first_carry = 0;  // ordinal of the sentence the previous chunk ended in; the number of that sentence's first word
    for (int c = 0; c < ((n + 63) >> 6); c++) {
      const int p = c * 64 + lane;
      const uint32_t t0 = p < n ? wt.get(p) : 0u;
      const bool ws = p < n && (t0 & TOK_WS);
      const unsigned long long W = __ballot(ws);
      if (WPACK) {
        const unsigned long long S = __ballot(ws && (t0 & ENC_SENT));
        const unsigned long long sle = S & (lt | (1ull << lane));
        const int so = sent_carry + __popcll(sle);
        int first = first_carry;
        if (sle) first = nw + __popcll(W & ((1ull << (63 - __clzll((long long)sle))) - 1ull));
        const unsigned long long sid = __shfl(my_sid, so < 0 ? 0 : so);
        const int wi = nw + __popcll(W & lt);
        if (ws) wsl.set(wi, (uint32_t)p | ((uint32_t)(wi - first) << 9) | ((uint32_t)(sid - sidx) << 18));
        if (S) first_carry = nw + __popcll(W & ((1ull << (63 - __clzll((long long)S))) - 1ull));
        sent_carry += __popcll(S);
      } else {
        if (ws) wsl.set(nw + __popcll(W & lt), (uint32_t)p);
      }
      nw += __popcll(W);
    }
  }
  if (WPACK && lane == 0) wsl.set(ENC_DROP_WORDS, 64u);  // the next word nobody has yet (lanes start with words 0 .. 63)
  // every pair's rule, lanes = positions: the words' first events (bpe.cpp:1556-1558) come from here -- one rule-hash round trip per 64
  // pairs, where a lane looking its word's pairs up one after the other made the wave wait for as many trips as its longest word has pairs.
  // (wm is free until a lane lays its word's links into it.)
  for (int c = 0; c < ((n + 63) >> 6); c += 2) {
    const int p0 = c * 64 + lane, p1 = p0 + 64;
    uint32_t a0 = 0, b0 = 0, a1 = 0, b1 = 0;
    bool w0 = false, w1 = false;
    if (p0 + 1 < n) {
      const uint32_t t = wt.get(p0 + 1);
      a0 = wt.get(p0) & ENC_IDM;
      b0 = t & ENC_IDM;
      w0 = !(t & TOK_WS);
    }
    if (p1 + 1 < n) {
      const uint32_t t = wt.get(p1 + 1);
      a1 = wt.get(p1) & ENC_IDM;
      b1 = t & ENC_IDM;
      w1 = !(t & TOK_WS);
    }
    uint32_t r0, r1;
    enc_pair_prio2(m, nullptr, w0, a0, b0, w1, a1, b1, &r0, &r1);
    if (p0 < n) wm.set(p0, r0);
    if (p1 < n) wm.set(p1, r1);
  }
  wave_sync();
  // The lane's word: set-up (its first events into the queue, its links), then pops until one finds every event skipped.
  int ws = 0, we = 0, cap = 0, ne = 0, head = 0;  // the queue is ev[head, head + ne) (sorted array) or ev[0, ne) (heap)
  bool heap = false;
  uint32_t draw = 0;            // the Weyl counter of the word's draws
  unsigned long long wkey = 0;  // the word's RNG stream: seed, sentence and number in the sentence, mixed once per word
  Q ev = evq;
  // (Measured and dropped in round 5: the events of short words in an UNSORTED bag -- append, swap-remove, a pop by a scan for the smallest:
  // the same examined order, draws and ids, but 195 ms per 10^7 sentences against the sorted array's 140: the scan's 64-bit compares per
  // event cost more than the array's shifts; these loops are bound by instructions, not by LDS latency.)
  // The sorted array never moves its front back: a pop of the smallest event is head++, not a shift of the whole array (round 5; it was the
  // largest loop of a pop), and an insertion shifts only the events behind the new one -- few: a merge's new pairs are younger rules than most
  // of what waits.  head + ne = the events ever inserted <= 3 (tokens - 1) < cap: the word's share of the space is never left.
  auto add = [&](uint32_t rule, uint32_t pos) {
    const typename Q::raw_t key = Q::pack(rule, pos);
    if (heap) heap_push(ev, ne, key);
    else ev_insert(ev.at((size_t)head), ne, key);
  };
  // A draw: the murmur3 finalizer over a Weyl sequence that starts at the word's well-mixed key -- two 32-bit multiplies, where mix64 of
  // (key + draw) cost eight of them (v_mul_lo_u32 runs at a quarter of the VALU's rate, and a draw is the innermost step of a pop).  The
  // threshold is the top 32 bits of p * 2^64: p is resolved to 2.3e-10.
  auto skip = [&]() {
    if (d.always_skip) return true;
    draw += 0x9e3779b9u;
    uint32_t h = draw ^ (uint32_t)(wkey >> 32);
    h ^= h >> 16;
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    h *= 0xc2b2ae35u;
    h ^= h >> 16;
    return h < (uint32_t)(d.thr >> 32);
  };
  auto setup = [&](int w) {
    const uint32_t e0 = wsl.get(w);
    ws = WPACK ? (int)(e0 & 511u) : (int)e0;
    we = w + 1 < nw ? (WPACK ? (int)(wsl.get(w + 1) & 511u) : (int)wsl.get(w + 1)) : n;
    wkey = mix64(drop_key(d, WPACK ? sidx + (unsigned long long)(e0 >> 18) : sidx, WPACK ? (e0 >> 9) & 511u : (uint32_t)w));
    ev = evq.at(3 * (size_t)ws);
    cap = 3 * (we - ws);             // the word's share of the queue space: every event it can ever hold
    heap = we - ws >= d.heap_from;   // (skipped events of a pop wait at the top end of that space)
    ne = 0;
    head = 0;
    draw = (uint32_t)wkey;
    for (int i = ws; i < we; i++) {
      const uint32_t r = wm.get(i);
      if (i + 1 < we && r != ENC_INF) add(r, (uint32_t)i);
      if (PACK) {
        wr.set(i, (i + 1 < we ? (uint32_t)(i + 1) : NIL16) | ((i > ws ? (uint32_t)(i - 1) : NIL16) << 16));
        wm.set(i, i + 1 < we ? r : ENC_INF);  // the rule of the pair (i, i + 1)
      } else {
        wr.set(i, i + 1 < we ? (uint32_t)(i + 1) : NIL);
        wm.set(i, i > ws ? (uint32_t)(i - 1) : NIL);
      }
    }
  };
  auto pop = [&]() -> bool {  // false: every event was skipped, the word is finished (bpe.cpp:1431-1437)
    typename Q::raw_t e = 0;
    bool found = false;
    if (heap) {
      int ns = 0;
      while (ne > 0) {
        e = heap_pop(ev, ne);
        if (!skip()) { found = true; break; }
        ev.set(cap - 1 - ns, e);
        ns++;
      }
      for (int k = 0; k < ns; k++) heap_push(ev, ne, ev.get(cap - 1 - k));
    } else {
      int acc = -1;
      for (int j = 0; j < ne; j++) {
        if (!skip()) { acc = j; break; }
      }
      if (acc >= 0) {
        found = true;
        e = ev.get(head + acc);
        for (int j = acc; j > 0; j--) ev.set(head + j, ev.get(head + j - 1));  // (the skipped ones before it move up one; acc is 0 nine times of ten)
        head++;
        ne--;
      }
    }
    if (!found) return false;
    const uint32_t rule = Q::rule(e);
    const int p1 = (int)Q::pos(e);
    const uint32_t t1 = wt.get(p1);
    if (PACK) {
      if (t1 == DEAD || wm.get(p1) != rule) return true;  // stale (:1569-1572): the position is gone, or its pair is no longer this rule's
      const uint32_t l1 = wr.get(p1);
      const uint32_t p2 = l1 & 0xffffu, p0 = l1 >> 16;   // (p2 exists: the last position of a word never has a rule)
      const uint32_t p3 = wr.get((int)p2) & 0xffffu;
      const uint32_t zt = enc_rule_z(m, rule);
      wt.set((int)p2, DEAD);
      wt.set(p1, zt | (t1 & (TOK_WS | ENC_SENT)));
      wr.set(p1, p3 | (p0 << 16));
      if (p3 != NIL16) wr.set((int)p3, (wr.get((int)p3) & 0xffffu) | ((uint32_t)p1 << 16));
      // the two pairs the merge made (:1580-1585), their rule-hash loads in flight together
      uint32_t rl, rr;
      enc_pair_prio2(m, nullptr, p0 != NIL16, p0 != NIL16 ? wt.get((int)p0) & ENC_IDM : 0u, zt, p3 != NIL16, zt, p3 != NIL16 ? wt.get((int)p3) & ENC_IDM : 0u, &rl, &rr);
      if (p0 != NIL16) wm.set((int)p0, rl);
      wm.set(p1, rr);
      if (rl != ENC_INF) add(rl, (uint32_t)p0);
      if (rr != ENC_INF) add(rr, (uint32_t)p1);
      return true;
    }
    const uint32_t p2 = wr.get(p1);
    const unsigned long long xy = m.rule_xy[rule];
    if (t1 == DEAD || (t1 & ENC_IDM) != (uint32_t)(xy >> 32) || p2 == NIL || (wt.get((int)p2) & ENC_IDM) != (uint32_t)xy) return true;  // :1569-1572
    const uint32_t p0 = wm.get(p1), p3 = wr.get((int)p2);
    wt.set((int)p2, DEAD);
    wr.set((int)p2, NIL);
    wt.set(p1, enc_rule_z(m, rule) | (t1 & (TOK_WS | ENC_SENT)));
    wr.set(p1, p3);
    if (p3 != NIL) wm.set((int)p3, (uint32_t)p1);
    {  // the two pairs the merge made (:1580-1585), their rule-hash loads in flight together
      const uint32_t zt = wt.get(p1) & ENC_IDM;
      uint32_t rl, rr;
      enc_pair_prio2(m, nullptr, p0 != NIL, p0 != NIL ? wt.get((int)p0) & ENC_IDM : 0u, zt, p3 != NIL, zt, p3 != NIL ? wt.get((int)p3) & ENC_IDM : 0u, &rl, &rr);
      if (rl != ENC_INF) add(rl, (uint32_t)p0);
      if (rr != ENC_INF) add(rr, (uint32_t)p1);
    }
    return true;
  };
  if constexpr (!WPACK) {  // (a long sentence on the HBM scratch: a lane finishes its word, the wave its longest word, then every lane takes its next one)
    for (int w = lane; w < nw; w += 64) {
      setup(w);
      while (pop()) {}
    }
  } else {
    // One flat loop: an iteration is one pop of the lane's word (or, for a lane that has just taken a word, the word's set-up first): a lane
    // that is done takes the next word from the counter while the others go on -- the wave waits for its longest LANE, not once per word.
    int w = lane;
    bool fresh = true;
    while (w < nw) {
      if (fresh) {
        setup(w);
        fresh = false;
      }
      if (!pop()) {
        w = (int)atomicAdd(&wsl.p[ENC_DROP_WORDS], 1u);
        fresh = true;
      }
    }
  }
  wave_sync();
  // drop dead nodes (order is preserved, so this is the linked-list traversal of bpe.cpp:1597)
  int base = 0;
  for (int c = 0; c < ((n + 63) >> 6); c++) {
    const int p = c * 64 + lane;
    const uint32_t t0 = p < n ? wt.get(p) : DEAD;
    const bool alive = t0 != DEAD;
    const unsigned long long AM = __ballot(alive);
    wave_sync();
    if (alive) wt.set(base + __popcll(AM & lt), t0);
    base += __popcll(AM);
    wave_sync();
  }
  return base;
}

// UTF-8 decode + char -> token, word starts, unknown-run collapse (bpe.cpp:1497-1530).  Appends the sentence's tokens to
// wt[n0...) and returns the new end.  A sentence of B bytes yields at most B+1 tokens.
template <class A>
__device__ int enc_tokenize(const EncModel &m, const uint8_t *__restrict__ s, unsigned long long nbytes, A wt, int n0) {
  const int lane = lane_id();
  const unsigned long long lt = lanemask_lt();
  int n = n0;
  bool carry_space = true, carry_unk = false;  // class of the last valid char before this step (start of text acts like a space)
  for (unsigned long long b0 = 0; b0 < nbytes; b0 += 64) {
    const unsigned long long i = b0 + (unsigned long long)lane;
    bool valid = false, space = false, unk = false;
    uint32_t id = 0;
    if (i < nbytes && u8_is_start(s, i, nbytes)) {
      uint32_t len;
      const uint32_t cp = u8_decode_at(s, i, nbytes, &len);
      if (cp != INVALID_CP) {  // invalid bytes are dropped (utf8.cpp:111-128)
        valid = true;
        id = m.cpmap[cp];
        space = id == CP_SPACE;
        unk = id == CP_UNK;
      }
    }
    const unsigned long long V = __ballot(valid), S = __ballot(space), U = __ballot(unk);
    bool prev_space = carry_space, prev_unk = carry_unk;
    const unsigned long long pv = V & lt;
    if (pv) {
      const int j = 63 - __clzll((long long)pv);
      prev_space = (S >> j) & 1ull;
      prev_unk = (U >> j) & 1ull;
    }
    int emit = 0;
    if (valid && !space) {
      const bool wstart = prev_space;
      if (unk && prev_unk && !wstart) emit = 0;  // continues an unknown run
      else emit = wstart ? 2 : 1;
    }
    const unsigned long long e1 = __ballot(emit >= 1), e2 = __ballot(emit == 2);
    const int pos = n + __popcll(e1 & lt) + __popcll(e2 & lt);
    const uint32_t tv = unk ? ENC_UNKP : id;
    if (emit == 2) {
      wt.set(pos, m.space_id | TOK_WS);  // every word starts with "▁" (bpe.cpp:1514)
      wt.set(pos + 1, tv);
    } else if (emit == 1) {
      wt.set(pos, tv);
    }
    n += __popcll(e1) + __popcll(e2);
    if (V) {
      const int j = 63 - __clzll((long long)V);
      carry_space = (S >> j) & 1ull;
      carry_unk = (U >> j) & 1ull;
    }
  }
  return n;
}


// Merge rounds of the deterministic encoder over the n tokens in wt (words = TOK_WS segments; sentence boundaries are word
// boundaries, so several sentences can share the arrays).  Returns the new token count.
template <class A>
__device__ int merge_rounds(const EncModel &m, const uint32_t *bloom, A wt, A wr, A wm, int n) {
  const int lane = lane_id();
  const unsigned long long lt = lanemask_lt();
  // wr[p] = priority (rule index) of the pair (p,p+1), ENC_INF if it has no rule, ENC_DIRTY if unknown.  Only pairs next
  // to a merge change, so after the first round a handful of pairs per word are looked up again; a pair is first
  // tested against the LDS-resident Bloom filter of all rules, and only a positive goes to the rule hash in L2/HBM.
  for (int c = 0; c < ((n + 63) >> 6); c++) {
    const int p = c * 64 + lane;
    if (p < n) {
      wr.set(p, ENC_DIRTY);
      if (wt.get(p) & TOK_WS) wm.set(p, ENC_INF);
    }
  }
  wave_sync();
  for (;;) {
    const int nchunks = (n + 63) >> 6;
    // phase 1+2: (re)compute dirty pairs and fold every priority into its word's minimum (kept at the word's first
    // position; reset to ENC_INF when that position was written)
    int carry_ws = 0;
    for (int c = 0; c < nchunks; c++) {
      const int p = c * 64 + lane;
      const uint32_t t0 = p < n ? wt.get(p) : 0u;
      const bool ws = p < n && (t0 & TOK_WS);
      const unsigned long long W = __ballot(ws);
      int wsp = carry_ws;
      const unsigned long long wle = W & ((2ull << lane) - 1ull);
      if (wle) wsp = c * 64 + 63 - __clzll((long long)wle);
      if (W) carry_ws = c * 64 + 63 - __clzll((long long)W);
      if (p < n) {
        uint32_t r = wr.get(p);
        if (r == ENC_DIRTY) {
          r = ENC_INF;
          if (p + 1 < n) {
            const uint32_t t1 = wt.get(p + 1);
            const uint32_t a = t0 & ENC_IDM, b = t1 & ENC_IDM;
            if (!(t1 & TOK_WS) && a != ENC_UNKP && b != ENC_UNKP) r = enc_pair_prio(m, bloom, a, b);
          }
          wr.set(p, r);
        }
        if (r != ENC_INF) wm.amin(wsp, r);
      }
    }
    wave_sync();
    // phase 3: merge sites = pairs carrying their word's minimum; x==x pairs only at even offsets from the run start
    // (= the left-to-right greedy of the reference).  Marked in place (ENC_SITE) before anything moves.
    bool any = false;
    carry_ws = 0;
    for (int c = 0; c < nchunks; c++) {
      const int p = c * 64 + lane;
      const uint32_t t0 = p < n ? wt.get(p) : 0;
      const bool ws = p < n && (t0 & TOK_WS);
      const unsigned long long W = __ballot(ws);
      int wsp = carry_ws;
      const unsigned long long wle = W & ((2ull << lane) - 1ull);
      if (wle) wsp = c * 64 + 63 - __clzll((long long)wle);
      if (W) carry_ws = c * 64 + 63 - __clzll((long long)W);
      bool site = false;
      if (p < n) {
        const uint32_t r = wr.get(p);
        if (r != ENC_INF && r == wm.get(wsp)) {
          site = true;
          const uint32_t a = t0 & ENC_IDM;
          if ((wt.get(p + 1) & ENC_IDM) == a) {
            int q = p;
            while (q > 0 && !(wt.get(q) & TOK_WS) && (wt.get(q - 1) & ENC_IDM) == a) q--;
            site = ((p - q) & 1) == 0;
          }
          if (site) wr.set(p, r | ENC_SITE);
        }
      }
      any = any || __ballot(site) != 0;
    }
    wave_sync();
    if (!any) break;
    // phase 4: apply + compact in place (ascending chunks; writes never pass unread data).  A surviving pair keeps its
    // priority unless one of its two tokens changed.
    int base = 0;
    bool prev_site = false;  // site flag of the last position of the previous chunk
    for (int c = 0; c < nchunks; c++) {
      const int p = c * 64 + lane;
      uint32_t t0 = 0, r = ENC_INF, r_next = ENC_INF;
      if (p < n) {
        t0 = wt.get(p);
        r = wr.get(p);
        if (p + 1 < n) r_next = wr.get(p + 1);
      }
      const bool site = p < n && r != ENC_INF && r != ENC_DIRTY && (r & ENC_SITE);
      const bool next_site = r_next != ENC_INF && r_next != ENC_DIRTY && (r_next & ENC_SITE);
      const unsigned long long SM = __ballot(site);
      const bool dead = lane == 0 ? prev_site : ((SM >> (lane - 1)) & 1ull);
      const bool alive = p < n && !dead;
      const unsigned long long AM = __ballot(alive);
      uint32_t nt = t0, nr = r;
      if (site) {
        nt = enc_rule_z(m, r & ~ENC_SITE) | (t0 & (TOK_WS | ENC_SENT));
        nr = ENC_DIRTY;
      } else if (next_site) {
        nr = ENC_DIRTY;
      }
      wave_sync();  // all lanes have read their inputs before anyone overwrites lower positions
      if (alive) {
        const int np = base + __popcll(AM & lt);
        wt.set(np, nt);
        wr.set(np, nr);
        if (nt & TOK_WS) wm.set(np, ENC_INF);
      }
      base += __popcll(AM);
      prev_site = (SM >> 63) & 1ull;
      wave_sync();
    }
    n = base;
  }
  return n;
}

// ---- merge rounds, one word per lane --------------------------------------------------------------------------------------
// The rounds above cost three passes over every chunk of the pack, and the pack needs as many rounds as its longest word has merges,
// however few lanes still have anything to do.  When the pack is many words of moderate length (the word cache's distinct words; the
// words of a few packed sentences) a lane takes a word instead and walks it alone: per round a scan of ITS pairs for the smallest rule's
// leftmost site, the merge, the tail moved up one place, and the look-ups of the two pairs the merge made (the lanes' rule-hash loads of a
// round go out together).  No ballots, no atomics, no wave-wide passes until the words are done; the order of merges inside a word is the
// reference's (bpe.cpp:1560-1589): by rule, then left to right.
constexpr uint32_t ENC_DEAD = 0xffffffffu;  // a position a finished word no longer uses (never a token: id bits above ENC_UNKP)

// One lane, one word: tokens wt[ws, we), pair priorities wr[ws, we) (the last one ENC_INF).  Returns the new end.
// Both loops go four places at a time -- the LDS reads of a step in flight together, one wait: a lane's walk is a chain of LDS round
// trips, and with one word per lane there is little else to hide them behind.  (Indices past the word's end are clamped to its last place,
// whose priority is ENC_INF: read twice, never the minimum.)
__device__ inline int lane_rounds(const EncModel &m, const uint32_t *bloom, LdsArr wt, LdsArr wr, int ws, int we) {
  for (;;) {
    // the word's smallest rule, leftmost site
    uint32_t mn = ENC_INF;
    int at = ws;
    const int last = we - 1;
    for (int i = ws; i < we; i += 4) {
      const int i1 = i + 1 < last ? i + 1 : last, i2 = i + 2 < last ? i + 2 : last, i3 = i + 3 < last ? i + 3 : last;
      const uint32_t r0 = wr.get(i), r1 = wr.get(i1), r2 = wr.get(i2), r3 = wr.get(i3);
      if (r0 < mn) { mn = r0; at = i; }
      if (r1 < mn) { mn = r1; at = i1; }
      if (r2 < mn) { mn = r2; at = i2; }
      if (r3 < mn) { mn = r3; at = i3; }
    }
    if (mn == ENC_INF) break;
    // (at, at + 1) -> z, the tail moves up one place.  Another site of the same rule further right is the next round's leftmost
    // minimum -- the pairs a merge makes rank behind the rule that made their token -- so sites go left to right like the reference's.
    const uint32_t z = enc_rule_z(m, mn);
    const uint32_t flags = wt.get(at) & (TOK_WS | ENC_SENT);
    const uint32_t tr = at + 2 < we ? wt.get(at + 2) : TOK_WS;  // the token right of the pair, if the word has one
    wt.set(at, z | flags);
    for (int i = at + 1; i < last; i += 4) {  // place i takes what place i + 1 holds
      const int s1 = i + 1, s2 = i + 2 < last ? i + 2 : last, s3 = i + 3 < last ? i + 3 : last, s4 = i + 4 < last ? i + 4 : last;
      const uint32_t t1 = wt.get(s1), t2 = wt.get(s2), t3 = wt.get(s3), t4 = wt.get(s4);
      const uint32_t r1 = wr.get(s1), r2 = wr.get(s2), r3 = wr.get(s3), r4 = wr.get(s4);
      wt.set(i, t1);
      wr.set(i, r1);
      if (i + 1 < last) { wt.set(i + 1, t2); wr.set(i + 1, r2); }
      if (i + 2 < last) { wt.set(i + 2, t3); wr.set(i + 2, r3); }
      if (i + 3 < last) { wt.set(i + 3, t4); wr.set(i + 3, r4); }
    }
    we--;
    const bool hl = at > ws && !(flags & TOK_WS), hr = !(tr & TOK_WS);  // (no pair across a word start)
    uint32_t pl, pr;
    enc_pair_prio2(m, bloom, hl, hl ? wt.get(at - 1) & ENC_IDM : 0u, z, hr, z, tr & ENC_IDM, &pl, &pr);
    if (hl) wr.set(at - 1, pl);
    wr.set(at, pr);
  }
  return we;
}

// Returns the new token count, or -1 when the pack is not the shape for this (a word longer than lane_max tokens: one lane would walk it
// while 63 wait) -- nothing but wr / wm has been written then and merge_rounds takes over.
__device__ int merge_lanes(const EncModel &m, const uint32_t *bloom, LdsArr wt, LdsArr wr, LdsArr wm, int n, int lane_max) {
  const int lane = lane_id();
  const unsigned long long lt = lanemask_lt();
  const int nchunks = (n + 63) >> 6;
  // the words: wm[w] = first token of word w, wm[nw] = n (a word has at least two tokens, so the list fits)
  int nw = 0;
  for (int c = 0; c < nchunks; c++) {
    const int p = c * 64 + lane;
    const bool ws = p < n && (wt.get(p) & TOK_WS);
    const unsigned long long W = __ballot(ws);
    if (ws) wm.set(nw + (int)__popcll(W & lt), (uint32_t)p);
    nw += (int)__popcll(W);
  }
  if (lane == 0) wm.set(nw, (uint32_t)n);
  wave_sync();
  int longest = 0;
  for (int w = lane; w < nw; w += 64) {
    const int len = (int)(wm.get(w + 1) - wm.get(w));
    longest = len > longest ? len : longest;
  }
  if (__ballot(longest > lane_max) != 0ull) return -1;
  // every pair's priority, lanes = positions (the last pair of a word has none)
  for (int c = 0; c < nchunks; c++) {
    const int p = c * 64 + lane;
    if (p < n) {
      uint32_t r = ENC_INF;
      if (p + 1 < n) {
        const uint32_t t0 = wt.get(p), t1 = wt.get(p + 1);
        const uint32_t a = t0 & ENC_IDM, b = t1 & ENC_IDM;
        if (!(t1 & TOK_WS) && a != ENC_UNKP && b != ENC_UNKP) r = enc_pair_prio(m, bloom, a, b);
      }
      wr.set(p, r);
    }
  }
  wave_sync();
  for (int w0 = 0; w0 < nw; w0 += 64) {
    const int w = w0 + lane;
    int ws = 0, we = 0;
    if (w < nw) {
      ws = (int)wm.get(w);
      we = (int)wm.get(w + 1);
    }
    const int we0 = we;
    we = lane_rounds(m, bloom, wt, wr, ws, we);
    for (int i = we; i < we0; i++) wt.set(i, ENC_DEAD);
  }
  wave_sync();
  // close the gaps the words left
  int base = 0;
  for (int c = 0; c < nchunks; c++) {
    const int p = c * 64 + lane;
    const uint32_t t0 = p < n ? wt.get(p) : ENC_DEAD;
    const bool alive = t0 != ENC_DEAD;
    const unsigned long long AM = __ballot(alive);
    wave_sync();
    if (alive) wt.set(base + (int)__popcll(AM & lt), t0);
    base += (int)__popcll(AM);
    wave_sync();
  }
  return base;
}

// Cooperative path: one wavefront encodes one sentence, lanes = token positions.  wt = tokens (bit31 = first token of a
// word), wr = priority of the pair that starts at p, wm = per-word minimum priority stored at the word's first position.
// Sentences too long for the LDS arrays run the same code on HBM scratch (GlbArr).
template <class A>
__device__ void encode_wave(const EncModel &m, const uint32_t *bloom, const uint8_t *__restrict__ s, unsigned long long nbytes, A wt, A wr,
                            A wm, int bos, int eos, int reverse, int32_t *__restrict__ out, uint32_t *__restrict__ count_out,
                            const DropoutArgs &drop, unsigned long long sidx) {
  const int lane = lane_id();
  const unsigned long long lt = lanemask_lt();
  int n = enc_tokenize<A>(m, s, nbytes, wt, 0);
  wave_sync();
  // ---- B. merge rounds -----------------------------------------------------------------------------------------------
  if (drop.enabled) n = dropout_merge<A, GlbArr, EvGlb>(m, wt, wr, wm, n, drop, sidx, GlbArr{drop.wsl}, EvGlb{drop.ev});  // (HBM scratch; the fence of old: GlbArr reads at agent scope)
  else {
    n = merge_rounds<A>(m, bloom, wt, wr, wm, n);
  }
  // ---- C. output (bpe.cpp:1591-1630): unknown runs -> unk_id; the id-0 quirk drops an unmerged leading "▁" whose id is 0
  const int nb = bos ? 1 : 0;
  int total = nb;
  for (int c = 0; c < ((n + 63) >> 6); c++) {
    const int p = c * 64 + lane;
    bool emit = false;
    uint32_t t0 = 0;
    if (p < n) {
      t0 = wt.get(p);
      emit = !(t0 == TOK_WS);  // == (id 0 | TOK_WS): only the space token can be a word start with id 0
    }
    const unsigned long long E = __ballot(emit);
    if (emit) {
      const uint32_t id = t0 & ENC_IDM;
      wm.set(total - nb + __popcll(E & lt), id == ENC_UNKP ? (uint32_t)m.unk_id : id);
    }
    total += __popcll(E);
  }
  wave_sync();
  const int n_ids = total + (eos ? 1 : 0);
  for (int k = lane; k < n_ids; k += 64) {
    int32_t v;
    if (bos && k == 0) v = m.bos_id;
    else if (eos && k == n_ids - 1) v = m.eos_id;
    else v = (int32_t)wm.get(k - nb);
    out[reverse ? (n_ids - 1 - k) : k] = v;
  }
  if (lane == 0) *count_out = (uint32_t)n_ids;
}

// What K5 encodes: sentence j = bytes [off[j], off[j + 1]) of the text, its ids go to scratch + 2 off[j] + 2 j (room for 2 B + 2 ids:
// B + 1 tokens, bos, eos).  Or, for the word cache (k_wcache.hip), word j = bytes [off[j], end[j]) in any order, ids at scratch + 2 off[j].
struct SentView {
  const unsigned long long *off, *end;  // end == nullptr: sentences back to back
  __device__ unsigned long long lo(unsigned long long j) const { return off[j]; }
  __device__ unsigned long long hi(unsigned long long j) const { return end ? end[j] : off[j + 1]; }
  __device__ unsigned long long spos(unsigned long long j) const { return end ? 2 * off[j] : 2 * off[j] + 2 * j; }
};

// Several consecutive sentences share one wavefront's arrays (dropout off, LDS path): a 128-byte sentence is 129 tokens --
// two full chunks and a third with one lane busy -- and the fixed cost of a round is per chunk.  Sentences [s, e) are
// tokenized back to back while they fit, merged together (words are independent), and written out one by one; returns how
// many sentences were consumed (>= 1: the caller made sure the first one fits).
__device__ int encode_pack(const EncModel &m, const uint32_t *bloom, const uint8_t *__restrict__ text,
                           const SentView &sv, unsigned long long s, unsigned long long e, LdsArr wt, LdsArr wr,
                           LdsArr wm, int bos, int eos, int reverse, int32_t *__restrict__ scratch_ids, uint32_t *__restrict__ counts,
                           const DropoutArgs &drop, int wcap /* tokens a pack may hold */, LdsArr dws, EvLds dq /* dropout: word starts, event queues */,
                           int lane_max /* words of up to this many tokens: one per lane (merge_lanes); 0 = never */) {
  const int lane = lane_id();
  const unsigned long long lt = lanemask_lt();
  int n = 0, consumed = 0, k = 0;
  unsigned long long my_sid = 0;  // lane j: index of the j-th non-empty sentence of the pack
  const int kmax = drop.enabled && drop.pack_sent > 0 && drop.pack_sent < 64 ? drop.pack_sent : 64;
  for (unsigned long long j = s; j < e && k < kmax; j++) {
    const unsigned long long b0 = sv.lo(j), nbytes = sv.hi(j) - b0;
    if (nbytes + 1 > (unsigned long long)(wcap - n)) break;
    const int n0 = n;
    n = enc_tokenize<LdsArr>(m, text + b0, nbytes, wt, n0);
    wave_sync();
    if (n == n0) {  // no token at all: only bos / eos
      const int n_ids = (bos ? 1 : 0) + (eos ? 1 : 0);
      int32_t *out = scratch_ids + sv.spos(j);
      if (lane == 0) {
        if (bos) out[reverse ? n_ids - 1 : 0] = m.bos_id;
        if (eos) out[reverse ? 0 : n_ids - 1] = m.eos_id;
        counts[j] = (uint32_t)n_ids;
      }
    } else {
      if (lane == 0) wt.set(n0, wt.get(n0) | ENC_SENT);
      if (lane == k) my_sid = j;
      k++;
    }
    consumed++;
  }
  wave_sync();
  if (k == 0) return consumed;
  if (drop.enabled) {  // BPE-dropout: a word per lane at a time (the RNG stream of a word is keyed by its sentence and its number there)
    n = dropout_merge<LdsArr, LdsArr, EvLds>(m, wt, wr, wm, n, drop, s, dws, dq, my_sid);
  } else {
    const int nl = lane_max > 0 ? merge_lanes(m, bloom, wt, wr, wm, n, lane_max) : -1;
    n = nl >= 0 ? nl : merge_rounds<LdsArr>(m, bloom, wt, wr, wm, n);
  }
  wave_sync();
  // ---- output (bpe.cpp:1591-1630).  wm[o] = ids of sentence o of the pack, wr[o] = ids emitted before its first token
  if (lane < k) wm.set(lane, 0);
  wave_sync();
  const int nchunks = (n + 63) >> 6;
  int ord_carry = -1, emitted = 0;  // ordinal of the sentence the previous chunk ended in; ids emitted so far
  for (int c = 0; c < nchunks; c++) {
    const int p = c * 64 + lane;
    const uint32_t t0 = p < n ? wt.get(p) : 0u;
    const bool sent = p < n && (t0 & ENC_SENT);
    const bool emit = p < n && (t0 & ~ENC_SENT) != TOK_WS;  // (id 0 | TOK_WS): an unmerged "▁" with id 0 is dropped
    const unsigned long long SB = __ballot(sent), E = __ballot(emit);
    const int ord = ord_carry + __popcll(SB & (lt | (1ull << lane)));
    if (sent) wr.set(ord, (uint32_t)(emitted + __popcll(E & lt)));
    if (emit) atomicAdd(&wm.p[ord], 1u);
    ord_carry += __popcll(SB);
    emitted += __popcll(E);
  }
  wave_sync();
  ord_carry = -1;
  emitted = 0;
  const int nb = bos ? 1 : 0;
  for (int c = 0; c < nchunks; c++) {
    const int p = c * 64 + lane;
    const uint32_t t0 = p < n ? wt.get(p) : 0u;
    const bool sent = p < n && (t0 & ENC_SENT);
    const bool emit = p < n && (t0 & ~ENC_SENT) != TOK_WS;
    const unsigned long long SB = __ballot(sent), E = __ballot(emit);
    const int ord = ord_carry + __popcll(SB & (lt | (1ull << lane)));
    const unsigned long long sid = __shfl(my_sid, ord < 0 ? 0 : ord);
    if (p < n) {
      const int n_ids = (int)wm.get(ord) + nb + (eos ? 1 : 0);
      int32_t *out = scratch_ids + sv.spos(sid);
      if (emit) {
        const int q = nb + emitted + __popcll(E & lt) - (int)wr.get(ord);
        const uint32_t id = t0 & ENC_IDM;
        out[reverse ? (n_ids - 1 - q) : q] = (int32_t)(id == ENC_UNKP ? (uint32_t)m.unk_id : id);
      }
      if (sent) {
        if (bos) out[reverse ? n_ids - 1 : 0] = m.bos_id;
        if (eos) out[reverse ? 0 : n_ids - 1] = m.eos_id;
        counts[sid] = (uint32_t)n_ids;
      }
    }
    ord_carry += __popcll(SB);
    emitted += __popcll(E);
  }
  wave_sync();
  return consumed;
}

// DROP: the BPE-dropout instantiation -- no Bloom filter of the rules (its look-ups go to the rule hash), the wave's LDS share holds the
// working arrays, word starts and event queues of a pack of up to ENC_DROP_WCAP tokens instead (80 KB per workgroup: two per CU, as without dropout).
template <bool DROP>
__global__ __launch_bounds__(ENC_THREADS) void k5_encode(EncModel m, const uint8_t *__restrict__ text,
                                                   SentView sv, unsigned long long n_sent, int bos,
                                                   int eos, int reverse, int32_t *__restrict__ scratch_ids,
                                                   uint32_t *__restrict__ counts, uint32_t *__restrict__ work,
                                                   unsigned long long work_stride, DropoutArgs drop, unsigned long long drop_stride, unsigned int group, int lane_max) {
  // 80 KB, two workgroups per CU (one byte more and it is one).  Without dropout: 48 KB of working arrays (3 x 512 tokens per wave) + the
  // rules' Bloom filter (32 KB).  With dropout: per wave three arrays of ENC_DROP_WCAP tokens, the event queues and the word starts.
  __shared__ uint32_t smem[ENC_WAVES * 3 * ENC_WCAP + ENC_BLOOM_WORDS];
  static_assert(ENC_WAVES * ENC_DROP_WAVE_WORDS <= ENC_WAVES * 3 * ENC_WCAP + ENC_BLOOM_WORDS, "the dropout layout fits the same 80 KB");
  static_assert(ENC_DROP_WCAP <= 512, "positions of a pack are packed into 9 bits");
  uint32_t *bloom = smem + ENC_WAVES * 3 * ENC_WCAP;
  if (!DROP) {
    for (int i = (int)threadIdx.x; i < ENC_BLOOM_WORDS; i += ENC_THREADS) bloom[i] = m.bloom[i];
    __syncthreads();
  }
  const int wave = uni((int)(threadIdx.x >> 6));
  const unsigned long long gw = (unsigned long long)blockIdx.x * ENC_WAVES + wave;
  const unsigned long long n_waves = (unsigned long long)gridDim.x * ENC_WAVES;
  // tokens a pack may hold: the LDS arrays; with dropout what the LDS event queues take (0: every sentence through the HBM scratch --
  // a model with 2^23 rules or more, whose rule indices do not fit the packed events)
  const int wcap = DROP ? (drop.lds_queues ? ENC_DROP_WCAP : 0) : ENC_WCAP;
  // a wavefront owns groups of `group` consecutive sentences and packs as many of a group at a time as fit its LDS arrays
  const unsigned long long n_groups = (n_sent + group - 1) / group;
  for (unsigned long long grp = gw; grp < n_groups; grp += n_waves) {
    unsigned long long sidx = grp * group;
    const unsigned long long grp_end = sidx + group < n_sent ? sidx + group : n_sent;
    while (sidx < grp_end) {
      const unsigned long long b0 = sv.lo(sidx), b1 = sv.hi(sidx);
      const unsigned long long nbytes = b1 - b0;
      uint32_t *const wbase = DROP ? smem + wave * ENC_DROP_WAVE_WORDS : smem + wave * 3 * ENC_WCAP;
      constexpr int WSTR = DROP ? ENC_DROP_WCAP : ENC_WCAP;
      LdsArr a{wbase}, b{wbase + WSTR}, c{wbase + 2 * WSTR};
      DropoutArgs d = drop;
      if (d.enabled) {  // per-wave slice of the dropout scratch (long sentences): word starts, then the event queues
        d.wsl = drop.wsl + gw * 7 * drop_stride;
        d.ev = reinterpret_cast<unsigned long long *>(drop.wsl + gw * 7 * drop_stride + drop_stride);
      }
      if (nbytes + 1 <= (unsigned long long)wcap) {
        sidx += (unsigned long long)encode_pack(m, bloom, text, sv, sidx, grp_end, a, b, c, bos, eos, reverse, scratch_ids, counts, d, wcap,
                                                LdsArr{wbase + 6 * ENC_DROP_WCAP}, EvLds{wbase + 3 * ENC_DROP_WCAP}, lane_max);
        continue;
      }
      // too long for the LDS arrays: one sentence at a time on the wavefront's HBM scratch
      int32_t *out = scratch_ids + sv.spos(sidx);  // capacity 2*nbytes + 2 ids per sentence
      uint32_t *w = work + gw * 3 * work_stride;
      GlbArr ga{w}, gb{w + work_stride}, gc{w + 2 * work_stride};
      encode_wave(m, bloom, text + b0, nbytes, ga, gb, gc, bos, eos, reverse, out, &counts[sidx], d, sidx);
      wave_sync();
      sidx++;
    }
  }
}

// ---- K5 for the word cache's distinct words (k_wcache.hip) -------------------------------------------------------------------
// Item j = bytes [ustart[j], uend[j]) of the text: one word, its ids to scratch + 2 ustart[j], where and how many to its table slot (pub); no bos / eos,
// no dropout.  A wavefront packs up to 64 consecutive ones -- one per lane -- into its share of the
// LDS: tokens and pair priorities only (lane_rounds), so a pack holds ENCW_LANE_TOKENS tokens, and one workgroup of 16 waves per CU
// shares a single copy of the rules' Bloom filter: 16 x 7.5 KB + 32 KB = 152 KB (+ 512 B: the ASCII half of the char map).  An item of more than lane_max tokens is walked by the
// whole wave instead (encode_wave: thirds of the share, or the HBM scratch when even that is too small).
__device__ inline void word_publish(const WordPublish &pub, unsigned long long u, unsigned long long ids_at, uint32_t n) {
  if (u < pub.n_table) {
    pub.slot[pub.uslot[u]] = (ids_at << 20) | n;  // (a cached word has at most 65 536 ids)
  } else {
    pub.extra[2 * (u - pub.n_table)] = ids_at;
    pub.extra[2 * (u - pub.n_table) + 1] = n;
  }
}
constexpr int ENCW_WAVES = 16;
constexpr int ENCW_POOL = 1920;                    // LDS words per wave
constexpr int ENCW_LANE_TOKENS = ENCW_POOL / 2;    // a pack of words, one per lane: tokens + priorities
constexpr int ENCW_WAVE_TOKENS = ENCW_POOL / 3;    // a word the wave walks together: tokens + priorities + minima
__global__ __launch_bounds__(ENCW_WAVES * 64) void k5_words(EncModel m, const uint8_t *__restrict__ text, const unsigned long long *__restrict__ ustart,
                                                            const unsigned long long *__restrict__ uend, unsigned long long n_words,
                                                            int32_t *__restrict__ scratch_ids, uint32_t *__restrict__ counts, uint32_t *__restrict__ work,
                                                            unsigned long long work_stride, unsigned int group, int lane_max, WordPublish pub) {
  __shared__ uint32_t pool[ENCW_WAVES][ENCW_POOL];
  __shared__ uint32_t bloom[ENC_BLOOM_WORDS];
  __shared__ uint32_t cp_ascii[128];  // cpmap[0..128)
  for (int i = (int)threadIdx.x; i < ENC_BLOOM_WORDS; i += ENCW_WAVES * 64) bloom[i] = m.bloom[i];
  if (threadIdx.x < 128) cp_ascii[threadIdx.x] = m.cpmap[threadIdx.x];
  __syncthreads();
  const int wave = uni((int)(threadIdx.x >> 6)), lane = lane_id();  // (uni: the wave's number is the same in its lanes -- what follows from it stays in scalar registers)
  const unsigned long long gw = (unsigned long long)blockIdx.x * ENCW_WAVES + wave;
  const unsigned long long n_waves = (unsigned long long)gridDim.x * ENCW_WAVES;
  uint32_t *mine = pool[wave];
  const LdsArr wt{mine}, wr{mine + ENCW_LANE_TOKENS};
  DropoutArgs nodrop{};
  const unsigned long long n_groups = (n_words + group - 1) / group;
  for (unsigned long long grp = gw; grp < n_groups; grp += n_waves) {
    unsigned long long sidx = grp * group;
    const unsigned long long grp_end = sidx + group < n_words ? sidx + group : n_words;
    while (sidx < grp_end) {
      const unsigned long long avail = grp_end - sidx < 64ull ? grp_end - sidx : 64ull;
      unsigned long long lo = 0, len64 = 0;
      if ((unsigned long long)lane < avail) {
        lo = ustart[sidx + lane];
        len64 = uend[sidx + lane] - lo;
      }
      const uint32_t len = len64 > 0x7ffffffeull ? 0x7ffffffeu : (uint32_t)len64;
      const uint32_t need = (unsigned long long)lane < avail ? len + 1u : 0u;  // a word of B bytes: at most B + 1 tokens
      const unsigned long long LONG = __ballot(need > (uint32_t)lane_max);
      if (LONG & 1ull) {  // the first item is a long word: the whole wave on it
        const unsigned long long b0 = ((unsigned long long)from_lane0((uint32_t)(lo >> 32)) << 32) | from_lane0((uint32_t)lo);
        const unsigned long long nbytes = ((unsigned long long)from_lane0((uint32_t)(len64 >> 32)) << 32) | from_lane0((uint32_t)len64);
        int32_t *out = scratch_ids + 2 * b0;
        if (nbytes + 1 <= (unsigned long long)ENCW_WAVE_TOKENS) {
          encode_wave(m, bloom, text + b0, nbytes, LdsArr{mine}, LdsArr{mine + ENCW_WAVE_TOKENS}, LdsArr{mine + 2 * ENCW_WAVE_TOKENS}, 0, 0, 0, out,
                      &counts[sidx], nodrop, sidx);
        } else {
          uint32_t *w = work + gw * 3 * work_stride;
          encode_wave(m, bloom, text + b0, nbytes, GlbArr{w}, GlbArr{w + work_stride}, GlbArr{w + 2 * work_stride}, 0, 0, 0, out, &counts[sidx], nodrop,
                      sidx);
        }
        wave_sync();
        if (lane == 0) word_publish(pub, sidx, 2 * b0, counts[sidx]);
        sidx++;
        continue;
      }
      // the items before the first long one, as many as fit (by their upper bounds, bytes + 1): a lane tokenizes ITS word into its own
      // stretch of the token array -- eight bytes per load while they are ASCII, ids from an LDS copy of the map's ASCII half; byte by byte
      // with the exact decode otherwise -- and closes what it does not use with ENC_DEAD (a word start to every test below: no pair with
      // it).  (Laying the pack's bytes end to end and tokenizing them 64 per step, a binary search of the item per byte, was 4 of the
      // kernel's 9.8 ms.)
      const uint32_t pre = wave_incl_scan(need);
      const unsigned long long before_long = LONG ? (LONG & (0ull - LONG)) - 1ull : ~0ull;
      const unsigned long long FIT = __ballot((unsigned long long)lane < avail && pre <= (uint32_t)ENCW_LANE_TOKENS) & before_long;
      const int cnt = (int)__popcll(FIT);  // (a prefix of the lanes; >= 1: the first item is not long)
      const int n = (int)__shfl((int)pre, cnt - 1);  // places in all
      const int ws = (int)(pre - need);
      int we = ws;
      if (lane < cnt) {
        const uint8_t *sp = text + lo;
        bool prev_space = true, prev_unk = false;
        auto put = [&](uint32_t id) {  // one valid char (bpe.cpp:1497-1530): "▁" in front of a word's first, a run of unknown chars is one token
          const bool space = id == CP_SPACE, unk = id == CP_UNK;
          if (!space && !(unk && prev_unk && !prev_space)) {
            if (prev_space) wt.set(we++, m.space_id | TOK_WS);
            wt.set(we++, unk ? ENC_UNKP : id);
          }
          prev_space = space;
          prev_unk = unk;
        };
        uint32_t i = 0;
        while (i < len) {
          // eight bytes from sp + i (an aligned pair of loads when both lie inside the text, else byte by byte)
          unsigned long long w = 0;
          const unsigned long long at = lo + i;
          uint32_t got = len - i < 8u ? len - i : 8u;
          if ((at & ~7ull) + 16 <= pub.text_bytes) {
            const unsigned long long *q = reinterpret_cast<const unsigned long long *>(text + (at & ~7ull));
            const unsigned long long lo8 = q[0], hi8 = q[1];
            const unsigned int sh = (unsigned int)(at & 7ull) * 8u;
            w = sh ? (lo8 >> sh) | (hi8 << (64u - sh)) : lo8;
          } else {
            for (uint32_t k = 0; k < got; k++) w |= (unsigned long long)sp[i + k] << (8 * k);
          }
          if (got < 8u) w &= (1ull << (8 * got)) - 1ull;
          if ((w & 0x8080808080808080ull) == 0ull) {
            for (uint32_t k = 0; k < got; k++) put(cp_ascii[(uint32_t)(w >> (8 * k)) & 0x7fu]);
            i += got;
            continue;
          }
          // a byte beyond ASCII: the ASCII bytes in front of it, then one char the exact way
          uint32_t k = 0;
          for (; k < got && !((w >> (8 * k)) & 0x80ull); k++) put(cp_ascii[(uint32_t)(w >> (8 * k)) & 0x7fu]);
          i += k;
          uint32_t clen;
          const uint32_t cp = u8_decode_at(sp, i, len, &clen);
          if (cp != INVALID_CP) put(m.cpmap[cp]);
          i += clen;
        }
        for (int q = we; q < ws + (int)need; q++) wt.set(q, ENC_DEAD);
      }
      const uint32_t ntok = (uint32_t)(we - ws);
      wave_sync();
      // every pair's priority, lanes = positions
      for (int c = 0; c < ((n + 63) >> 6); c++) {
        const int p = c * 64 + lane;
        if (p < n) {
          uint32_t r = ENC_INF;
          if (p + 1 < n) {
            const uint32_t ta = wt.get(p), tb = wt.get(p + 1);
            const uint32_t a = ta & ENC_IDM, b = tb & ENC_IDM;
            if (!(tb & TOK_WS) && a != ENC_UNKP && b != ENC_UNKP) r = enc_pair_prio(m, bloom, a, b);
          }
          wr.set(p, r);
        }
      }
      wave_sync();
      if (ntok) we = lane_rounds(m, bloom, wt, wr, ws, we);
      // output (bpe.cpp:1591-1630): a lane writes its word's ids; an unmerged space token with id 0 is dropped, a run of unknown chars is unk_id
      if (lane < cnt) {
        int32_t *out = scratch_ids + 2 * lo;
        uint32_t q = 0;
        for (int i = ws; i < we; i++) {
          const uint32_t tk = wt.get(i);
          if (tk != TOK_WS) {
            const uint32_t id = tk & ENC_IDM;
            out[q++] = (int32_t)(id == ENC_UNKP ? (uint32_t)m.unk_id : id);
          }
        }
        word_publish(pub, sidx + lane, 2 * lo, q);
      }
      wave_sync();
      sidx += (unsigned long long)cnt;
    }
  }
}

// copy ids from the over-allocated scratch to the packed output (one wave per sentence)
__global__ __launch_bounds__(BLOCK) void k5_gather(const int32_t *__restrict__ scratch_ids, SentView sv,
                                                   const unsigned long long *__restrict__ out_off, unsigned long long n_sent,
                                                   int32_t *__restrict__ ids_out) {
  const unsigned long long gw = (unsigned long long)blockIdx.x * NWAVES + (unsigned long long)uni((int)(threadIdx.x >> 6));
  const unsigned long long n_waves = (unsigned long long)gridDim.x * NWAVES;
  for (unsigned long long sidx = gw; sidx < n_sent; sidx += n_waves) {
    const int32_t *src = scratch_ids + sv.spos(sidx);
    const unsigned long long o0 = out_off[sidx], o1 = out_off[sidx + 1];
    for (unsigned long long k = lane_id(); k < o1 - o0; k += 64) ids_out[o0 + k] = src[k];
  }
}

void launch_encode(const EncModel &m, const uint8_t *text, const unsigned long long *offsets, const unsigned long long *ends,
                   unsigned long long n_sent, int bos,
                   int eos, int reverse, int32_t *scratch_ids, uint32_t *counts, uint32_t *work, unsigned long long work_stride,
                   unsigned int n_blocks, double dropout_prob, unsigned long long seed, uint32_t *drop_scratch,
                   unsigned long long drop_stride, hipStream_t st, const WordPublish *pub) {
  if (!n_sent) return;
  DropoutArgs d{};
  d.enabled = dropout_prob > 0;
  d.always_skip = dropout_prob >= 1.0;
  d.thr = d.always_skip ? ~0ull : (unsigned long long)(dropout_prob * 18446744073709551616.0);
  d.seed = seed;
  const std::shared_ptr<const Config> C = cfg();  // (the snapshot the encoder's creation took)
  d.heap_from = (int)C->dropout_heap_from.i;  // (tests: 0 = every word)
  d.wsl = drop_scratch;
  d.ev = nullptr;
  // sentences per wavefront group: large enough that packs are full, small enough that every wavefront of the launch has work
  unsigned long long group = n_sent / ((unsigned long long)n_blocks * ENC_WAVES * 4);
  if (group < 1) group = 1;
  if (group > 24) group = 24;
  if (C->k5_group.u) group = C->k5_group.u;  // (tests: packs of several sentences in a small batch)
  d.lds_queues = m.n_rules < (1u << 23) && !C->dropout_hbm_queues.set;
  d.pack_links = !C->dropout_no_pack.set;
  d.pack_sent = (int)C->dropout_pack_sent.u;
  // one word per lane (merge_lanes) for packs whose words have at most this many tokens; 0 = the wave-wide rounds only.
  // YTTM_K5_LANE_WORDS: the word cache's distinct words, YTTM_K5_LANE_SENT: packed sentences
  const int lane_max = ends ? (int)C->k5_lane_words.i : (int)C->k5_lane_sent.i;
  if (ends && !d.enabled && pub) {  // the word cache's distinct words
    unsigned int wblocks = (n_blocks + 1) / 2;  // (16 waves each: never more waves than n_blocks + 1 of k5_encode's -- the HBM scratch is sized for those)
    if (wblocks > 256) wblocks = 256;
    const unsigned long long waves = (unsigned long long)wblocks * ENCW_WAVES;
    unsigned long long wgroup = n_sent / (waves * 8);  // items a wave takes at a time: packs are cut at a group's end, so many packs per group,
    wgroup = wgroup < 64 ? 64 : wgroup > 512 ? 512 : wgroup;  // and several groups per wave
    hipLaunchKernelGGL(k5_words, dim3(wblocks), dim3(ENCW_WAVES * 64), 0, st, m, text, offsets, ends, n_sent, scratch_ids, counts, work, work_stride,
                       (unsigned int)wgroup, lane_max < 0 ? 0 : lane_max > ENCW_LANE_TOKENS ? ENCW_LANE_TOKENS : lane_max /* 0: every word by the whole wave (tests) */,
                       *pub);
    return;
  }
  if (d.enabled)
    hipLaunchKernelGGL(k5_encode<true>, dim3(n_blocks), dim3(ENC_THREADS), 0, st, m, text, SentView{offsets, ends}, n_sent, bos, eos, reverse, scratch_ids, counts,
                       work, work_stride, d, drop_stride, (unsigned int)group, lane_max);
  else
    hipLaunchKernelGGL(k5_encode<false>, dim3(n_blocks), dim3(ENC_THREADS), 0, st, m, text, SentView{offsets, ends}, n_sent, bos, eos, reverse, scratch_ids, counts,
                       work, work_stride, d, drop_stride, (unsigned int)group, lane_max);
}
void launch_encode_gather(const int32_t *scratch_ids, const unsigned long long *offsets, const unsigned long long *ends,
                          const unsigned long long *out_off, unsigned long long n_sent, int32_t *ids_out, hipStream_t st) {
  if (!n_sent) return;
  unsigned long long b = (n_sent + NWAVES - 1) / NWAVES;
  if (b > 256 * 16) b = 256 * 16;
  hipLaunchKernelGGL(k5_gather, dim3((unsigned int)b), dim3(BLOCK), 0, st, scratch_ids, SentView{offsets, ends}, out_off, n_sent, ids_out);
}

}  // namespace yttm
