// run_select.h -- carry arithmetic on 64-bit lane masks (scalar unit): which pairs of a run of equal tokens count, which links
// of a chain of new tokens count, where a run that lost its first token ends.  Used by the class-A merge-apply kernel
// (k_apply.hip); plain integer code, checked against brute force by tests/test_run_select.py.
//
// Background (bpe.cpp:461-475): a run of L equal tokens counts floor(L/2) for its self pair -- the pairs at an
// even offset from the start of the run.  With lane l <-> token position 64 c + l, bit q of a mask E = "tokens q and q+1 are
// equal and belong to the same word"; a run of tokens is a run of set bits, and the masks of a tile's chunks form one stream
// (the carries below go from a chunk to the next).
#pragma once
#include <stdint.h>

#ifndef __host__
#define __host__
#define __device__
#endif

namespace yttm {

// Pairs at an even offset from the start of their run.  A run that starts at an even bit position (or goes on from the previous
// mask with its next pair at an even offset) is made to carry out of E + S_e: D marks exactly those runs; they take their even
// positions, all other runs their odd ones.
struct RunCarry {
  bool cont = false, cont_even = false;  // the run at the top of the previous mask goes on / its next pair is at an even offset
};
__host__ __device__ inline unsigned long long even_offset_select(unsigned long long E, RunCarry &rc) {
  const unsigned long long S = E & ~((E << 1) | (rc.cont ? 1ull : 0ull));
  const unsigned long long S_e = (S & 0x5555555555555555ull) | (rc.cont_even ? (E & 1ull) : 0ull);
  const unsigned long long D = (E + S_e) ^ E;
  const unsigned long long sel = (D & E & 0x5555555555555555ull) | (~D & E & 0xaaaaaaaaaaaaaaaaull);
  rc.cont = (E >> 63) != 0ull;
  rc.cont_even = rc.cont && !(sel >> 63);
  return sel;
}

// The same for chains at stride 2 (bit q = "the new tokens made at q and at q + 2 are equal and adjacent"): even and odd bit
// positions are two independent classes; within a class the gaps are filled with ones so that a carry travels along the chain.
// Selects the links at an even offset from the start of their chain: a chain of L equal new tokens has L - 1 links, of which
// ceil((L-1)/2) = floor(L/2) are selected.
struct Chain2Carry {
  bool ce = false, ce_even = false, co = false, co_even = false;
};
__host__ __device__ inline unsigned long long stride2_select(unsigned long long Zall, Chain2Carry &cc) {
  const unsigned long long ME = 0x5555555555555555ull, MO = 0xaaaaaaaaaaaaaaaaull;
  unsigned long long sel = 0;
  {
    const unsigned long long Z = Zall & ME, G = Z | MO;
    const unsigned long long S = Z & ~((Z << 2) | (cc.ce ? 1ull : 0ull));
    const unsigned long long S_e = (S & 0x1111111111111111ull) | (cc.ce_even ? (Z & 1ull) : 0ull);
    const unsigned long long D = (G + S_e) ^ G;
    const unsigned long long s = (D & Z & 0x1111111111111111ull) | (~D & Z & 0x4444444444444444ull);
    cc.ce = ((Z >> 62) & 1ull) != 0ull;
    cc.ce_even = cc.ce && !((s >> 62) & 1ull);
    sel |= s;
  }
  {
    const unsigned long long Z = Zall & MO, G = Z | ME;
    const unsigned long long S = Z & ~((Z << 2) | (cc.co ? 2ull : 0ull));
    const unsigned long long S_e = (S & 0x2222222222222222ull) | (cc.co_even ? (Z & 2ull) : 0ull);
    const unsigned long long D = (G + S_e) ^ G;
    const unsigned long long s = (D & Z & 0x2222222222222222ull) | (~D & Z & 0x8888888888888888ull);
    cc.co = (Z >> 63) != 0ull;
    cc.co_even = cc.co && !(s >> 63);
    sel |= s;
  }
  return sel;
}

// Runs of E whose first bit is marked (or that go on from the previous mask with `cons` set): the mask of the first zero bit
// behind each such run = the position of the run's last token.  prev_top: bit 63 of the previous mask.
__host__ __device__ inline unsigned long long marked_run_ends(unsigned long long E, unsigned long long mark, bool prev_top, bool &cons) {
  const unsigned long long RS = E & ~((E << 1) | (prev_top ? 1ull : 0ull));
  const unsigned long long add = (RS & mark) | (cons ? 1ull : 0ull);
  const unsigned long long C = E + add;
  cons = C < E;
  return C & ~E;
}

}  // namespace yttm
