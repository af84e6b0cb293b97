// gpu_frontend.cpp -- K1 (char histogram), K2 (word table -> unique-word token tiles), tile repack: the host side of k_frontend.hip.
// (Round 5: cut out of gpu_ctx.cpp, code motion only; gpu_ctx_internal.h says what went where.)
#include "gpu_ctx_internal.h"

namespace yttm {

// ------------------------------------------------------------------------------------------------- K1
void GpuCtx::char_hist(std::vector<uint32_t> &cps, std::vector<unsigned long long> &cnts, unsigned long long &n_codepoints) {
  HIP_CHECK(hipSetDevice(device_));
  tl_stream = strm();
  tl_device = device_;
  const bool have_k1 = spec_.hist_done;  // (upload_overlapped ran K1 on the parts of the text as they arrived; multi-GPU: of this rank's shard -- the sum over the ranks follows below)
  spec_.hist_done = false;
  if (!have_k1) {
  if (!d_hist_) d_hist_ = dmalloc<unsigned long long>(N_CODEPOINTS);
  HIP_CHECK(hipMemsetAsync(d_hist_, 0, (size_t)N_CODEPOINTS * 8, strm()));
  HIP_CHECK(hipMemsetAsync(d_counters_, 0, 64 * 8, strm()));
  // a look at four 4 KB samples of the text: lead bytes of three- and four-byte chars (>= 0xE0) above 1 % pick the kernel variant that
  // counts such chars in LDS (only speed depends on the verdict)
  bool wide_chars = false;
  if (n_text_ >= (1u << 16)) {
    static thread_local uint8_t smp[4][4096];
    for (int i = 0; i < 4; i++) HIP_CHECK(hipMemcpyAsync(smp[i], d_text_ + (n_text_ / 4) * (unsigned long long)i, 4096, hipMemcpyDeviceToHost, strm()));
    sync();
    unsigned int wide = 0;
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 4096; j++) wide += smp[i][j] >= 0xE0u;
    wide_chars = wide * 100u > 4u * 4096u;
  } else {
    wide_chars = true;  // (small inputs: tests of both variants run on them through YTTM_K1_WIDE)
  }
  if (cfg_->k1_wide.set) wide_chars = cfg_->k1_wide.i != 0;
  t_begin(KT_CHAR_HIST);
  DFREE(d_chunk_segs_);
  d_chunk_segs_ = dmalloc<uint32_t>(fe_chunks(n_text_) + 1);
  if (n_text_) launch_char_hist(d_text_, n_text_, d_hist_, d_counters_, wide_chars, d_chunk_segs_, strm());
  t_end(KT_CHAR_HIST, n_text_);
  }
  unsigned long long h_cnt[2] = {0, 0};
  HIP_CHECK(hipMemcpyAsync(h_cnt, d_counters_, 16, hipMemcpyDeviceToHost, strm()));
  sync();
  n_segments = h_cnt[1];  // local segments (before any cross-rank reduction)
  if (multi()) {
    comm_->allreduce_sum_u64(d_hist_, N_CODEPOINTS, strm());
    comm_->allreduce_sum_u64(d_counters_, 1, strm());
    HIP_CHECK(hipMemcpyAsync(h_cnt, d_counters_, 8, hipMemcpyDeviceToHost, strm()));
    sync();
  }
  n_codepoints = h_cnt[0];
  // compact the non-zero bins
  uint32_t *d_cps = dmalloc<uint32_t>(N_CODEPOINTS);
  unsigned long long *d_cnts = dmalloc<unsigned long long>(N_CODEPOINTS);
  unsigned int *d_n = (unsigned int *)(d_counters_ + 8);
  HIP_CHECK(hipMemsetAsync(d_n, 0, 4, strm()));
  launch_hist_compact(d_hist_, d_cps, d_cnts, d_n, N_CODEPOINTS, strm());
  unsigned int k = 0;
  HIP_CHECK(hipMemcpyAsync(&k, d_n, 4, hipMemcpyDeviceToHost, strm()));
  sync();
  cps.resize(k);
  cnts.resize(k);
  if (k) {
    HIP_CHECK(hipMemcpyAsync(cps.data(), d_cps, (size_t)k * 4, hipMemcpyDeviceToHost, strm()));
    HIP_CHECK(hipMemcpyAsync(cnts.data(), d_cnts, (size_t)k * 8, hipMemcpyDeviceToHost, strm()));
    sync();
  }
  seen_cps_ = cps;
  DFREE(d_cps);
  DFREE(d_cnts);
}

// ------------------------------------------------------------------------------------------------- K2
void GpuCtx::build_word_table(const uint32_t *cp, const uint32_t *id, uint32_t n_alpha, uint32_t space_id, uint32_t n_ids_cap) {
  max_id_ = space_id;  // largest token id that can occur in a tile (alphabet now, new ids as they are made)
  for (uint32_t a = 0; a < n_alpha; a++) max_id_ = std::max(max_id_, id[a]);
  id_min_ = space_id;  // (K3 counts the pairs of a small id range in a dense table)
  for (uint32_t a = 0; a < n_alpha; a++) id_min_ = std::min(id_min_, id[a]);
  id_max_ = max_id_;
  HIP_CHECK(hipSetDevice(device_));
  tl_stream = strm();
  tl_device = device_;
  // code point -> class map
  {
    std::vector<uint32_t> cpmap(N_CODEPOINTS, CP_DROP);
    for (uint32_t i = 0; i < n_alpha; i++)
      if (cp[i] < N_CODEPOINTS) cpmap[cp[i]] = id[i];
    const uint32_t spaces[] = {9, 10, 11, 12, 13, 32, 9601};
    for (uint32_t s : spaces) cpmap[s] = CP_SPACE;
    if (!d_cpmap_) d_cpmap_ = dmalloc<uint32_t>(N_CODEPOINTS);
    HIP_CHECK(hipMemcpyAsync(d_cpmap_, cpmap.data(), (size_t)N_CODEPOINTS * 4, hipMemcpyHostToDevice, strm()));
    sync();
  }
  free_words();
  free_class(cls_[0]); free_class(cls_[1]); free_class(cls_[2]);
  cls_[0].nom = TILE_NOM_A; cls_[0].slot = TILE_SLOT_A;
  cls_[1].nom = TILE_NOM_B; cls_[1].slot = TILE_SLOT_B;
  n_alpha_ = n_alpha;
  n_unique = 0; n_tokens0 = 0; n_tiles = 0;
  id_cap_ = n_ids_cap + 64;

  const unsigned long long n_segs = n_segments;
  if (n_segs == 0 || n_text_ == 0) { drop_spec(); return; }
  // The word table upload_overlapped made under the upload is this text's iff words compared by code points are words compared by ids:
  // every char that occurs (and is no space) has an id of its own.  And the table must not have overflowed or filled beyond what the sizing
  // below accepts.
  bool take_spec = spec_.words_done && spec_.n_segs == n_segs;  // (multi-GPU: seen_cps_ is the chars of ALL shards -- a superset of this one's)
  if (take_spec) {
    std::vector<uint32_t> kept(cp, cp + n_alpha);
    std::sort(kept.begin(), kept.end());
    for (uint32_t c : seen_cps_) {
      const bool space = c == 32 || (c >= 9 && c <= 13) || c == 9601;
      if (!space && !std::binary_search(kept.begin(), kept.end(), c)) { take_spec = false; break; }
    }
    if (spec_.h_status[6] || (!spec_.long_segments && (unsigned long long)spec_.h_status[0] * 2 > spec_.ht_cap)) take_spec = false;
  }
  unsigned long long *ht = nullptr;
  unsigned long long ht_cap = 0;
  unsigned int *d_status = (unsigned int *)(d_counters_ + 24);
  unsigned int h_status[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (take_spec) {
    ht = spec_.ht;
    ht_cap = spec_.ht_cap;
    memcpy(h_status, spec_.h_status, sizeof h_status);
    spec_.ht = nullptr;
    front_end_overlapped = true;
  }
  drop_spec();
  if (!take_spec && chunked_) {
    // the text was taken in chunks and is gone; the words it left were compared by code points, which is not this alphabet's partition
    // (coverage dropped chars): the source once more, words compared by the alphabet's ids (d_cpmap_ is in place)
    front_end_chunked(false);
    if (spec_.n_segs != n_segs) throw GpuError{"chunked front end: the second pass over the source found another text"};
    ht = spec_.ht;
    ht_cap = spec_.ht_cap;
    memcpy(h_status, spec_.h_status, sizeof h_status);
    spec_.ht = nullptr;
    drop_spec();
    take_spec = true;
  }
  if (!take_spec) {
  // segment starts
  unsigned long long *d_seg = dmalloc<unsigned long long>(n_segs);
  {
    // where each 4 KB chunk's segments go: exclusive scan of the counts K1 left (no cursor, and the starts come out in text order)
    const unsigned long long nch = fe_chunks(n_text_);
    unsigned long long *d_chunk_off = dmalloc<unsigned long long>(nch + 1);
    unsigned long long *scan_tmp = dmalloc<unsigned long long>(scan_scratch_blocks(nch));
    t_begin(KT_SEGS);
    launch_exclusive_scan(d_chunk_segs_, nch, d_chunk_off, scan_tmp, d_counters_ + 16, strm());
    launch_seg_write(d_text_, n_text_, d_seg, d_chunk_off, strm());
    t_end(KT_SEGS, n_text_ + 8 * n_segs);
    sync();
    DFREE(d_chunk_off);
    DFREE(scan_tmp);
  }
  // hash dedup.  The table is sized for an eighth as many distinct words as there are occurrences (natural text and the
  // benchmark corpora have far fewer: Heaps' law) -- the compaction pass streams it, and a small table keeps the frequent words'
  // slots cache-resident; a corpus of mostly distinct words overflows it (probe chains beyond WH_MAX_PROBES) and is redone
  // with the worst-case size.
  for (int attempt = 0;; attempt++) {
    // (long segments -- CJK-shaped text: clauses of dozens of chars between white space -- are nearly all distinct: the estimate is bound to
    // fail there and the whole dedup would run twice; K1 knows the average segment length)
    const bool long_segments = n_text_ / n_segs >= 16;
    ht_cap = attempt == 0 && !long_segments && !cfg_->word_table_full.set ? pow2_at_least(std::max<unsigned long long>(n_segs / 4, 1ull << 16))
                                                                                 : pow2_at_least(n_segs + n_segs / 2 + 1024);
    ht = dmalloc<unsigned long long>(3 * ht_cap);  // keys, counts, positions of the short words' representatives (k_frontend.hip: WH_SHORT)
    launch_word_table_clear(ht, ht_cap, strm());
    HIP_CHECK(hipMemsetAsync(d_status, 0, 32, strm()));
    t_begin(KT_DEDUP);
    launch_insert_words(d_text_, n_text_, d_cpmap_, d_seg, n_segs, ht, ht_cap - 1, d_status, strm());
    t_end(KT_DEDUP, n_text_ + 8 * n_segs);
    HIP_CHECK(hipMemcpyAsync(h_status, d_status, 32, hipMemcpyDeviceToHost, strm()));
    sync();
    // (more than half full counts as overflow too: the merge loop's tiles do not care, but probe chains do)
    if (!h_status[6] && (attempt || long_segments || (unsigned long long)h_status[0] * 2 <= ht_cap)) break;
    if (attempt) { DFREE(ht); DFREE(d_seg); throw GpuError{"word table overflow"}; }
    DFREE(ht);
    word_table_retries++;
  }
  DFREE(d_seg);
  }
  if (h_status[5] >= (1u << 28)) {
    DFREE(ht);
    throw GpuError{"a word of 2^28 or more characters is not supported"};
  }
  const unsigned int U = h_status[0], UC = h_status[4], UB = h_status[2] - UC, UA = U - UB - UC;
  if (UC) {  // very long words: same layout, slot sized by the longest of them, one workgroup per tile (k_giant.hip)
    cls_[2].nom = h_status[5];
    cls_[2].slot = (2 * h_status[5] + 3u) & ~3u;
  }
  // A tile holds whole words in a fixed slot and only ever shrinks, so the slack a slot needs is one word: pack the
  // slots as full as the longest word allows (HBM pages are then read densely and there are fewer tiles to visit).
  if (h_status[3] > 0 && h_status[3] < (unsigned int)TILE_NOM_A) cls_[0].nom = (unsigned int)TILE_SLOT_A - h_status[3];
  n_unique = U;
  if (U == 0) { DFREE(ht); return; }
  // (room for the extra copies of words seen more than 2^32 - 1 times: k2c_compact_words)
  constexpr unsigned int HX = 4 * HEAVY_CAP;
  unsigned long long *posA = dmalloc<unsigned long long>(UA + HX), *posB = dmalloc<unsigned long long>(UB + HX), *posC = dmalloc<unsigned long long>(UC + HX);
  uint32_t *lenA = dmalloc<uint32_t>(UA + HX), *lenB = dmalloc<uint32_t>(UB + HX), *lenC = dmalloc<uint32_t>(UC + HX);
  cls_[2].d_wcnt = dmalloc<uint32_t>(UC + HX + 256);
  cls_[0].d_wcnt = dmalloc<uint32_t>(UA + HX + 256);  // padding: k_tiles loads SLOT/2 frequencies from a tile's first word unconditionally
  cls_[1].d_wcnt = dmalloc<uint32_t>(UB + HX + 256);
  unsigned int *d_cursor = (unsigned int *)(d_counters_ + 32);
  HIP_CHECK(hipMemsetAsync(d_cursor, 0, 16, strm()));
  unsigned long long *d_heavy = dmalloc<unsigned long long>(3 * HEAVY_CAP);
  const unsigned long long wmax = cfg_->test_wcnt_max.u;  // (tests: heavy words at toy sizes)
  t_begin(KT_BUILD);
  launch_compact_words(d_text_, n_text_, d_cpmap_, ht, ht_cap, posA, cls_[0].d_wcnt, lenA, posB, cls_[1].d_wcnt, lenB, posC, cls_[2].d_wcnt, lenC, d_cursor,
                       d_status, wmax, d_heavy, strm());
  unsigned int h_cursor[4] = {0, 0, 0, 0};
  HIP_CHECK(hipMemcpyAsync(h_status, d_status, 16, hipMemcpyDeviceToHost, strm()));
  HIP_CHECK(hipMemcpyAsync(h_cursor, d_cursor, 16, hipMemcpyDeviceToHost, strm()));
  sync();
  DFREE(ht);
  unsigned int UA2 = UA, UB2 = UB, UC2 = UC;
  if (h_cursor[3] && !(h_status[1] & 2u)) {
    // words seen more than wmax times (the reference counts in uint64, bpe.cpp:382-385): more copies of the word until the weights add up
    // to its count -- every pair count is a sum over words, so the merge loop computes what it would with one word of the whole weight
    std::vector<unsigned long long> hv(3 * (size_t)h_cursor[3]);
    HIP_CHECK(hipMemcpy(hv.data(), d_heavy, hv.size() * 8, hipMemcpyDeviceToHost));
    std::vector<unsigned long long> xp[3];
    std::vector<uint32_t> xl[3], xc[3];
    bool too_many = false;
    for (unsigned int i = 0; i < h_cursor[3]; i++) {
      const uint32_t len = (uint32_t)hv[3 * i + 1];
      const int ci = len > (uint32_t)TILE_NOM_B ? 2 : len > (uint32_t)TILE_NOM_A ? 1 : 0;
      for (unsigned long long left = hv[3 * i + 2]; left;) {
        const unsigned long long c = std::min(left, wmax);
        xp[ci].push_back(hv[3 * i]); xl[ci].push_back(len); xc[ci].push_back((uint32_t)c);
        left -= c;
        if (xp[ci].size() > HX) { too_many = true; break; }
      }
    }
    if (too_many) h_status[1] |= 2u;
    else {
      unsigned long long *pos[3] = {posA, posB, posC};
      uint32_t *len[3] = {lenA, lenB, lenC};
      unsigned int *U2[3] = {&UA2, &UB2, &UC2};
      for (int ci = 0; ci < 3; ci++) {
        if (xp[ci].empty()) continue;
        HIP_CHECK(hipMemcpy(pos[ci] + *U2[ci], xp[ci].data(), xp[ci].size() * 8, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(len[ci] + *U2[ci], xl[ci].data(), xl[ci].size() * 4, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(cls_[ci].d_wcnt + *U2[ci], xc[ci].data(), xc[ci].size() * 4, hipMemcpyHostToDevice));
        *U2[ci] += (unsigned int)xp[ci].size();
      }
      n_unique = (unsigned long long)UA2 + UB2 + UC2;
    }
  }
  DFREE(d_heavy);
  if (h_status[1] & 2u) { DFREE(posA); DFREE(posB); DFREE(lenA); DFREE(lenB); DFREE(posC); DFREE(lenC); throw GpuError{"too many words seen 2^32 times or more"}; }
  build_class(0, posA, lenA, UA2, space_id);
  build_class(1, posB, lenB, UB2, space_id);
  build_class(2, posC, lenC, UC2, space_id);
  if (cls_[2].n_tiles) cls_[2].d_scratch = dmalloc<uint32_t>((size_t)cls_[2].n_tiles * 4 * cls_[2].slot);
  t_end(KT_BUILD, n_text_ / 8 + 4 * (cls_[0].n_tokens0 + cls_[1].n_tokens0 + cls_[2].n_tokens0) + 16ull * U);
  sync();
  DFREE(posA); DFREE(posB); DFREE(lenA); DFREE(lenB); DFREE(posC); DFREE(lenC);
  n_tokens0 = cls_[0].n_tokens0 + cls_[1].n_tokens0 + cls_[2].n_tokens0;
  n_tiles = cls_[0].n_tiles + cls_[1].n_tiles + cls_[2].n_tiles;
}

void GpuCtx::free_class(WordClass &c) {
  DFREE(c.d_tok); DFREE(c.d_tile_len); DFREE(c.d_tile_word0); DFREE(c.d_wcnt); DFREE(c.d_work_n); DFREE(c.d_scratch);
  c.ts = TileSet{};
  c.n_unique = c.n_tokens0 = 0;
  c.n_tiles = 0;
}

// offsets -> tiles -> token slots for one class of unique words
void GpuCtx::build_class(int ci, unsigned long long *uw_pos, uint32_t *uw_len, unsigned int U, uint32_t space_id) {
  WordClass &c = cls_[ci];
  c.n_unique = U;
  if (U == 0) return;
  unsigned long long *uw_off = dmalloc<unsigned long long>(U);
  unsigned long long *scan_tmp = dmalloc<unsigned long long>(scan_scratch_blocks(U));
  launch_exclusive_scan(uw_len, U, uw_off, scan_tmp, d_counters_ + 40, strm());
  unsigned long long total = 0, last_off = 0;
  HIP_CHECK(hipMemcpyAsync(&total, d_counters_ + 40, 8, hipMemcpyDeviceToHost, strm()));
  HIP_CHECK(hipMemcpyAsync(&last_off, uw_off + (U - 1), 8, hipMemcpyDeviceToHost, strm()));
  sync();
  DFREE(scan_tmp);
  c.n_tokens0 = total;
  c.n_tiles = (unsigned int)(last_off / c.nom) + 1;
  unsigned long long *tile_start = dmalloc<unsigned long long>(c.n_tiles);
  c.d_tile_word0 = dmalloc<uint32_t>(c.n_tiles);
  c.d_tile_len = dmalloc<uint32_t>(c.n_tiles);
  c.d_work_n = dmalloc<unsigned int>(16);  // word mode: the round's worklist length [0], "take every word" [WL_PARTS + 1]
  HIP_CHECK(hipMemsetAsync(c.d_work_n, 0, 64, strm()));
  if (ci == 0 && !multi()) {
    // The pair table is allocated and cleared HERE, ahead of the token fill, not right before K3: K3 then does not start on the
    // dirty lines of a 1 GB memset (measured: 0.435 -> 0.395 ms at 1 GB).
    free_table(pt_);
    pt_cap_ = 0;
    ensure_table_capacity(initial_table_keys(total));
    pt_fresh_ = true;
  }
  c.d_tok = dmalloc<uint32_t>((size_t)c.n_tiles * c.slot + 64);
  // slots are read 16 B wide past the live prefix and the staged ids index the flag table: never leave them undefined
  HIP_CHECK(hipMemsetAsync(c.d_tok, 0, ((size_t)c.n_tiles * c.slot + 64) * 4, strm()));
  launch_tiles(uw_off, U, c.nom, tile_start, c.d_tile_word0, strm());
  launch_tile_len(tile_start, c.n_tiles, total, c.d_tile_len, strm());
  launch_fill_tokens(d_text_, n_text_, d_cpmap_, space_id, uw_pos, uw_off, U, c.nom, c.slot, tile_start, c.d_tok, strm(), total);
  sync();
  DFREE(uw_off);
  DFREE(tile_start);
  c.ts.tok = c.d_tok;
  c.ts.tile_len = c.d_tile_len;
  c.ts.tile_word0 = c.d_tile_word0;
  c.ts.wcnt = c.d_wcnt;
  c.ts.n_tiles = c.n_tiles;
}

void GpuCtx::download_word_table(std::vector<uint32_t> &tok, std::vector<unsigned long long> &off, std::vector<uint32_t> &cnt) {
  tok.clear(); off.clear(); cnt.clear();
  off.push_back(0);
  join_class_b();
  for (int ci = 0; ci < 3; ci++) {
    WordClass &c = cls_[ci];
    if (!c.n_tiles) continue;
    std::vector<uint32_t> all((size_t)c.n_tiles * c.slot), tl(c.n_tiles), wc(c.n_unique);
    HIP_CHECK(hipMemcpyAsync(all.data(), c.d_tok, all.size() * 4, hipMemcpyDeviceToHost, strm()));
    HIP_CHECK(hipMemcpyAsync(tl.data(), c.d_tile_len, (size_t)c.n_tiles * 4, hipMemcpyDeviceToHost, strm()));
    HIP_CHECK(hipMemcpyAsync(wc.data(), c.d_wcnt, (size_t)c.n_unique * 4, hipMemcpyDeviceToHost, strm()));
    sync();
    if (ci == 0 && word_mode_) {  // class A in word mode: the words are where wmeta says
      std::vector<unsigned long long> wm(c.n_unique);
      HIP_CHECK(hipMemcpy(wm.data(), d_wmeta_, (size_t)c.n_unique * 8, hipMemcpyDeviceToHost));
      for (unsigned long long w = 0; w < c.n_unique; w++) {
        const unsigned long long o = wm[w] >> 16, len = wm[w] & 0xffffull;
        for (unsigned long long p = 0; p < len; p++) {
          const uint32_t v = all[o + p];
          if (p == 0 && !tok.empty()) off.push_back(tok.size());
          tok.push_back(v & TOK_MASK);
        }
      }
      cnt.insert(cnt.end(), wc.begin(), wc.end());
      continue;
    }
    for (unsigned int t = 0; t < c.n_tiles; t++) {
      for (uint32_t p = 0; p < tl[t]; p++) {
        uint32_t v = all[(size_t)t * c.slot + p];
        if ((v & TOK_WS) && !tok.empty()) off.push_back(tok.size());
        tok.push_back(v & TOK_MASK);
      }
    }
    cnt.insert(cnt.end(), wc.begin(), wc.end());
  }
  off.push_back(tok.size());
  if (tok.empty()) { off.assign(1, 0); }
}

// Re-deal the live words of a tile class into fresh, full tiles when the average fill has dropped below half.
void GpuCtx::maybe_repack(int ci) {
  WordClass &c = cls_[ci];
  if (c.n_tiles < 2) return;
  if (ci == 0 && word_mode_) return;  // (the words live in fixed slots now)
  if (rp_known_[ci]) {
    // The look itself costs three launches, a copy and a stream synchronisation.  The class holds at least what the last look counted minus
    // every merge site since (a site removes one token): the sites the mailbox has reported (a round or two old) plus the summed pair counts
    // of the last rounds' batches, which bound what it may lack.  While that is more than half the nominal fill there is nothing to look at.
    // (Round 5: in word mode the trigger's "tokens streamed last round" is small against the nominal size of ALL tiles, so a corpus with
    // class-B tiles -- CJK-shaped text -- took this look, and its synchronisation, every second round: 75 .. 400 us of host time each.)
    const unsigned long long gone = (sites_cum_ - rp_sites_at_[ci]) + rp_recent_[0] + rp_recent_[1] + rp_recent_[2];
    if (rp_total_[ci] > gone && (rp_total_[ci] - gone) * 2 > (unsigned long long)c.n_tiles * c.nom) return;
  }
  repack_looks++;
  chain_event_ = nullptr;  // work between two timed intervals: they no longer share an event
  if (ci == 1) join_class_b();  // (the look reads the tiles' lengths, the repack their tokens: behind the END of the last class-B launch on the second stream)
  unsigned long long *off = dmalloc<unsigned long long>(c.n_tiles);
  unsigned long long *scan_tmp = dmalloc<unsigned long long>(scan_scratch_blocks(c.n_tiles));
  launch_exclusive_scan(c.d_tile_len, c.n_tiles, off, scan_tmp, d_counters_ + 48, strm());
  unsigned long long total = 0;
  HIP_CHECK(hipMemcpyAsync(&total, d_counters_ + 48, 8, hipMemcpyDeviceToHost, strm()));
  sync();
  DFREE(scan_tmp);
  rp_known_[ci] = true;
  rp_total_[ci] = total;
  rp_sites_at_[ci] = sites_cum_;
  if (total == 0 || total * 2 > (unsigned long long)c.n_tiles * c.nom) { DFREE(off); return; }
  const unsigned int n_new = (unsigned int)((total - 1) / c.nom) + 1;
  uint32_t *new_tok = dmalloc<uint32_t>((size_t)n_new * c.slot + 64);
  uint32_t *new_len = dmalloc<uint32_t>(n_new), *new_word0 = dmalloc<uint32_t>(n_new);
  unsigned long long *gstart = dmalloc<unsigned long long>(n_new);
  HIP_CHECK(hipMemsetAsync(new_tok, 0, ((size_t)n_new * c.slot + 64) * 4, strm()));
  HIP_CHECK(hipMemsetAsync(gstart, 0xff, (size_t)n_new * 8, strm()));
  HIP_CHECK(hipMemsetAsync(new_word0, 0xff, (size_t)n_new * 4, strm()));
  launch_repack(ci, c.ts, off, c.nom, total, gstart, n_new, new_tok, new_len, new_word0, strm());
  sync();
  DFREE(off); DFREE(gstart);
  DFREE(c.d_tok); DFREE(c.d_tile_len); DFREE(c.d_tile_word0);
  c.d_tok = new_tok; c.d_tile_len = new_len; c.d_tile_word0 = new_word0;
  c.n_tiles = n_new;
  c.ts.tok = new_tok; c.ts.tile_len = new_len; c.ts.tile_word0 = new_word0; c.ts.n_tiles = n_new;
  n_tiles = cls_[0].n_tiles + cls_[1].n_tiles + cls_[2].n_tiles;
  repacks++;
  if (ci == 0) {  // tile numbers changed: the pair index is void until it is built again
    idx_valid_ = false;
    idx_pending_ = true;
  }
}

}  // namespace yttm
