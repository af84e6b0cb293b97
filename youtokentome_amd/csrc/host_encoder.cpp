// host_encoder.cpp -- BaseEncoder (bpe.h:22-82): model load, HBM-resident rule tables, batch encode through K5,
// and the tiny host-side helpers of the drop-in surface (id<->subword, decode, vocabulary).
#include <cstdlib>
#include <cstdio>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <thread>
#include <sys/mman.h>
#include <chrono>
#include <mutex>
#include <random>

#include "gpu_ctx.h"
#include "host_core.h"

namespace yttm {

static const std::string UNK_TOKEN = "<UNK>", PAD_TOKEN = "<PAD>", BOS_TOKEN = "<BOS>", EOS_TOKEN = "<EOS>";  // bpe.h:12-15

template <class T>
static T *dalloc(size_t n) {
  void *p = nullptr;
  HIP_CHECK(hipMalloc(&p, (n ? n : 1) * sizeof(T)));
  return (T *)p;
}

// One batch in flight: a stream and the batch buffers, reused and grown on demand.  Two lanes per encoder, so that two host
// threads (Python threads calling encode() on one BPE object -- ctypes releases the GIL, the reference's Cython binding did
// not --, or the two workers of encode_cli) overlap their copies and kernels instead of racing on one set of buffers.
struct EncodeLane {
  std::mutex mu;  // held for the whole of upload -> encode -> fetch
  hipStream_t st = nullptr;
  uint32_t *d_drop = nullptr; size_t cap_drop = 0;
  uint8_t *d_bytes = nullptr; size_t cap_bytes = 0;
  unsigned long long *d_off = nullptr; size_t cap_off = 0;
  int32_t *d_scratch = nullptr; size_t cap_scratch = 0;
  uint32_t *d_counts = nullptr; size_t cap_counts = 0;
  unsigned long long *d_out_off = nullptr; size_t cap_out_off = 0;
  unsigned long long *d_scan_tmp = nullptr; size_t cap_scan_tmp = 0;
  unsigned long long *d_total = nullptr;
  int32_t *d_ids = nullptr; size_t cap_ids = 0;
  uint32_t *d_work = nullptr; size_t cap_work = 0;
  // word cache (k_wcache.hip): the batch's table of distinct words, the slot of every occurrence, the list K5 encodes, its ids
  unsigned long long *d_wc_slot = nullptr; size_t cap_wc_slot = 0;
  unsigned long long *d_wc_pos = nullptr; size_t cap_wc_pos = 0;
  uint32_t *d_wc_occ = nullptr; size_t cap_wc_occ = 0;
  unsigned long long *d_wc_extra = nullptr; size_t cap_wc_extra = 0;
  unsigned int *d_wc_misc = nullptr;  // [0] number of uncached words, [1] status
  uint32_t *d_wc_blk = nullptr; size_t cap_wc_blk = 0;
  unsigned long long *d_wc_blk_off = nullptr; size_t cap_wc_blk_off = 0;
  unsigned long long *d_ustart = nullptr; size_t cap_ustart = 0;
  unsigned long long *d_uend = nullptr; size_t cap_uend = 0;
  uint32_t *d_uslot = nullptr; size_t cap_uslot = 0;
  uint32_t *d_ucounts = nullptr; size_t cap_ucounts = 0;
  unsigned long long last_n_ids = 0, last_n_sent = 0;
  unsigned long long last_distinct_words = 0;  // of the last cached batch (0: the batch went straight through K5)

  template <class T>
  void grow(T *&p, size_t &cap, size_t need) {
    if (need <= cap) return;
    if (p) (void)hipFree(p);
    p = nullptr;
    size_t c = need + need / 4 + 64;
    p = dalloc<T>(c);
    cap = c;
  }
  ~EncodeLane() {
    for (void *p : {(void *)d_drop, (void *)d_bytes, (void *)d_off, (void *)d_scratch, (void *)d_counts, (void *)d_out_off, (void *)d_scan_tmp,
                    (void *)d_total, (void *)d_ids, (void *)d_work, (void *)d_wc_slot, (void *)d_wc_pos, (void *)d_wc_occ, (void *)d_wc_extra,
                    (void *)d_wc_misc, (void *)d_wc_blk, (void *)d_wc_blk_off, (void *)d_ustart, (void *)d_uend, (void *)d_uslot, (void *)d_ucounts})
      if (p) (void)hipFree(p);
    if (st) (void)hipStreamDestroy(st);
  }
};

struct EncoderDevice {
  uint32_t *d_cpmap = nullptr;
  RuleSlot *d_rules = nullptr;
  uint32_t *d_rule_z = nullptr;
  unsigned long long *d_rule_xy = nullptr;
  uint32_t *d_bloom = nullptr;
  EncModel m{};
  std::shared_ptr<const Config> cfg;  // the hooks as they stood at creation (BaseEncoder::config)
  // BPE-dropout draws: counter-based, seeded per call from a per-encoder random salt (the reference draws from a
  // std::random_device-independent global mt19937, bpe.cpp:1415; YTTM_DROPOUT_SEED pins the salt for reproducible runs)
  std::atomic<unsigned long long> dropout_calls{0};
  unsigned long long seed_salt = 0;
  // word cache: 0 = never, 1 = whenever it applies (no dropout), 2 = for batches of at least cache_min_bytes (YTTM_ENCODE_CACHE = 0 | 1;
  // YTTM_ENCODE_CACHE_MIN_MB moves the threshold)
  int cache_mode = 2;
  unsigned long long cache_min_bytes = 8ull << 20;  // (tools/dbg/cache_crossover.py: text 0.9x at 4 MB, 1.1x at 8, 2.3x at 32, 3.2x at 128; random words break even at ~10 MB)
  static constexpr int N_LANES = 2;
  EncodeLane lane[N_LANES];
  std::atomic<unsigned int> next_lane{0};
  std::atomic<unsigned long long> last_distinct_words{0};  // of the most recent batch (cache_words())
  // a free lane, locked (falls back to waiting for the caller's turn-based choice)
  // (Lane 0 last: the device-resident pair encode_device / fetch_device_result keeps its result there, unlocked, between the two
  // calls -- a host-to-host encode from another thread in between takes another lane while one is free.)
  EncodeLane &acquire(std::unique_lock<std::mutex> &lk) {
    for (int k = N_LANES - 1; k >= 0; k--) {
      lk = std::unique_lock<std::mutex>(lane[k].mu, std::try_to_lock);
      if (lk.owns_lock()) return lane[k];
    }
    EncodeLane &l = lane[N_LANES - 1 - next_lane.fetch_add(1) % N_LANES];
    lk = std::unique_lock<std::mutex>(l.mu);
    return l;
  }
  ~EncoderDevice() {
    for (void *p : {(void *)d_cpmap, (void *)d_rules, (void *)d_rule_z, (void *)d_rule_xy, (void *)d_bloom})
      if (p) (void)hipFree(p);
  }
};

BaseEncoder::BaseEncoder(const std::string &model_path, int _n_threads, Status *ret_status, int device) : n_threads(_n_threads), device_(device) {
  Status status = bpe_state.load(model_path);  // bpe.cpp:1643-1656
  if (!status.ok()) { *ret_status = status; return; }
  fill_from_state();
  try {
    HIP_CHECK(hipSetDevice(device_));
    dev_ = new EncoderDevice();
    for (EncodeLane &l : dev_->lane) {
      HIP_CHECK(hipStreamCreateWithFlags(&l.st, hipStreamNonBlocking));
      l.d_total = dalloc<unsigned long long>(2);
      l.d_wc_misc = dalloc<unsigned int>(2);
    }
    cfg_refresh();  // the environment hooks are read here, once per encoder (yttm_config.h) ...
    const std::shared_ptr<const Config> C = cfg();
    dev_->cfg = C;  // ... and kept: the entry points below bind this snapshot, not whatever a later refresh made the process-wide one
    if (C->encode_cache.set) dev_->cache_mode = C->encode_cache.i ? 1 : 0;
    if (C->encode_cache_min_mb.set) dev_->cache_min_bytes = C->encode_cache_min_mb.u << 20;
    if (C->dropout_seed.set) {
      dev_->seed_salt = C->dropout_seed.u;
    } else {
      std::random_device rd;
      dev_->seed_salt = ((unsigned long long)rd() << 32) ^ (unsigned long long)rd();
    }
    // code point -> final id / CP_SPACE / CP_UNK.  is_space wins over char2id (words are split first, bpe.cpp:1509-1510).
    std::vector<uint32_t> cpmap(N_CODEPOINTS, CP_UNK);
    for (auto &c : bpe_state.char2id)
      if (c.first < N_CODEPOINTS) cpmap[c.first] = c.second;
    for (uint32_t s : {9u, 10u, 11u, 12u, 13u, 32u, 9601u}) cpmap[s] = CP_SPACE;
    dev_->d_cpmap = dalloc<uint32_t>(N_CODEPOINTS);
    HIP_CHECK(hipMemcpy(dev_->d_cpmap, cpmap.data(), (size_t)N_CODEPOINTS * 4, hipMemcpyHostToDevice));
    {  // working tokens of the encode kernel keep two flag bits: ids must stay below 2^30 - 16
      uint32_t max_id = 0;
      for (auto &c : char2id) max_id = std::max(max_id, c.second);
      for (auto &r : bpe_state.rules) max_id = std::max(max_id, std::max(r.z, std::max(r.x, r.y)));
      if (max_id >= 0x3ffffff0u) {
        *ret_status = Status(1, "token ids of 2^30 and above are not supported by the MI355X encoder");
        return;
      }
    }
    // rule hash: (x,y) -> rule index; later rules overwrite earlier duplicates like rule2id (bpe.cpp:1672-1674)
    const size_t nr = bpe_state.rules.size();
    unsigned int cap = 64;
    while (cap < 4 * nr + 2) cap <<= 1;  // load factor <= 1/4: nine lookups in ten end at the first slot
    std::vector<RuleSlot> slots(cap);
    for (auto &s : slots) { s.key = PT_EMPTY; s.z = 0; s.pad = 0; }
    std::vector<uint32_t> rz(nr ? nr : 1, 0);
    std::vector<unsigned long long> rxy(nr ? nr : 1, 0);
    for (size_t i = 0; i < nr; i++) {
      const BPE_Rule &r = bpe_state.rules[i];
      rz[i] = r.z;
      rxy[i] = pair_key(r.x, r.y);
      const unsigned long long key = pair_key(r.x, r.y);
      unsigned int h = enc_hash(r.x, r.y) & (cap - 1);
      while (slots[h].key != PT_EMPTY && slots[h].key != key) h = (h + 1) & (cap - 1);
      slots[h].key = key;
      slots[h].z = r.z;
      slots[h].pad = (uint32_t)i;
    }
    dev_->d_rules = dalloc<RuleSlot>(cap);
    HIP_CHECK(hipMemcpy(dev_->d_rules, slots.data(), (size_t)cap * sizeof(RuleSlot), hipMemcpyHostToDevice));
    dev_->d_rule_z = dalloc<uint32_t>(rz.size());
    HIP_CHECK(hipMemcpy(dev_->d_rule_z, rz.data(), rz.size() * 4, hipMemcpyHostToDevice));
    dev_->d_rule_xy = dalloc<unsigned long long>(rxy.size());
    HIP_CHECK(hipMemcpy(dev_->d_rule_xy, rxy.data(), rxy.size() * 8, hipMemcpyHostToDevice));
    std::vector<uint32_t> bloom(ENC_BLOOM_WORDS, 0);
    for (size_t i = 0; i < nr; i++) {
      const uint32_t h = enc_hash(bpe_state.rules[i].x, bpe_state.rules[i].y);
      bloom[enc_bloom_word(h)] |= enc_bloom_bits(h);
    }
    dev_->d_bloom = dalloc<uint32_t>(ENC_BLOOM_WORDS);
    HIP_CHECK(hipMemcpy(dev_->d_bloom, bloom.data(), (size_t)ENC_BLOOM_WORDS * 4, hipMemcpyHostToDevice));
    EncModel &m = dev_->m;
    m.bloom = dev_->d_bloom;
    m.cpmap = dev_->d_cpmap;
    m.rules = dev_->d_rules;
    m.rule_z = dev_->d_rule_z;
    m.rule_xy = dev_->d_rule_xy;
    m.rule_mask = cap - 1;
    m.n_rules = (uint32_t)nr;
    {  // merged tokens are numbered in rule order with the special ids skipped (bpe.cpp:814-837): z(r) without a load
      m.z_affine = nr > 0;
      m.z_base = nr ? rz[0] : 0;
      for (uint32_t &b : m.z_bp) b = 0xffffffffu;
      int nbp = 0;
      for (size_t i = 1; i < nr && m.z_affine; i++) {
        const long long step = (long long)rz[i] - (long long)rz[i - 1] - 1;
        if (step < 0 || nbp + step > 4) m.z_affine = 0;
        else for (long long k = 0; k < step; k++) m.z_bp[nbp++] = (uint32_t)i;
      }
    }
    auto it = char2id.find(SPACE_TOKEN);
    m.space_id = it == char2id.end() ? 0u : it->second;
    m.unk_id = bpe_state.special_tokens.unk_id;
    m.bos_id = bpe_state.special_tokens.bos_id;
    m.eos_id = bpe_state.special_tokens.eos_id;
  } catch (const GpuError &e) {
    *ret_status = Status(2, "GPU error: " + e.msg);
    return;
  }
  if (n_threads == -1) n_threads = 1;  // accepted for API compatibility; encode runs on the GPU
  *ret_status = Status();
}

BaseEncoder::~BaseEncoder() {
  if (dev_) {
    (void)hipSetDevice(device_);
    delete dev_;
  }
}

void BaseEncoder::fill_from_state() {  // bpe.cpp:1667-1690
  for (auto &x : bpe_state.char2id) { char2id[x.first] = x.second; id2char[x.second] = x.first; }
  for (auto &x : id2char) recipe[x.first] = {x.first};
  for (auto &r : bpe_state.rules) {
    std::vector<uint32_t> v = recipe[r.x];
    const std::vector<uint32_t> &w = recipe[r.y];
    v.insert(v.end(), w.begin(), w.end());
    recipe[r.z] = std::move(v);
  }
  for (auto &kv : recipe) {
    std::vector<uint32_t> cps;
    for (uint32_t id : kv.second) cps.push_back(id2char.at(id));
    reversed_recipe[encode_utf8(cps)] = kv.first;
  }
  reversed_recipe[BOS_TOKEN] = (uint32_t)bpe_state.special_tokens.bos_id;
  reversed_recipe[EOS_TOKEN] = (uint32_t)bpe_state.special_tokens.eos_id;
}

int BaseEncoder::vocab_size() const {
  return (int)(bpe_state.rules.size() + bpe_state.char2id.size() + bpe_state.special_tokens.n_special_tokens());
}

// One K5 launch over n_items items of the text (sentences back to back, or the word cache's distinct words [offsets[j], ends[j])):
// ids into the lane's scratch, counts into `counts`.
static void k5_pass(EncoderDevice &D, EncodeLane &d, const void *d_bytes, const unsigned long long *d_offsets, const unsigned long long *d_ends,
                    unsigned long long n_items, unsigned long long total_bytes, unsigned long long max_item_bytes, bool bos, bool eos, bool reverse,
                    double dropout_prob, uint32_t *counts, const WordPublish *pub = nullptr) {
  d.grow(d.d_scratch, d.cap_scratch, (size_t)(2 * total_bytes + 2 * n_items + 2));
  unsigned int max_blocks = 256 * 2;  // 2 workgroups per CU (80 KB LDS each)
  const unsigned long long tok_cap = std::max<unsigned long long>(ENC_LDS_TOKENS, 2 * max_item_bytes + 2);  // tokens per item
  unsigned long long stride = 0, drop_stride = 0;
  {
    // per-wave HBM scratch: 3 working arrays for items that do not fit LDS, + (dropout) word starts and event queues
    unsigned long long per_wave = 0;
    if (tok_cap > (unsigned long long)ENC_LDS_TOKENS) per_wave += 3 * tok_cap * 4;
    if (dropout_prob > 0) per_wave += 7 * tok_cap * 4;
    if (per_wave) {
      unsigned long long waves = std::max<unsigned long long>(1, (4ull << 30) / per_wave);
      max_blocks = (unsigned int)std::max<unsigned long long>(1, std::min<unsigned long long>(max_blocks, waves / ENC_WAVES_PER_BLOCK));
    }
  }
  unsigned long long nb = (n_items + ENC_WAVES_PER_BLOCK - 1) / ENC_WAVES_PER_BLOCK;
  const unsigned int n_blocks = (unsigned int)std::min<unsigned long long>(nb, max_blocks);
  if (tok_cap > (unsigned long long)ENC_LDS_TOKENS) {
    stride = tok_cap;
    d.grow(d.d_work, d.cap_work, (size_t)(3 * stride * (unsigned long long)(n_blocks + 1) * ENC_WAVES_PER_BLOCK));  // (+ 1: k5_words rounds its waves up to 16)
  }
  if (dropout_prob > 0) {
    drop_stride = tok_cap;
    d.grow(d.d_drop, d.cap_drop, (size_t)(7 * drop_stride * (unsigned long long)n_blocks * ENC_WAVES_PER_BLOCK));
  }
  const unsigned long long seed = mix64(D.seed_salt + 0x5bd1e995ull * (D.dropout_calls.fetch_add(1) + 1));
  launch_encode(D.m, (const uint8_t *)d_bytes, d_offsets, d_ends, n_items, bos, eos, reverse, d.d_scratch, counts, d.d_work, stride, n_blocks,
                dropout_prob, seed, d.d_drop, drop_stride, d.st, pub);
}

// counts -> offsets (exclusive scan, the total behind the last one and on the host)
static unsigned long long scan_counts(EncodeLane &d, const uint32_t *counts, unsigned long long n, unsigned long long *off) {
  d.grow(d.d_scan_tmp, d.cap_scan_tmp, (size_t)scan_scratch_blocks(n));
  launch_exclusive_scan(counts, n, off, d.d_scan_tmp, d.d_total, d.st);
  unsigned long long total = 0;
  HIP_CHECK(hipMemcpyAsync(off + n, d.d_total, 8, hipMemcpyDeviceToDevice, d.st));  // (device to device: no host source that must outlive the call)
  HIP_CHECK(hipMemcpyAsync(&total, d.d_total, 8, hipMemcpyDeviceToHost, d.st));
  HIP_CHECK(hipStreamSynchronize(d.st));
  return total;
}

// N4, the word cache (k_wcache.hip): distinct words of the batch -> K5 -> the sentences' ids by lookup.  Leaves ids + offsets in the
// lane's buffers like the direct path.
static void encode_cached(EncoderDevice &D, EncodeLane &d, const void *d_bytes, const unsigned long long *d_offsets, unsigned long long n_sent,
                          unsigned long long total_bytes, unsigned long long max_sentence_bytes, bool bos, bool eos, bool reverse,
                          unsigned long long *n_ids_out) {
  const uint8_t *text = (const uint8_t *)d_bytes;
  unsigned long long cap = 1024;
  // a slot per 16 bytes of text or more: text has far fewer distinct words (1e7 random 'abcd ' sentences: one per 64 bytes), and the passes
  // over the table -- clearing it, listing its words -- cost by its size; a batch that does fill it is inserted again into twice the slots
  while (cap < total_bytes / 16 && cap < (1ull << 31)) cap <<= 1;
  WordCache wc{};
  unsigned long long n_table = 0;
  unsigned int misc[2] = {0, 0};
  unsigned long long short_cap = 1ull << 20;  // slots the words of up to 7 bytes start in (k_wcache.hip wc_insert_word): 8 MB of keys
  if (cfg()->wc_short_slots.set) {  // (measurements, tests) -- rounded up to a power of two: the region's size is used as a mask
    const unsigned long long want = std::max<unsigned long long>(16, cfg()->wc_short_slots.u);
    short_cap = 16;
    while (short_cap < want && short_cap < (1ull << 40)) short_cap <<= 1;
  }
  for (;;) {
    d.grow(d.d_wc_slot, d.cap_wc_slot, (size_t)cap);
    d.grow(d.d_wc_pos, d.cap_wc_pos, (size_t)cap);
    d.grow(d.d_wc_occ, d.cap_wc_occ, (size_t)((total_bytes + n_sent) / 2 + 2));
    const unsigned long long extra_cap = total_bytes / 65536 + 2;
    d.grow(d.d_wc_extra, d.cap_wc_extra, (size_t)(2 * extra_cap));
    wc.slot = d.d_wc_slot;
    wc.pos = d.d_wc_pos;
    wc.mask = cap - 1;
    wc.short_mask = std::min(cap, short_cap) - 1;
    wc.occ = d.d_wc_occ;
    wc.extra = d.d_wc_extra;
    wc.extra_n = d.d_wc_misc;
    wc.extra_cap = (unsigned int)extra_cap;
    wc.status = d.d_wc_misc + 1;
    launch_fill_u64(wc.slot, PT_EMPTY, cap, d.st);
    HIP_CHECK(hipMemsetAsync(wc.occ, 0xff, (size_t)((total_bytes + n_sent) / 2 + 2) * 4, d.st));  // (no word starts anywhere yet)
    HIP_CHECK(hipMemsetAsync(d.d_wc_misc, 0, 8, d.st));
    launch_wcache_insert(D.m, text, total_bytes, d_offsets, n_sent, wc, d.st);
    const unsigned long long n_blk = wcache_count_blocks(wc);
    d.grow(d.d_wc_blk, d.cap_wc_blk, (size_t)n_blk);
    d.grow(d.d_wc_blk_off, d.cap_wc_blk_off, (size_t)n_blk + 1);
    launch_wcache_count_slots(wc, d.d_wc_blk, d.st);
    HIP_CHECK(hipMemcpyAsync(misc, d.d_wc_misc, 8, hipMemcpyDeviceToHost, d.st));
    n_table = scan_counts(d, d.d_wc_blk, n_blk, d.d_wc_blk_off);  // (syncs)
    if (!(misc[1] & 1u)) break;
    // too full for the probe limit: start over -- with the short words spread wider if it WAS one of them that found no slot (the kernel
    // says so: status bit 1), else with twice the slots at once (a batch of mostly distinct LONG words, the usual cause of a full
    // table, paid up to three insert passes over gigabytes of text for a region that was not the problem)
    if ((misc[1] & 2u) && short_cap < cap) {
      short_cap = std::min(cap, short_cap << 3);
      continue;
    }
    if (cap >= (1ull << 31)) throw GpuError{"encode: the word table does not fit"};
    cap <<= 1;
    short_cap = cap;
  }
  const unsigned long long n_extra = misc[0], n_words = n_table + n_extra;
  d.last_distinct_words = n_words;
  D.last_distinct_words.store(n_words);
  d.grow(d.d_ustart, d.cap_ustart, (size_t)n_words + 1);
  d.grow(d.d_uend, d.cap_uend, (size_t)n_words + 1);
  d.grow(d.d_uslot, d.cap_uslot, (size_t)n_table + 1);
  d.grow(d.d_ucounts, d.cap_ucounts, (size_t)n_words + 1);
  launch_wcache_list(wc, d.d_wc_blk_off, n_table, d.d_ustart, d.d_uend, d.d_uslot, d.st);
  if (n_words) {  // (the distinct words' ids stay where K5 puts them, in the lane's scratch: the sentences are assembled from there)
    // (k5_words leaves in every word's table slot where its ids are and how many)
    const WordPublish pub{wc.slot, d.d_uslot, n_table, wc.extra, total_bytes};
    k5_pass(D, d, d_bytes, d.d_ustart, d.d_uend, n_words, total_bytes, max_sentence_bytes, false, false, false, 0.0, d.d_ucounts, &pub);
  }
  launch_wcache_count(d_offsets, n_sent, wc, (bos ? 1 : 0) + (eos ? 1 : 0), d.d_counts, d.st);
  const unsigned long long total = scan_counts(d, d.d_counts, n_sent, d.d_out_off);
  d.grow(d.d_ids, d.cap_ids, (size_t)total + 1);
  launch_wcache_scatter(D.m, d_offsets, n_sent, wc, d.d_scratch, bos, eos, reverse, d.d_out_off, d.d_ids, d.st);
  HIP_CHECK(hipStreamSynchronize(d.st));
  d.last_n_ids = total;
  if (n_ids_out) *n_ids_out = total;
}

// K5 on one lane (locked by the caller): input already in HBM, ids + offsets left in the lane's buffers
static Status encode_on_lane(const BaseEncoder &enc, EncoderDevice &D, EncodeLane &d, int device, const void *d_bytes, const void *d_offsets,
                             unsigned long long n_sent, unsigned long long total_bytes, unsigned long long max_sentence_bytes, bool bos, bool eos,
                             bool reverse, double dropout_prob, unsigned long long *n_ids_out, double *kernel_ms) {
  // bpe.cpp:1702-1707
  if (bos && enc.bpe_state.special_tokens.bos_id == -1) return Status(1, "Can't add <BOS> token. Model was trained without it.");
  if (eos && enc.bpe_state.special_tokens.eos_id == -1) return Status(1, "Can't add <EOS> token. Model was trained without it.");
  try {
    HIP_CHECK(hipSetDevice(device));
    d.last_n_sent = n_sent;
    d.last_n_ids = 0;
    if (n_ids_out) *n_ids_out = 0;
    if (n_sent == 0) return Status();
    d.grow(d.d_counts, d.cap_counts, (size_t)n_sent);
    d.grow(d.d_out_off, d.cap_out_off, (size_t)n_sent + 1);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (kernel_ms) {
      HIP_CHECK(hipEventCreate(&e0));
      HIP_CHECK(hipEventCreate(&e1));
      HIP_CHECK(hipEventRecord(e0, d.st));
    }
    // the word cache: no dropout (its draws are per occurrence), a batch worth the extra passes, a buffer the 8-byte loads can walk
    const bool cached = dropout_prob <= 0 && D.cache_mode != 0 && (D.cache_mode == 1 || total_bytes >= D.cache_min_bytes) &&
                        ((uintptr_t)d_bytes & 7u) == 0 && total_bytes < (1ull << 40) && total_bytes > 0;
    d.last_distinct_words = 0;
    D.last_distinct_words.store(0);
    if (cached) {
      encode_cached(D, d, d_bytes, (const unsigned long long *)d_offsets, n_sent, total_bytes, max_sentence_bytes, bos, eos, reverse, n_ids_out);
      if (kernel_ms) {
        HIP_CHECK(hipEventRecord(e1, d.st));
        HIP_CHECK(hipEventSynchronize(e1));
        float ms = 0;
        HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        *kernel_ms = ms;
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
      }
      return Status();
    }
    k5_pass(D, d, d_bytes, (const unsigned long long *)d_offsets, nullptr, n_sent, total_bytes, max_sentence_bytes, bos, eos, reverse, dropout_prob,
            d.d_counts);
    if (kernel_ms) HIP_CHECK(hipEventRecord(e1, d.st));
    const unsigned long long total = scan_counts(d, d.d_counts, n_sent, d.d_out_off);
    d.grow(d.d_ids, d.cap_ids, (size_t)total);
    launch_encode_gather(d.d_scratch, (const unsigned long long *)d_offsets, nullptr, d.d_out_off, n_sent, d.d_ids, d.st);
    HIP_CHECK(hipStreamSynchronize(d.st));
    if (kernel_ms) {
      float ms = 0;
      HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
      *kernel_ms = ms;
      (void)hipEventDestroy(e0);
      (void)hipEventDestroy(e1);
    }
    d.last_n_ids = total;
    if (n_ids_out) *n_ids_out = total;
  } catch (const GpuError &e) {
    return Status(2, "GPU error: " + e.msg);
  }
  return Status();
}

// Host arrays of a large batch cross the link through the trainer's pinned chunks (gpu_ctx.cpp staged_transfer; 1e7 sentences are 1.3 GB up
// and 1.2 GB down: a plain copy from / to pageable memory moves them at a fraction of the link's rate, and the first touch of a freshly
// allocated result array is paid by one thread); small ones as plain copies on the lane's stream.
constexpr size_t ENC_CHUNK = 2u << 20;  // (the encoder's arrays in chunks of 2 MB: 10^7 sentences host -> host 79 -> 75 ms against 8 MB, 64 against 70 in sub-batches)
static size_t staged_from() {
  const std::shared_ptr<const Config> C = cfg();  // (tests: every copy through the chunks)
  return C->enc_staged_from.set ? (size_t)C->enc_staged_from.u : (size_t)(16u << 20);
}
static void copy_up(int device, void *d_dst, const void *src, size_t n, hipStream_t st) {
  if (n >= staged_from() && n) {
    staged_transfer(device, (uint8_t *)d_dst, n, true, [&](void *chunk, unsigned long long off, size_t len) {
      memcpy(chunk, (const uint8_t *)src + off, len);
      return true;
    }, nullptr, ENC_CHUNK);
  } else if (n) {
    HIP_CHECK(hipMemcpyAsync(d_dst, src, n, hipMemcpyHostToDevice, st));
  }
}
static void copy_down(int device, void *dst, const void *d_src, size_t n, hipStream_t st) {
  if (n >= staged_from() && n) {
    staged_transfer(device, (uint8_t *)const_cast<void *>(d_src), n, false, [&](void *chunk, unsigned long long off, size_t len) {
      memcpy((uint8_t *)dst + off, chunk, len);
      return true;
    }, nullptr, ENC_CHUNK);
  } else if (n) {
    HIP_CHECK(hipMemcpyAsync(dst, d_src, n, hipMemcpyDeviceToHost, st));
  }
}

static Status fetch_lane(EncodeLane &d, int device, int32_t *ids, unsigned long long *out_off, unsigned long long n_sent) {
  try {
    HIP_CHECK(hipSetDevice(device));
    if (n_sent == 0) { if (out_off) out_off[0] = 0; return Status(); }
    // (the lane's stream is idle: encode_on_lane synchronised it)
    if (ids && d.last_n_ids) copy_down(device, ids, d.d_ids, (size_t)d.last_n_ids * 4, d.st);
    if (out_off) copy_down(device, out_off, d.d_out_off, (size_t)(n_sent + 1) * 8, d.st);
    HIP_CHECK(hipStreamSynchronize(d.st));
  } catch (const GpuError &e) {
    return Status(2, "GPU error: " + e.msg);
  }
  return Status();
}

// The device-resident pair (what bench.py times): encode_device leaves its result in lane 0, fetch_device_result copies it
// out.  Each call locks the lane; the PAIR is not atomic -- one host thread at a time may use these two on an encoder.
Status BaseEncoder::encode_device(const void *d_bytes, const void *d_offsets, unsigned long long n_sent, unsigned long long total_bytes,
                                  unsigned long long max_sentence_bytes, bool bos, bool eos, bool reverse, double dropout_prob,
                                  unsigned long long *n_ids_out, double *kernel_ms) const {
  if (!dev_) return Status(2, "encoder has no device state");
  const CfgBind bind(dev_->cfg);
  std::lock_guard<std::mutex> lk(dev_->lane[0].mu);
  return encode_on_lane(*this, *dev_, dev_->lane[0], device_, d_bytes, d_offsets, n_sent, total_bytes, max_sentence_bytes, bos, eos, reverse,
                        dropout_prob, n_ids_out, kernel_ms);
}

std::shared_ptr<const Config> BaseEncoder::config() const { return dev_ ? dev_->cfg : nullptr; }
void BaseEncoder::set_cache(int mode, unsigned long long min_bytes) const {
  if (!dev_) return;
  dev_->cache_mode = mode < 0 ? 0 : (mode > 2 ? 2 : mode);
  dev_->cache_min_bytes = min_bytes;
}
unsigned long long BaseEncoder::cache_words() const {
  if (!dev_) return 0;
  return dev_->last_distinct_words.load();  // (of the most recent batch, whichever lane ran it)
}

Status BaseEncoder::fetch_device_result(int32_t *ids, unsigned long long *out_off, unsigned long long n_sent) const {
  if (!dev_) return Status(2, "fetch_device_result: no matching result");
  const CfgBind bind(dev_->cfg);
  std::lock_guard<std::mutex> lk(dev_->lane[0].mu);
  if (n_sent != dev_->lane[0].last_n_sent) return Status(2, "fetch_device_result: no matching result");
  return fetch_lane(dev_->lane[0], device_, ids, out_off, n_sent);
}

// upload -> K5 -> download on one lane; the output arrays come from the caller's allocator once their sizes are known
template <class AllocIds, class AllocOff>
static Status encode_host_to_host(const BaseEncoder &enc, EncoderDevice *dev, int device, const uint8_t *bytes, const unsigned long long *offsets,
                                  unsigned long long n_sent, bool bos, bool eos, bool reverse, double dropout_prob, AllocIds alloc_ids, AllocOff alloc_off) {
  if (bos && enc.bpe_state.special_tokens.bos_id == -1) return Status(1, "Can't add <BOS> token. Model was trained without it.");
  if (eos && enc.bpe_state.special_tokens.eos_id == -1) return Status(1, "Can't add <EOS> token. Model was trained without it.");
  if (n_sent == 0) {
    unsigned long long *o = alloc_off(1);
    if (o) o[0] = 0;
    (void)alloc_ids(0);
    return Status();
  }
  if (!dev) return Status(2, "encoder has no device state");
  const bool trace = cfg()->trace.set;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms_since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(now() - t).count(); };
  unsigned long long total_bytes = offsets[n_sent] - offsets[0], max_len = 0;
  // (the longest sentence sizes K5's scratch; a pass over 80 MB of offsets for 10^7 sentences -- on its own thread while the bytes go up)
  std::thread scan;
  auto longest = [&] { for (unsigned long long i = 0; i < n_sent; i++) max_len = std::max(max_len, offsets[i + 1] - offsets[i]); };
  if (n_sent >= (1u << 20)) scan = std::thread(longest);
  else longest();
  struct Joiner {
    std::thread &t;
    ~Joiner() { if (t.joinable()) t.join(); }
  } joiner{scan};
  std::unique_lock<std::mutex> lk;
  EncodeLane &d = dev->acquire(lk);  // held until the ids are back on the host
  const auto t1 = now();
  try {
    HIP_CHECK(hipSetDevice(device));
    d.grow(d.d_bytes, d.cap_bytes, (size_t)total_bytes + 16);
    d.grow(d.d_off, d.cap_off, (size_t)n_sent + 1);
    copy_up(device, d.d_bytes, bytes + offsets[0], (size_t)total_bytes, d.st);
    if (offsets[0] == 0) {
      copy_up(device, d.d_off, offsets, ((size_t)n_sent + 1) * 8, d.st);
      HIP_CHECK(hipStreamSynchronize(d.st));
    } else {  // offsets are rebased to the first byte of the batch
      std::vector<unsigned long long> rel((size_t)n_sent + 1);
      for (unsigned long long i = 0; i <= n_sent; i++) rel[i] = offsets[i] - offsets[0];
      copy_up(device, d.d_off, rel.data(), rel.size() * 8, d.st);
      HIP_CHECK(hipStreamSynchronize(d.st));
    }
  } catch (const GpuError &e) {
    return Status(2, "GPU error: " + e.msg);
  }
  if (scan.joinable()) scan.join();
  const double ms_up = ms_since(t1);
  const auto t2 = now();
  unsigned long long n_ids = 0;
  Status s = encode_on_lane(enc, *dev, d, device, d.d_bytes, d.d_off, n_sent, total_bytes, max_len, bos, eos, reverse, dropout_prob, &n_ids, nullptr);
  if (!s.ok()) return s;
  const double ms_enc = ms_since(t2);
  const auto t3 = now();
  int32_t *ids = alloc_ids((size_t)n_ids);
  unsigned long long *off = alloc_off((size_t)n_sent + 1);
  if ((n_ids && !ids) || !off) return Status(2, "out of memory");
  const double ms_alloc = ms_since(t3);
  const auto t4 = now();
  s = fetch_lane(d, device, ids, off, n_sent);
  if (trace)
    fprintf(stderr, "[yttm] encode host -> host: %llu sentences, %.1f MB up, %.1f MB down: copy up %.1f ms, encode %.1f ms, result arrays %.1f ms, copy down %.1f ms\n",
            n_sent, (double)(total_bytes + 8 * n_sent) / 1e6, (double)(4 * n_ids + 8 * n_sent) / 1e6, ms_up, ms_enc, ms_alloc, ms_since(t4));
  return s;
}

Status BaseEncoder::encode_as_ids(const uint8_t *bytes, const unsigned long long *offsets, unsigned long long n_sent, bool bos, bool eos,
                                  bool reverse, double dropout_prob, std::vector<int32_t> *ids, std::vector<unsigned long long> *out_off) const {
  ids->clear();
  out_off->clear();
  const CfgBind bind(config());
  return encode_host_to_host(
      *this, dev_, device_, bytes, offsets, n_sent, bos, eos, reverse, dropout_prob,
      [&](size_t n) { ids->resize(n); return ids->data(); }, [&](size_t n) { out_off->assign(n, 0); return out_off->data(); });
}

// A result array the caller releases with free().  A large one is asked for in huge pages: the first touch of 1.2 GB of fresh 4 KB pages --
// 3e5 page faults -- took 130 ms of a 10^7-sentence call, four times the link time of the copy that does the touching (MI355X box, transparent
// huge pages on "madvise").
static void *result_alloc(size_t bytes) {
  constexpr size_t HUGE = 2u << 20;
  if (bytes < 4 * HUGE) return malloc(bytes);
  void *p = nullptr;
  if (posix_memalign(&p, HUGE, (bytes + HUGE - 1) / HUGE * HUGE) != 0) return malloc(bytes);
#ifdef MADV_HUGEPAGE
  (void)madvise(p, (bytes + HUGE - 1) / HUGE * HUGE, MADV_HUGEPAGE);
#endif
  return p;
}

// A very large batch in sub-batches through both lanes: while sub-batch i is being encoded, i + 1 crosses the link upwards and the ids of
// i - 1 downwards (three threads: up, this one, down).  One after the other the three legs of 10^7 sentences are 25 + 28 + 25 ms; the
// link works both ways at once (measured: 79 -> 64 ms in sub-batches of 320 MB; smaller ones lose it again to the word cache's fixed costs
// and to words that recur across sub-batches).  The ids' array is asked for at its upper bound -- a sentence of B bytes has at most B + 1 tokens
// (enc_tokenize) -- in pages that are only ever touched up to the real size; returns false (nothing done) when that much address space is
// not to be had or a lane is busy: the caller then takes the plain path.
static bool encode_pipelined(const BaseEncoder &enc, EncoderDevice *dev, int device, const uint8_t *bytes, const unsigned long long *offsets,
                             unsigned long long n_sent, bool bos, bool eos, bool reverse, double dropout_prob, int32_t **ids_out,
                             unsigned long long **off_out, Status *result) {
  const unsigned long long total_bytes = offsets[n_sent] - offsets[0];
  const std::shared_ptr<const Config> C = cfg();
  unsigned long long sub_bytes = (unsigned long long)std::max<long long>(1, C->enc_sub_mb.i) << 20;
  if (C->enc_sub_kb.set) sub_bytes = (unsigned long long)std::max<long long>(1, C->enc_sub_kb.i) << 10;  // (tests)
  const unsigned long long min_bytes = C->enc_pipe_from.u;
  if (total_bytes < min_bytes || n_sent < 4) return false;
  // sub-batches of about sub_bytes each, cut at sentence starts
  std::vector<unsigned long long> cut{0};
  while (cut.back() < n_sent) {
    const unsigned long long want = offsets[cut.back()] + sub_bytes;
    unsigned long long s = (unsigned long long)(std::upper_bound(offsets + cut.back() + 1, offsets + n_sent + 1, want) - offsets) - 1;
    if (s <= cut.back()) s = cut.back() + 1;
    cut.push_back(std::min(s, n_sent));
  }
  const size_t K = cut.size() - 1;
  if (K < 2) return false;
  std::unique_lock<std::mutex> lk0(dev->lane[0].mu, std::try_to_lock), lk1(dev->lane[1].mu, std::try_to_lock);
  if (!lk0.owns_lock() || !lk1.owns_lock()) return false;
  const unsigned long long ids_cap = total_bytes + n_sent * (1ull + (bos ? 1 : 0) + (eos ? 1 : 0)) + 1;
  int32_t *ids = (int32_t *)result_alloc(ids_cap * sizeof(int32_t));
  unsigned long long *off = (unsigned long long *)result_alloc((n_sent + 1) * sizeof(unsigned long long));
  if (!ids || !off) {
    free(ids);
    free(off);
    return false;
  }
  std::mutex mu;
  std::condition_variable cv;
  std::vector<int> up_done(K, 0), enc_done(K, 0), down_done(K, 0);
  std::vector<unsigned long long> n_ids(K, 0), max_len(K, 0);
  bool failed = false;
  std::string error;
  auto fail = [&](const std::string &msg) {
    std::lock_guard<std::mutex> g(mu);
    if (!failed) error = msg;
    failed = true;
    cv.notify_all();
  };
  auto wait_flag = [&](std::vector<int> &flags, size_t i) {  // false: somebody failed
    std::unique_lock<std::mutex> g(mu);
    cv.wait(g, [&] { return failed || flags[i]; });
    return !failed;
  };
  auto set_flag = [&](std::vector<int> &flags, size_t i) {
    std::lock_guard<std::mutex> g(mu);
    flags[i] = 1;
    cv.notify_all();
  };
  auto has_failed = [&] {
    std::lock_guard<std::mutex> g(mu);
    return failed;
  };
  std::thread up([&] {
    const CfgBind bind(C);  // (the caller's snapshot: this thread's copies read the hooks too)
    try {
      HIP_CHECK(hipSetDevice(device));
      for (size_t i = 0; i < K; i++) {
        if (has_failed()) return;  // (nobody will encode what this thread would still stage)
        if (i >= 2 && !wait_flag(down_done, i - 2)) return;  // the lane's buffers are free again
        EncodeLane &d = dev->lane[i & 1];
        const unsigned long long s0 = cut[i], ns = cut[i + 1] - s0, b0 = offsets[s0], nb = offsets[s0 + ns] - b0;
        unsigned long long mx = 0;
        for (unsigned long long j = 0; j < ns; j++) mx = std::max(mx, offsets[s0 + j + 1] - offsets[s0 + j]);
        max_len[i] = mx;
        d.grow(d.d_bytes, d.cap_bytes, (size_t)nb + 16);
        d.grow(d.d_off, d.cap_off, (size_t)ns + 1);
        if (nb)
          staged_transfer(device, d.d_bytes, nb, true, [&](void *chunk, unsigned long long o, size_t len) {
            memcpy(chunk, bytes + b0 + o, len);
            return true;
          }, nullptr, ENC_CHUNK);
        staged_transfer(device, (uint8_t *)d.d_off, (ns + 1) * 8, true, [&](void *chunk, unsigned long long o, size_t len) {
          unsigned long long *dst = (unsigned long long *)chunk;  // (offsets relative to the sub-batch's first byte)
          const unsigned long long *src = offsets + s0 + o / 8;
          for (size_t j = 0; j < len / 8; j++) dst[j] = src[j] - b0;
          return true;
        }, nullptr, ENC_CHUNK);
        set_flag(up_done, i);
      }
    } catch (const GpuError &e) {
      fail("GPU error: " + e.msg);
    } catch (const std::exception &e) {  // (bad_alloc from a grow, ...: an error of this call, never std::terminate of the host process)
      fail(std::string("encode (upload thread): ") + e.what());
    } catch (...) {
      fail("encode (upload thread): unknown exception");
    }
  });
  std::thread down([&] {
    const CfgBind bind(C);
    try {
      HIP_CHECK(hipSetDevice(device));
      unsigned long long ids_base = 0;
      for (size_t i = 0; i < K; i++) {
        if (!wait_flag(enc_done, i)) return;
        EncodeLane &d = dev->lane[i & 1];
        const unsigned long long s0 = cut[i], ns = cut[i + 1] - s0;
        if (n_ids[i])
          staged_transfer(device, (uint8_t *)d.d_ids, n_ids[i] * 4, false, [&](void *chunk, unsigned long long o, size_t len) {
            memcpy((uint8_t *)(ids + ids_base) + o, chunk, len);
            return true;
          }, nullptr, ENC_CHUNK);
        staged_transfer(device, (uint8_t *)d.d_out_off, (ns + 1) * 8, false, [&](void *chunk, unsigned long long o, size_t len) {
          const unsigned long long *src = (const unsigned long long *)chunk;  // (the sub-batch's offsets start at 0: moved behind the ids so far;
          unsigned long long *dst = off + s0 + o / 8;                         //  its last entry is the next one's first, written twice, the same)
          for (size_t j = 0; j < len / 8; j++) dst[j] = src[j] + ids_base;
          return true;
        }, nullptr, ENC_CHUNK);
        ids_base += n_ids[i];
        set_flag(down_done, i);
      }
    } catch (const GpuError &e) {
      fail("GPU error: " + e.msg);
    } catch (const std::exception &e) {
      fail(std::string("encode (download thread): ") + e.what());
    } catch (...) {
      fail("encode (download thread): unknown exception");
    }
  });
  Status st;
  for (size_t i = 0; i < K; i++) {
    if (!wait_flag(up_done, i)) break;
    EncodeLane &d = dev->lane[i & 1];
    const unsigned long long s0 = cut[i], ns = cut[i + 1] - s0, nb = offsets[s0 + ns] - offsets[s0];
    unsigned long long got = 0;
    st = encode_on_lane(enc, *dev, d, device, d.d_bytes, d.d_off, ns, nb, max_len[i], bos, eos, reverse, dropout_prob, &got, nullptr);
    if (!st.ok()) {
      fail(st.message);
      break;
    }
    n_ids[i] = got;
    set_flag(enc_done, i);
  }
  up.join();
  down.join();
  if (failed) {
    free(ids);
    free(off);
    *result = st.ok() ? Status(2, error) : st;
    return true;
  }
  if (cfg()->trace.set) fprintf(stderr, "[yttm] encode host -> host: %llu sentences in %zu sub-batches through both lanes\n", n_sent, K);
  {
    // The array was asked for at its upper bound (four bytes per input byte and more); what the caller keeps until its free() is the ids
    // themselves: the tail -- never touched, so never backed by memory, but address space under RLIMIT_AS / strict overcommit -- goes back
    // now.  A shrinking realloc of an mmap'ed block is an mremap; the alignment of the block's start stays what it was.
    const unsigned long long used = off[n_sent];
    if (void *small = realloc(ids, (size_t)std::max<unsigned long long>(used, 1) * sizeof(int32_t))) ids = (int32_t *)small;
  }
  *ids_out = ids;
  *off_out = off;
  *result = Status();
  return true;
}

// the same into malloc'ed arrays (released by the caller with free()): what the C ABI hands out, without a copy in between
Status BaseEncoder::encode_as_ids_malloc(const uint8_t *bytes, const unsigned long long *offsets, unsigned long long n_sent, bool bos, bool eos,
                                         bool reverse, double dropout_prob, int32_t **ids, unsigned long long **out_off) const {
  *ids = nullptr;
  *out_off = nullptr;
  const CfgBind bind(config());
  if (dev_ && n_sent && !(bos && bpe_state.special_tokens.bos_id == -1) && !(eos && bpe_state.special_tokens.eos_id == -1)) {
    Status piped;
    if (encode_pipelined(*this, dev_, device_, bytes, offsets, n_sent, bos, eos, reverse, dropout_prob, ids, out_off, &piped)) return piped;
  }
  Status s = encode_host_to_host(
      *this, dev_, device_, bytes, offsets, n_sent, bos, eos, reverse, dropout_prob,
      [&](size_t n) { *ids = (int32_t *)result_alloc((n ? n : 1) * sizeof(int32_t)); return *ids; },
      [&](size_t n) { *out_off = (unsigned long long *)result_alloc((n ? n : 1) * sizeof(unsigned long long)); return *out_off; });
  if (!s.ok()) {
    free(*ids);
    free(*out_off);
    *ids = nullptr;
    *out_off = nullptr;
  }
  return s;
}

Status BaseEncoder::encode_as_subwords(const uint8_t *bytes, const unsigned long long *offsets, unsigned long long n_sent, bool bos, bool eos,
                                       bool reverse, double dropout_prob, std::vector<std::string> *pieces,
                                       std::vector<unsigned long long> *piece_off) const {
  // ids come from the GPU (forward order, no bos/eos); pieces are a host lookup (bpe.cpp:1597-1613).  The k-th unk id
  // of a sentence is the k-th run of unknown chars, whose text is recovered from the sentence itself.
  const CfgBind bind(config());
  if (bos && bpe_state.special_tokens.bos_id == -1) return Status(1, "Can't add <BOS> token. Model was trained without it.");
  if (eos && bpe_state.special_tokens.eos_id == -1) return Status(1, "Can't add <EOS> token. Model was trained without it.");
  std::vector<int32_t> ids;
  std::vector<unsigned long long> off;
  Status s = encode_as_ids(bytes, offsets, n_sent, false, false, false, dropout_prob, &ids, &off);
  if (!s.ok()) return s;
  pieces->clear();
  piece_off->assign(1, 0);
  const int unk = bpe_state.special_tokens.unk_id;
  for (unsigned long long i = 0; i < n_sent; i++) {
    std::vector<std::string> sent;
    if (bos) sent.push_back(BOS_TOKEN);
    // unknown runs of this sentence, in order
    std::vector<std::string> unk_runs;
    {
      std::vector<uint32_t> text = decode_utf8((const char *)bytes + offsets[i], (const char *)bytes + offsets[i + 1]);
      std::vector<uint32_t> run;
      for (uint32_t c : text) {
        const bool known = !is_space(c) && char2id.count(c);
        if (!is_space(c) && !known) { run.push_back(c); continue; }
        if (!run.empty()) { unk_runs.push_back(encode_utf8(run)); run.clear(); }
      }
      if (!run.empty()) unk_runs.push_back(encode_utf8(run));
    }
    size_t next_unk = 0;
    for (unsigned long long k = off[i]; k < off[i + 1]; k++) {
      const int id = ids[k];
      if (id == unk) {
        sent.push_back(next_unk < unk_runs.size() ? unk_runs[next_unk++] : std::string());
      } else {
        std::string piece;
        id_to_subword(id, &piece, false);
        sent.push_back(piece);
      }
    }
    if (eos) sent.push_back(EOS_TOKEN);
    if (reverse) std::reverse(sent.begin(), sent.end());
    for (auto &p : sent) pieces->push_back(std::move(p));
    piece_off->push_back(pieces->size());
  }
  return Status();
}

Status BaseEncoder::id_to_subword(int id, std::string *subword, bool replace_space) const {  // bpe.cpp:1774-1807
  if (id < 0 || vocab_size() <= id)
    return Status(1, "id must be in the range [0, vocab_size - 1]. Current value: vocab_size = " + std::to_string(vocab_size()) +
                         "; id=" + std::to_string(id) + ";");
  const SpecialTokens &sp = bpe_state.special_tokens;
  if (sp.unk_id == id) { *subword = UNK_TOKEN; return Status(); }
  if (sp.pad_id == id) { *subword = PAD_TOKEN; return Status(); }
  if (sp.bos_id == id) { *subword = BOS_TOKEN; return Status(); }
  if (sp.eos_id == id) { *subword = EOS_TOKEN; return Status(); }
  auto it = recipe.find((uint32_t)id);
  if (it == recipe.end()) { subword->clear(); return Status(); }
  std::vector<uint32_t> cps;
  for (uint32_t t : it->second) cps.push_back(id2char.at(t));
  if (replace_space && !cps.empty() && cps[0] == SPACE_TOKEN) {
    *subword = " " + encode_utf8(std::vector<uint32_t>(cps.begin() + 1, cps.end()));
    return Status();
  }
  *subword = encode_utf8(cps);
  return Status();
}

int BaseEncoder::subword_to_id(const std::string &token) const {  // bpe.cpp:1809-1826
  const SpecialTokens &sp = bpe_state.special_tokens;
  if (UNK_TOKEN == token) return sp.unk_id;
  if (PAD_TOKEN == token) return sp.pad_id;
  if (BOS_TOKEN == token) return sp.bos_id;
  if (EOS_TOKEN == token) return sp.eos_id;
  auto it = reversed_recipe.find(token);
  if (it != reversed_recipe.end()) return (int)it->second;
  return sp.unk_id;
}

Status BaseEncoder::decode(const std::vector<int> &ids, std::string *sentence, const std::unordered_set<int> *ignore_ids) const {
  // bpe.cpp:1843-1861
  bool first_iter = true;
  for (int id : ids) {
    std::string subword;
    if (!ignore_ids || ignore_ids->count(id) == 0) {
      Status status = id_to_subword(id, &subword, true);
      if (!status.ok()) return status;
      *sentence += subword;
      if (first_iter && !sentence->empty() && sentence->at(0) == ' ') *sentence = sentence->substr(1);
      first_iter = false;
    }
  }
  return Status();
}

std::vector<std::string> BaseEncoder::vocabulary() const {  // bpe.cpp:1884-1894
  int n = vocab_size();
  std::vector<std::string> vocab((size_t)n);
  for (int i = 0; i < n; i++) id_to_subword(i, &vocab[(size_t)i]);
  return vocab;
}

}  // namespace yttm
