/* yttm_gpu.h -- inner C ABI: the individual HIP kernel stages behind yttm_mi355x.h.
 *
 * A host (the reference's C++ trainer, or the parity tests) can drive the stages one by one; every stage cites the
 * reference code it replaces.  All functions return 0 on success; the message of the last failure is available from
 * yttm_gpu_last_error().  Opaque handle; the caller owns host buffers, the library owns device buffers; one host
 * thread per context, the stream is internal.
 */
#ifndef YTTM_GPU_H
#define YTTM_GPU_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct yttm_ctx yttm_ctx;

int yttm_gpu_ctx_create(int device, yttm_ctx **out);
void yttm_gpu_ctx_destroy(yttm_ctx *ctx);
const char *yttm_gpu_last_error(void);

/* Finished trainings keep their device buffers in a pool for the next call (hipMalloc/hipFree of ~10 GB cost several ms and
 * synchronise the device).  This returns the cached buffers to the driver.  YTTM_NO_POOL=1 disables the pool.  (No reference
 * counterpart: bpe.cpp allocates with new/std containers.) */
void yttm_release_device_memory(void);

/* Every YTTM_* environment hook of the library -- test and tuning knobs, none part of the drop-in surface -- as Markdown table rows
 * "| `NAME` | default | kind | what it does |" (csrc/yttm_config.h holds the one table; INTEGRATION.md prints it).  The hooks are read when a
 * trainer context or an encoder is created, never per round or per launch.  (No reference counterpart.) */
const char *yttm_config_table(void);

/* multi-GPU: attach a communicator (yttm_comm* from include/yttm_mi355x.h, passed as void*) before the stages run;
 * the context then holds ITS shard and the pair table holds GLOBAL counts */
int yttm_gpu_ctx_set_comm(yttm_ctx *ctx, void *comm);

/* corpus bytes: copy from host, or adopt a buffer already resident in HBM (16-byte aligned device pointer) */
int yttm_gpu_upload_corpus(yttm_ctx *ctx, const uint8_t *utf8, uint64_t n);
int yttm_gpu_attach_corpus(yttm_ctx *ctx, const void *device_ptr, uint64_t n);

/* K1 -- compute_char_count (bpe.cpp:839-857): histogram of valid non-space code points (*n_inout: capacity in,
 * count out; unsorted) and the number of decode steps (valid + invalid + space chars). */
int yttm_gpu_char_hist(yttm_ctx *ctx, uint32_t *cps, uint64_t *cnts, uint32_t *n_inout, uint64_t *n_codepoints);

/* K2 -- remove_rare_chars + compute_word_count + the layout half of build_linked_list (bpe.cpp:357-418, :436-451):
 * chars listed in cp[] map to compact ids id[]; every other non-space char is deleted; words = [space_id, ids...],
 * deduplicated exactly, stored as token tiles in HBM.  n_ids_cap = upper bound of token ids (vocab_size). */
int yttm_gpu_build_word_table(yttm_ctx *ctx, const uint32_t *cp, const uint32_t *id, uint32_t n_alpha, uint32_t space_id,
                              uint32_t n_ids_cap, uint64_t *n_unique, uint64_t *n_tokens);
/* test helper: current (compacted) word table; tok capacity = n_tokens, off = n_unique+1, cnt = n_unique */
int yttm_gpu_download_word_table(yttm_ctx *ctx, uint32_t *tok, uint64_t *off, uint32_t *cnt, uint64_t *n_tokens_now);

/* K3 -- pair2cnt of build_linked_list summed over shards (bpe.cpp:461-475, :1076-1088): weighted bigram histogram
 * into the HBM pair table (includes the cross-rank exchange on a multi-GPU context). */
int yttm_gpu_pair_count(yttm_ctx *ctx, uint64_t *n_pairs);
/* all pairs with count > 0 (unsorted); *n_inout: capacity in, count out */
int yttm_gpu_download_pairs(yttm_ctx *ctx, uint64_t *pairs /* x<<32|y */, uint64_t *counts, uint64_t *n_inout);

/* K4 -- worker_doing_merge (bpe.cpp:491-812) for a batch of k mutually non-intersecting rules xyz[3k]
 * (rule_intersection, bpe.cpp:145-147; at most one x==y rule, last): applies them to every word and updates the
 * pair table exactly. */
int yttm_gpu_merge_apply(yttm_ctx *ctx, const uint32_t *xyz, uint32_t k);
/* Measurement mode of K4 (bench.py's untimed pass behind roofline.algorithmic_bytes_8d): on != 0 makes the following
 * yttm_gpu_merge_apply calls also count the WORDS that hold a merge site and their tokens (W_touched, T_touched).
 * out (optional) receives the totals so far: [0] merge sites, [1] tiles with a site, [2] their tokens, [3] words with a site,
 * [4] their tokens, [5] rounds. */
int yttm_gpu_k4_measure(yttm_ctx *ctx, int on, uint64_t out[6]);
/* check_cnt (bpe.cpp:1099-1108): exact global count of given pairs */
int yttm_gpu_pair_query(yttm_ctx *ctx, const uint64_t *pairs, uint32_t n, uint64_t *counts);
/* candidate filter feeding the host's ordered pick (PriorityQueue, bpe.cpp:271-314): pairs with count > tau_cnt, or
 * count == tau_cnt and max(x,y) <= tau_mx.  *n_inout: capacity in, number found out (may exceed capacity). */
int yttm_gpu_candidates(yttm_ctx *ctx, uint64_t tau_cnt, uint32_t tau_mx, uint64_t *pairs, uint64_t *counts, uint32_t *n_inout);

#ifdef __cplusplus
}
#endif
#endif
