/* oracle/bpe_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the YouTokenToMe BPE train + encode hot path
 * (reference: /root/reference/youtokentome/cpp/bpe.cpp, utf8.cpp, utils.cpp;
 * each function in bpe_oracle.c cites the file:line it follows).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (youtokentome_amd/, libyttm_mi355x.so) never
 * links, imports or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle_vs_ref.py checks this restatement
 * against the unmodified reference compiled with -DDETERMINISTIC_QUEUE
 * (oracle/_ref/yttm_ref_det; byte-identical model files, identical ids, and
 * bit-identical dropout output at n_threads=1) whenever oracle/_ref exists, and
 * tests/test_oracle_golden.py checks it against the committed fixtures in
 * tests/golden/ (generated from the reference by tests/golden/make_golden.py).
 */
#ifndef BPE_ORACLE_H
#define BPE_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- training ------------------------------------------------------------ */

/* Train on an in-memory UTF-8 corpus and write the model file.
 * Returns 0 on success, 1 on a reference-style Status error (message copied to err). */
int oracle_train(const uint8_t *text, uint64_t n, int vocab_size, double coverage,
                 int pad_id, int unk_id, int bos_id, int eos_id,
                 const char *model_path, char *err, int errlen);

/* Intermediate results of the training front end / merge loop, for per-kernel parity tests.
 * All arrays are malloc'ed by the oracle and released by oracle_free(). */

/* K1: histogram of valid non-space code points (sorted by code point) and the number of
 * UTF8Iterator steps (valid + invalid + space).  bpe.cpp:839-857. */
int oracle_char_hist(const uint8_t *text, uint64_t n, uint32_t **cps, uint64_t **cnts,
                     uint64_t *n_chars, uint64_t *data_len);

/* Alphabet (bpe.cpp:316-355): compact ids.  cps_out/ids_out are in model-file (hash-slot) order of the
 * COMPACT ids (before rename).  removed_out = code points cut by coverage. */
int oracle_alphabet(const uint32_t *cps, const uint64_t *cnts, uint64_t n_chars, uint64_t data_len,
                    double coverage, int n_special, uint32_t **cps_out, uint32_t **ids_out, uint64_t *n_out,
                    uint32_t **removed_out, uint64_t *n_removed);

/* K2: unique-word table given a char->compact-id map (cp_map[i] -> id_map[i]); chars not in the map and
 * not spaces are deleted (removed / invalid).  Words are returned sorted lexicographically by token
 * sequence: tok = concatenated tokens, off[U+1], cnt[U].  bpe.cpp:357-418. */
int oracle_word_table(const uint8_t *text, uint64_t n, const uint32_t *cp_map, const uint32_t *id_map,
                      uint64_t n_map, uint32_t space_id, uint32_t **tok, uint64_t **off, uint64_t **cnt,
                      uint64_t *n_words);

/* K3: weighted pair counts of a word table (run rule: floor(L/2) per run of equal tokens),
 * sorted by (x,y).  bpe.cpp:436-478; stress_test.cpp:150-159. */
int oracle_pair_counts(const uint32_t *tok, const uint64_t *off, const uint64_t *cnt, uint64_t n_words,
                       uint32_t **xs, uint32_t **ys, uint64_t **cs, uint64_t *n_pairs);

/* K4: apply rules (x,y)->z sequentially, each left-to-right non-overlapping (stress_test.cpp:181-188),
 * in place on a word table (off is rewritten compacted). */
int oracle_apply_rules(uint32_t *tok, uint64_t *off, uint64_t n_words, const uint32_t *rules_xyz, uint64_t n_rules);

/* Greedy merge list on COMPACT ids (before rename): returns rules (x,y,z) and their counts. */
int oracle_learn_rules(const uint32_t *tok, const uint64_t *off, const uint64_t *cnt, uint64_t n_words,
                       uint32_t first_new_id, uint32_t max_rules, uint32_t **rules_xyz, uint64_t **rule_cnt,
                       uint64_t *n_rules);

/* Slot order of a ska::flat_hash_map<uint32_t,...> after inserting keys in the given order into a fresh
 * map and then copy-constructing it once (bpe.cpp:1289 -> utils.cpp:57-59). */
int oracle_ska_order(const uint32_t *keys, uint64_t n, uint32_t *order_out);

void oracle_free(void *p);

/* ---- encoding ------------------------------------------------------------ */

typedef struct oracle_model oracle_model;

oracle_model *oracle_model_load(const char *model_path, char *err, int errlen);
void oracle_model_free(oracle_model *m);
int oracle_model_vocab_size(const oracle_model *m);

/* Reset the emulated process-global std::mt19937 (default seed 5489), bpe.cpp:1415. */
void oracle_rng_reset(void);

/* Encode one sentence to ids (bpe.cpp:1455-1632).  Returns the number of ids (may exceed cap: then only
 * cap ids were written), or -1 with err set for the bos/eos Status errors of bpe.cpp:1702-1707. */
int64_t oracle_encode(const oracle_model *m, const uint8_t *sentence, uint64_t n, int bos, int eos, int reverse,
                      double dropout_prob, int32_t *out, uint64_t cap, char *err, int errlen);

/* Batch form: sentences given as bytes + offsets[S+1]; ids_out/out_off[S+1] malloc'ed by the oracle. */
int oracle_encode_batch(const oracle_model *m, const uint8_t *bytes, const uint64_t *offsets, uint64_t n_sent,
                        int bos, int eos, int reverse, double dropout_prob, int32_t **ids_out,
                        uint64_t **out_off, char *err, int errlen);

#ifdef __cplusplus
}
#endif
#endif
