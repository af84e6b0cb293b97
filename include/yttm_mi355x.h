/* yttm_mi355x.h -- the drop-in boundary: C ABI of libyttm_mi355x.so.
 *
 * These entry points are what a binding of the reference (youtokentome/cpp/yttm.pyx) binds instead of the C++
 * surface of youtokentome/cpp/bpe.h.  Each one cites the reference interface it replaces (file:line in
 * /root/reference).  Plain pointers and sizes only; no C++/torch types.  Errors follow the reference's
 * Status{code,message} (utils.h:56-64): return value 0 = ok, otherwise the message is copied to `err`
 * (user-visible wording kept verbatim); the binding raises ValueError(message) like yttm.pyx:61-62,:84-85.
 *
 * Sentences cross the boundary packed: UTF-8 bytes + offsets[n_sent+1] (sentence i = bytes[offsets[i]..offsets[i+1])).
 * Output arrays are malloc'ed by the library and released with yttm_free().
 * Threading: one host thread per encoder/ctx at a time; n_threads is accepted for API compatibility and ignored
 * (the hot path runs on the GPU).  The library needs a visible MI355X (gfx950) and fails loudly without one.
 */
#ifndef YTTM_MI355X_H
#define YTTM_MI355X_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* replaces: vkcom::Status train_bpe(const std::string& input_path, const std::string& model_path, int vocab_size,
 *           BpeConfig config)                                       bpe.h:19, bpe.cpp:1368; yttm.pyx:64-85 */
int yttm_train_bpe(const char *input_path, const char *model_path, int vocab_size, double coverage, int n_threads,
                   int pad_id, int unk_id, int bos_id, int eos_id, char *err, int errlen);

/* The same call with the device ordinal chosen by the caller and the report of the *_from_device variant below
 * (adds "seconds_upload": file -> pinned chunks -> HBM, which replaces fast_read_file_utf8, bpe.cpp:67-84). */
int yttm_train_bpe_ex(const char *input_path, const char *model_path, int vocab_size, double coverage, int n_threads,
                      int pad_id, int unk_id, int bos_id, int eos_id, int device, char *report_json, int report_len,
                      char *err, int errlen);

/* Same training on a corpus that is already in host memory (no file read), or already resident in HBM
 * (`d_text` = device pointer, 16-byte aligned).  `device` = HIP device ordinal.  `report_json` (optional, may be
 * NULL) receives a JSON object with wall-time phases and per-kernel GPU time / algorithmic bytes
 * (profile != 0 times every kernel with HIP events on the launch stream), and counters of the merge loop: "rounds",
 * "rules", "rounds_exhausted", "batch_extensions", "batch_splits" (word-mode batches of 129 .. 256 rules cut to their first
 * 128), "word_switch_round" / "word_rounds" / "word_fused_rounds" (rounds in K4's word mode; of those, one launch each) /
 * "word_all_rounds" (rounds that had to visit every word), "index_builds", "hot_rebuilds", "top_refills", "repacks", "front_end_overlapped" (1: the char histogram, the segment
 * starts and the word dedup ran on the parts of the text while the rest was still being uploaded, and that word table was taken). */
int yttm_train_bpe_from_memory(const uint8_t *text, uint64_t n, const char *model_path, int vocab_size, double coverage,
                               int pad_id, int unk_id, int bos_id, int eos_id, int device, char *report_json,
                               int report_len, char *err, int errlen);
int yttm_train_bpe_from_device(const void *d_text, uint64_t n, const char *model_path, int vocab_size, double coverage,
                               int pad_id, int unk_id, int bos_id, int eos_id, int device, int profile,
                               char *report_json, int report_len, char *err, int errlen);

/* replaces: class vkcom::BaseEncoder                                 bpe.h:22-82 */
typedef struct yttm_encoder yttm_encoder;

/* BaseEncoder(const std::string& model_path, int n_threads, Status*) bpe.h:33, bpe.cpp:1643; yttm.pyx:58-62 */
int yttm_encoder_create(const char *model_path, int n_threads, int device, yttm_encoder **out, char *err, int errlen);
/* yttm.pyx:55-56 __dealloc__ */
void yttm_encoder_destroy(yttm_encoder *enc);

/* Status encode_as_ids(const vector<string>&, vector<vector<int>>*, bool bos, bool eos, bool reverse,
 *                      double dropout_prob) const                    bpe.h:37-39, bpe.cpp:1740; yttm.pyx:87-109 */
int yttm_encode_as_ids(yttm_encoder *enc, const uint8_t *bytes, const uint64_t *offsets, uint64_t n_sent, int bos,
                       int eos, int reverse, double dropout_prob, int32_t **ids, uint64_t **out_offsets, char *err,
                       int errlen);

/* Status encode_as_subwords(...)                                      bpe.h:41-46, bpe.cpp:1757; yttm.pyx:110-124
 * pieces come back as one blob + piece_off[n_pieces+1] + sent_off[n_sent+1] (piece index ranges per sentence). */
int yttm_encode_as_subwords(yttm_encoder *enc, const uint8_t *bytes, const uint64_t *offsets, uint64_t n_sent, int bos,
                            int eos, int reverse, double dropout_prob, char **blob, uint64_t **piece_off,
                            uint64_t *n_pieces, uint64_t **sent_off, char *err, int errlen);

/* Device-resident batch encode (what bench.py times): input bytes/offsets already in HBM, ids stay in HBM inside
 * the encoder until yttm_encode_fetch copies them out.  kernel_ms (optional) = HIP-event time of the K5 launch. */
int yttm_encode_device(yttm_encoder *enc, const void *d_bytes, const void *d_offsets, uint64_t n_sent,
                       uint64_t total_bytes, uint64_t max_sentence_bytes, int bos, int eos, int reverse,
                       double dropout_prob, uint64_t *n_ids, double *kernel_ms, char *err, int errlen);
int yttm_encode_fetch(yttm_encoder *enc, int32_t *ids, uint64_t *out_offsets, uint64_t n_sent, char *err, int errlen);

/* Word-level encode cache (SURVEY.md 8f "N4"; the reference has no counterpart: bpe.cpp:1497-1632 encodes every word occurrence).
 * mode 0: every batch goes straight through the encode kernel; 1: distinct words are encoded once whenever that is possible
 * (dropout_prob == 0); 2 (default): the same for batches of at least min_bytes.  The ids are identical either way.
 * yttm_encode_cache_words: distinct words of the last yttm_encode_device batch, 0 if it did not go through the cache. */
int yttm_encoder_set_cache(yttm_encoder *enc, int mode, uint64_t min_bytes);
uint64_t yttm_encode_cache_words(yttm_encoder *enc);

/* Status id_to_subword(int id, string* subword, bool replace_space) bpe.h:48, bpe.cpp:1774; yttm.pyx:129-134 */
int yttm_id_to_subword(yttm_encoder *enc, int id, char **subword, char *err, int errlen);
/* int subword_to_id(const string& token) const                       bpe.h:50, bpe.cpp:1809; yttm.pyx:126-127 */
int yttm_subword_to_id(yttm_encoder *enc, const char *token);
/* Status decode(const vector<vector<int>>& ids, vector<string>* sentences, const unordered_set<int>* ignore_ids)
 *                                                                     bpe.h:52-54, bpe.cpp:1828; yttm.pyx:136-158 */
int yttm_decode(yttm_encoder *enc, const int32_t *ids, const uint64_t *offsets, uint64_t n_sent, const int32_t *ignore_ids,
                uint64_t n_ignore, char **blob, uint64_t **out_offsets, char *err, int errlen);
/* int vocab_size() const                                              bpe.h:62, bpe.cpp:1692; yttm.pyx:160-161 */
int yttm_vocab_size(yttm_encoder *enc);
/* vector<string> vocabulary() const                                   bpe.h:64, bpe.cpp:1884; yttm.pyx:163-165 */
int yttm_vocabulary(yttm_encoder *enc, char **blob, uint64_t **offsets, uint64_t *n);

/* The streaming loops of the `yttm` command line.  The reference's read std::cin and write std::cout; here the file
 * descriptors are arguments (the Python module passes 0 and 1).
 * Status encode_cli(const string& output_type, bool stream, bool bos, bool eos, bool reverse, double dropout_prob) const
 *                                                                     bpe.h:66-68, bpe.cpp:1942-2014; yttm.pyx:167-170
 * batch mode: stdin in batches of >= 10 MiB of line bytes -> H2D / K5 / D2H of one batch overlapped with the formatting of
 * the previous one (two encoder lanes) -> "<token> <token> ...\n" per sentence (utils.h:92-103); progress on stderr. */
int yttm_encode_cli(yttm_encoder *enc, const char *output_type, int stream, int bos, int eos, int reverse, double dropout_prob,
                    int in_fd, int out_fd, char *err, int errlen);
/* Status decode_cli(const unordered_set<int>* ignore_ids) const       bpe.h:70, bpe.cpp:2016-2028; yttm.pyx:172-178 */
int yttm_decode_cli(yttm_encoder *enc, const int32_t *ignore_ids, uint64_t n_ignore, int in_fd, int out_fd, char *err, int errlen);
/* void vocab_cli(bool verbose) const                                  bpe.h:71, bpe.cpp:1896-1940; yttm.pyx:180-181 */
int yttm_vocab_cli(yttm_encoder *enc, int verbose, int out_fd, char *err, int errlen);

/* ---- multi-GPU (one process per GPU) ---------------------------------------------------------------
 * The corpus shards across ranks; the only exchanged quantities are the char histogram (once) and sparse pair-count
 * deltas (after the initial count and after every merge round) -- the RCCL form of the reference's main thread summing
 * per-thread maps (bpe.cpp:1099-1108, :1245-1251).  Every rank passes ITS shard to the *_comm entry points; rank 0
 * writes the model file. */
typedef struct yttm_comm yttm_comm;
/* RCCL over xGMI: rank 0 makes the id, the application broadcasts its 128 bytes, every rank creates its communicator */
int yttm_comm_rccl_unique_id(uint8_t out[128]);
int yttm_comm_rccl_create(const uint8_t id[128], int rank, int world, int device, yttm_comm **out);
/* host-callback transport (torch.distributed/gloo, MPI, ...): in-place sum of n uint64; gather of the other ranks' bytes */
typedef int (*yttm_allreduce_u64_fn)(void *user, unsigned long long *buf, size_t n);
typedef int (*yttm_allgather_bytes_fn)(void *user, const void *send, size_t send_bytes, void *recv, size_t recv_cap,
                                       unsigned long long *recv_bytes);
int yttm_comm_callback_create(int rank, int world, yttm_allreduce_u64_fn allreduce, yttm_allgather_bytes_fn allgather,
                              void *user, yttm_comm **out);
void yttm_comm_destroy(yttm_comm *comm);
/* train_bpe (bpe.h:19) on a node of GPUs: every rank passes the SAME file and reads its own byte range of it, cut at white space like the
 * reference's per-thread split (bpe.cpp:864-873).  comm == NULL: one GPU (yttm_train_bpe_ex with the per-kernel timers: profile = 1). */
int yttm_train_bpe_comm(const char *input_path, const char *model_path, int vocab_size, double coverage, int n_threads, int pad_id,
                        int unk_id, int bos_id, int eos_id, int device, int profile, yttm_comm *comm, char *report_json,
                        int report_len, char *err, int errlen);
int yttm_train_bpe_from_device_comm(const void *d_text, uint64_t n, const char *model_path, int vocab_size, double coverage,
                                    int pad_id, int unk_id, int bos_id, int eos_id, int device, int profile,
                                    yttm_comm *comm, char *report_json, int report_len, char *err, int errlen);
int yttm_train_bpe_from_memory_comm(const uint8_t *text, uint64_t n, const char *model_path, int vocab_size, double coverage,
                                    int pad_id, int unk_id, int bos_id, int eos_id, int device, yttm_comm *comm,
                                    char *report_json, int report_len, char *err, int errlen);

/* Checksum of an encode result for comparisons across implementations: FNV-1a-64 over, per sentence, the little-endian
 * bytes of the uint32 length and of each int32 id (no reference counterpart; oracle/ref_driver.cpp hashes the reference's
 * encode_as_ids output the same way). */
unsigned long long yttm_ids_fnv1a64(const int32_t *ids, const uint64_t *offsets, uint64_t n_sent);

void yttm_free(void *p);
/* "gfx950 MI355X ..." or an error text when no usable GPU is visible */
int yttm_device_info(int device, char *buf, int buflen);

#ifdef __cplusplus
}
#endif
#endif
