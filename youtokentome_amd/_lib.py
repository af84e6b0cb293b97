"""ctypes loader for libyttm_mi355x.so (the C ABI of include/yttm_mi355x.h and include/yttm_gpu.h).

The library is built in-tree by `make -C youtokentome_amd/csrc` (hipcc, --offload-arch=gfx950).  There is NO CPU
fallback: if the shared library is missing this module raises, and if no MI355X is visible every call fails loudly.
(`YTTM_AMD_LIB` may point at another build of the same sources -- the CPU test-suite uses it to load the build made
against the HIP emulator in tests/hipsim/; that build exists for logic tests only.)"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("YTTM_AMD_LIB") or os.path.join(_HERE, "libyttm_mi355x.so")

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
i32p = C.POINTER(C.c_int32)
ERRLEN = 2048

EXPORTS = [
    # include/yttm_mi355x.h
    "yttm_train_bpe", "yttm_train_bpe_ex", "yttm_train_bpe_from_memory", "yttm_train_bpe_from_device", "yttm_encoder_create",
    "yttm_encoder_destroy", "yttm_encode_as_ids", "yttm_encode_as_subwords", "yttm_encode_device", "yttm_encode_fetch",
    "yttm_id_to_subword", "yttm_subword_to_id", "yttm_decode", "yttm_vocab_size", "yttm_vocabulary", "yttm_free", "yttm_ids_fnv1a64", "yttm_encode_cli", "yttm_decode_cli", "yttm_vocab_cli",
    "yttm_encoder_set_cache", "yttm_encode_cache_words",
    "yttm_device_info", "yttm_comm_rccl_unique_id", "yttm_comm_rccl_create", "yttm_comm_callback_create",
    "yttm_comm_destroy", "yttm_train_bpe_comm", "yttm_train_bpe_from_device_comm", "yttm_train_bpe_from_memory_comm",
    # include/yttm_gpu.h
    "yttm_gpu_ctx_create", "yttm_gpu_ctx_destroy", "yttm_gpu_ctx_set_comm", "yttm_gpu_last_error", "yttm_release_device_memory", "yttm_config_table",
    "yttm_gpu_upload_corpus",
    "yttm_gpu_attach_corpus", "yttm_gpu_char_hist", "yttm_gpu_build_word_table", "yttm_gpu_download_word_table",
    "yttm_gpu_pair_count", "yttm_gpu_download_pairs", "yttm_gpu_merge_apply", "yttm_gpu_pair_query",
    "yttm_gpu_candidates", "yttm_gpu_k4_measure",
]

_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()' "
            "or make -C youtokentome_amd/csrc).  youtokentome_amd has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    cs, ci, cd, cvp = C.c_char_p, C.c_int, C.c_double, C.c_void_p
    L.yttm_train_bpe.argtypes = [cs, cs, ci, cd, ci, ci, ci, ci, ci, cs, ci]
    L.yttm_train_bpe_ex.argtypes = [cs, cs, ci, cd, ci, ci, ci, ci, ci, ci, cs, ci, cs, ci]
    L.yttm_train_bpe_from_memory.argtypes = [cs, C.c_uint64, cs, ci, cd, ci, ci, ci, ci, ci, cs, ci, cs, ci]
    L.yttm_train_bpe_from_device.argtypes = [cvp, C.c_uint64, cs, ci, cd, ci, ci, ci, ci, ci, ci, cs, ci, cs, ci]
    L.yttm_encoder_create.argtypes = [cs, ci, ci, C.POINTER(cvp), cs, ci]
    L.yttm_encoder_destroy.argtypes = [cvp]
    L.yttm_encoder_destroy.restype = None
    L.yttm_encode_as_ids.argtypes = [cvp, cs, u64p, C.c_uint64, ci, ci, ci, cd, C.POINTER(i32p), C.POINTER(u64p), cs, ci]
    L.yttm_encode_as_subwords.argtypes = [cvp, cs, u64p, C.c_uint64, ci, ci, ci, cd, C.POINTER(cvp), C.POINTER(u64p),
                                          u64p, C.POINTER(u64p), cs, ci]
    L.yttm_encode_device.argtypes = [cvp, cvp, cvp, C.c_uint64, C.c_uint64, C.c_uint64, ci, ci, ci, cd, u64p,
                                     C.POINTER(cd), cs, ci]
    L.yttm_encode_fetch.argtypes = [cvp, i32p, u64p, C.c_uint64, cs, ci]
    L.yttm_encoder_set_cache.argtypes = [cvp, ci, C.c_uint64]
    L.yttm_encode_cache_words.argtypes = [cvp]
    L.yttm_encode_cache_words.restype = C.c_uint64
    L.yttm_id_to_subword.argtypes = [cvp, ci, C.POINTER(cvp), cs, ci]
    L.yttm_subword_to_id.argtypes = [cvp, cs]
    L.yttm_decode.argtypes = [cvp, i32p, u64p, C.c_uint64, i32p, C.c_uint64, C.POINTER(cvp), C.POINTER(u64p), cs, ci]
    L.yttm_vocab_size.argtypes = [cvp]
    L.yttm_vocabulary.argtypes = [cvp, C.POINTER(cvp), C.POINTER(u64p), u64p]
    L.yttm_encode_cli.argtypes = [cvp, cs, ci, ci, ci, ci, cd, ci, ci, cs, ci]
    L.yttm_decode_cli.argtypes = [cvp, i32p, C.c_uint64, ci, ci, cs, ci]
    L.yttm_vocab_cli.argtypes = [cvp, ci, ci, cs, ci]
    L.yttm_ids_fnv1a64.argtypes = [i32p, u64p, C.c_uint64]
    L.yttm_ids_fnv1a64.restype = C.c_ulonglong
    L.yttm_free.argtypes = [cvp]
    L.yttm_free.restype = None
    L.yttm_device_info.argtypes = [ci, cs, ci]
    ALLREDUCE_FN = C.CFUNCTYPE(ci, cvp, C.POINTER(C.c_ulonglong), C.c_size_t)
    ALLGATHER_FN = C.CFUNCTYPE(ci, cvp, cvp, C.c_size_t, cvp, C.c_size_t, C.POINTER(C.c_ulonglong))
    L.ALLREDUCE_FN, L.ALLGATHER_FN = ALLREDUCE_FN, ALLGATHER_FN
    L.yttm_comm_callback_create.argtypes = [ci, ci, ALLREDUCE_FN, ALLGATHER_FN, cvp, C.POINTER(cvp)]
    L.yttm_comm_destroy.argtypes = [cvp]
    L.yttm_comm_destroy.restype = None
    L.yttm_train_bpe_from_device_comm.argtypes = [cvp, C.c_uint64, cs, ci, cd, ci, ci, ci, ci, ci, ci, cvp, cs, ci, cs, ci]
    L.yttm_train_bpe_from_memory_comm.argtypes = [cs, C.c_uint64, cs, ci, cd, ci, ci, ci, ci, ci, cvp, cs, ci, cs, ci]
    L.yttm_train_bpe_comm.argtypes = [cs, cs, ci, cd, ci, ci, ci, ci, ci, ci, ci, cvp, cs, ci, cs, ci]
    if hasattr(L, "yttm_comm_rccl_create"):  # absent from the emulator build (no RCCL there)
        L.yttm_comm_rccl_unique_id.argtypes = [u8p]
        L.yttm_comm_rccl_create.argtypes = [u8p, ci, ci, ci, C.POINTER(cvp)]
    L.yttm_gpu_ctx_create.argtypes = [ci, C.POINTER(cvp)]
    L.yttm_gpu_ctx_destroy.argtypes = [cvp]
    L.yttm_gpu_ctx_set_comm.argtypes = [cvp, cvp]
    L.yttm_gpu_ctx_destroy.restype = None
    L.yttm_gpu_last_error.restype = cs
    L.yttm_release_device_memory.restype = None
    L.yttm_release_device_memory.argtypes = []
    L.yttm_gpu_upload_corpus.argtypes = [cvp, cs, C.c_uint64]
    L.yttm_gpu_attach_corpus.argtypes = [cvp, cvp, C.c_uint64]
    L.yttm_gpu_char_hist.argtypes = [cvp, u32p, u64p, u32p, u64p]
    L.yttm_gpu_build_word_table.argtypes = [cvp, u32p, u32p, C.c_uint32, C.c_uint32, C.c_uint32, u64p, u64p]
    L.yttm_gpu_download_word_table.argtypes = [cvp, u32p, u64p, u32p, u64p]
    L.yttm_gpu_pair_count.argtypes = [cvp, u64p]
    L.yttm_gpu_download_pairs.argtypes = [cvp, u64p, u64p, u64p]
    L.yttm_gpu_merge_apply.argtypes = [cvp, u32p, C.c_uint32]
    L.yttm_gpu_pair_query.argtypes = [cvp, u64p, C.c_uint32, u64p]
    L.yttm_gpu_k4_measure.argtypes = [cvp, ci, u64p]
    L.yttm_gpu_candidates.argtypes = [cvp, C.c_uint64, C.c_uint32, u64p, u64p, u32p]
    _lib = L
    return L


def device_info(device=0):
    buf = C.create_string_buffer(512)
    rc = load().yttm_device_info(device, buf, 512)
    return rc, buf.value.decode(errors="replace")
