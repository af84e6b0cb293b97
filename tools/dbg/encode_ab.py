"""GPU tuning aid: K5 by merge-round scheme (wave-wide rounds / one word per lane) on the
bench's 1e7 random 'abcd ' sentences and on Zipf text lines; FNV of the ids per variant (they must agree)."""
import ctypes as C, os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import gen, torch
from youtokentome_amd import _lib
L = _lib.load()
err, rep = C.create_string_buffer(2048), C.create_string_buffer(16384)
VARIANTS = [("rounds, table order", "0", "0", "1", ""), ("rounds, classes", "0", "0", "8", ""), ("lanes, table order", "48", "48", "1", ""), ("lanes, classes", "48", "48", "8", ""),
            ("lanes 32, classes", "32", "32", "8", ""), ("lanes 96, classes", "96", "96", "8", ""),
            ("table order, short 2^18", "48", "48", "1", str(1 << 18)), ("table order, short 2^22", "48", "48", "1", str(1 << 22)),
            ("table order, no short region", "48", "48", "1", str(1 << 31)), ("table order, short 2^16", "48", "48", "1", str(1 << 16))]
n_abcd = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
KINDS = sys.argv[2].split(",") if len(sys.argv) > 2 else ["abcd", "zipf"]
if len(sys.argv) > 3:  # variants by index
    VARIANTS = [VARIANTS[int(x)] for x in sys.argv[3].split(",")]
STEPS = int(sys.argv[4]) if len(sys.argv) > 4 else 3
for kind in KINDS:
    if kind == "abcd":
        train = gen.abcd_corpus(1_000_000_000, seed=19, line=100, survey_stream=False)
        text = gen.abcd_corpus(n_abcd * 129, seed=123, line=128, survey_stream=False)
    else:
        train = text = gen.zipf_corpus_fast(1_000_000_000, seed=7, vocab=400000)
    d = torch.frombuffer(bytearray(train), dtype=torch.uint8).cuda()
    model = "/tmp/ab_%s.model" % kind
    assert L.yttm_train_bpe_from_device(C.c_void_p(d.data_ptr()), d.numel(), model.encode(), 32000, 1.0, 0, 1, 2, 3, 0, 0, rep, 16384, err, 2048) == 0, err.value
    if text is not train:
        del d
        d = torch.frombuffer(bytearray(text), dtype=torch.uint8).cuda()
    del train
    h = C.c_void_p()
    assert L.yttm_encoder_create(model.encode(), 1, 0, C.byref(h), err, 2048) == 0
    arr = np.frombuffer(text, dtype=np.uint8)
    ends = np.flatnonzero(arr == 10).astype(np.int64) + 1
    k = len(ends)
    off = np.zeros(k + 1, np.int64); off[1:] = ends
    d_off = torch.from_numpy(off).cuda()
    mx = int((off[1:] - off[:-1]).max())
    n_ids, kms = C.c_uint64(), C.c_double()
    for name, lw, ls, cl, sh in VARIANTS:
        os.environ["YTTM_K5_LANE_WORDS"], os.environ["YTTM_K5_LANE_SENT"], os.environ["YTTM_K5_CLASSES"] = lw, ls, cl
        os.environ.pop("YTTM_WC_SHORT_SLOTS", None)
        if sh:
            os.environ["YTTM_WC_SHORT_SLOTS"] = sh
        out = []
        for mode in (0, 1):
            L.yttm_encoder_set_cache(h, mode, 0)
            def step():
                assert L.yttm_encode_device(h, C.c_void_p(d.data_ptr()), C.c_void_p(d_off.data_ptr()), k, int(off[-1]), mx, 0, 0, 0, 0.0, C.byref(n_ids), C.byref(kms), err, 2048) == 0, err.value
            step()
            ms = []
            for _ in range(STEPS):
                step()
                ms.append(kms.value)
            ids = np.zeros(n_ids.value, dtype=np.int32)
            o64 = np.zeros(k + 1, dtype=np.uint64)
            L.yttm_encode_fetch(h, ids.ctypes.data_as(_lib.i32p), o64.ctypes.data_as(_lib.u64p), k, err, 2048)
            out.append((min(ms), "%016x" % L.yttm_ids_fnv1a64(ids.ctypes.data_as(_lib.i32p), o64.ctypes.data_as(_lib.u64p), k)))
        print("%s %-22s direct %8.3f ms  cached %8.3f ms  (%d sentences, %d distinct words)  fnv %s %s" % (
            kind, name, out[0][0], out[1][0], k, L.yttm_encode_cache_words(h), out[0][1], out[1][1]), flush=True)
    ms = []
    for _ in range(3):  # BPE-dropout 0.1 (never through the word cache)
        assert L.yttm_encode_device(h, C.c_void_p(d.data_ptr()), C.c_void_p(d_off.data_ptr()), k, int(off[-1]), mx, 0, 0, 0, 0.1, C.byref(n_ids), C.byref(kms), err, 2048) == 0, err.value
        ms.append(kms.value)
    print("%s dropout 0.1: %s ms, %.3f ids per sentence" % (kind, ["%.1f" % x for x in ms], n_ids.value / k), flush=True)
    L.yttm_encoder_destroy(h)
    del d, d_off
    torch.cuda.empty_cache()
