"""GPU tuning aid: batch encode with and without the word cache by batch size (Zipf text lines and random 'abcd ' sentences)."""
import ctypes as C, os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import gen, torch
from youtokentome_amd import _lib
L = _lib.load()
err, rep = C.create_string_buffer(2048), C.create_string_buffer(16384)
for kind in ("zipf", "abcd"):
    text = gen.zipf_corpus_fast(200_000_000, seed=7, vocab=400000) if kind == "zipf" else gen.abcd_corpus(200_000_000, seed=19, line=128, survey_stream=True)
    d = torch.frombuffer(bytearray(text), dtype=torch.uint8).cuda()
    model = "/tmp/cc_%s.model" % kind
    assert L.yttm_train_bpe_from_device(C.c_void_p(d.data_ptr()), d.numel(), model.encode(), 32000, 1.0, 0, 1, 2, 3, 0, 0, rep, 16384, err, 2048) == 0, err.value
    h = C.c_void_p()
    assert L.yttm_encoder_create(model.encode(), 1, 0, C.byref(h), err, 2048) == 0
    arr = np.frombuffer(text, dtype=np.uint8)
    ends = np.flatnonzero(arr == 10).astype(np.int64) + 1
    for mb in (0.25, 1, 2, 4, 8, 32, 128):
        k = int(np.searchsorted(ends, int(mb * 1e6)))
        off = np.zeros(k + 1, np.int64); off[1:] = ends[:k]
        d_off = torch.from_numpy(off).cuda()
        mx = int((off[1:] - off[:-1]).max())
        n_ids, kms = C.c_uint64(), C.c_double()
        res = []
        for mode in (0, 1):
            L.yttm_encoder_set_cache(h, mode, 0)
            def step():
                assert L.yttm_encode_device(h, C.c_void_p(d.data_ptr()), C.c_void_p(d_off.data_ptr()), k, int(off[-1]), mx, 0, 0, 0, 0.0, C.byref(n_ids), C.byref(kms), err, 2048) == 0, err.value
            step(); step()
            t0 = time.perf_counter()
            for _ in range(5):
                step()
            res.append((time.perf_counter() - t0) / 5 * 1e3)
        print("%s %6.2f MB %8d sentences: direct %.3f ms, cached %.3f ms (%d distinct words) -> %.2fx" % (kind, off[-1] / 1e6, k, res[0], res[1], L.yttm_encode_cache_words(h), res[0] / res[1]))
    L.yttm_encoder_destroy(h)
    del d
