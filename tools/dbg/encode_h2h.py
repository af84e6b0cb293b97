"""GPU tuning aid: host -> host encode (yttm_encode_as_ids: packed host bytes + offsets in, malloc'ed ids out) of 1e7 random 'abcd ' sentences,
plain copies against the pinned-chunk pipeline, by number of IO threads."""
import ctypes as C, os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import gen, torch
from youtokentome_amd import _lib
L = _lib.load()
err, rep = C.create_string_buffer(2048), C.create_string_buffer(16384)
n_sent = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
train = gen.abcd_corpus(300_000_000, seed=19, line=100, survey_stream=False)
d = torch.frombuffer(bytearray(train), dtype=torch.uint8).cuda()
model = "/tmp/h2h.model"
assert L.yttm_train_bpe_from_device(C.c_void_p(d.data_ptr()), d.numel(), model.encode(), 32000, 1.0, 0, 1, 2, 3, 0, 0, rep, 16384, err, 2048) == 0, err.value
del d, train
torch.cuda.empty_cache()
text = gen.abcd_corpus(n_sent * 129, seed=123, line=128, survey_stream=False)
h_off = (np.arange(n_sent + 1, dtype=np.uint64) * 129)
h = C.c_void_p()
assert L.yttm_encoder_create(model.encode(), 1, 0, C.byref(h), err, 2048) == 0
for name, env in (("one batch, chunks", {"YTTM_ENC_PIPE_FROM": str(1 << 60)}), ("sub-batches of 192 MB", {"YTTM_ENC_SUB_MB": "192"}), ("sub-batches of 320 MB", {"YTTM_ENC_SUB_MB": "320"}), ("sub-batches of 440 MB", {"YTTM_ENC_SUB_MB": "440"}), ("sub-batches of 650 MB", {"YTTM_ENC_SUB_MB": "650"}), ("sub-batches of 320 MB, 2 MB chunks", {"YTTM_ENC_SUB_MB": "320", "YTTM_IO_CHUNK_MB": "2"}), ("sub-batches of 320 MB, 2 threads", {"YTTM_ENC_SUB_MB": "320", "YTTM_IO_THREADS": "2"})):
    for k in ("YTTM_ENC_STAGED_FROM", "YTTM_IO_THREADS", "YTTM_IO_CHUNK_MB", "YTTM_ENC_PIPE_FROM", "YTTM_ENC_SUB_MB"):
        os.environ.pop(k, None)
    os.environ.update(env)
    os.environ["YTTM_TRACE"] = "1"
    ts = []
    for _ in range(4):
        p_ids, p_off = _lib.i32p(), _lib.u64p()
        t0 = time.perf_counter()
        rc = L.yttm_encode_as_ids(h, text, h_off.ctypes.data_as(_lib.u64p), n_sent, 0, 0, 0, 0.0, C.byref(p_ids), C.byref(p_off), err, 2048)
        ts.append(time.perf_counter() - t0)
        assert rc == 0, err.value
        n_ids = p_off[n_sent]
        fnv = "%016x" % L.yttm_ids_fnv1a64(p_ids, p_off, n_sent)
        L.yttm_free(C.cast(p_ids, C.c_void_p))
        L.yttm_free(C.cast(p_off, C.c_void_p))
    print("%-28s best %.1f ms of %s -> %.3g sentences/s  (%d ids, fnv %s)" % (name, min(ts) * 1e3, ["%.0f" % (t * 1e3) for t in ts], n_sent / min(ts), n_ids, fnv), flush=True)
