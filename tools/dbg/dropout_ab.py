"""GPU tuning aid: BPE-dropout encode (BASELINE configs[4]: 1e7 sentences of 128 chars, p = 0.1) under environment hooks, one process.
usage: python tools/dbg/dropout_ab.py [n_sentences] -- NAME:K=V,K=V NAME2: ...     (hooks are read when an encoder is created; the pseudo-hook
P=<prob> sets the dropout probability of a variant: 1.0 = every word ends at its first pop, the fixed part of the kernel)"""
import ctypes as C, os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import gen, torch
from youtokentome_amd import _lib
L = _lib.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] != "--" else 10_000_000
variants = [(a.partition(":")[0], dict(x.split("=", 1) for x in a.partition(":")[2].split(",") if x)) for a in sys.argv[sys.argv.index("--") + 1:]] if "--" in sys.argv else [("base", {})]
err, rep = C.create_string_buffer(2048), C.create_string_buffer(16384)
train = gen.abcd_corpus(1_000_000_000, seed=19, survey_stream=True)
d = torch.frombuffer(bytearray(train), dtype=torch.uint8).cuda()
del train
model = "/tmp/dropout_ab.model"
assert L.yttm_train_bpe_from_device(C.c_void_p(d.data_ptr()), d.numel(), model.encode(), 32000, 1.0, 0, 1, 2, 3, 0, 0, rep, 16384, err, 2048) == 0, err.value
del d
line = 128
host = gen.abcd_corpus(n * (line + 1), seed=123, line=line, survey_stream=True)
n = len(host) // (line + 1)
db = torch.frombuffer(bytearray(host), dtype=torch.uint8).cuda()
do = torch.arange(n + 1, dtype=torch.int64, device="cuda") * (line + 1)
for name, env in variants:
    prob = float(env.pop("P", "0.1"))
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    os.environ["YTTM_DROPOUT_SEED"] = "777"
    h = C.c_void_p()
    assert L.yttm_encoder_create(model.encode(), 1, 0, C.byref(h), err, 2048) == 0, err.value
    n_ids, kms = C.c_uint64(), C.c_double()
    ks = []
    for i in range(4):
        assert L.yttm_encode_device(h, C.c_void_p(db.data_ptr()), C.c_void_p(do.data_ptr()), n, db.numel(), line + 1, 0, 0, 0, prob, C.byref(n_ids), C.byref(kms), err, 2048) == 0, err.value
        if i:
            ks.append(kms.value)
    ids = np.zeros(n_ids.value, dtype=np.int32); off = np.zeros(n + 1, dtype=np.uint64)
    L.yttm_encode_fetch(h, ids.ctypes.data_as(_lib.i32p), off.ctypes.data_as(_lib.u64p), n, err, 2048)
    print("%-10s kernel ms %s  ids/sentence %.4f  fnv %016x" % (name, ["%.2f" % x for x in ks], n_ids.value / n, L.yttm_ids_fnv1a64(ids.ctypes.data_as(_lib.i32p), off.ctypes.data_as(_lib.u64p), n)), flush=True)
    L.yttm_encoder_destroy(h)
    for k, v in saved.items():
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = v
