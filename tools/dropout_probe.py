"""Throughput of the BPE-dropout encode path next to the deterministic one (tuning aid; run on the GPU box).
Packed API (bytes + offsets), so the Python list marshalling is not what is measured."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen  # noqa: E402
import youtokentome_amd as yttm  # noqa: E402

open("/tmp/c.txt", "wb").write(bytes(gen.abcd_corpus(50_000_000, seed=19)))
yttm.BPE.train("/tmp/c.txt", "/tmp/m.model", 8000)
bpe = yttm.BPE("/tmp/m.model")
rng = np.random.default_rng(123)
n = 2_000_000
a = np.frombuffer(b"abcd ", dtype=np.uint8)[rng.integers(0, 5, size=(n, 128))]
blob = a.tobytes()
offs = np.arange(n + 1, dtype=np.uint64) * 128
core = bpe.bpe_cython if hasattr(bpe, "bpe_cython") else bpe
for p in (0.0, 0.1, 0.5, 1.0):
    core.encode_packed(blob, offs, dropout_prob=p)
    t0 = time.perf_counter()
    ids, off = core.encode_packed(blob, offs, dropout_prob=p)
    dt = time.perf_counter() - t0
    print("dropout %.1f: %.3f s for %d sentences = %.2e sentences/s, %.1f ids/sentence" % (p, dt, n, n / dt, len(ids) / n), flush=True)
