"""Times the Python boundary of an installed `youtokentome` (whichever one PYTHONPATH selects: the reference's Cython module under
oracle/_ref/pyref, or the drop-in's shim/):  BPE(model, n_threads).encode(list[str]) -> list[list[int]]   (yttm.pyx:87-109).
usage: python tools/python_boundary.py MODEL LINES_FILE N_SENTENCES N_THREADS [RUNS]   -> one JSON line"""
import json, sys, time
import youtokentome as yt

model, path, n, nt = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
runs = int(sys.argv[5]) if len(sys.argv) > 5 else 3
sents = []
with open(path, "r") as f:
    for line in f:
        sents.append(line.rstrip("\n"))
        if len(sents) >= n:
            break
bpe = yt.BPE(model, n_threads=nt)
bpe.encode(sents[:1000], output_type=yt.OutputType.ID)
secs, ids = [], 0
for _ in range(runs):
    t0 = time.perf_counter()
    out = bpe.encode(sents, output_type=yt.OutputType.ID)
    secs.append(time.perf_counter() - t0)
    ids = sum(map(len, out))
    del out
secs.sort()
h = 0xcbf29ce484222325
print(json.dumps({"module": yt.__file__, "sentences": len(sents), "ids": ids, "seconds": secs, "median_seconds": secs[len(secs) // 2]}))
