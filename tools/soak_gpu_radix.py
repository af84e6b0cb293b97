"""GPU soak of K3 by radix partition (k_pairradix.hip): random CJK-shaped corpora -- alphabets of 70 .. 8000 ideographs, 1 .. 24 MB, lexicons of every
size -- counted both ways on the MI355X; the two trainings must write the same model bytes (the general kernel is the path every earlier pin was made
with), and the radix run must say it took the path.  usage: python tools/soak_gpu_radix.py [seconds] [seed]"""
import ctypes as C, json, os, random, sys, tempfile, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import gen
from youtokentome_amd import _lib
L = _lib.load()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
tmp = tempfile.mkdtemp()
t0, n, tokens = time.time(), 0, 0
while time.time() - t0 < budget:
    n_chars = rng.choice([70, 130, 500, 2000, 4096, 8000])
    size = rng.choice([1, 2, 5, 12, 24]) * 1_000_000
    text = gen.cjk_corpus_fast(size, seed=rng.randint(0, 10 ** 6), n_chars=n_chars, lexicon=rng.choice([2000, 50000, 300000]))
    vocab = n_chars + rng.choice([500, 4000, 12000])
    corpus = os.path.join(tmp, "c.txt")
    open(corpus, "wb").write(text)
    models = []
    for radix_min, want in (("0", 1), ("1000000000000", 0)):
        os.environ["YTTM_K3_RADIX_MIN"] = radix_min
        err, rep, model = C.create_string_buffer(2048), C.create_string_buffer(16384), os.path.join(tmp, "m%d.model" % want)
        rc = L.yttm_train_bpe_ex(corpus.encode(), model.encode(), vocab, 1.0, 8, 0, 1, 2, 3, 0, rep, 16384, err, 2048)
        assert rc == 0, (err.value, n_chars, size)
        r = json.loads(rep.value.decode())
        assert r["k3_radix"] == want, (r["k3_radix"], want, n_chars, size)
        models.append(open(model, "rb").read())
    if models[0] != models[1]:
        open(os.path.join(R, "gpurun_out", "FAIL_radix_%d.txt" % n), "wb").write(text)
        raise SystemExit("MISMATCH corpus %d: %d ideographs, %d bytes, vocab %d" % (n, n_chars, size, vocab))
    n += 1
    tokens += len(text) // 3
print("gpu radix soak ok: %d corpora (%.0f M chars) counted both ways, same models, in %.0f s" % (n, tokens / 1e6, time.time() - t0), flush=True)
