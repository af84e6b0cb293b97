"""GPU debugging aid: train the same corpus with the fused round tail and with the separate scan kernel, compare the models."""
import os, sys, subprocess, hashlib
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import gen
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 30
vocab = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
kind = sys.argv[3] if len(sys.argv) > 3 else "abcd"
text = gen.abcd_corpus(mb * 1_000_000, seed=21) if kind == "abcd" else gen.zipf_corpus_fast(mb * 1_000_000, seed=7, vocab=400000)
open("/tmp/fc.txt", "wb").write(text)
code = "import sys; sys.path.insert(0,%r); import youtokentome_amd as y; y.BPE.train('/tmp/fc.txt', sys.argv[1], %d)" % (R, vocab)
out = {}
for tag, env in (("nofuse", {"YTTM_NO_FUSE": "1"}), ("fuse1", {}), ("fuse2", {})):
    m = "/tmp/fc_%s.model" % tag
    subprocess.run([sys.executable, "-c", code, m], env=dict(os.environ, YTTM_DBG_CAND="/tmp/fc_%s.cand" % tag, **env), capture_output=True)
    out[tag] = open(m).read().split("\n")
    print(tag, hashlib.md5("\n".join(out[tag]).encode()).hexdigest(), len(out[tag]))
for tag in ("fuse1", "fuse2"):
    a, b = out["nofuse"], out[tag]
    for i, (x, y) in enumerate(zip(a, b)):
        if x != y:
            print(tag, "first difference at line", i, "nofuse:", x, "|", tag + ":", y)
            break

a = open("/tmp/fc_nofuse.cand").read().split("\n")
b = open("/tmp/fc_fuse2.cand").read().split("\n")
import re
strip = lambda l: re.sub(r"fused=\d ", "", l)
for i, (x, y) in enumerate(zip(a, b)):
    if strip(x) != strip(y):
        print("first differing scan line", i)
        for j in range(max(0, i - 3), min(len(a), i + 3)):
            print("  nofuse:", a[j]); print("  fuse2 :", b[j] if j < len(b) else None)
        break
