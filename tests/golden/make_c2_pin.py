"""Writes tests/golden/c2_100mb_pin.json: md5 of the 100 MB variant of SURVEY.md's C2 corpus and of the model the unmodified
reference (oracle/_ref/yttm_ref_det: bpe.cpp with -DDETERMINISTIC_QUEUE, n_threads=8) trains on it at vocab_size 32000; the
oracle (oracle/bpe_oracle.c) must give the same file.  Run in the build container (needs /root/reference for oracle/_ref).
usage: python tests/golden/make_c2_pin.py"""
import hashlib
import json
import os
import sys
import tempfile

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, "tests"))
import gen  # noqa: E402
import oracle_lib as O  # noqa: E402
import refbin  # noqa: E402

text = gen.abcd_corpus(100_000_000, seed=19, survey_stream=True)
d = tempfile.mkdtemp()
corpus = os.path.join(d, "c2_100mb.txt")
open(corpus, "wb").write(text)
m_ref, m_ora = os.path.join(d, "ref.model"), os.path.join(d, "ora.model")
assert refbin.available("det")
refbin.train(corpus, m_ref, 32000, n_threads=8, kind="det")
O.train(text, m_ora, 32000, 1.0, 0, 1, 2, 3)
md5 = lambda p: hashlib.md5(open(p, "rb").read()).hexdigest()  # noqa: E731
assert md5(m_ref) == md5(m_ora), "oracle and reference disagree"
pin = {"corpus": "SURVEY.md Appendix C gen_abcd(seed=19), 990 099 rows of 100 chars", "corpus_bytes": len(text),
       "corpus_md5": hashlib.md5(text).hexdigest(), "vocab_size": 32000, "model_md5": md5(m_ref), "model_bytes": os.path.getsize(m_ref),
       "reference": "oracle/_ref/yttm_ref_det (unmodified bpe.cpp, -DDETERMINISTIC_QUEUE), n_threads=8; oracle/bpe_oracle.c gives the same file"}
json.dump(pin, open(os.path.join(R, "tests", "golden", "c2_100mb_pin.json"), "w"), indent=1)
print(pin)
