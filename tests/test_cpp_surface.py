"""The C++ surface of the reference's bpe.h (bpe.h:19-82) on top of the product: shim/cpp/vkcom_adapter.{h,cpp} forward `vkcom::train_bpe`,
`vkcom::BaseEncoder` and the helpers the reference's tests call to the C ABI of libyttm_mi355x.so.  Two reference programs are compiled
against it UNCHANGED by oracle/Makefile (outputs under the git-ignored oracle/_ref/, which travels to the GPU box):

  * oracle/_ref/cyshim/_youtokentome_cython.so -- /root/reference/youtokentome/cpp/yttm.pyx through Cython, the adapter in place of bpe.cpp;
    the reference's own Python package (youtokentome/*.py) beside it.  The reference's unit tests (test_python_api.py, test_cli.py) run on it.
  * oracle/_ref/stress_shim/stress -- /root/reference/tests/unit_tests/stress_test.cpp: `base N` trains on the GPU and compares rules and
    char2id with the reference's brute-force learn_bpe_slow, encodes and compares ids and pieces with decode_slow (asserts abort on a
    difference); `parallel N` = batch encode == sentence by sentence; `manual`.

`-m gpu`: everything on the MI355X.  Without a GPU the stress harness and a smoke run of the binding go through the HIP emulator build
(tests/hipsim) -- the adapter's code is the same."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CYSHIM = os.path.join(ROOT, "oracle", "_ref", "cyshim")
STRESS = os.path.join(ROOT, "oracle", "_ref", "stress_shim", "stress")
SUITE = os.path.join(ROOT, "oracle", "_ref", "unit_tests")
FILES = ("test_python_api.py", "test_cli.py", "utils_for_testing.py")


def _need(path):
    if not os.path.exists(path):
        pytest.skip(os.path.relpath(path, ROOT) + " not built (make -C oracle ref, where /root/reference exists)")


def _stress(args, lib=None, timeout=900):
    _need(STRESS)
    env = dict(os.environ)
    if lib:
        env["YTTM_AMD_LIB"] = lib
    else:
        env.pop("YTTM_AMD_LIB", None)
    r = subprocess.run([STRESS] + args, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, "stress %s failed (rc %d):\n%s" % (" ".join(args), r.returncode, r.stderr[-3000:])
    return r.stderr


def _binding_env(lib=None):
    env = dict(os.environ)
    env["PATH"] = os.path.join(CYSHIM, "bin") + os.pathsep + env.get("PATH", "")
    env["PYTHONPATH"] = os.pathsep.join([CYSHIM, env.get("PYTHONPATH", "")])
    if lib:
        env["YTTM_AMD_LIB"] = lib
    else:
        env.pop("YTTM_AMD_LIB", None)
    return env


SMOKE = r"""
import sys, os, hashlib
import _youtokentome_cython, youtokentome as yttm
assert _youtokentome_cython.__file__.startswith(sys.argv[1]), _youtokentome_cython.__file__   # the reference's binding, not shim/
assert yttm.__file__.startswith(sys.argv[1]), yttm.__file__
corpus, model = sys.argv[2], sys.argv[3]
bpe = yttm.BPE.train(data=corpus, model=model, vocab_size=300, n_threads=1)
sents = [l.rstrip("\n") for l in open(corpus, encoding="utf-8")][:50]
ids = bpe.encode(sents, output_type=yttm.OutputType.ID, bos=True, eos=True)
sub = bpe.encode(sents, output_type=yttm.OutputType.SUBWORD)
assert bpe.decode(ids, ignore_ids=[2, 3]) == [" ".join(s.split()) for s in sents]
assert [bpe.subword_to_id(t) for t in sub[0]] == bpe.encode([sents[0]])[0]
assert len(bpe.vocab()) == bpe.vocab_size() == 300
try:
    yttm.BPE(model="/nonexistent/model")
    raise SystemExit("no error for a missing model")
except ValueError as e:
    assert "Can not open file with model" in str(e), str(e)
try:
    yttm.BPE.train(data=corpus, model=model, vocab_size=5)
    raise SystemExit("no error for a tiny vocabulary")
except ValueError as e:
    assert "Vocabulary size too small" in str(e), str(e)
print("IDS", hashlib.md5(repr(ids).encode()).hexdigest(), "MODEL", hashlib.md5(open(model, "rb").read()).hexdigest())
"""


def _binding_smoke(tmp_path, lib):
    _need(os.path.join(CYSHIM, "_youtokentome_cython.so"))
    import gen
    import oracle_lib as O
    text = gen.readme_corpus(300, 80, "abcde ", seed=3)
    corpus, model = tmp_path / "c.txt", tmp_path / "m.model"
    corpus.write_bytes(text)
    r = subprocess.run([sys.executable, "-c", SMOKE, CYSHIM, str(corpus), str(model)], env=_binding_env(lib), capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    # the same corpus through the oracle: byte-identical model, identical ids
    import hashlib
    m_ora = tmp_path / "ora.model"
    O.train(text, str(m_ora), 300)
    sents = [l for l in text.split(b"\n") if l][:50]
    want = O.Model(str(m_ora)).encode(sents, True, True)
    line = [l for l in r.stdout.splitlines() if l.startswith("IDS")][0].split()
    assert line[3] == hashlib.md5(m_ora.read_bytes()).hexdigest(), "model written through the reference's binding differs from the oracle's"
    assert line[1] == hashlib.md5(repr(want).encode()).hexdigest(), "ids through the reference's binding differ from the oracle's"


def test_stress_harness_on_emulator(sim_lib):
    _stress(["manual"], sim_lib)
    out = _stress(["base", "40"], sim_lib)
    assert out.count("new test") == 40
    _stress(["parallel", "3"], sim_lib)


def test_reference_binding_smoke_on_emulator(tmp_path, sim_lib):
    _binding_smoke(tmp_path, sim_lib)


@pytest.mark.gpu
def test_stress_harness_gpu():
    _stress(["manual"])
    out = _stress(["base", "300"], timeout=1500)  # tests/unit_tests/test_stress.py:34 runs `base 1000`; 300 here for the box's time
    assert out.count("new test") == 300
    _stress(["parallel", "20"], timeout=1500)  # test_stress.py:37: `parallel 50`


@pytest.mark.gpu
def test_reference_binding_smoke_gpu(tmp_path):
    _binding_smoke(tmp_path, None)


@pytest.mark.gpu
def test_reference_unit_tests_through_the_reference_binding(tmp_path):
    """test_python_api.py + test_cli.py, unchanged, on yttm.pyx compiled unchanged: only the C++ below bpe.h is ours."""
    _need(os.path.join(CYSHIM, "_youtokentome_cython.so"))
    if not all(os.path.exists(os.path.join(SUITE, f)) for f in FILES):
        pytest.skip("oracle/_ref/unit_tests not staged")
    work = tmp_path / "unit_tests"
    work.mkdir()
    for f in FILES:
        shutil.copy(os.path.join(SUITE, f), work / f)
    r = subprocess.run([sys.executable, "-m", "pytest", "test_python_api.py", "test_cli.py", "-x", "-q", "-p", "no:cacheprovider"], cwd=str(work),
                       env=_binding_env(None), capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout
