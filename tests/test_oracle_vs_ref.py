"""Pins oracle/bpe_oracle.c against the UNMODIFIED reference (oracle/_ref, built from /root/reference by
oracle/Makefile).  Train parity target = the -DDETERMINISTIC_QUEUE build; encode parity =
any build; dropout at n_threads=1 in a fresh process is bit-exact.

Skipped when oracle/_ref is absent (e.g. a checkout without /root/reference and without the prebuilt binaries)."""
import filecmp
import os
import random

import pytest

import gen
import oracle_lib as O
import refbin

pytestmark = pytest.mark.skipif(not refbin.available("det"), reason="oracle/_ref not built")


def _cmp_train(tmp_path, text, vocab, coverage=1.0, ids=(0, 1, 2, 3), n_threads=1, name="c"):
    corpus = str(tmp_path / f"{name}.txt")
    with open(corpus, "wb") as f:
        f.write(text)
    m_ref = str(tmp_path / f"{name}.ref.model")
    m_ora = str(tmp_path / f"{name}.ora.model")
    pad, unk, bos, eos = ids
    ref_err = ora_err = None
    try:
        refbin.train(corpus, m_ref, vocab, coverage, n_threads, pad, unk, bos, eos, kind="det")
    except ValueError as e:
        ref_err = str(e)
    try:
        O.train(text, m_ora, vocab, coverage, pad, unk, bos, eos)
    except ValueError as e:
        ora_err = str(e)
    assert ref_err == ora_err
    if ref_err is None:
        assert filecmp.cmp(m_ref, m_ora, shallow=False), f"model files differ for {name}"
    return m_ref if ref_err is None else None


def test_readme_corpus_c1(tmp_path):
    text = gen.readme_corpus()
    _cmp_train(tmp_path, text, 5000, name="c1")


def test_stress_texts(tmp_path):
    rng = random.Random(1234)
    for it in range(120):
        text = gen.stress_text(rng, 1000, True).encode()
        vocab = len(set(text.decode()) | {" "}) + 4 + rng.randint(0, 40)
        cov = 1.0 if rng.randint(0, 1) == 0 else 1 - rng.random() * 0.4
        _cmp_train(tmp_path, text, vocab, cov, n_threads=rng.choice([1, 3, 8]), name=f"s{it}")


def test_unicode_and_special_ids(tmp_path):
    rng = random.Random(99)
    layouts = [(0, 1, 2, 3), (-1, 0, -1, -1), (5, 7, -1, 2), (3, 2, 1, 0), (-1, 3, 1, -1), (0, 40, 29, 35)]
    for it in range(60):
        kind = rng.choice(list(gen.UNICODE_ALPHABETS))
        text = gen.unicode_text(rng, rng.randint(50, 3000), kind)
        cov = rng.choice([1.0, 1.0, 0.9, 0.7, 0.999])
        ids = rng.choice(layouts)
        vocab = rng.randint(45, 120)
        _cmp_train(tmp_path, text, vocab, cov, ids, n_threads=rng.choice([1, 8]), name=f"u{it}")


def test_invalid_utf8_with_coverage(tmp_path):
    # with coverage < 1 removing something, the reference drops invalid bytes (bpe.cpp:357-380)
    rng = random.Random(5)
    for it in range(20):
        text = gen.unicode_text(rng, 2000, "mix", p_invalid=0.05)
        _cmp_train(tmp_path, text, 80, 0.8, name=f"inv{it}")


def test_config_errors(tmp_path):
    text = b"aaa bbb abab"
    for kw in [dict(coverage=0.0), dict(coverage=1.5), dict(ids=(0, 300, 2, 3)), dict(ids=(0, 1, 1, 3)),
               dict(ids=(-2, 1, 2, 3)), dict(vocab=5)]:
        vocab = kw.pop("vocab", 50)
        _cmp_train(tmp_path, text, vocab, kw.get("coverage", 1.0), kw.get("ids", (0, 1, 2, 3)), name="err")


def _sentences(rng, n, alphabet="abcde  "):
    out = []
    for _ in range(n):
        k = rng.randint(0, 60)
        out.append("".join(rng.choice(alphabet) for _ in range(k)).encode())
    return out


def test_encode_ids_vs_ref(tmp_path):
    rng = random.Random(7)
    text = gen.readme_corpus(2000, 100)
    cases = [((0, 1, 2, 3), 800), ((5, 7, -1, 2), 300), ((-1, 0, -1, -1), 200), ((0, 60, 29, 35), 500)]
    for ci, (ids, vocab) in enumerate(cases):
        model = _cmp_train(tmp_path, text, vocab, ids=ids, name=f"e{ci}")
        sents = _sentences(rng, 300) + [b"", b"   ", b"a", b" a ", "жж a ж".encode(), b"\xff\xfea b\x80"]
        lines = str(tmp_path / f"e{ci}.lines")
        # the reference driver reads lines with getline: no newlines inside sentences
        with open(lines, "wb") as f:
            f.write(b"\n".join(sents) + b"\n")
        m = O.Model(model)
        pad, unk, bos_id, eos_id = ids
        for bos, eos, rev in [(0, 0, 0), (1, 1, 0), (0, 0, 1), (1, 1, 1)]:
            if (bos and bos_id == -1) or (eos and eos_id == -1):
                with pytest.raises(ValueError) as e1:
                    refbin.encode(model, lines, 1, bos, eos, rev)
                with pytest.raises(ValueError) as e2:
                    m.encode(sents, bos, eos, rev)
                assert str(e1.value) == str(e2.value)
                continue
            want = refbin.encode(model, lines, 1, bos, eos, rev)
            got = m.encode(sents, bos, eos, rev)
            assert got == want
            want8 = refbin.encode(model, lines, 8, bos, eos, rev)
            assert got == want8


def test_encode_dropout_bit_exact_single_thread(tmp_path):
    rng = random.Random(11)
    text = gen.readme_corpus(2000, 100)
    model = _cmp_train(tmp_path, text, 1000, name="d")
    sents = _sentences(rng, 300, "abcd  ")
    lines = str(tmp_path / "d.lines")
    with open(lines, "wb") as f:
        f.write(b"\n".join(sents) + b"\n")
    m = O.Model(model)
    for p in [0.1, 0.5, 1.0]:
        want = refbin.encode(model, lines, 1, dropout=p)  # fresh process: mt19937 default seed 5489
        O.rng_reset()
        got = m.encode(sents, dropout_prob=p)
        assert got == want
