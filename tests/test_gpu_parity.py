"""Parity tests proper: the HIP path on a real MI355X, through the C ABI, against the oracle (bit exact), the committed
golden fixtures, the unmodified reference binaries (oracle/_ref, when present) and size-independent properties."""
import filecmp
import os
import random

import numpy as np
import pytest

import gen
import oracle_lib as O
import refbin
import stage_checks as S

pytestmark = pytest.mark.gpu


def test_device_is_gfx950():
    from youtokentome_amd import _lib
    rc, info = _lib.device_info(0)
    assert rc == 0, info
    assert "gfx950" in info, info


def test_k1_char_hist():
    for t in S.texts_small(0, n=8, size=20000) + [gen.readme_corpus(3000, 100), gen.zipf_corpus(300000, vocab=5000)]:
        S.check_char_hist(t)


def test_front_end_three_byte_fast_path(monkeypatch):
    """The same check as tests/test_sim_stages.py on the MI355X, larger texts (hundreds of workgroups; the wide-char K1 variant picked by
    the sample of the text as well as forced either way)."""
    rng = random.Random(6)
    for n in (4097, 70000, 300000, 1200000):
        t = S.three_byte_text(rng, n)
        S.check_char_hist(t)
        for wide in ("0", "1"):
            monkeypatch.setenv("YTTM_K1_WIDE", wide)
            S.check_char_hist(t)
        monkeypatch.delenv("YTTM_K1_WIDE")
        if n <= 300000 and t.strip():
            S.check_word_table_and_pairs(t)


def test_k2_k3_word_table_and_pair_count():
    for i, t in enumerate(S.texts_small(1, n=8, size=20000)):
        S.check_word_table_and_pairs(t, coverage=1.0 if i % 2 == 0 else 0.9)
    S.check_word_table_and_pairs(gen.readme_corpus(3000, 100, seed=3))
    S.check_word_table_and_pairs(gen.zipf_corpus(400000, vocab=8000))
    for t in S.texts_by_alphabet_size(n_words=5000):  # K3's kernels: up to 32 symbols, up to 64, beyond
        S.check_word_table_and_pairs(t)


def test_k3_radix_partition(tmp_path, monkeypatch):
    """K3 of large alphabets by two-level radix partition (k_pairradix.hip, round 6) against the oracle's count: forced on at small sizes
    (YTTM_K3_RADIX_MIN=0) on alphabets of 66 .. 5000 symbols with runs of one symbol, then 4 MB of the CJK-shaped corpus -- thousands of
    workgroups, every level-1 group, chunks that span several first tokens -- and the same text counted by the general kernel."""
    import ctypes as C
    import json
    from youtokentome_amd import _lib
    monkeypatch.setenv("YTTM_K3_RADIX_MIN", "0")
    for t in S.texts_by_alphabet_size(sizes=(66, 130, 300, 1500, 5000), n_words=3000):
        S.check_word_table_and_pairs(t)
    big = gen.cjk_corpus_fast(4_000_000, seed=3)
    S.check_word_table_and_pairs(big)
    S.check_word_table_and_pairs(gen.cjk_corpus_fast(1_000_000, seed=4), coverage=0.95)
    L = _lib.load()
    corpus = str(tmp_path / "c.txt")
    open(corpus, "wb").write(big)
    models = []
    for radix_min, want in (("0", 1), ("1000000000000", 0)):
        monkeypatch.setenv("YTTM_K3_RADIX_MIN", radix_min)
        err, rep, model = C.create_string_buffer(2048), C.create_string_buffer(16384), str(tmp_path / ("m%d.model" % want))
        assert L.yttm_train_bpe_ex(corpus.encode(), model.encode(), 6000, 1.0, 8, 0, 1, 2, 3, 0, rep, 16384, err, 2048) == 0, err.value
        assert json.loads(rep.value.decode())["k3_radix"] == want
        models.append(open(model, "rb").read())
    assert models[0] == models[1]


def test_k5_dropout_heap_equals_array():
    S.check_dropout_heap_equals_array()


def test_front_end_under_the_upload(tmp_path, monkeypatch):
    monkeypatch.setenv("YTTM_FE_OVERLAP_MIN", "0")
    monkeypatch.setenv("YTTM_FE_PART_KB", "4")
    monkeypatch.setenv("YTTM_IO_CHUNK_KB", "4")
    S.check_front_end_under_upload(tmp_path, rounds=24)


def test_k5_word_cache():
    S.check_encode_word_cache(n_sent=2000)


def test_k5_word_cache_fuzz(tmp_path):
    S.check_encode_word_cache_fuzz(tmp_path, trials=12)


def test_k5_host_to_host_in_sub_batches(monkeypatch):
    """a very large host -> host batch goes through both lanes in sub-batches, up / encode / down at once (host_encoder.cpp encode_pipelined):
    here every batch, in sub-batches of 1 KB, arrays through 4 KB chunks"""
    monkeypatch.setenv("YTTM_ENC_PIPE_FROM", "1")
    monkeypatch.setenv("YTTM_ENC_SUB_KB", "1")
    monkeypatch.setenv("YTTM_IO_CHUNK_KB", "4")
    S.check_encode_mixed_shapes(n_sent=2000, seed=67)
    S.check_encode_word_cache(n_sent=500, seed=71)


def test_k5_host_arrays_through_pinned_chunks(monkeypatch):
    """host -> host encode of a large batch moves its arrays through the trainer's pinned chunks, several threads at once (host_encoder.cpp
    copy_up / copy_down, gpu_ctx.cpp staged_transfer): here every array of small batches, in chunks of 4 KB"""
    monkeypatch.setenv("YTTM_ENC_STAGED_FROM", "1")
    monkeypatch.setenv("YTTM_IO_CHUNK_KB", "4")
    S.check_encode_mixed_shapes(n_sent=2000, seed=53)
    S.check_encode_word_cache(n_sent=500, seed=59)


@pytest.mark.parametrize("sblk", [3, 64])
def test_k5_word_cache_sentence_blocks(sblk, tmp_path, monkeypatch):
    """the word cache's three walks over the text (k_wcache.hip: insert, count, scatter) take blocks of consecutive sentences as one run --
    64 per wavefront in large batches, one in small ones like these tests' unless told otherwise: sentences of every length, empty ones,
    ones that begin and end inside UTF-8 sequences, against the oracle"""
    monkeypatch.setenv("YTTM_WC_SBLK", str(sblk))
    S.check_encode_word_cache(n_sent=1500, seed=43)
    S.check_encode_word_cache_fuzz(tmp_path, trials=8, seed=47)


def test_k5_word_cache_crowded_short_region(tmp_path, monkeypatch):
    """more distinct short words than the table's short region has slots (k_wcache.hip WC_SHORT_PROBES)"""
    monkeypatch.setenv("YTTM_WC_SHORT_SLOTS", "16")
    S.check_encode_word_cache(n_sent=1000, seed=37)
    S.check_encode_word_cache_fuzz(tmp_path, trials=4, seed=41)


@pytest.mark.parametrize("lane_max", [0, 5, 1000])
def test_k5_one_word_per_lane(lane_max, monkeypatch):
    """the wave-wide merge rounds and merge_lanes (k_encode.hip) against the oracle: lanes never / for short words only / always"""
    monkeypatch.setenv("YTTM_K5_LANE_WORDS", str(lane_max))
    monkeypatch.setenv("YTTM_K5_LANE_SENT", str(lane_max))
    S.check_encode_mixed_shapes(n_sent=1500, seed=29)
    S.check_encode_word_cache(n_sent=1000, seed=31)


def test_k4_many_words_per_tile(tmp_path):
    S.check_many_words_per_tile(tmp_path)


def test_k4_merge_apply_rounds():
    for i, t in enumerate(S.texts_small(2, n=6, size=8000)):
        if t.strip():
            S.check_merge_rounds(t, rounds=8, seed=i)
    S.check_merge_rounds(gen.readme_corpus(1500, 100, seed=9), rounds=25, seed=3)
    t = ("aaaa aaaaa aaaaaaa abababab aabbaabb abcabcabc bbbbbb ab aaab baaa " + "a" * 700 + " " + "ab" * 500 + " ") * 3
    S.check_merge_rounds(t.encode(), rounds=14, seed=1)
    S.check_site_placements(trials=200, seed=9)  # a rare pair at many positions of its tile, sites 2 and 3 apart, runs next to a site
    # ids >= 32768: flags from the HBM table instead of the LDS bitmap
    for i, t in enumerate(S.texts_small(5, n=3, size=8000)):
        if t.strip():
            S.check_merge_rounds(t, rounds=8, seed=i, id_shift=40000)
    # more than 64 / 128 merge sites in one class-A tile (phase 2 takes them 64 per pass)
    words = ["ab" * k for k in range(60, 125, 7)] + ["a" * k for k in range(150, 250, 13)] + ["abc" * k for k in (50, 70, 80)]
    S.check_merge_rounds((" ".join(words) + " ").encode(), rounds=8, seed=4)


def test_k4_measurement_pass():
    """The instrument behind bench.py's roofline.algorithmic_bytes_8d (words that held a merge site and their tokens, SURVEY.md 8d:
    W_touched / T_touched) against a count on the oracle's word table -- on the MI355X, not only on the emulator."""
    for i, t in enumerate(S.texts_small(3, n=3, size=8000) + [gen.readme_corpus(1500, 100, seed=4)]):
        if t.strip():
            S.check_k4_measure(t, rounds=10, seed=i)
    words = ["ab" * k for k in range(60, 125, 7)] + ["a" * k for k in range(150, 250, 13)]
    S.check_k4_measure((" ".join(words) + " ").encode(), rounds=6, seed=1)


def test_k4_word_mode(tmp_path, monkeypatch):
    """Word mode (one launch per round, k_words<FUSED>, and k_wgather + k_words + k_delta_apply) forced on from the second round whatever the corpus size: golden
    corpora, random text of three scripts and a Zipf corpus against the oracle; then with tiny hot lists (index rebuilds, rounds over
    every word), overflowing record regions and record log; then the 100 MB pins of configs[1] and of the CJK-shaped corpus with the
    default switch rule (only the size floor lifted), single GPU and through an RCCL communicator of one rank."""
    import ctypes as C
    import hashlib
    import json
    from youtokentome_amd import _lib
    L = _lib.load()
    for k in ("YTTM_WORD_MIN_TILES", "YTTM_WORD_MIN_TOKENS", "YTTM_WORD_DIV"):
        monkeypatch.setenv(k, "0")
    rng = random.Random(5)
    for cfg in (None, {"YTTM_WORDS_FUSE_MAX": 0}, {"YTTM_HOT_TARGET": 40, "YTTM_HOT_MIN": 4, "YTTM_HOT_CAP": 400, "YTTM_WORD_DREC": 64, "YTTM_INDEX_AGG_MIN": 0},
                {"YTTM_WORD_LOG": 3000, "YTTM_WORDS_INLINE_MAX": 0, "YTTM_WORDS_FUSE_MAX": 0}, {"YTTM_WORD_LOG": 3000, "YTTM_WORD_DREC": 16, "YTTM_WORDS_GRID": 3}):
        for k, v in (cfg or {}).items():
            monkeypatch.setenv(k, str(v))
        for name in ("readme_small", "runs", "mix_cov", "zipf"):
            S.check_golden_train(name, tmp_path)
        S.check_train_vs_oracle(gen.zipf_corpus(2_000_000, seed=3, vocab=30000), 4000, tmp_path, tag="wm")
        for kind in ("ascii", "cyr", "cjk"):
            S.check_train_vs_oracle(gen.unicode_text(rng, 300000, kind), 1500, tmp_path, tag="wm" + kind)
        S.check_train_vs_oracle(gen.readme_corpus(3000, 100, seed=2), 3000, tmp_path, tag="wmr")
        for k in (cfg or {}):
            monkeypatch.delenv(k)
    monkeypatch.setenv("YTTM_WORD_DIV", "200")
    pins = _full_pins()
    err, rep = C.create_string_buffer(2048), C.create_string_buffer(16384)
    idbuf = (C.c_uint8 * 128)()
    assert L.yttm_comm_rccl_unique_id(idbuf) == 0
    comm = C.c_void_p()
    assert L.yttm_comm_rccl_create(idbuf, 0, 1, 0, C.byref(comm)) == 0
    try:
        for pin_name, text in (("c2_100mb", gen.abcd_corpus(100_000_000, seed=19, survey_stream=True)), ("c6_cjk_100mb", gen.cjk_corpus_fast(100_000_000, seed=11))):
            pin = pins[pin_name]
            text = text[:pin["corpus_bytes"]]
            assert hashlib.md5(text).hexdigest() == pin["corpus_md5"], pin_name
            corpus, model = str(tmp_path / "wm.txt"), str(tmp_path / "wm.model")
            open(corpus, "wb").write(text)
            rc = L.yttm_train_bpe_ex(corpus.encode(), model.encode(), 32000, 1.0, 1, 0, 1, 2, 3, 0, rep, 16384, err, 2048)
            assert rc == 0, err.value
            r = json.loads(rep.value.decode())
            assert r["word_rounds"] > 100 and r["word_all_rounds"] == 0, (pin_name, r["word_rounds"], r["word_all_rounds"])
            assert hashlib.md5(open(model, "rb").read()).hexdigest() == pin["model_md5"], pin_name
            rc = L.yttm_train_bpe_from_memory_comm(text, len(text), model.encode(), 32000, 1.0, 0, 1, 2, 3, 0, comm, rep, 16384, err, 2048)
            assert rc == 0, err.value
            assert json.loads(rep.value.decode())["word_rounds"] > 100
            assert hashlib.md5(open(model, "rb").read()).hexdigest() == pin["model_md5"], pin_name + " (RCCL)"
    finally:
        L.yttm_comm_destroy(comm)


@pytest.mark.parametrize("name", S.golden_train_names())
def test_golden_train(name, tmp_path):
    S.check_golden_train(name, tmp_path)


@pytest.mark.parametrize("name", S.golden_encode_names())
def test_golden_encode(name):
    S.check_golden_encode(name)


def test_train_encode_vs_oracle_random(tmp_path):
    rng = random.Random(5)
    layouts = [(0, 1, 2, 3), (-1, 0, -1, -1), (5, 7, -1, 2), (3, 2, 1, 0), (0, 40, 29, 35)]
    for it in range(30):
        kind = rng.choice(list(gen.UNICODE_ALPHABETS))
        inv = it % 3 == 0
        text = gen.unicode_text(rng, rng.randint(200, 20000), kind, p_invalid=0.02 if inv else 0.0)
        cov = rng.choice([0.9, 0.7, 0.99]) if inv else rng.choice([1.0, 1.0, 0.95])
        model = S.check_train_vs_oracle(text, rng.randint(45, 300), tmp_path, cov, layouts[it % 5], tag=f"r{it}")
        if model:
            sents = [gen.unicode_text(rng, rng.randint(0, 200), kind, p_invalid=0.01).decode(errors="ignore").replace("\n", " ")
                     for _ in range(100)] + ["", "  "]
            S.check_encode_vs_oracle(model, sents)


def test_stress_texts_vs_oracle(tmp_path):
    rng = random.Random(77)
    for it in range(60):
        text = gen.stress_text(rng, 1000, True).encode()
        vocab = len(set(text.decode()) | {" "}) + 4 + rng.randint(0, 40)
        cov = 1.0 if rng.randint(0, 1) == 0 else 1 - rng.random() * 0.4
        model = S.check_train_vs_oracle(text, vocab, tmp_path, cov, tag=f"s{it}")
        if model:
            S.check_encode_vs_oracle(model, [gen.stress_text(rng, 1000, False) for _ in range(4)], flags=((0, 0, 0),))


def test_encode_mixed_shapes():
    S.check_encode_mixed_shapes(n_sent=3000)


def test_config_errors(tmp_path):
    for kw in [dict(coverage=0.0), dict(coverage=1.5), dict(ids=(0, 300, 2, 3)), dict(ids=(0, 1, 1, 3)), dict(ids=(-2, 1, 2, 3)),
               dict(vocab=5)]:
        S.check_train_vs_oracle(b"aaa bbb abab", kw.get("vocab", 50), tmp_path, kw.get("coverage", 1.0), kw.get("ids", (0, 1, 2, 3)), tag="e")


def test_c1_readme_corpus_full(tmp_path):
    """BASELINE.json configs[0]: 10k lines x 100 chars over 'abcd ', vocab 5000 (tests/unit_tests/utils_for_testing.py:23-36)."""
    text = gen.readme_corpus()
    model = S.check_train_vs_oracle(text, 5000, tmp_path, tag="c1")
    sents = [ln.decode() for ln in gen.readme_corpus(3000, 100, "abcde ", seed=4).split(b"\n") if ln]
    S.check_encode_vs_oracle(model, sents)


def test_zipf_corpus_vs_oracle(tmp_path):
    text = gen.zipf_corpus(3_000_000, vocab=30000)
    model = S.check_train_vs_oracle(text, 6000, tmp_path, tag="z")
    sents = [ln.decode() for ln in gen.zipf_corpus(200000, seed=11, vocab=30000).split(b"\n") if ln]
    S.check_encode_vs_oracle(model, sents, flags=((0, 0, 0), (1, 1, 1)))


def test_vocab_above_32768_vs_oracle(tmp_path):
    """vocab_size 40000: token ids leave the kernels' 32768-id LDS flag bitmap (flags from the HBM table, batches uploaded
    instead of passed as kernel arguments); the encoder works with 40000 rules."""
    text = gen.zipf_corpus(6_000_000, seed=5, vocab=150000)
    model = S.check_train_vs_oracle(text, 40000, tmp_path, tag="v40k")
    sents = [ln.decode() for ln in gen.zipf_corpus(100000, seed=12, vocab=150000).split(b"\n") if ln]
    S.check_encode_vs_oracle(model, sents, flags=((0, 0, 0), (1, 1, 1)))


def test_long_words_and_long_sentences(tmp_path):
    rng = random.Random(3)
    words = ["".join(rng.choice("abc") for _ in range(n)) for n in (2046, 1500, 1024, 700, 65, 64, 63, 1)]
    text = (" ".join(words) + "\n") * 3 + gen.readme_corpus(50, 80).decode()
    model = S.check_train_vs_oracle(text.encode(), 200, tmp_path, tag="lw")
    # sentences longer than the LDS budget of the encode kernel take its HBM-scratch path
    sents = [" ".join(words), "ab" * 3000, ("abc " * 2000).strip(), "a"]
    S.check_encode_vs_oracle(model, sents)


def test_very_long_words(tmp_path):
    S.check_very_long_words(tmp_path, lengths=(2047, 2048, 2049, 5000, 20000, 70000))


@pytest.mark.skipif(not refbin.available("det"), reason="oracle/_ref not present")
def test_medium_corpus_vs_reference_binary(tmp_path):
    """30 MB 'abcd ' corpus, vocab 8000: too big for the oracle's comfort, checked against the unmodified reference
    (-DDETERMINISTIC_QUEUE) running on the host cores."""
    import youtokentome_amd as yttm
    text = gen.abcd_corpus(30_000_000, seed=21)
    corpus = str(tmp_path / "m.txt")
    open(corpus, "wb").write(text)
    m_gpu, m_ref = str(tmp_path / "gpu.model"), str(tmp_path / "ref.model")
    yttm.BPE.train(corpus, m_gpu, 8000)
    refbin.train(corpus, m_ref, 8000, n_threads=8, kind="det")
    assert filecmp.cmp(m_gpu, m_ref, shallow=False)
    # encode 200k sentences: ids identical to the reference encoder
    lines = str(tmp_path / "enc.txt")
    sent_bytes = gen.abcd_corpus(200_000 * 129, seed=5, line=128)
    open(lines, "wb").write(sent_bytes)
    want = refbin.encode_bench(m_ref, lines, n_threads=8)
    bpe = yttm.BPE(m_gpu)
    offs = np.arange(200_001, dtype=np.uint64) * 129
    ids, off = bpe.bpe_cython.encode_packed(sent_bytes, offs)
    assert len(ids) == want["ids"]
    import ctypes as C
    from youtokentome_amd import _lib
    ids = np.ascontiguousarray(ids, np.int32)
    off = np.ascontiguousarray(off, np.uint64)
    got = _lib.load().yttm_ids_fnv1a64(ids.ctypes.data_as(_lib.i32p), off.ctypes.data_as(_lib.u64p), len(off) - 1)
    assert "%016x" % got == want["fnv1a64"]


def test_encode_properties_at_scale(tmp_path):
    """Size-independent properties on 1M sentences: batch == sum of chunks, decode(encode(x)) == normalised x,
    reverse == reversed, bos/eos only add the two ids."""
    import youtokentome_amd as yttm
    text = gen.readme_corpus(5000, 100)
    corpus = str(tmp_path / "p.txt")
    open(corpus, "wb").write(text)
    bpe = yttm.BPE.train(corpus, str(tmp_path / "p.model"), 3000)
    n = 1_000_000
    blob = gen.abcd_corpus(n * 65, seed=8, line=64)
    offs = np.arange(n + 1, dtype=np.uint64) * 65
    core = bpe.bpe_cython
    ids, off = core.encode_packed(blob, offs)
    ids2, off2 = core.encode_packed(blob, offs, bos=True, eos=True, reverse=True)
    assert len(ids2) == len(ids) + 2 * n
    k = 200_000
    part_ids, part_off = core.encode_packed(blob[: k * 65], offs[: k + 1])
    assert np.array_equal(part_ids, ids[: int(off[k])]) and np.array_equal(part_off, off[: k + 1])
    for i in (0, 1, 999_999, 123_456):
        a, b = int(off[i]), int(off[i + 1])
        a2, b2 = int(off2[i]), int(off2[i + 1])
        assert ids2[a2:b2].tolist() == [3] + ids[a:b].tolist()[::-1] + [2]
        sent = blob[i * 65:(i + 1) * 65].decode()
        assert bpe.decode([ids[a:b].tolist()])[0] == " ".join(sent.split())


def test_dropout_extremes_and_roundtrip():
    model = os.path.join(S.G, "train_readme_small.model")
    rng = random.Random(4)
    sents = ["".join(rng.choice("abcd  ") for _ in range(rng.randint(0, 200))) for _ in range(2000)] + ["", " ", "abcd" * 400]
    S.check_dropout_extremes(model, sents)


@pytest.mark.parametrize("p", [0.1, 0.5])
def test_dropout_distribution_vs_reference_semantics(p, tmp_path):
    """BASELINE.json configs[4] (scaled to 200k sentences): distribution match against the oracle's bit-exact emulation of
    the reference at n_threads=1."""
    import youtokentome_amd as yttm
    text = gen.readme_corpus(4000, 100)
    corpus = str(tmp_path / "d.txt")
    open(corpus, "wb").write(text)
    model = str(tmp_path / "d.model")
    yttm.BPE.train(corpus, model, 2000)
    sents = [ln.decode() for ln in gen.abcd_corpus(200_000 * 129, seed=77, line=128).split(b"\n") if ln]
    mean_g, mean_w, ks, chi = S.check_dropout_distribution(model, sents, p, 2000)
    print(f"dropout p={p}: ids/sentence gpu {mean_g:.3f} oracle {mean_w:.3f} KS {ks:.5f} chi2/dof {chi:.3f}")


def test_zz_c2_100mb_model_pin(tmp_path):
    """BASELINE.json configs[1] at a tenth of its size: the 100 MB variant of SURVEY.md's C2 file (byte-identical: md5
    f35ed066...), vocab 32000.  The model must be the one the unmodified reference (-DDETERMINISTIC_QUEUE, n_threads=8)
    and the oracle both produce: md5 222ef3e6..., pinned by tests/golden/c2_100mb_pin.json (made by tests/golden/make_c2_pin.py
    in the build container, where /root/reference exists).  Last in the file: it takes the longest."""
    import hashlib
    import json
    import youtokentome_amd as yttm
    pin = json.load(open(os.path.join(S.G, "c2_100mb_pin.json")))
    text = gen.abcd_corpus(pin["corpus_bytes"] + 1, seed=19, survey_stream=True)
    assert len(text) == pin["corpus_bytes"] and hashlib.md5(text).hexdigest() == pin["corpus_md5"]
    corpus = str(tmp_path / "c2.txt")
    open(corpus, "wb").write(text)
    model = str(tmp_path / "c2.model")
    yttm.BPE.train(corpus, model, pin["vocab_size"])
    assert hashlib.md5(open(model, "rb").read()).hexdigest() == pin["model_md5"]


def _full_pins():
    import json
    return json.load(open(os.path.join(S.G, "full_size_pins.json")))


def test_zz_c3_100mb_zipf_model_pin(tmp_path):
    """BASELINE.json configs[2] at a tenth of its size: 100 MB of the Zipf corpus bench.py uses (gen.zipf_corpus_fast), vocab
    32000 -- thousands of short merge rounds.  Model md5 as the unmodified reference (det queue, n_threads=8) writes it
    (tests/golden/full_size_pins.json, made by tests/golden/make_full_pins.py)."""
    import hashlib
    import youtokentome_amd as yttm
    pin = _full_pins()["c3_100mb"]
    text = gen.zipf_corpus_fast(100_000_000, seed=7, vocab=400000)
    assert len(text) == pin["corpus_bytes"] and hashlib.md5(text).hexdigest() == pin["corpus_md5"]
    corpus = str(tmp_path / "c3.txt")
    open(corpus, "wb").write(text)
    model = str(tmp_path / "c3.model")
    yttm.BPE.train(corpus, model, pin["vocab_size"])
    assert hashlib.md5(open(model, "rb").read()).hexdigest() == pin["model_md5"]


@pytest.mark.skipif(os.environ.get("YTTM_FULL_PINS") == "0", reason="YTTM_FULL_PINS=0: a quick local run without the minute of corpus generation (the driver's run never sets it)")
def test_zz_full_size_pins(tmp_path):
    """BASELINE.json configs[1] and configs[3] at FULL size, in the test-suite (they lived only in bench.py): the 1 GB `abcd `
    corpus (md5 63857720...) -> vocab 32000 -> the model the unmodified reference writes (md5 73e74d66..., tests/golden/full_size_pins.json
    c2_1gb), through the file path (front end under the upload) AND from memory; then all 10 M sentences of C4 encoded with that model:
    FNV-1a-64 of every (length, ids) against the reference's encode_as_ids (pin c4_10m)."""
    import ctypes as C
    import hashlib
    import numpy as np
    import youtokentome_amd as yttm
    from youtokentome_amd import _lib
    pins = _full_pins()
    pin = pins["c2_1gb"]
    text = gen.abcd_corpus(pin["corpus_bytes"] + 1, seed=19, survey_stream=True)
    assert len(text) == pin["corpus_bytes"] and hashlib.md5(text).hexdigest() == pin["corpus_md5"]
    corpus, model = str(tmp_path / "c2.txt"), str(tmp_path / "c2.model")
    open(corpus, "wb").write(text)
    yttm.BPE.train(corpus, model, pin["vocab_size"])
    assert hashlib.md5(open(model, "rb").read()).hexdigest() == pin["model_md5"]
    L = _lib.load()
    err, model2 = C.create_string_buffer(_lib.ERRLEN), str(tmp_path / "c2m.model")
    assert L.yttm_train_bpe_from_memory(text, len(text), model2.encode(), pin["vocab_size"], 1.0, 0, 1, 2, 3, 0, None, 0, err, _lib.ERRLEN) == 0, err.value
    assert hashlib.md5(open(model2, "rb").read()).hexdigest() == pin["model_md5"]
    del text
    os.remove(corpus)
    p4 = pins["c4_10m"]
    line, n = 128, p4["n_sentences"]
    sents = gen.abcd_corpus(n * (line + 1), seed=123, line=line, survey_stream=True)
    assert hashlib.md5(sents).hexdigest() == p4["input_md5"]
    off_in = (np.arange(n + 1, dtype=np.uint64) * np.uint64(line + 1))
    h = C.c_void_p()
    assert L.yttm_encoder_create(model.encode(), 1, 0, C.byref(h), err, _lib.ERRLEN) == 0, err.value
    ids, off = _lib.i32p(), _lib.u64p()
    # (sentence i = bytes [off[i], off[i+1]): the newline rides at the end of each, white space like any other)
    assert L.yttm_encode_as_ids(h, sents, off_in.ctypes.data_as(_lib.u64p), n, 0, 0, 0, 0.0, C.byref(ids), C.byref(off), err, _lib.ERRLEN) == 0, err.value
    assert int(off[n]) == p4["n_ids"]
    assert "%016x" % L.yttm_ids_fnv1a64(ids, off, n) == p4["fnv1a64"]
    L.yttm_free(ids)
    L.yttm_free(off)
    L.yttm_encoder_destroy(h)


@pytest.mark.parametrize("name,chunk_mb", [("c2_100mb", "16"), ("c3_100mb", "8")])
def test_zz_chunked_front_end_100mb_pins(name, chunk_mb, tmp_path, monkeypatch):
    """The chunked front end (gpu_ctx.cpp front_end_chunked) at a size where chunks are thousands of workgroups: the 100 MB variants of
    configs[1] / configs[2] in chunks of 16 / 8 MB -- the lexicon and the word table grow on the way -- must give the reference's models."""
    import ctypes as C
    import hashlib
    import json
    from youtokentome_amd import _lib
    pin = _full_pins()[name]
    text = gen.abcd_corpus(pin["corpus_bytes"] + 1, seed=19, survey_stream=True) if name == "c2_100mb" else gen.zipf_corpus_fast(100_000_000, seed=7, vocab=400000)
    assert hashlib.md5(text).hexdigest() == pin["corpus_md5"]
    corpus, model = str(tmp_path / "c.txt"), str(tmp_path / "c.model")
    open(corpus, "wb").write(text)
    monkeypatch.setenv("YTTM_FE_CHUNK_MB", chunk_mb)
    L = _lib.load()
    err, rep = C.create_string_buffer(2048), C.create_string_buffer(16384)
    assert L.yttm_train_bpe_ex(corpus.encode(), model.encode(), pin["vocab_size"], 1.0, 8, 0, 1, 2, 3, 0, rep, 16384, err, 2048) == 0, err.value
    r = json.loads(rep.value.decode())
    assert r["front_end_chunks"] >= 100 // int(chunk_mb) and r["front_end_overlapped"] == 1
    assert hashlib.md5(open(model, "rb").read()).hexdigest() == pin["model_md5"]


def test_zz_rccl_world_of_one(tmp_path):
    """The RCCL transport of the multi-GPU path on the one GPU there is: a communicator of size 1 still runs every collective
    of a round (ncclAllReduce of the char histogram and of the hot-list verdict, the grouped send/recv all-gather after K3,
    ncclAllGather of the delta blocks) and the block fold -- the model must not change.  (N>1 logic: tests/test_multi_rank_gloo.py.)"""
    import ctypes as C
    import hashlib
    from youtokentome_amd import _lib
    L = _lib.load()
    idbuf = (C.c_uint8 * 128)()
    assert L.yttm_comm_rccl_unique_id(idbuf) == 0
    comm = C.c_void_p()
    assert L.yttm_comm_rccl_create(idbuf, 0, 1, 0, C.byref(comm)) == 0
    err, rep = C.create_string_buffer(2048), C.create_string_buffer(16384)
    try:
        for name, text, vocab in (("readme", gen.readme_corpus(2000, 100, seed=11), 2000), ("zipf", gen.zipf_corpus(3_000_000, vocab=20000), 6000)):
            m_gpu, m_ora = str(tmp_path / f"{name}.rccl.model"), str(tmp_path / f"{name}.ora.model")
            rc = L.yttm_train_bpe_from_memory_comm(text, len(text), m_gpu.encode(), vocab, 1.0, 0, 1, 2, 3, 0, comm, rep, 16384, err, 2048)
            assert rc == 0, err.value.decode()
            O.train(text, m_ora, vocab)
            assert filecmp.cmp(m_gpu, m_ora, shallow=False), name
        # and at a tenth of the headline size, against the reference's pin
        pin = _full_pins()["c2_100mb"]
        text = gen.abcd_corpus(pin["corpus_bytes"] + 1, seed=19, survey_stream=True)
        m_gpu = str(tmp_path / "c2.rccl.model")
        rc = L.yttm_train_bpe_from_memory_comm(text, len(text), m_gpu.encode(), 32000, 1.0, 0, 1, 2, 3, 0, comm, rep, 16384, err, 2048)
        assert rc == 0, err.value.decode()
        assert hashlib.md5(open(m_gpu, "rb").read()).hexdigest() == pin["model_md5"]
    finally:
        L.yttm_comm_destroy(comm)


@pytest.mark.parametrize("use_comm,pin_name", [(0, "c2_100mb"), (1, "c2_100mb"), (0, "c6_cjk_100mb")])
def test_zz_fused_tail_ordering(tmp_path, use_comm, pin_name):
    """The round's candidate scan rides in the tail of the round's last kernel: the LAST workgroup to take its ticket reads what every
    other workgroup -- on other XCDs -- published before taking its own (device-scope atomics, write-through stores, a workgroup-scope
    release + s_waitcnt vmcnt(0); k_tiles.hip / k_words.hip / k_pairtable.hip k_fold_list).  The emulator cannot show that ordering; this does: the
    100 MB variant of configs[1] (1 500 workgroups per launch in the tile rounds, word mode after them) trained with the fused tail and
    with the scan as a kernel of its own (YTTM_NO_FUSE=1: ordered by a kernel boundary), ten times (four through the communicator): the candidate traces
    (YTTM_DBG_CAND: one line per scan -- thresholds, list lengths, key count, a hash of the candidates) must agree line by line and the
    models must be the reference's pin.  use_comm=1: through an RCCL communicator of one rank -- the multi-GPU round, whose scan sits in
    the fold kernel's tail behind the all-gather.  c6_cjk_100mb (round 5): long clauses -- class-B tiles, whose launch precedes k_words in every
    word-mode round (two kernels before the tail), and a large alphabet."""
    import hashlib
    import re
    import subprocess
    import sys
    pin = _full_pins()[pin_name]
    text = gen.abcd_corpus(pin["corpus_bytes"] + 1, seed=19, survey_stream=True) if pin_name == "c2_100mb" else gen.cjk_corpus_fast(100_000_000, seed=11)[:pin["corpus_bytes"]]
    assert hashlib.md5(text).hexdigest() == pin["corpus_md5"]
    corpus = str(tmp_path / "c2.txt")
    open(corpus, "wb").write(text)
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpu_train_worker.py")

    def run(tag, env):
        model, trace = str(tmp_path / (tag + ".model")), str(tmp_path / (tag + ".cand"))
        r = subprocess.run([sys.executable, worker, corpus, model, "32000", str(use_comm)], env=dict(os.environ, YTTM_DBG_CAND=trace, **env), capture_output=True, text=True)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        assert hashlib.md5(open(model, "rb").read()).hexdigest() == pin["model_md5"], tag
        lines = [re.sub(r"fused=\d ", "", ln) for ln in open(trace).read().split("\n") if ln]
        return lines, json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("REPORT ")][-1][7:])
    import json
    # (YTTM_NO_REFINE=1 on both sides: the fused scan otherwise raises the host's threshold by itself -- fewer candidates travel, same batches --
    # and the traces would differ in their candidate counts for that reason alone)
    # YTTM_WORD_MIN_TOKENS=0: word mode (and its one-launch rounds) also at a tenth of the headline size, where a pass over the tiles is
    # still cheap enough for the library to stay on them by itself
    # YTTM_NO_BATCH_SPLIT=1: a batch of up to twice what the kernel arguments hold is cut in two only where rounds are ONE launch
    # (host_trainer.cpp) -- with the scan as a kernel of its own they never are, and the two sides would not have the same rounds
    hooks = {"YTTM_NO_REFINE": "1", "YTTM_WORD_MIN_TOKENS": "0", "YTTM_NO_BATCH_SPLIT": "1"}
    ref, rep0 = run("nofuse", dict(hooks, YTTM_NO_FUSE="1"))
    assert rep0["fused_rounds"] == 0
    for i in range(10 if not use_comm and pin_name == "c2_100mb" else 4):  # (the ordering rests on an empirical check -- more runs of it, ten on the headline path)
        # (class-B tiles, the CJK-shaped pin: runs 0 and 2 with their launch of a word-mode round BESIDE k_words on a second stream -- the tail waits
        # for its flag, ScanArgs::peer_flag --, runs 1 and 3 the default: before it on the main stream)
        beside = pin_name == "c6_cjk_100mb" and i % 2 == 0
        got, rep = run("fuse%d" % i, dict(hooks, YTTM_CLASSB_BESIDE="1") if beside else hooks)
        assert rep["fused_rounds"] > 100 and rep["word_fused_rounds"] > 100, (rep["fused_rounds"], rep["word_fused_rounds"])
        if pin_name == "c6_cjk_100mb":
            assert (rep["classb_overlapped"] > 100) == beside, rep["classb_overlapped"]
        assert len(got) == len(ref), (len(got), len(ref))
        for n, (a, b) in enumerate(zip(ref, got)):
            assert a == b, "scan %d differs (run %d):\n  separate scan: %s\n  fused tail:    %s" % (n, i, a, b)


def test_zz_encode_concurrent_threads(tmp_path):
    """Two Python threads encode on ONE BPE object (ctypes releases the GIL): each call owns an encoder lane, results are the
    single-threaded ones."""
    import threading
    import youtokentome_amd as yttm
    model = os.path.join(S.G, "train_readme_small.model")
    bpe = yttm.BPE(model)
    rng = random.Random(12)
    batches = [["".join(rng.choice("abcd  ") for _ in range(rng.randint(0, 300))) for _ in range(rng.randint(1, 400))] for _ in range(24)]
    want = [bpe.encode(b, yttm.OutputType.ID, bos=True) for b in batches]
    got = [None] * len(batches)

    def work(k):
        for i in range(k, len(batches), 4):
            got[i] = bpe.encode(batches[i], yttm.OutputType.ID, bos=True)
    th = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert got == want
