"""Kernel-logic tests on GPU-less machines: the UNMODIFIED product sources (youtokentome_amd/csrc) built against the
HIP emulator (tests/hipsim), driven through the same C ABI and checked against the oracle / golden fixtures.
(The parity tests proper are tests/test_gpu_parity.py, -m gpu, on a real MI355X.)"""
import os
import random

import pytest

import gen
import stage_checks as S

pytestmark = pytest.mark.usefixtures("sim_lib")


def test_char_hist():
    for t in S.texts_small(0, n=4, size=1500) + [gen.readme_corpus(100, 100)]:
        S.check_char_hist(t)


def test_front_end_three_byte_fast_path(monkeypatch):
    """K1 / K2a classify a window of ASCII + three-byte chars four bytes per instruction (round 5); anything else in a lane's 24-byte window
    takes the exact byte-by-byte path.  Counts, decode steps, segment starts and words must not depend on which path a lane took -- both K1
    variants (wide chars in an LDS hash, or global atomics), texts below and above one 4 KB chunk."""
    rng = random.Random(5)
    for n in (50, 333, 4096, 4097, 9000, 20000):
        for _ in range(2):
            t = S.three_byte_text(rng, n)
            for wide in ("0", "1"):
                monkeypatch.setenv("YTTM_K1_WIDE", wide)
                S.check_char_hist(t)
            monkeypatch.delenv("YTTM_K1_WIDE")
            if t.strip():
                S.check_word_table_and_pairs(t)


def test_chunked_front_end(tmp_path, monkeypatch):
    """Round 5: a corpus that does not fit the HBM left for it crosses the device in chunks cut at white space; only the distinct words' bytes
    stay (a lexicon behind the chunk region), the word table grows by rehashing, and when coverage drops chars the source is read a second
    time (gpu_ctx.cpp front_end_chunked).  Forced onto toy files with chunks of 4 KB and 8 KB: the model must be the oracle's -- ASCII, mixed
    scripts with invalid bytes, coverage < 1, three-byte text, a file without a trailing newline, words seen more often than a weight holds;
    the report says how many chunks there were."""
    import ctypes as C
    import filecmp
    import json
    import oracle_lib as O
    from youtokentome_amd import _lib
    rng = random.Random(3)
    cases = [(gen.readme_corpus(300, 90, "abcdef ", seed=8), 300, 1.0), (gen.unicode_text(rng, 20000, "mix", p_invalid=0.01), 150, 1.0),
             (gen.unicode_text(rng, 20000, "mix", p_invalid=0.01), 120, 0.85), (gen.zipf_corpus(60000, vocab=3000), 400, 1.0),
             (S.three_byte_text(rng, 30000), 200, 1.0), (gen.readme_corpus(200, 90, "abcdef ", seed=9).rstrip(b"\n"), 200, 1.0)]
    L = _lib.load()
    for kb, serial in (("4", False), ("8", False), ("4", True)):  # (serial: no landing buffer, a chunk is uploaded, then worked on)
        monkeypatch.setenv("YTTM_FE_CHUNK_KB", kb)
        if serial:
            monkeypatch.setenv("YTTM_FE_CHUNK_SERIAL", "1")
        for i, (text, vocab, cov) in enumerate(cases):
            corpus, m_gpu, m_ora = str(tmp_path / "c.txt"), str(tmp_path / "g.model"), str(tmp_path / "o.model")
            open(corpus, "wb").write(text)
            err, rep = C.create_string_buffer(2048), C.create_string_buffer(16384)
            assert L.yttm_train_bpe_ex(corpus.encode(), m_gpu.encode(), vocab, cov, 1, 0, 1, 2, 3, 0, rep, 16384, err, 2048) == 0, err.value
            O.train(text, m_ora, vocab, cov)
            assert filecmp.cmp(m_gpu, m_ora, shallow=False), (kb, i)
            r = json.loads(rep.value.decode())
            assert r["front_end_chunks"] >= len(text) // (int(kb) << 10) and r["peak_device_bytes"] > 0, r
    monkeypatch.delenv("YTTM_FE_CHUNK_SERIAL")
    # a word longer than a chunk is refused, and says so (the documented limit: INTEGRATION.md)
    monkeypatch.setenv("YTTM_FE_CHUNK_KB", "4")
    corpus = str(tmp_path / "long.txt")
    open(corpus, "wb").write(b"ab " * 3000 + b"abcd" * 1300 + b" ab\n")
    err = C.create_string_buffer(2048)
    assert L.yttm_train_bpe_ex(corpus.encode(), str(tmp_path / "l.model").encode(), 50, 1.0, 1, 0, 1, 2, 3, 0, None, 0, err, 2048) != 0
    assert b"a word longer than the front end's chunk" in err.value, err.value
    # from host memory, and a word heavier than a weight holds (the copies are made from the lexicon's bytes)
    monkeypatch.setenv("YTTM_FE_CHUNK_KB", "4")
    monkeypatch.setenv("YTTM_TEST_WCNT_MAX", "7")
    text = gen.readme_corpus(300, 90, "ab ", seed=4)
    err = C.create_string_buffer(2048)
    m_gpu, m_ora = str(tmp_path / "g2.model"), str(tmp_path / "o2.model")
    assert L.yttm_train_bpe_from_memory(text, len(text), m_gpu.encode(), 60, 1.0, 0, 1, 2, 3, 0, None, 0, err, 2048) == 0, err.value
    O.train(text, m_ora, 60)
    assert filecmp.cmp(m_gpu, m_ora, shallow=False)


def test_word_table_and_pair_count():
    for i, t in enumerate(S.texts_small(1, n=4, size=2000)):
        S.check_word_table_and_pairs(t, coverage=1.0 if i % 2 == 0 else 0.9)


def test_pair_count_by_alphabet_size():
    """K3 has a kernel of its own for alphabets of up to 64 symbols (two table sizes: up to 32, up to 64) and the general tile kernel
    beyond; runs of equal tokens cross its 64-token chunks at every offset."""
    for t in S.texts_by_alphabet_size():
        S.check_word_table_and_pairs(t)


def test_word_table_fast_and_exact_scans_agree():
    """K2 scans the common word -- a few ASCII letters, nothing dropped -- 16 bytes at a time and compares such words as bytes;
    everything else takes the exact char-by-char walk.  The same word must dedup across the two paths: with and without an
    invalid byte inside, with a char that coverage drops, at every length around the 8/16-byte loads, at the very end of the text."""
    rng = random.Random(13)
    base = ["a" * k for k in (1, 7, 8, 9, 15, 16, 17, 23, 24, 25, 31, 32, 33, 40)] + ["abcdefgh" * 3, "abcab", "x"]
    words = []
    for w in base:
        words += [w.encode()] * rng.randint(1, 3)
        words.append(w[: len(w) // 2].encode() + b"\xff" + w[len(w) // 2:].encode())      # invalid byte inside: same tokens
        words.append(w.encode() + b"\xc3")                                                # truncated char at the end: same tokens
        words.append(w[:1].encode() + "й".encode() + w[1:].encode())                       # a rare char (dropped below at coverage 0.9)
        words.append(w.encode() + "日".encode())
    rng.shuffle(words)
    for tail in (b"", b" ab", b" abcabcabcabcabcabcabcab", b"\n"):
        text = b" ".join(words) + tail
        S.check_word_table_and_pairs(text, coverage=1.0)
        S.check_word_table_and_pairs(text, coverage=0.9)


def test_word_table_multi_tile():
    S.check_word_table_and_pairs(gen.readme_corpus(150, 100, seed=3))  # ~10k dedup tokens: several tiles


def test_merge_apply_rounds():
    for i, t in enumerate(S.texts_small(2, n=3, size=1500)):
        if t.strip():
            S.check_merge_rounds(t, rounds=5, seed=i)


def test_k4_measurement_pass():
    for i, t in enumerate(S.texts_small(3, n=2, size=1500) + [gen.readme_corpus(120, 100, seed=4)]):
        if t.strip():
            S.check_k4_measure(t, rounds=6, seed=i)
    words = ["ab" * k for k in range(60, 125, 7)] + ["a" * k for k in range(150, 250, 13)]
    S.check_k4_measure((" ".join(words) + " ").encode(), rounds=6, seed=1)


def test_merge_apply_runs():
    t = ("aaaa aaaaa aaaaaaa abababab aabbaabb abcabcabc bbbbbb ab aaab baaa aaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaa " * 3).encode()
    S.check_merge_rounds(t, rounds=12, seed=1)


def test_merge_apply_ids_above_the_lds_bitmap():
    """Token ids >= 32768 (vocab_size 50000, large alphabets): flags come from the HBM table, batches are uploaded."""
    for i, t in enumerate(S.texts_small(5, n=2, size=1500)):
        if t.strip():
            S.check_merge_rounds(t, rounds=6, seed=i, id_shift=40000)
    t = ("aaaa aaaaa abababab aabbaabb abcabcabc bbbbbb ab aaab baaa " * 3).encode()
    S.check_merge_rounds(t, rounds=10, seed=1, id_shift=33000)


def test_merge_apply_small_alphabets_random():
    """Words over 2..5 letters (runs of equal tokens, runs of new tokens, many sites per tile), random weights and batch sizes."""
    rng = random.Random(2024)
    for trial in range(40):
        alpha = "abcde"[: rng.choice([2, 2, 3, 3, 4, 5])]
        words = ["".join(rng.choice(alpha) for _ in range(rng.choice([1, 2, 3, 5, 8, 13, 30, 80, 200]))) for _ in range(rng.randint(5, 250))]
        words = [w for w in words for _ in range(rng.randint(1, 3))]
        rng.shuffle(words)
        S.check_merge_rounds((" ".join(words) + " ").encode(), rounds=rng.randint(3, 14), seed=trial)


def test_merge_apply_site_placements():
    S.check_site_placements(trials=100, seed=3)


def test_merge_apply_many_sites_per_tile():
    """More than 64 (and more than 128) merge sites in one class-A tile: phase 2 of K4 takes them 64 per pass."""
    words = ["ab" * k for k in range(60, 125, 7)] + ["a" * k for k in range(150, 250, 13)] + ["abc" * k for k in (50, 70, 80)]
    S.check_merge_rounds((" ".join(words) + " ").encode(), rounds=8, seed=4)


def test_word_mode_batch_split(tmp_path, monkeypatch):
    """A batch of 129 .. 256 rules in word mode goes as two rounds (its first 128 rules are one launch; host_trainer.cpp): corpora whose
    second round merges a few hundred disjoint pairs at once -- word i is three fresh ideographs a b c (and a b, a b c d beside it) -- same
    models as the oracle, with the split on (at least one batch cut) and off."""
    import ctypes as C
    import filecmp
    import json
    from youtokentome_amd import _lib
    import oracle_lib as O
    L = _lib.load()
    for k in ("YTTM_WORD_MIN_TILES", "YTTM_WORD_MIN_TOKENS", "YTTM_WORD_DIV"):
        monkeypatch.setenv(k, "0")
    monkeypatch.setenv("YTTM_WORDS_GRID", "3")
    splits = {}
    for off in (False, True):
        if off:
            monkeypatch.setenv("YTTM_NO_BATCH_SPLIT", "1")
        splits[off] = 0
        for n, vocab in ((200, 4 + 800 + 500), (300, 4 + 1200 + 700)):
            text = gen.disjoint_words_corpus(n)
            cp, mg, mo = str(tmp_path / "c.txt"), str(tmp_path / "g.model"), str(tmp_path / "o.model")
            open(cp, "wb").write(text)
            err, rep = C.create_string_buffer(2048), C.create_string_buffer(16384)
            assert L.yttm_train_bpe_ex(cp.encode(), mg.encode(), vocab, 1.0, 1, 0, 1, 2, 3, 0, rep, 16384, err, 2048) == 0, err.value
            O.train(text, mo, vocab)
            assert filecmp.cmp(mg, mo, shallow=False), (n, off)
            r = json.loads(rep.value.decode())
            assert r["word_fused_rounds"] > 100, r
            splits[off] += r["batch_splits"]
    assert splits[False] >= 2 and splits[True] == 0, splits


@pytest.mark.parametrize("name", S.golden_train_names())
def test_golden_train(name, tmp_path):
    S.check_golden_train(name, tmp_path)


@pytest.mark.parametrize("name", S.golden_encode_names())
def test_golden_encode(name):
    S.check_golden_encode(name)


def test_train_encode_vs_oracle_random(tmp_path):
    rng = random.Random(5)
    layouts = [(0, 1, 2, 3), (-1, 0, -1, -1), (5, 7, -1, 2), (3, 2, 1, 0)]
    for it in range(6):
        kind = rng.choice(list(gen.UNICODE_ALPHABETS))
        text = gen.unicode_text(rng, rng.randint(200, 2500), kind, p_invalid=0.02 if it % 3 == 0 else 0.0)
        cov = rng.choice([1.0, 0.9, 0.7]) if it % 3 == 0 else rng.choice([1.0, 1.0, 0.95])
        ids = layouts[it % 4]
        if cov == 1.0 and it % 3 == 0:
            cov = 0.9  # invalid bytes + coverage 1 would crash the reference; both drop them here anyway
        model = S.check_train_vs_oracle(text, rng.randint(40, 90), tmp_path, cov, ids, tag=f"r{it}")
        if model:
            sents = [gen.unicode_text(rng, rng.randint(0, 60), kind).decode().replace("\n", " ") for _ in range(20)] + ["", "  "]
            S.check_encode_vs_oracle(model, sents)


def test_encode_mixed_shapes():
    S.check_encode_mixed_shapes(n_sent=150)


def test_front_end_under_the_upload(tmp_path, monkeypatch):
    monkeypatch.setenv("YTTM_FE_OVERLAP_MIN", "0")
    monkeypatch.setenv("YTTM_FE_PART_KB", "4")
    monkeypatch.setenv("YTTM_IO_CHUNK_KB", "4")
    S.check_front_end_under_upload(tmp_path, rounds=12)


def test_encode_word_cache():
    S.check_encode_word_cache()


def test_encode_word_cache_fuzz(tmp_path):
    S.check_encode_word_cache_fuzz(tmp_path)


def test_encode_host_to_host_in_sub_batches(monkeypatch):
    """a very large host -> host batch goes through both lanes in sub-batches, up / encode / down at once (host_encoder.cpp encode_pipelined):
    here every batch, in sub-batches of 1 KB, arrays through 4 KB chunks"""
    monkeypatch.setenv("YTTM_ENC_PIPE_FROM", "1")
    monkeypatch.setenv("YTTM_ENC_SUB_KB", "1")
    monkeypatch.setenv("YTTM_IO_CHUNK_KB", "4")
    S.check_encode_mixed_shapes(n_sent=150, seed=67)
    S.check_encode_word_cache(n_sent=40, seed=71)


def test_encode_host_arrays_through_pinned_chunks(monkeypatch):
    """host -> host encode of a large batch moves its arrays through the trainer's pinned chunks, several threads at once (host_encoder.cpp
    copy_up / copy_down, gpu_ctx.cpp staged_transfer): here every array of small batches, in chunks of 4 KB"""
    monkeypatch.setenv("YTTM_ENC_STAGED_FROM", "1")
    monkeypatch.setenv("YTTM_IO_CHUNK_KB", "4")
    S.check_encode_mixed_shapes(n_sent=150, seed=53)
    S.check_encode_word_cache(n_sent=40, seed=59)


@pytest.mark.parametrize("sblk", [3, 64])
def test_encode_word_cache_sentence_blocks(sblk, tmp_path, monkeypatch):
    """the word cache's three walks over the text (k_wcache.hip: insert, count, scatter) take blocks of consecutive sentences as one run --
    64 per wavefront in large batches, one in small ones like these tests' unless told otherwise: sentences of every length, empty ones,
    ones that begin and end inside UTF-8 sequences, against the oracle"""
    monkeypatch.setenv("YTTM_WC_SBLK", str(sblk))
    S.check_encode_word_cache(n_sent=80, seed=43)
    S.check_encode_word_cache_fuzz(tmp_path, trials=4, seed=47)


def test_encode_word_cache_crowded_short_region(tmp_path, monkeypatch):
    """more distinct short words than the table's short region has slots: they go on in the whole table (k_wcache.hip WC_SHORT_PROBES)"""
    monkeypatch.setenv("YTTM_WC_SHORT_SLOTS", "16")
    S.check_encode_word_cache(n_sent=60, seed=37)
    S.check_encode_word_cache_fuzz(tmp_path, trials=3, seed=41)


@pytest.mark.parametrize("lane_max", [0, 5, 1000])
def test_encode_one_word_per_lane(lane_max, monkeypatch):
    """K5's two ways through the merge rounds -- the wave-wide rounds and one word per lane (merge_lanes) -- are the oracle's ids both:
    never the lanes (0), the lanes for packs of short words only (5: most packs fall back), the lanes whatever the words (1000; the
    default is 48).  The shapes have runs of one letter, a rule several times in a word, unknown chars, words of 1..60 chars."""
    monkeypatch.setenv("YTTM_K5_LANE_WORDS", str(lane_max))
    monkeypatch.setenv("YTTM_K5_LANE_SENT", str(lane_max))
    S.check_encode_mixed_shapes(n_sent=60, seed=29)
    S.check_encode_word_cache(n_sent=40, seed=31)


def test_many_words_per_tile(tmp_path):
    S.check_many_words_per_tile(tmp_path)


def test_hot_list_rebuilds(tmp_path, monkeypatch):
    """The candidate filter reads a hot list of pairs instead of the whole pair table; shrink the list so that tiny corpora
    go through its rebuild, overflow and whole-table fallback paths, and demand the same models."""
    for target, mn, cap, nofuse in ((8, 3, 64, "0"), (4, 2, 16, "0"), (64, 8, 4096, "0"), (8, 3, 64, "1"), (64, 8, 4096, "1")):
        monkeypatch.setenv("YTTM_NO_FUSE", nofuse)  # 1: the candidate scan as a kernel of its own instead of the apply kernel's tail
        monkeypatch.setenv("YTTM_HOT_TARGET", str(target))
        monkeypatch.setenv("YTTM_HOT_MIN", str(mn))
        monkeypatch.setenv("YTTM_HOT_CAP", str(cap))
        for name in ("readme_small", "runs", "mix_cov"):
            S.check_golden_train(name, tmp_path)
        rng = random.Random(target)
        text = gen.unicode_text(rng, 3000, "ascii")
        S.check_train_vs_oracle(text, 150, tmp_path, tag=f"hot{target}")


def test_list_overflow_in_a_fused_round(tmp_path, monkeypatch):
    """The top list overflows while the candidate scan rides in the apply kernel's tail (which then has no complete list to zero
    the finished batch's pairs through): the refill that follows must still see those pairs at zero.  Same for the hot list."""
    import ctypes as C
    import json
    import filecmp
    from youtokentome_amd import _lib
    L = _lib.load()
    seen = fused = 0
    # (hot target, hot cap, top target, top cap, top min)
    for ht, hc, tt, tc, tm in ((400, 4096, 8, 24, 1), (400, 4096, 30, 80, 1), (40, 128, 8, 24, 1), (100, 256, 16, 40, 4)):
        for k, v in (("YTTM_HOT_TARGET", ht), ("YTTM_HOT_CAP", hc), ("YTTM_HOT_MIN", 1), ("YTTM_TOP_TARGET", tt), ("YTTM_TOP_CAP", tc), ("YTTM_TOP_MIN", tm)):
            monkeypatch.setenv(k, str(v))
        for name in ("zipf", "readme_small", "mix_cov"):
            a = json.load(open(os.path.join(S.G, f"train_{name}.args.json")))
            out = str(tmp_path / f"{name}.model")
            err, rep = C.create_string_buffer(2048), C.create_string_buffer(16384)
            rc = L.yttm_train_bpe_ex(os.path.join(S.G, f"train_{name}.txt").encode(), out.encode(), a["vocab"], a["coverage"], 1, a["pad"], a["unk"],
                                     a["bos"], a["eos"], 0, rep, 16384, err, 2048)
            assert rc == 0, err.value
            assert filecmp.cmp(out, os.path.join(S.G, f"train_{name}.model"), shallow=False), (name, ht, hc, tt, tc)
            r = json.loads(rep.value.decode())
            fused += r["fused_rounds"]
            seen += r["fused_overflows"]
    assert fused > 0 and seen > 0, "no configuration overflowed the top list during a fused round: the test does not test what it says"


def test_words_seen_more_often_than_a_weight_holds(tmp_path, monkeypatch):
    """The reference counts word frequencies in uint64 (bpe.cpp:382-385); the tiles keep uint32 weights.  A word seen more often than a
    weight holds becomes several equal words whose weights add up to its count (every pair count is a sum over words).  At full scale
    that takes a word seen 2^32 times (tests/golden/full_size_pins.json c8_heavy_word, checked on the MI355X); here the largest weight is
    lowered to a few dozen, so that many words of ordinary corpora are split into many copies -- short, long (class B) and very long
    (class C) ones: same models as the oracle."""
    rng = random.Random(5)
    long_b = "".join(rng.choice("abc") for _ in range(700))
    long_c = "".join(rng.choice("ab") for _ in range(2500))
    rare = " ".join("".join(rng.choice("abcd") for _ in range(rng.randint(3, 9))) for _ in range(400))
    heavy = ["ab", "abab", "ba", "aab", "dcba", "cab", long_b, long_c]
    for wmax, reps in (("3", 11), ("1", 4), ("37", 120), ("1000", 1001)):  # (up to a few hundred copies in all: the list of such words is short by design)
        monkeypatch.setenv("YTTM_TEST_WCNT_MAX", wmax)
        text = (" ".join(heavy[:6] * reps + heavy[6:] * min(reps, 11)) + " " + rare + "\n").encode()
        S.check_train_vs_oracle(text, 120, tmp_path, tag="heavy" + wmax)


def test_very_long_words(tmp_path):
    S.check_very_long_words(tmp_path)


def test_memory_pool_reuse_and_release(tmp_path):
    """Trainings reuse the device buffers of earlier ones (pool in gpu_ctx.cpp): same models with dirty, recycled memory;
    yttm_release_device_memory() hands the cache back and training still works afterwards."""
    from youtokentome_amd import _lib
    for rep in range(2):
        for name in ("readme_small", "runs"):
            S.check_golden_train(name, tmp_path)
    _lib.load().yttm_release_device_memory()
    S.check_golden_train("mix_cov", tmp_path)


def test_config_errors(tmp_path):
    for kw in [dict(coverage=0.0), dict(ids=(0, 300, 2, 3)), dict(ids=(0, 1, 1, 3)), dict(vocab=5)]:
        S.check_train_vs_oracle(b"aaa bbb abab", kw.get("vocab", 50), tmp_path, kw.get("coverage", 1.0), kw.get("ids", (0, 1, 2, 3)), tag="e")


def test_long_words_class_b(tmp_path):
    """words of 513..2047 chars live in the second tile class (slot 4096, one wave per workgroup)"""
    rng = random.Random(3)
    words = ["".join(rng.choice("abc") for _ in range(n)) for n in (2046, 1500, 700, 513, 512, 511, 65, 64, 63, 1)]
    text = (" ".join(words) + "\n") * 2 + "ab abc aabb\n"
    S.check_merge_rounds(text.encode(), rounds=6, seed=2)
    model = S.check_train_vs_oracle(text.encode(), 60, tmp_path, tag="lw")
    S.check_encode_vs_oracle(model, [" ".join(words[2:]), "ab" * 700, "a"], flags=((0, 0, 0), (1, 1, 1)))


def test_dropout_extremes_and_roundtrip():
    import os
    model = os.path.join(S.G, "train_readme_small.model")
    rng = random.Random(4)
    sents = ["".join(rng.choice("abcd  ") for _ in range(rng.randint(0, 70))) for _ in range(40)] + ["", " ", "abcd" * 40]
    S.check_dropout_extremes(model, sents)


def test_dropout_heap_equals_array():
    S.check_dropout_heap_equals_array()


def test_dropout_distribution_small():
    import os
    model = os.path.join(S.G, "train_readme_small.model")
    rng = random.Random(6)
    sents = ["".join(rng.choice("abcd  ") for _ in range(60)) for _ in range(1500)]
    S.check_dropout_distribution(model, sents, 0.1, 600)


def test_encode_concurrent_threads():
    """Two encoder lanes behind one BPE object: concurrent encode() calls from Python threads give the single-threaded results."""
    import os
    import threading
    import youtokentome_amd as yttm
    bpe = yttm.BPE(os.path.join(S.G, "train_readme_small.model"))
    rng = random.Random(12)
    batches = [["".join(rng.choice("abcd  ") for _ in range(rng.randint(0, 80))) for _ in range(rng.randint(1, 12))] for _ in range(12)]
    want = [bpe.encode(b, yttm.OutputType.ID, bos=True) for b in batches]
    got = [None] * len(batches)

    def work(k):
        for i in range(k, len(batches), 3):
            got[i] = bpe.encode(batches[i], yttm.OutputType.ID, bos=True)
    th = [threading.Thread(target=work, args=(k,)) for k in range(3)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert got == want


def test_word_table_overflow_is_redone(tmp_path):
    """K2 sizes its word table for far fewer distinct words than occurrences; a corpus of (nearly) all distinct words overflows
    it and the dedup is redone with the worst-case size -- same model as the oracle's."""
    import ctypes as C
    import filecmp
    import json
    from youtokentome_amd import _lib
    import oracle_lib as O
    rng = random.Random(31)
    words = set()
    while len(words) < 36000:
        words.add("".join(rng.choice("abcdefgh") for _ in range(rng.randint(6, 9))))
    text = (" ".join(words) + "\n").encode()
    corpus, m_gpu, m_ora = str(tmp_path / "u.txt"), str(tmp_path / "gpu.model"), str(tmp_path / "ora.model")
    open(corpus, "wb").write(text)
    L = _lib.load()
    err, rep = C.create_string_buffer(2048), C.create_string_buffer(16384)
    rc = L.yttm_train_bpe_ex(corpus.encode(), m_gpu.encode(), 40, 1.0, 1, 0, 1, 2, 3, 0, rep, 16384, err, 2048)
    assert rc == 0, err.value
    assert json.loads(rep.value.decode())["word_table_retries"] == 1
    O.train(text, m_ora, 40)
    assert filecmp.cmp(m_gpu, m_ora, shallow=False)


def test_word_mode_with_class_b_tiles_beside(tmp_path, monkeypatch):
    """Word mode with class-B tiles (words of 257 .. 2 048 tokens -- the long clauses of unsegmented scripts -- stay in tiles of their own): the
    round's class-B launch goes to a second stream beside k_words, its last workgroup raises a flag the round's tail waits for (gpu_ctx.cpp
    merge_apply, ScanArgs::peer_flag) -- with YTTM_CLASSB_BESIDE=1; since round 6 the default is one stream (measured faster: profiles/r6_classb_order.txt).
    Same model as the oracle either way; the report counts the rounds that ran side by side."""
    import ctypes as C
    import filecmp
    import json
    from youtokentome_amd import _lib
    import oracle_lib as O
    L = _lib.load()
    for k in ("YTTM_WORD_MIN_TILES", "YTTM_WORD_MIN_TOKENS", "YTTM_WORD_DIV"):
        monkeypatch.setenv(k, "0")
    monkeypatch.setenv("YTTM_WORDS_GRID", "3")
    rng = random.Random(5)
    ws = []
    for _ in range(60):  # long clauses, some of them periodic (runs of one pair), among many short words
        n_ch = rng.choice([rng.randint(260, 420), rng.randint(500, 900), rng.randint(1500, 2040)])
        w = (rng.choice("ab") * rng.randint(1, 4) + rng.choice("abc") * rng.randint(1, 4)) * (n_ch // 2) if rng.random() < 0.3 else "".join(rng.choice("abc") for _ in range(n_ch))
        ws.append(w[:n_ch])
    ws += ["".join(rng.choice("abcd") for _ in range(rng.randint(1, 9))) for _ in range(3000)]
    rng.shuffle(ws)
    text = (" ".join(ws) + "\n").encode()
    O.train(text, str(tmp_path / "o.model"), 260)
    seen = {}
    for off in (False, True):
        if off:
            monkeypatch.delenv("YTTM_CLASSB_BESIDE")
        else:
            monkeypatch.setenv("YTTM_CLASSB_BESIDE", "1")
        cp, mg = str(tmp_path / "c.txt"), str(tmp_path / "g.model")
        open(cp, "wb").write(text)
        err, rep = C.create_string_buffer(2048), C.create_string_buffer(16384)
        assert L.yttm_train_bpe_ex(cp.encode(), mg.encode(), 260, 1.0, 1, 0, 1, 2, 3, 0, rep, 16384, err, 2048) == 0, err.value
        assert filecmp.cmp(mg, str(tmp_path / "o.model"), shallow=False), off
        r = json.loads(rep.value.decode())
        assert r["word_rounds"] > 20, r
        seen[off] = r["classb_overlapped"]
    assert seen[False] > 20 and seen[True] == 0, seen


def test_word_mode(tmp_path, monkeypatch):
    """Word mode (class-A words in fixed slots, a round visits the words that hold a merge site, found through the pair index at word
    granularity or the instance list of the pair's younger token) forced on from the second round, on corpora of several tiles: the round
    as ONE launch (k_words<FUSED>, the default) and as k_wgather + k_words + k_delta_apply (YTTM_WORDS_FUSE_MAX=0); then with tiny hot lists
    (an index build per rebuild, rounds over every word while the list is overflowed), a record log and record regions that overflow, in
    both forms, the count updates through k_delta_apply instead of the small rounds' own tail, few workgroups (several gather passes per
    workgroup): same models as the oracle."""
    import ctypes as C
    import filecmp
    import json
    from youtokentome_amd import _lib
    import oracle_lib as O
    L = _lib.load()
    monkeypatch.setenv("YTTM_WORD_MIN_TILES", "0")
    monkeypatch.setenv("YTTM_WORD_MIN_TOKENS", "0")
    monkeypatch.setenv("YTTM_WORD_DIV", "0")
    rng = random.Random(78)
    cases = [(gen.readme_corpus(300, 100, seed=6), 900), (gen.zipf_corpus(120000, vocab=3000), 700),
             (gen.unicode_text(rng, 30000, "ascii"), 400), (("aaaa aaaaa abababab aabbaabb bbbbbb ab aaab baaa " * 400).encode(), 60),
             (gen.unicode_text(rng, 20000, "cjk"), 600)]
    word_rounds = all_rounds = builds = fused = 0
    # (the emulator's time goes with the workgroups it runs: the grids the launchers would pick on two corpora only, a few workgroups --
    # several gather passes each -- elsewhere)
    small = {"YTTM_WORDS_GRID": 3, "YTTM_WGATHER_GRID": 2}
    tiny_hot = {"YTTM_HOT_TARGET": 40, "YTTM_HOT_MIN": 4, "YTTM_HOT_CAP": 400, "YTTM_WORD_DREC": 64}
    for cfg, which in ((None, (1, 3)), ({"YTTM_WORDS_FUSE_MAX": 0}, (3, 4)), (small, (0, 1, 2, 3, 4)), ({"YTTM_WORDS_FUSE_MAX": 0, **small}, (0, 1, 2, 3, 4)),
                       ({"YTTM_INDEX_AGG_MIN": 0, **tiny_hot, **small}, (0, 1, 2, 3, 4)), ({"YTTM_WORDS_FUSE_MAX": 0, **tiny_hot, **small}, (0, 2, 4)),
                       ({"YTTM_WORD_LOG": 300, "YTTM_WORDS_INLINE_MAX": 0, "YTTM_WORDS_FUSE_MAX": 0, **small}, (0, 1, 2, 4)),
                       ({"YTTM_WORD_LOG": 300, "YTTM_WORD_DREC": 16, **small}, (0, 1, 2, 4)),
                       ({"YTTM_WORDS_FUSE_MAX": 4400, "YTTM_WORDS_GRID": 1, "YTTM_WGATHER_GRID": 1}, (0, 1, 2, 4))):
        for k, v in (cfg or {}).items():
            monkeypatch.setenv(k, str(v))
        for i, (text, vocab) in enumerate(cases):
            if i not in which:
                continue
            corpus, m_gpu, m_ora = str(tmp_path / f"c{i}.txt"), str(tmp_path / f"g{i}.model"), str(tmp_path / f"o{i}.model")
            open(corpus, "wb").write(text)
            err, rep = C.create_string_buffer(2048), C.create_string_buffer(16384)
            rc = L.yttm_train_bpe_ex(corpus.encode(), m_gpu.encode(), vocab, 1.0, 1, 0, 1, 2, 3, 0, rep, 16384, err, 2048)
            assert rc == 0, err.value
            O.train(text, m_ora, vocab)
            assert filecmp.cmp(m_gpu, m_ora, shallow=False), (i, cfg)
            r = json.loads(rep.value.decode())
            word_rounds += r["word_rounds"]
            all_rounds += r["word_all_rounds"]
            builds += r["index_builds"]
            fused += r["word_fused_rounds"]
        for k in (cfg or {}):
            monkeypatch.delenv(k)
    assert word_rounds > 1000 and all_rounds > 10 and builds > 20 and 500 < fused < word_rounds - 500, (word_rounds, all_rounds, builds, fused)


def test_k3_radix_partition(tmp_path, monkeypatch):
    """K3 of large alphabets (k_pairradix.hip, round 6): (pair, weight) records partitioned by their first token in two levels, summed in a dense
    LDS array per first token -- forced on at toy sizes (YTTM_K3_RADIX_MIN=0).  The whole pair table must equal the oracle's count (alphabets of 66 ..
    1500 symbols: one, two and more level-1 groups per chunk; runs of one symbol, odd and even: the floor(L/2) rule), and a training that
    counted this way must write the oracle's model."""
    import ctypes as C
    import filecmp
    import json
    from youtokentome_amd import _lib
    import oracle_lib as O
    monkeypatch.setenv("YTTM_K3_RADIX_MIN", "0")
    for t in S.texts_by_alphabet_size(sizes=(66, 130, 300, 1500), n_words=300):
        S.check_word_table_and_pairs(t)
    S.check_word_table_and_pairs(gen.cjk_corpus_fast(40000, seed=3))  # (up to 4096 ideographs, clauses without spaces)
    S.check_word_table_and_pairs(gen.cjk_corpus_fast(40000, seed=4), coverage=0.95)
    L = _lib.load()
    for i, (text, vocab) in enumerate(((S.texts_by_alphabet_size(sizes=(200,), n_words=300)[0], 500), (gen.cjk_corpus_fast(40000, seed=5), 4600))):
        corpus, m_gpu, m_ora = str(tmp_path / f"c{i}.txt"), str(tmp_path / f"g{i}.model"), str(tmp_path / f"o{i}.model")
        open(corpus, "wb").write(text)
        err, rep = C.create_string_buffer(2048), C.create_string_buffer(16384)
        assert L.yttm_train_bpe_ex(corpus.encode(), m_gpu.encode(), vocab, 1.0, 1, 0, 1, 2, 3, 0, rep, 16384, err, 2048) == 0, err.value
        r = json.loads(rep.value.decode())
        assert r["k3_radix"] == 1, r
        O.train(text, m_ora, vocab)
        assert filecmp.cmp(m_gpu, m_ora, shallow=False), i
    # the path's scratch (16 bytes per class-A token) must fit the free device memory with room to spare: short of it, the general kernel counts
    monkeypatch.setenv("YTTM_TEST_FREE_BYTES", "100000")
    text = S.texts_by_alphabet_size(sizes=(200,), n_words=300)[0]
    corpus, m_gpu, m_ora = str(tmp_path / "cm.txt"), str(tmp_path / "gm.model"), str(tmp_path / "om.model")
    open(corpus, "wb").write(text)
    err, rep = C.create_string_buffer(2048), C.create_string_buffer(16384)
    assert L.yttm_train_bpe_ex(corpus.encode(), m_gpu.encode(), 500, 1.0, 1, 0, 1, 2, 3, 0, rep, 16384, err, 2048) == 0, err.value
    assert json.loads(rep.value.decode())["k3_radix"] == 0
    O.train(text, m_ora, 500)
    assert filecmp.cmp(m_gpu, m_ora, shallow=False)
    monkeypatch.delenv("YTTM_TEST_FREE_BYTES")
    monkeypatch.setenv("YTTM_K3_RADIX_MIN", "1000000000000")
    S.check_word_table_and_pairs(S.texts_by_alphabet_size(sizes=(130,), n_words=300)[0])
