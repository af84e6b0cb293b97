"""CPU soak: random corpora through the product sources under the HIP emulator against the oracle (byte-identical model files, identical
encode ids with and without the word cache).  usage: python tools/soak_sim.py [seconds] [seed] [big|long|rounds|words]
(words: the larger tables of `big` with K4's word mode forced on from the second round and the hooks that force its rare paths)"""
import os, pathlib, random, sys, tempfile, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("YTTM_AMD_LIB", os.path.join(R, "tests", "hipsim", "_build", "libyttm_sim.so"))
os.environ.setdefault("YTTM_WORD_HINT_FLOOR", "2048")  # (the emulator's time goes with the workgroups: smaller grids for the small word-mode rounds)
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import gen
import stage_checks as S

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
words_mode = len(sys.argv) > 3 and sys.argv[3] == "words"
big = len(sys.argv) > 3 and sys.argv[3] in ("big", "words")
long_words = len(sys.argv) > 3 and sys.argv[3] == "long"
rounds_mode = len(sys.argv) > 3 and sys.argv[3] == "rounds"
rng = random.Random(seed)
tmp = pathlib.Path(tempfile.mkdtemp())
t0, n = time.time(), 0
devnull = os.open(os.devnull, os.O_WRONLY)
os.dup2(devnull, 2)  # (the trainers print the reference's progress lines)
while time.time() - t0 < budget:
    kind = rng.choice(list(gen.UNICODE_ALPHABETS))
    r = rng.random()
    if not big and not long_words and rng.random() < 0.25:  # the reference's own stress generator (stress_test.cpp:272-311), longer
        text = ("\n".join(gen.stress_text(rng, rng.randint(50, 1000), True) for _ in range(rng.randint(1, 12))) + "\n").encode()
        cov = 1.0
    elif r < 0.4:
        text = gen.unicode_text(rng, rng.randint(200, 6000), kind, p_invalid=0.02 if rng.random() < 0.3 else 0.0)
        cov = rng.choice([1.0, 1.0, 0.95, 0.9, 0.7])
        if b"\xff" in text or cov == 1.0 and any(b >= 0x80 for b in text) and rng.random() < 0.0:
            pass
        if cov == 1.0 and any(x in text for x in (b"\x80", b"\xbf", b"\xc0", b"\xe2", b"\xf0", b"\xff", b"\xed")) and kind == "ascii":
            cov = 0.9
    elif r < 0.7:
        text = gen.readme_corpus(rng.randint(20, 400), rng.randint(20, 120), rng.choice(["abcd ", "ab ", "abcdefgh  "]), seed=rng.randint(0, 10 ** 6))
        cov = 1.0
    else:
        text = gen.zipf_corpus(rng.randint(5000, 80000), vocab=rng.randint(50, 3000), seed=rng.randint(0, 10 ** 6))
        cov = 1.0
    vocab = rng.randint(30, 400)
    if rounds_mode:  # K4 alone: random batches, the whole pair table and word table against an oracle recount after every round; ids
        # around and beyond the 32 768 that fit the kernels' LDS flag bitmap
        try:
            S.check_merge_rounds(text, rounds=rng.randint(3, 60), seed=rng.randint(0, 10 ** 6), coverage=cov if cov in (1.0, 0.9) else 1.0,
                                 id_shift=rng.choice([0, 0, 32700, 32760, 40000, 1 << 20]))
        except Exception:
            open(tmp / f"FAIL_{n}.txt", "wb").write(text)
            print("FAIL rounds", n, kind, tmp, flush=True)
            raise
        n += 1
        continue
    if long_words:  # words of the tile classes B (257 .. 2048 tokens) and C (longer), many of them, merged far down: repacks of class B
        sigma = rng.choice(["ab", "abc", "abcd"])
        ws = []
        for _ in range(rng.randint(20, 260)):
            n_ch = rng.choice([rng.randint(250, 300), rng.randint(300, 700), rng.randint(1, 30), rng.randint(2040, 2060), rng.randint(2100, 5000)] if rng.random() < 0.15
                              else [rng.randint(256, 420), rng.randint(1, 40)])
            if rng.random() < 0.3:
                w = (rng.choice(sigma) * rng.randint(1, 5) + rng.choice(sigma) * rng.randint(1, 5)) * (n_ch // 4 + 1)
            else:
                w = "".join(rng.choice(sigma) for _ in range(n_ch))
            ws.append(w[:n_ch])
        text = (" ".join(ws) + "\n").encode() * rng.randint(1, 3)
        cov = 1.0
        vocab = rng.randint(100, 2500)
        # half of them in word mode from the second round: class-B tiles before k_words or (YTTM_CLASSB_BESIDE) beside it on a second stream, the tail waiting for their flag
        for k in ("YTTM_WORD_MIN_TILES", "YTTM_WORD_MIN_TOKENS", "YTTM_WORD_DIV", "YTTM_CLASSB_BESIDE"):
            os.environ.pop(k, None)
        if rng.random() < 0.5:
            os.environ.update({"YTTM_WORD_MIN_TILES": "0", "YTTM_WORD_MIN_TOKENS": "0", "YTTM_WORD_DIV": "0"})
            if rng.random() < 0.5:
                os.environ["YTTM_CLASSB_BESIDE"] = "1"
    if big:  # larger tables (several tiles, repacks, the pair index) and the tuning hooks that force the rare paths
        if r >= 0.7:
            text = gen.zipf_corpus(rng.randint(60000, 400000), vocab=rng.randint(500, 20000), seed=rng.randint(0, 10 ** 6))
        vocab = rng.randint(200, 3000)
        for k in ("YTTM_K4_DIRECT", "YTTM_NO_FUSE", "YTTM_HOT_TARGET", "YTTM_HOT_MIN", "YTTM_HOT_CAP", "YTTM_TOP_TARGET", "YTTM_TOP_MIN", "YTTM_TOP_CAP",
                  "YTTM_WORD_TABLE_FULL"):
            os.environ.pop(k, None)
        hooks = rng.choice([{}, {"YTTM_K4_DIRECT": "0"}, {"YTTM_NO_FUSE": "1"},
                            {"YTTM_HOT_TARGET": "8", "YTTM_HOT_MIN": "3", "YTTM_HOT_CAP": "32"}, {"YTTM_TOP_TARGET": "4", "YTTM_TOP_MIN": "2", "YTTM_TOP_CAP": "16"},
                            {"YTTM_K4_DIRECT": "0", "YTTM_TOP_TARGET": "16", "YTTM_TOP_MIN": "4", "YTTM_TOP_CAP": "64"}])
        os.environ.update(hooks)
        if words_mode:
            for k in ("YTTM_WORD_LOG", "YTTM_WORD_DREC", "YTTM_WORDS_INLINE_MAX", "YTTM_INDEX_AGG_MIN", "YTTM_WORDS_WPI", "YTTM_WORDS_FUSE_MAX", "YTTM_WORDS_GRID"):
                os.environ.pop(k, None)
            os.environ.update({"YTTM_WORD_MIN_TILES": "0", "YTTM_WORD_MIN_TOKENS": "0", "YTTM_WORD_DIV": rng.choice(["0", "0", "2", "8"])})
            os.environ.pop("YTTM_K4_DIRECT", None)
            os.environ.update(rng.choice([{}, {"YTTM_WORD_LOG": str(rng.choice([50, 300, 2000]))}, {"YTTM_WORD_DREC": str(rng.choice([8, 64]))},
                                          {"YTTM_WORDS_INLINE_MAX": "0"}, {"YTTM_INDEX_AGG_MIN": "0"}, {"YTTM_WORDS_FUSE_MAX": "0"},
                                          {"YTTM_WORDS_FUSE_MAX": str(rng.choice([4200, 4500, 6000]))}, {"YTTM_WORDS_GRID": str(rng.choice([1, 2, 5]))},
                                          {"YTTM_WORDS_FUSE_MAX": "4400", "YTTM_WORD_LOG": "500", "YTTM_WORDS_GRID": "2"},
                                          {"YTTM_WORD_LOG": "200", "YTTM_WORD_DREC": "16", "YTTM_WORDS_INLINE_MAX": "0", "YTTM_INDEX_AGG_MIN": "0"}]))
    if words_mode and rng.random() < 0.1:  # batches of hundreds of disjoint rules (the trainer's batch split)
        nw = rng.randint(130, 400)
        text, cov, vocab = gen.disjoint_words_corpus(nw, shuffle_seed=rng.randint(0, 10 ** 6)), 1.0, 8 + 4 * nw + rng.randint(nw, 3 * nw)
    # round 5: a third of the corpora cross the device in chunks of 4 / 8 / 16 KB (the chunked front end: lexicon, table growth, the second pass
    # when coverage drops chars), some of the three-byte kind (K1 / K2a's window classification against the exact path)
    os.environ.pop("YTTM_FE_CHUNK_KB", None)
    if rng.random() < 0.33:  # (a word longer than a chunk is refused, loudly: the `long` mode's words have up to 5 000 chars)
        os.environ["YTTM_FE_CHUNK_KB"] = rng.choice(["8", "16"] if long_words else ["4", "8", "16"])
    if not big and rng.random() < 0.15:
        text, cov = S.three_byte_text(rng, rng.choice([300, 5000, 30000])), rng.choice([1.0, 1.0, 0.9])
        if not text.strip():
            text = b"ab " + text
    ids = rng.choice([(0, 1, 2, 3), (3, 2, 1, 0), (-1, 0, -1, -1), (5, 7, -1, 2)])
    try:
        model = S.check_train_vs_oracle(text, vocab, tmp, cov, ids, tag=f"s{n}")
    except Exception:
        open(tmp / f"FAIL_{n}.txt", "wb").write(text)
        print("FAIL train", n, kind, vocab, cov, ids, tmp, {k: v for k, v in os.environ.items() if k.startswith("YTTM_")}, flush=True)
        raise
    if model and rng.random() < 0.5:
        sents = [gen.unicode_text(rng, rng.randint(0, 80), kind).decode(errors="ignore").replace("\n", " ") for _ in range(12)] + ["", " "]
        for cache in ("0", "1"):
            os.environ["YTTM_ENCODE_CACHE"] = cache
            S.check_encode_vs_oracle(model, sents)
    n += 1
print("soak ok:", n, "corpora in %.0f s" % (time.time() - t0), flush=True)
