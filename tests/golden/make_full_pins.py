"""Writes tests/golden/full_size_pins.json: md5 pins of the FULL-SIZE workloads of BASELINE.json, produced by the unmodified
reference (oracle/_ref/yttm_ref_det = bpe.cpp with -DDETERMINISTIC_QUEUE, n_threads=8; encode: oracle/_ref/yttm_ref_prod).

  c2_1gb    configs[1]: 1 GB random 'abcd ' corpus (SURVEY.md Appendix C gen_abcd, seed 19), vocab 32000 -> model md5
  c2_100mb  its 100 MB variant
  c3_1gb    configs[2]: 1 GB Zipf ASCII corpus (tests/gen.py zipf_corpus_fast, seed 7, lexicon 400 000), vocab 32000 -> model md5
  c3_100mb  its 100 MB variant
  c5        configs[4]: histograms (sentence lengths, unigram ids) of the reference's BPE-dropout output, p = 0.1, n_threads=1, on the
            first 1 M sentences of the C4 stream with the c2_1gb model -> tests/golden/c5_dropout_pin.json
  c4_10m    configs[3]: 10 M sentences of 128 chars (gen_abcd stream, seed 123) encoded with the c2_1gb model -> FNV-1a-64
            of (length, ids...) per sentence over all 10 M, and over the first 1 M

  c3_8gb         the C3 stream at 8 GB (more than 2^32 bytes)
  c8_heavy_word  a corpus in which one word occurs 4.4e9 times (more than 2^32)

bench.py regenerates the same corpora from the same seeds on the GPU box, compares the md5 of what it generated and of
what the GPU produced with these pins, prints the verdict in its JSON line and exits non-zero on a mismatch.
Run in the build container (needs /root/reference for oracle/_ref): python tests/golden/make_full_pins.py [name ...]"""
import hashlib
import json
import os
import sys
import tempfile
import time

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, "tests"))
import gen  # noqa: E402
import refbin  # noqa: E402

OUT = os.path.join(R, "tests", "golden", "full_size_pins.json")
md5f = lambda p: hashlib.md5(open(p, "rb").read()).hexdigest()  # noqa: E731
REF = "oracle/_ref/yttm_ref_det (unmodified bpe.cpp, -DDETERMINISTIC_QUEUE), n_threads=8"


def train_pin(name, text, desc, d):
    corpus = os.path.join(d, name + ".txt")
    open(corpus, "wb").write(text)
    model = os.path.join(d, name + ".model")
    t0 = time.time()
    refbin.train(corpus, model, 32000, n_threads=8, kind="det")
    dt = time.time() - t0
    os.remove(corpus)
    return {"corpus": desc, "corpus_bytes": len(text), "corpus_md5": hashlib.md5(text).hexdigest(), "vocab_size": 32000,
            "model_md5": md5f(model), "model_bytes": os.path.getsize(model), "reference": REF,
            "reference_train_seconds_build_container": round(dt, 1)}, model


def main():
    want = set(sys.argv[1:])
    pins = json.load(open(OUT)) if os.path.exists(OUT) else {}
    assert refbin.available("det") and refbin.available("prod")
    d = tempfile.mkdtemp(prefix="yttm_pins_")
    c2_model = None
    for name, nbytes in (("c2_100mb", 100_000_000), ("c2_1gb", 1_000_000_000)):
        if want and name not in want and not (name == "c2_1gb" and ("c4_10m" in want or "c5" in want)):
            continue
        text = gen.abcd_corpus(nbytes, seed=19, survey_stream=True)
        pins[name], m = train_pin(name, text, f"SURVEY.md Appendix C gen_abcd(seed=19), {len(text)//101} rows of 100 chars", d)
        if name == "c2_1gb":
            c2_model = m
        print(name, pins[name], flush=True)
        json.dump(pins, open(OUT, "w"), indent=1)
    for name, nbytes in (("c3_100mb", 100_000_000), ("c3_1gb", 1_000_000_000)):
        if want and name not in want:
            continue
        text = gen.zipf_corpus_fast(nbytes, seed=7, vocab=400000)
        pins[name], _ = train_pin(name, text, "tests/gen.py zipf_corpus_fast(seed=7, vocab=400000, exponent=1.05, 16 words per line)", d)
        print(name, pins[name], flush=True)
        json.dump(pins, open(OUT, "w"), indent=1)
    # SURVEY.md 8a / the round-2 verdict: workloads between and beyond the two BASELINE corpora -- a large alphabet (the reference's slowest
    # published cases are zh / ja) and an enwik-like Zipf instance (lexicon 4 * 10^6, exponent 1.0: U ~ 2-3 * 10^6, T ~ 2-3 * 10^7)
    for name, make, desc in (("c6_cjk_1gb", lambda: gen.cjk_corpus_fast(1_000_000_000, seed=11), "tests/gen.py cjk_corpus_fast(seed=11, n_chars=4096, lexicon=300000)"),
                             ("c7_zipf4m_1gb", lambda: gen.zipf_corpus_fast(1_000_000_000, seed=7, vocab=4_000_000, exponent=1.0),
                              "tests/gen.py zipf_corpus_fast(seed=7, vocab=4000000, exponent=1.0, 16 words per line)"),
                             ("c6_cjk_100mb", lambda: gen.cjk_corpus_fast(100_000_000, seed=11), "tests/gen.py cjk_corpus_fast(seed=11, n_chars=4096, lexicon=300000)")):
        if name not in want:
            continue
        text = make()
        pins[name], _ = train_pin(name, text, desc, d)
        print(name, pins[name], flush=True)
        json.dump(pins, open(OUT, "w"), indent=1)
    # Beyond 2^32: a corpus of more than 4 GiB (every byte offset, segment number and token index past 32 bits) and a word that occurs
    # more than 2^32 times (the reference counts word frequencies in uint64, bpe.cpp:382-385).  Streamed to a file, never whole in
    # host memory here; the reference itself reads the file into one std::string (bpe.cpp:67-84).
    for name, chunks, desc in (("c3_8gb", lambda: gen._zipf_chunks(8_000_000_000, seed=7, vocab=400000), "tests/gen.py zipf_corpus_fast_to_file(8e9, seed=7, vocab=400000): the C3 stream, 8 GB"),
                               ("c8_heavy_word", lambda: gen.heavy_word_chunks(4_400_000_000), "tests/gen.py heavy_word_chunks(4 400 000 000): 2 MB of Zipf text, then the word 'a' 4.4e9 times")):
        if name not in want:
            continue
        corpus = os.path.join(d, name + ".txt")
        nbytes, md5 = gen.stream_to_file(corpus, chunks())
        model = os.path.join(d, name + ".model")
        t0 = time.time()
        refbin.train(corpus, model, 32000, n_threads=8, kind="det")
        dt = time.time() - t0
        os.remove(corpus)
        pins[name] = {"corpus": desc, "corpus_bytes": nbytes, "corpus_md5": md5, "vocab_size": 32000, "model_md5": md5f(model),
                      "model_bytes": os.path.getsize(model), "reference": REF, "reference_train_seconds_build_container": round(dt, 1)}
        print(name, pins[name], flush=True)
        json.dump(pins, open(OUT, "w"), indent=1)
    if not want or "c4_10m" in want:
        assert c2_model is not None
        line = 128
        host = gen.abcd_corpus(10_000_000 * (line + 1), seed=123, line=line, survey_stream=True)
        lines = os.path.join(d, "c4.txt")
        open(lines, "wb").write(host)
        full = refbin.encode_bench(c2_model, lines, n_threads=8, max_lines=-1)
        first = refbin.encode_bench(c2_model, lines, n_threads=8, max_lines=1_000_000)
        os.remove(lines)
        pins["c4_10m"] = {"sentences": "gen_abcd stream: default_rng(123), 10 000 000 rows of 128 chars over 'abcd '", "input_md5": hashlib.md5(host).hexdigest(),
                          "model": "c2_1gb", "model_md5": pins["c2_1gb"]["model_md5"], "n_sentences": full["sentences"], "n_ids": full["ids"],
                          "fnv1a64": full["fnv1a64"], "first_1m": {"n_ids": first["ids"], "fnv1a64": first["fnv1a64"]},
                          "hash": "FNV-1a-64 over, per sentence, the little-endian bytes of uint32 length then of each int32 id (oracle/ref_driver.cpp encode_bench)",
                          "reference": "oracle/_ref/yttm_ref_prod encode_as_ids, n_threads=8, dropout 0"}
        print("c4_10m", pins["c4_10m"], flush=True)
    if "c5" in want:
        # configs[4]: BPE-dropout p = 0.1 on the first 10^6 sentences of the C4 stream with the c2_1gb model -- the reference as shipped,
        # n_threads=1, a fresh process (one global mt19937, seed 5489: bpe.cpp:1415; with threads its output is a data race).
        # The fixture holds the histograms (sentence lengths, unigram ids) bench.py compares its own 10 M-sentence output with.
        import subprocess
        assert c2_model is not None
        line, m = 128, 1_000_000
        host = gen.abcd_corpus(m * (line + 1), seed=123, line=line, survey_stream=True)
        lines = os.path.join(d, "c5.txt")
        open(lines, "wb").write(host)
        hist = os.path.join(d, "c5_hist.json")
        t0 = time.time()
        r = subprocess.run([os.path.join(R, "oracle", "_ref", "yttm_ref_prod"), "encode_hist", c2_model, lines, "1", "0.1", str(m), hist], capture_output=True, text=True, check=True)
        h = json.load(open(hist))
        os.remove(lines)
        pin = {"what": "BASELINE.json configs[4]: histograms of encode_as_ids(dropout_prob=0.1) over the first 1 000 000 sentences of the C4 stream (gen_abcd, default_rng(123), 128 chars)",
               "model": "c2_1gb", "model_md5": pins["c2_1gb"]["model_md5"], "input_md5_first_1m": hashlib.md5(host).hexdigest(), "dropout_prob": 0.1,
               "reference": "oracle/_ref/yttm_ref_prod (unmodified bpe.cpp as shipped), n_threads=1, fresh process (std::mt19937 seed 5489)",
               "reference_seconds_build_container": round(time.time() - t0, 1), "sentences": h["sentences"], "ids": h["ids"],
               "ids_per_sentence": round(h["ids"] / h["sentences"], 4), "len_hist": h["len_hist"], "id_hist": h["id_hist"]}
        json.dump(pin, open(os.path.join(R, "tests", "golden", "c5_dropout_pin.json"), "w"))
        print("c5", {k: v for k, v in pin.items() if k not in ("len_hist", "id_hist")}, flush=True)
        json.dump(pins, open(OUT, "w"), indent=1)


if __name__ == "__main__":
    main()
