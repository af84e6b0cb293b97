"""Seeded corpus generators used by tests and bench.  Test/bench fixture code."""
import random

import numpy as np


def readme_corpus(n_lines=10000, n_chars=100, alphabet="abcd ", seed=19):
    """C1: exactly tests/unit_tests/utils_for_testing.py:23-36 of the reference."""
    random.seed(seed)
    out = []
    for _ in range(n_lines):
        out.append("".join([random.choice(alphabet) for _ in range(n_chars)]))
    return ("\n".join(out) + "\n").encode()


def abcd_corpus(nbytes, seed=19, line=100, alphabet=b"abcd ", survey_stream=False):
    """C2 / C4 family: uniform over the alphabet, `line` chars per row + newline (numpy default_rng), generated in row
    chunks so that 1 GB needs ~1 GB of host memory.  survey_stream=False draws uint8 (fast; what the tests and their golden
    files use).  survey_stream=True draws int64 like SURVEY.md Appendix C's gen_abcd -- chunking does not change that
    stream -- and reproduces its files byte for byte: 999 999 990 B, seed 19 -> md5 63857720bbc611c198a8e0a3d609fbcc (C2),
    99 999 999 B -> f35ed06647d5646121ac66dae1071cd0; bench.py uses this."""
    rng = np.random.default_rng(seed)
    alpha = np.frombuffer(alphabet, dtype=np.uint8)
    nlines = max(1, nbytes // (line + 1))
    out = np.empty((nlines, line + 1), dtype=np.uint8)
    step = 1 << 16 if survey_stream else 1 << 20
    for r0 in range(0, nlines, step):
        r1 = min(nlines, r0 + step)
        if survey_stream:
            out[r0:r1, :line] = alpha[rng.integers(0, len(alpha), size=(r1 - r0, line))]
        else:
            out[r0:r1, :line] = alpha[rng.integers(0, len(alpha), size=(r1 - r0, line), dtype=np.uint8)]
    out[:, line] = 10
    return out.tobytes()


def zipf_corpus(nbytes, seed=7, vocab=20000, line_words=16, exponent=1.05):
    """C3 family: Zipfian word ids over a synthetic lowercase lexicon."""
    rng = np.random.default_rng(seed)
    letters = np.frombuffer(b"etaoinshrdlcumwfgypbvkjxqz", dtype=np.uint8)
    p = np.array([12.7, 9.1, 8.2, 7.5, 7.0, 6.7, 6.3, 6.1, 6.0, 4.3, 4.0, 2.8, 2.8, 2.4, 2.4, 2.2, 2.0, 2.0, 1.9,
                  1.5, 1.0, 0.8, 0.15, 0.15, 0.1, 0.07])
    p /= p.sum()
    lens = np.clip(rng.poisson(5.5, size=vocab) + 1, 1, 20)
    lex = [bytes(letters[rng.choice(26, size=l, p=p)]) for l in lens]
    ranks = np.arange(1, vocab + 1)
    w = 1 / ranks ** exponent
    w /= w.sum()
    avg = float((w * (lens + 1)).sum())
    nwords = max(1, int(nbytes / avg))
    ids = rng.choice(vocab, size=nwords, p=w)
    toks = [lex[i] for i in ids]
    lines = [b" ".join(toks[i:i + line_words]) for i in range(0, nwords, line_words)]
    return b"\n".join(lines) + b"\n"


def _zipf_chunks(nbytes, seed=7, vocab=400000, line_words=16, exponent=1.05, chunk_words=4_000_000):
    """The byte chunks of zipf_corpus_fast, in order (the stream is defined chunk by chunk: the draws are chunked)."""
    rng = np.random.default_rng(seed)
    letters = np.frombuffer(b"etaoinshrdlcumwfgypbvkjxqz", dtype=np.uint8)
    p = np.array([12.7, 9.1, 8.2, 7.5, 7.0, 6.7, 6.3, 6.1, 6.0, 4.3, 4.0, 2.8, 2.8, 2.4, 2.4, 2.2, 2.0, 2.0, 1.9,
                  1.5, 1.0, 0.8, 0.15, 0.15, 0.1, 0.07])
    p /= p.sum()
    lens = np.clip(rng.poisson(5.5, size=vocab) + 1, 1, 20).astype(np.int64)
    lex = letters[rng.choice(26, size=(vocab, 20), p=p)]  # word i = lex[i, :lens[i]]
    ranks = np.arange(1, vocab + 1)
    w = 1 / ranks ** exponent
    w /= w.sum()
    cdf = np.cumsum(w)
    cdf[-1] = 1.0
    avg = float((w * (lens + 1)).sum())
    nwords = max(1, int(nbytes / avg))
    nwords -= nwords % line_words  # whole lines
    nwords = max(line_words, nwords)
    col = np.arange(21, dtype=np.int64)
    for w0 in range(0, nwords, chunk_words):
        n = min(chunk_words, nwords - w0)
        ids = np.searchsorted(cdf, rng.random(n), side="right").astype(np.int64)
        np.minimum(ids, vocab - 1, out=ids)
        l = lens[ids]
        # each word contributes its letters + one separator (space, or newline after every line_words-th word)
        mat = np.empty((n, 21), dtype=np.uint8)
        mat[:, :20] = lex[ids]
        sep = np.where((np.arange(w0, w0 + n) % line_words) == line_words - 1, 10, 32).astype(np.uint8)
        mat[np.arange(n), l] = sep
        keep = col[None, :] <= l[:, None]
        yield mat[keep].tobytes()


def zipf_corpus_fast(nbytes, seed=7, vocab=400000, line_words=16, exponent=1.05, chunk_words=4_000_000):
    """C3 (BASELINE.json configs[2]): the same Zipfian lexicon text as zipf_corpus, generated with array operations in fixed
    chunks of `chunk_words` words so that 1 GB takes seconds, not minutes (bench.py --corpus zipf and the full-size pins
    of tests/golden/full_size_pins.json use THIS stream; its bytes differ from zipf_corpus's because the draws are chunked)."""
    return b"".join(_zipf_chunks(nbytes, seed, vocab, line_words, exponent, chunk_words))


def stream_to_file(path, chunks):
    """Writes an iterable of byte chunks to `path`; returns (bytes written, md5 of the file) -- corpora beyond host-memory comfort."""
    import hashlib
    h = hashlib.md5()
    n = 0
    with open(path, "wb") as f:
        for c in chunks:
            f.write(c)
            h.update(c)
            n += len(c)
    return n, h.hexdigest()


def zipf_corpus_fast_to_file(path, nbytes, **kw):
    """zipf_corpus_fast(nbytes, ...) written straight to a file (the 8 GB corpus of the > 4 GiB pin: never whole in host memory)."""
    return stream_to_file(path, _zipf_chunks(nbytes, **kw))


def heavy_word_chunks(reps, seed=3):
    """A corpus in which ONE word occurs more than 2^32 times (the reference counts word frequencies in uint64, bpe.cpp:382-385): two
    megabytes of Zipf text, then `reps` times "a " (a line break every 2^20 words)."""
    yield zipf_corpus_fast(2_000_000, seed=seed, vocab=5000)
    blk = b"a " * ((1 << 20) - 1) + b"a\n"
    full, rest = divmod(reps, 1 << 20)
    for _ in range(full):
        yield blk
    if rest:
        yield b"a " * (rest - 1) + b"a\n"


def cjk_corpus_fast(nbytes, seed=11, n_chars=4096, lexicon=300000, chunk_words=4_000_000):
    """A large-alphabet corpus in the shape of Chinese / Japanese text (the reference's slowest published cases, benchmark.md:23,39):
    `n_chars` ideographs (U+4E00 ..., three UTF-8 bytes each, Zipfian), a lexicon of 1-3 char words (Zipfian), written WITHOUT spaces
    between words; a clause ends after about ten words with a comma (no space) or a full stop followed by a space or a newline -- so the
    space-delimited "words" the trainer sees are whole sentences of some tens of chars, nearly all of them unique."""
    rng = np.random.default_rng(seed)
    cw = 1 / np.arange(1, n_chars + 1) ** 0.9
    cw /= cw.sum()
    lens = rng.choice([1, 2, 3], size=lexicon, p=[0.3, 0.5, 0.2]).astype(np.int64)
    cp = 0x4E00 + rng.choice(n_chars, size=(lexicon, 3), p=cw)
    lexb = np.empty((lexicon, 9), dtype=np.uint8)  # word i = lexb[i, :3 * lens[i]]
    for k in range(3):
        c = cp[:, k]
        lexb[:, 3 * k] = 0xE0 | (c >> 12)
        lexb[:, 3 * k + 1] = 0x80 | ((c >> 6) & 0x3F)
        lexb[:, 3 * k + 2] = 0x80 | (c & 0x3F)
    w = 1 / np.arange(1, lexicon + 1) ** 1.0
    w /= w.sum()
    cdf = np.cumsum(w)
    cdf[-1] = 1.0
    # separators behind a word: nothing (0.9), "," U+FF0C (0.05), "." U+3002 + space (0.04), "." + newline (0.01)
    seps = np.zeros((4, 4), dtype=np.uint8)
    seps[1, :3] = (0xEF, 0xBC, 0x8C)
    seps[2] = (0xE3, 0x80, 0x82, 32)
    seps[3] = (0xE3, 0x80, 0x82, 10)
    sep_len = np.array([0, 3, 4, 4], dtype=np.int64)
    avg = float((w * 3 * lens).sum()) + 0.05 * 3 + 0.05 * 4
    nwords = max(1, int(nbytes / avg))
    parts = []
    col = np.arange(13, dtype=np.int64)
    for w0 in range(0, nwords, chunk_words):
        n = min(chunk_words, nwords - w0)
        ids = np.searchsorted(cdf, rng.random(n), side="right").astype(np.int64)
        np.minimum(ids, lexicon - 1, out=ids)
        kind = np.searchsorted(np.array([0.9, 0.95, 0.99, 1.0]), rng.random(n), side="right").astype(np.int64)
        np.minimum(kind, 3, out=kind)
        if w0 + n >= nwords:
            kind[-1] = 3  # the text ends with a full stop and a newline
        l = 3 * lens[ids]
        mat = np.zeros((n, 13), dtype=np.uint8)
        mat[:, :9] = lexb[ids]
        rows = np.arange(n)
        for k in range(4):
            mat[rows, l + k] = np.where(k < sep_len[kind], seps[kind, k], mat[rows, np.minimum(l + k, 12)])
        keep = col[None, :] < (l + sep_len[kind])[:, None]
        parts.append(mat[keep].tobytes())
    return b"".join(parts)


def disjoint_words_corpus(n_words=200, first_cp=0x4E00, shuffle_seed=None):
    """Word i = four fresh code points a b c d; the text holds "a b" three times, "a b c" 2 .. 4 times and "a b c d" once or twice.  The
    first merge round takes every (a, b) at once, the second every (ab, c): batches of n_words disjoint rules -- the batch sizes a
    corpus of natural or random text only reaches at gigabytes (host_trainer.cpp: batches of 129 .. 256 rules in word mode are cut in two)."""
    parts = []
    for i in range(n_words):
        a, b, c, d = (chr(first_cp + 4 * i + j) for j in range(4))
        parts.append((a + b + " ") * 3 + (a + b + c + " ") * (2 + i % 3) + (a + b + c + d + " ") * (1 + i % 2))
    if shuffle_seed is not None:
        random.Random(shuffle_seed).shuffle(parts)
    return "".join(parts).encode()


def stress_text(rng: random.Random, n_limit=1000, train=True):
    """Random text in the spirit of the reference stress generator (tests/unit_tests/stress_test.cpp:272-311):
    short alphabet, single chars mixed with repeated segments so that long runs of equal symbols occur."""
    sigma = "abc " if train else "abcd "
    n = min(rng.randint(1, 1000), n_limit)
    s = [sigma[0]]
    while len(s) < n:
        if rng.randint(0, 1):
            s.append(rng.choice(sigma))
        else:
            rep = rng.randint(2, 6)
            seg = [rng.choice(sigma) for _ in range(rng.randint(1, 4))]
            s.extend(seg * rep)
    s = s[:n]
    while s and s[-1] == " ":
        s.pop()
    while len(s) < n:
        s.append(sigma[0])
    return "".join(s)


UNICODE_ALPHABETS = {
    "ascii": "abcdefghij  ",
    "cyr": "абвгдежзик  ",
    "cjk": "日本語山川木気水火  ",
    "mix": "aбc日😀é ß\t\n  ",
}


def unicode_text(rng: random.Random, n, kind="mix", p_run=0.2, p_invalid=0.0):
    """Random UTF-8 bytes over a small multi-script alphabet, with symbol runs and optional invalid bytes."""
    sigma = UNICODE_ALPHABETS[kind]
    out = bytearray()
    i = 0
    while i < n:
        ch = rng.choice(sigma)
        k = rng.randint(2, 7) if rng.random() < p_run else 1
        out += (ch * k).encode()
        if p_invalid and rng.random() < p_invalid:
            out += bytes([rng.choice([0x80, 0xBF, 0xC0, 0xE2, 0xF0, 0xFF, 0xED])])
        i += k
    return bytes(out)
