"""GPU soak of K4's word mode: random corpora of tens of megabytes (thousands of workgroups: what the CPU emulator cannot show -- the ordering of the
kernels' tickets and tails across XCDs) trained twice on the MI355X, with word mode under random switch rules / rare-path hooks and with the tiles to
the end (YTTM_WORD_MODE=0: the path pinned against the reference); the two model files must be the same bytes.
usage: python tools/soak_gpu_words.py [seconds] [seed] [kinds, e.g. cjk: long clauses -- class-B tiles beside k_words on a second stream]"""
import ctypes as C, filecmp, json, os, random, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import gen
import torch
from youtokentome_amd import _lib
L = _lib.load()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
HOOKS = ("YTTM_WORD_MODE", "YTTM_WORD_DIV", "YTTM_WORD_MIN_TOKENS", "YTTM_WORD_MIN_TILES", "YTTM_WORDS_INLINE_MAX", "YTTM_WORD_DREC", "YTTM_WORD_LOG",
         "YTTM_WORDS_FUSE_MAX", "YTTM_WORDS_GRID", "YTTM_NO_BATCH_SPLIT", "YTTM_INDEX_AGG_MIN", "YTTM_HOT_TARGET", "YTTM_HOT_TARGET_WORDS", "YTTM_NO_FUSE", "YTTM_NO_REFINE")


def train(d, vocab, out, env):
    for k in HOOKS:
        os.environ.pop(k, None)
    os.environ.update(env)
    err, rep = C.create_string_buffer(2048), C.create_string_buffer(16384)
    rc = L.yttm_train_bpe_from_device(C.c_void_p(d.data_ptr()), d.numel(), out.encode(), vocab, 1.0, 0, 1, 2, 3, 0, 0, rep, 16384, err, 2048)
    assert rc == 0, err.value
    return json.loads(rep.value.decode())


t0, n, wr, ar, fr = time.time(), 0, 0, 0, 0
while time.time() - t0 < budget:
    mb = rng.choice([3, 8, 20, 40, 60])
    kind = rng.choice(sys.argv[3].split(",") if len(sys.argv) > 3 else ["abcd", "abcd", "ab", "zipf", "zipfbig", "cjk", "disjoint"])
    seed = rng.randint(0, 10 ** 6)
    if kind == "abcd":
        text = gen.abcd_corpus(mb * 1_000_000, seed=seed, survey_stream=True)
    elif kind == "ab":
        text = gen.readme_corpus(mb * 10000, 100, "ab ", seed=seed) if mb <= 8 else gen.abcd_corpus(mb * 1_000_000, seed=seed)
    elif kind == "zipf":
        text = gen.zipf_corpus_fast(mb * 1_000_000, seed=seed, vocab=rng.choice([3000, 50000, 400000]))
    elif kind == "zipfbig":
        text = gen.zipf_corpus_fast(mb * 1_000_000, seed=seed, vocab=4_000_000, exponent=1.0)
    elif kind == "disjoint":  # batches of hundreds of disjoint rules (the trainer's batch split; a small corpus: tens of thousands of words)
        nw = rng.choice([200, 1000, 5000])
        text = gen.disjoint_words_corpus(nw, shuffle_seed=seed) * rng.choice([1, 50])
    else:
        text = gen.cjk_corpus_fast(mb * 1_000_000, seed=seed)
    vocab = rng.choice([8000, 32000] if kind == "cjk" else [300, 2000, 8000, 32000])  # (the CJK-shaped alphabet alone is 4096 chars)
    if kind == "disjoint":
        vocab = 8 + 4 * nw + rng.randint(nw, 3 * nw)
    d = torch.frombuffer(bytearray(text), dtype=torch.uint8).cuda()
    env = {"YTTM_WORD_MIN_TOKENS": "0", "YTTM_WORD_MIN_TILES": "0", "YTTM_WORD_DIV": rng.choice(["0", "4", "50", "200"])}
    env.update(rng.choice([{}, {}, {"YTTM_NO_BATCH_SPLIT": "1"}, {"YTTM_WORDS_INLINE_MAX": "0"}, {"YTTM_WORD_DREC": str(rng.choice([64, 1024]))}, {"YTTM_WORD_LOG": str(rng.choice([20000, 500000]))},
                           {"YTTM_WORDS_FUSE_MAX": "0"}, {"YTTM_WORDS_FUSE_MAX": "200000"}, {"YTTM_WORDS_FUSE_MAX": "200000", "YTTM_WORDS_GRID": str(rng.choice([1, 7, 64]))},
                           {"YTTM_INDEX_AGG_MIN": "0"}, {"YTTM_HOT_TARGET": "512", "YTTM_HOT_TARGET_WORDS": "1024"}, {"YTTM_NO_FUSE": "1"}, {"YTTM_NO_REFINE": "1"}]))
    r = train(d, vocab, "/tmp/sgw_w.model", env)
    train(d, vocab, "/tmp/sgw_t.model", {"YTTM_WORD_MODE": "0"})
    if not filecmp.cmp("/tmp/sgw_w.model", "/tmp/sgw_t.model", shallow=False):
        open("/tmp/sgw_FAIL.txt", "wb").write(text)
        print("FAIL", n, kind, mb, seed, vocab, env, r["word_rounds"], flush=True)
        sys.exit(1)
    n += 1
    wr += r["word_rounds"]
    ar += r["word_all_rounds"]
    fr += r["word_fused_rounds"]
    del d
print("gpu word-mode soak ok: %d corpora, %d word-mode rounds (%d of them one launch each, %d over every word) in %.0f s" % (n, wr, fr, ar, time.time() - t0), flush=True)
