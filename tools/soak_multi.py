"""CPU soak of the N>1 path: random corpora trained by 2-4 ranks (gloo, product sources under the HIP emulator) against the oracle on the
whole corpus.  usage: python tools/soak_multi.py [seconds] [seed] [big]"""
import filecmp, os, pathlib, random, sys, tempfile, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import gen
import oracle_lib as O
import test_multi_rank_gloo as M

lib = os.path.join(R, "tests", "hipsim", "_build", "libyttm_sim.so")
os.environ.setdefault("YTTM_WORD_HINT_FLOOR", "2048")  # (inherited by the ranks: smaller grids for the small word-mode rounds under the emulator)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
big = len(sys.argv) > 3 and sys.argv[3] == "big"
tmp = pathlib.Path(tempfile.mkdtemp())
t0, n = time.time(), 0
while time.time() - t0 < budget:
    r = rng.random()
    if r < 0.4:
        text = gen.unicode_text(rng, rng.randint(500, 6000), rng.choice(list(gen.UNICODE_ALPHABETS)))
    elif r < 0.7:
        text = gen.readme_corpus(rng.randint(20, 300), rng.randint(20, 120), rng.choice(["abcd ", "ab ", "abcdefgh  "]), seed=rng.randint(0, 10 ** 6))
    else:
        text = gen.zipf_corpus(rng.randint(5000, 60000), vocab=rng.randint(50, 2000), seed=rng.randint(0, 10 ** 6))
    vocab, world = rng.randint(30, 400), rng.choice([2, 2, 3, 4])
    if big:
        if r >= 0.7:
            text = gen.zipf_corpus(rng.randint(60000, 300000), vocab=rng.randint(500, 20000), seed=rng.randint(0, 10 ** 6))
        vocab = rng.randint(200, 2500)
    words = {"YTTM_WORD_MIN_TILES": "0", "YTTM_WORD_MIN_TOKENS": "0", "YTTM_WORD_DIV": "0", "YTTM_WORDS_GRID": "3", "YTTM_WGATHER_GRID": "2"}
    env = rng.choice([{}, {"YTTM_XCHG_BLK_MIN": "2", "YTTM_XCHG_MARGIN": "0.05"}, {"YTTM_XCHG_NOTES": "2"}, words, dict(words, YTTM_XCHG_NOTES="2", YTTM_XCHG_BLK_MIN="2", YTTM_XCHG_MARGIN="0.05"),
                      dict(words, YTTM_HOT_TARGET="8", YTTM_HOT_MIN="3", YTTM_HOT_CAP="32"), {"YTTM_HOT_TARGET": "8", "YTTM_HOT_MIN": "3", "YTTM_HOT_CAP": "32"},
                      {"YTTM_TOP_TARGET": "4", "YTTM_TOP_MIN": "2", "YTTM_TOP_CAP": "16"}, dict(words, YTTM_WORD_DREC="16", YTTM_WORD_LOG="200")])
    corpus, m_mp, m_ora = str(tmp / f"c{n}.txt"), str(tmp / f"mp{n}.model"), str(tmp / f"ora{n}.model")
    open(corpus, "wb").write(text)
    try:
        O.train(text, m_ora, vocab, 1.0)
    except ValueError:
        continue
    M.run_world(corpus, m_mp, vocab, 1.0, world, lib, env)
    if not filecmp.cmp(m_mp, m_ora, shallow=False):
        print("FAIL", n, world, vocab, env, corpus, flush=True)
        sys.exit(1)
    n += 1
print("multi-rank soak ok:", n, "corpora in %.0f s" % (time.time() - t0), flush=True)
