"""CPU soak of the batch encoder: random sentences (every script, invalid bytes, cuts inside UTF-8 sequences, words of 1..60 chars, runs of one
letter) through the product sources under the HIP emulator against the oracle, with and without the word cache, under random settings of
the hooks that steer K5's paths (sentences per wavefront of the word cache's walks, one word per lane or the whole wave, a crowded short
region of the word table); a fifth of the rounds: BPE-dropout's differential check on random sentences.  usage: python tools/soak_encode.py [seconds] [seed]"""
import os, pathlib, random, sys, tempfile, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("YTTM_AMD_LIB", os.path.join(R, "tests", "hipsim", "_build", "libyttm_sim.so"))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import stage_checks as S

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = random.Random(seed)
tmp = pathlib.Path(tempfile.mkdtemp())
t0, n = time.time(), 0
devnull = os.open(os.devnull, os.O_WRONLY)
os.dup2(devnull, 2)  # (the trainers print the reference's progress lines)
HOOKS = {"YTTM_WC_SBLK": ["", "1", "2", "5", "64"], "YTTM_K5_LANE_WORDS": ["", "0", "3", "17", "960"], "YTTM_K5_LANE_SENT": ["", "0", "4", "30", "1000"],
         "YTTM_WC_SHORT_SLOTS": ["", "16", "1024"]}
while time.time() - t0 < budget:
    env = {}
    for k, vals in HOOKS.items():
        v = rng.choice(vals)
        os.environ.pop(k, None)
        if v:
            os.environ[k] = env[k] = v
    s = rng.randint(0, 10 ** 9)
    try:
        kind = rng.random()
        if kind < 0.2:  # BPE-dropout: the same ids however the batch is cut into packs, wherever the queues live, array or heap (round 5)
            S.check_dropout_heap_equals_array(seed=s)
        elif kind < 0.4:
            S.check_encode_word_cache_fuzz(tmp, trials=rng.randint(1, 3), seed=s)
        elif kind < 0.7:
            S.check_encode_word_cache(n_sent=rng.randint(5, 150), seed=s)
        else:
            S.check_encode_mixed_shapes(n_sent=rng.randint(10, 80), seed=s)
    except BaseException as e:
        print("MISMATCH seed %d round %d sub-seed %d hooks %s: %r" % (seed, n, s, env, e), flush=True)
        sys.exit(1)
    n += 1
print("encode soak ok: %d rounds in %.0f s" % (n, time.time() - t0), flush=True)
