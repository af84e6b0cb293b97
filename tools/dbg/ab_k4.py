"""GPU tuning aid: A/B runs of the trainer on the pinned corpora under different environment hooks, one process.
usage: python tools/dbg/ab_k4.py OUT.json [abcd|zipf|both] [MB] -- NAME:K=V,K=V NAME2: ...
Per variant: 2 trainings from the HBM-resident corpus (the second is reported): wall, rounds, per-family kernel ms (HIP events),
K4 ms by ranges of rounds, the model's md5 against tests/golden/full_size_pins.json."""
import ctypes as C, hashlib, json, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import gen, torch
out_path, which, mb = sys.argv[1], sys.argv[2], int(sys.argv[3])
variants = []
for a in sys.argv[5:]:
    name, _, kv = a.partition(":")
    variants.append((name, dict(x.split("=", 1) for x in kv.split(",") if x)))
pins = json.load(open(os.path.join(R, "tests", "golden", "full_size_pins.json")))
from youtokentome_amd import _lib
L = _lib.load()
res = {}
for kind in (["abcd", "zipf"] if which == "both" else which.split("+")):
    text = {"abcd": lambda: gen.abcd_corpus(mb * 1_000_000, seed=19, survey_stream=True), "zipf": lambda: gen.zipf_corpus_fast(mb * 1_000_000, seed=7, vocab=400000),
            "cjk": lambda: gen.cjk_corpus_fast(mb * 1_000_000, seed=11), "zipf4m": lambda: gen.zipf_corpus_fast(mb * 1_000_000, seed=7, vocab=4_000_000, exponent=1.0)}[kind]()
    pin = pins.get({"abcd": "c2_", "zipf": "c3_", "cjk": "c6_cjk_", "zipf4m": "c7_zipf4m_"}[kind] + ("1gb" if mb == 1000 else "%dmb" % mb), {})
    d = torch.frombuffer(bytearray(text), dtype=torch.uint8).cuda()
    del text
    for name, env in variants:
        saved = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        os.environ["YTTM_TRACE"] = "/tmp/ab.trace"
        err, rep = C.create_string_buffer(2048), C.create_string_buffer(16384)
        walls = []
        for i in range(2):
            t = time.time()
            rc = L.yttm_train_bpe_from_device(C.c_void_p(d.data_ptr()), d.numel(), b"/tmp/ab.model", 32000, 1.0, 0, 1, 2, 3, 0, 1, rep, 16384, err, 2048)
            walls.append(time.time() - t)
            assert rc == 0, err.value
        r = json.loads(rep.value.decode())
        md5 = hashlib.md5(open("/tmp/ab.model", "rb").read()).hexdigest()
        k4 = [float(l.split()[1]) for l in open("/tmp/ab.trace") if l.startswith("5 ")]
        segs = {}
        for a, b in ((0, 11), (11, 46), (46, 100), (100, 200), (200, 300), (300, 450), (450, 700), (700, 100000)):
            seg = k4[a:b]
            if seg:
                segs["%d-%d" % (a + 1, min(b, len(k4)))] = [round(sum(seg), 2), round(1e3 * sum(seg) / len(seg), 1)]
        row = {"wall_s": round(min(walls), 4), "rounds": r["rounds"], "seconds_merge": r["seconds_merge"], "seconds_frontend": r["seconds_frontend"],
               "gathered_rounds": r.get("gathered_rounds"), "word_rounds": r.get("word_rounds"), "word_switch_round": r.get("word_switch_round"), "word_all_rounds": r.get("word_all_rounds"), "word_fused_rounds": r.get("word_fused_rounds"), "index_builds": r.get("index_builds"), "repacks": r.get("repacks"), "rounds_exhausted": r.get("rounds_exhausted"), "batch_extensions": r.get("batch_extensions"), "batch_splits": r.get("batch_splits"),
               "cand_rescans": r.get("cand_rescans"), "top_refills": r.get("top_refills"), "hot_rebuilds": r.get("hot_rebuilds"),
               "kernels_ms": {k: [round(v["ms"], 3), v["launches"]] for k, v in r["kernels"].items() if v["launches"]}, "touched_tiles": r["touched_tiles"],
               "touched_tile_tokens": r["touched_tile_tokens"], "merge_sites": r["merge_sites"], "k4_launches": len(k4), "k4_ms_by_rounds[sum,avg_us]": segs,
               "model_md5": md5, "matches_pin": (md5 == pin.get("model_md5")) if pin else None, "env": env}
        res["%s/%s" % (kind, name)] = row
        print(kind, name, json.dumps(row), flush=True)
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        json.dump(res, open(out_path, "w"), indent=1)
    del d
    torch.cuda.empty_cache()
