"""Generates the committed golden fixtures from the UNMODIFIED reference (oracle/_ref/yttm_ref_{det,prod}, built by
oracle/Makefile from /root/reference).  Run in the build container (needs /root/reference):

    python tests/golden/make_golden.py

Fixtures (small, committed):
  train_<name>.txt / train_<name>.args.json / train_<name>.model   corpus, config, det-reference model file
  encode_<name>.lines / encode_<name>.json                          sentences + reference ids for several flag sets
  dropout_<name>.json                                               reference ids at n_threads=1 (fresh process RNG)
The three reference golden texts of tests/unit_tests/test_manual.py (ru/en/ja) are included as corpora; their
expected SUBWORD lists pin the PRODUCTION tie order only and are recorded in
manual_expected.json for reporting, not gating.
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import gen  # noqa: E402
import refbin  # noqa: E402

RU = "\n        собирать cборник сборище отобранный сборщица \n        "
EN = ("\n        anachronism\n        synchronous  \n        chronology\n        chronic\n        chronophilia\n"
      "        chronoecological\n        chronocoulometry\n        ")
JA = ("\n        むかし、 むかし、 ある ところ に\n        おじいさん と おばあさん が いました。\n"
      "        おじいさん が 山（やま） へ 木（き） を きり に いけば、\n"
      "        おばあさん は 川（かわ） へ せんたく に でかけます。\n"
      "        「おじいさん、 はよう もどって きなされ。」\n        「おばあさん も き を つけて な。」\n"
      "        まい日（にち） やさしく いい あって でかけます \n    ")


def main():
    rng = random.Random(2024)
    trains = {
        "manual_ru": (RU.encode(), dict(vocab=50)),
        "manual_en": (EN.encode(), dict(vocab=200)),
        "manual_ja": (JA.encode(), dict(vocab=100)),
        "stress_manual": (b"baba baaab", dict(vocab=9)),  # stress_test.cpp:313-337
        "readme_small": (gen.readme_corpus(300, 100), dict(vocab=600)),
        "readme_rename": (gen.readme_corpus(300, 100), dict(vocab=600, coverage=0.999, bos=29, eos=148, unk=292)),
        "nopad": (gen.readme_corpus(200, 80), dict(vocab=300, pad=-1, unk=0, bos=-1, eos=-1)),
        "runs": (("aaaa aaaaa aaaaaaa abababab aabbaabb abcabcabc bbbbbb ab aaab baaa " * 5).encode(), dict(vocab=40)),
        "mix_cov": (gen.unicode_text(rng, 4000, "mix", p_invalid=0.02), dict(vocab=90, coverage=0.8)),
        "cjk": (gen.unicode_text(rng, 3000, "cjk"), dict(vocab=120)),
        "zipf": (gen.zipf_corpus(60000, vocab=2000), dict(vocab=1500)),
    }
    for i in range(8):
        t = gen.stress_text(rng, 1000, True).encode()
        trains[f"stress{i}"] = (t, dict(vocab=len(set(t.decode()) | {" "}) + 4 + rng.randint(0, 40),
                                        coverage=1.0 if i % 2 else 1 - rng.random() * 0.4))
    for name, (text, a) in trains.items():
        corpus = os.path.join(HERE, f"train_{name}.txt")
        with open(corpus, "wb") as f:
            f.write(text)
        args = dict(vocab=a["vocab"], coverage=a.get("coverage", 1.0), pad=a.get("pad", 0), unk=a.get("unk", 1),
                    bos=a.get("bos", 2), eos=a.get("eos", 3))
        with open(os.path.join(HERE, f"train_{name}.args.json"), "w") as f:
            json.dump(args, f)
        refbin.train(corpus, os.path.join(HERE, f"train_{name}.model"), args["vocab"], args["coverage"], 8,
                     args["pad"], args["unk"], args["bos"], args["eos"], kind="det")

    # encode fixtures
    enc_cases = {
        "readme_small": [rng.choice(["", " "]) + "".join(rng.choice("abcde  ") for _ in range(rng.randint(0, 80)))
                         for _ in range(200)] + ["", "   ", "e", "eee e", "abcd" * 30],
        "readme_rename": ["".join(rng.choice("abcde ") for _ in range(rng.randint(0, 60))) for _ in range(100)],
        "nopad": ["".join(rng.choice("abcdx ") for _ in range(rng.randint(0, 60))) for _ in range(100)] + ["xx", "x a"],
        "manual_ru": ["\t собранный собрание прибор", "сбор zzz сборник"],
        "manual_en": ["chronocline synchroscope ", "chronic\tpain ▁ok"],
        "manual_ja": [" おばあさん が  川 で せん ", "山川　やま"],
        "mix_cov": [gen.unicode_text(rng, rng.randint(1, 80), "mix").decode().replace("\n", " ") for _ in range(80)],
        "zipf": [ln.decode() for ln in gen.zipf_corpus(8000, seed=3, vocab=2000).split(b"\n") if ln],
    }
    for name, sents in enc_cases.items():
        model = os.path.join(HERE, f"train_{name}.model")
        lines = os.path.join(HERE, f"encode_{name}.lines")
        with open(lines, "wb") as f:
            f.write(("\n".join(sents) + "\n").encode())
        a = json.load(open(os.path.join(HERE, f"train_{name}.args.json")))
        out = {}
        for bos, eos, rev in [(0, 0, 0), (1, 1, 0), (0, 0, 1), (1, 1, 1)]:
            if (bos and a["bos"] == -1) or (eos and a["eos"] == -1):
                continue
            out[f"{bos}{eos}{rev}"] = refbin.encode(model, lines, 8, bos, eos, rev, kind="prod")
        out["subword_000"] = refbin.encode(model, lines, 1, subword=True, kind="prod")
        with open(os.path.join(HERE, f"encode_{name}.json"), "w") as f:
            json.dump(out, f, ensure_ascii=False)

    # dropout, n_threads=1, fresh process per call (std::mt19937 default seed, bpe.cpp:1415)
    for name in ["readme_small", "zipf"]:
        model = os.path.join(HERE, f"train_{name}.model")
        lines = os.path.join(HERE, f"encode_{name}.lines")
        out = {str(p): refbin.encode(model, lines, 1, dropout=p, kind="prod") for p in (0.1, 0.5, 1.0)}
        with open(os.path.join(HERE, f"dropout_{name}.json"), "w") as f:
            json.dump(out, f)

    manual = {
        "manual_ru": {"test": "\n        собранный собрание прибор\n        ",
                      "expected_prod": ["▁с", "обранный", "▁с", "об", "ран", "и", "е", "▁", "п", "р", "и", "бор"]},
        "manual_en": {"test": "chronocline synchroscope ",
                      "expected_prod": ["▁chrono", "c", "l", "i", "n", "e", "▁", "sy", "n", "ch", "r", "o", "s", "co", "p", "e"]},
        "manual_ja": {"test": " おばあさん が  川 で せん ",
                      "expected_prod": ["▁おばあさん", "▁が", "▁", "川", "▁", "で", "▁", "せ", "ん"]},
    }
    with open(os.path.join(HERE, "manual_expected.json"), "w") as f:
        json.dump(manual, f, ensure_ascii=False, indent=1)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
