"""Oracle vs the committed golden fixtures (tests/golden/, generated from the unmodified reference by
tests/golden/make_golden.py).  Runs anywhere -- needs neither /root/reference nor oracle/_ref."""
import filecmp
import glob
import json
import os

import pytest

import oracle_lib as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TRAIN = sorted(os.path.basename(p)[len("train_"):-len(".txt")] for p in glob.glob(os.path.join(G, "train_*.txt")))
ENC = sorted(os.path.basename(p)[len("encode_"):-len(".json")] for p in glob.glob(os.path.join(G, "encode_*.json")))


def read_lines(path):
    return open(path, "rb").read().split(b"\n")[:-1]


@pytest.mark.parametrize("name", TRAIN)
def test_train_model_bytes(name, tmp_path):
    a = json.load(open(os.path.join(G, f"train_{name}.args.json")))
    text = open(os.path.join(G, f"train_{name}.txt"), "rb").read()
    out = str(tmp_path / "m.model")
    O.train(text, out, a["vocab"], a["coverage"], a["pad"], a["unk"], a["bos"], a["eos"])
    assert filecmp.cmp(out, os.path.join(G, f"train_{name}.model"), shallow=False)


@pytest.mark.parametrize("name", ENC)
def test_encode_ids(name):
    m = O.Model(os.path.join(G, f"train_{name}.model"))
    sents = read_lines(os.path.join(G, f"encode_{name}.lines"))
    want = json.load(open(os.path.join(G, f"encode_{name}.json")))
    for key, ids in want.items():
        if key.startswith("subword"):
            continue
        bos, eos, rev = (int(c) for c in key)
        assert m.encode(sents, bos, eos, rev) == ids, (name, key)


@pytest.mark.parametrize("name", ["readme_small", "zipf"])
def test_dropout_bit_exact(name):
    m = O.Model(os.path.join(G, f"train_{name}.model"))
    sents = read_lines(os.path.join(G, f"encode_{name}.lines"))
    want = json.load(open(os.path.join(G, f"dropout_{name}.json")))
    for p, ids in want.items():
        O.rng_reset()
        assert m.encode(sents, dropout_prob=float(p)) == ids
