"""`python -m youtokentome_amd.yttm_cli` keeps the reference CLI's commands, flags and output formats
(reference: tests/unit_tests/test_cli.py).  Kernels run under the HIP emulator here and on the GPU with -m gpu."""
import os
import subprocess
import sys

import pytest

import gen
import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_cli(args, stdin=b"", env=None):
    e = dict(os.environ, PYTHONPATH=ROOT)
    e.update(env or {})
    r = subprocess.run([sys.executable, "-m", "youtokentome_amd.yttm_cli"] + args, input=stdin, capture_output=True, env=e)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    return r.stdout


def _roundtrip(tmp_path):
    text = gen.readme_corpus(120, 80)
    test = gen.readme_corpus(20, 60, "abcde ", seed=3)
    corpus, model = str(tmp_path / "c.txt"), str(tmp_path / "m.model")
    open(corpus, "wb").write(text)
    run_cli(["bpe", f"--data={corpus}", f"--model={model}", "--vocab_size=300", "--coverage=0.999", "--bos_id=29", "--eos_id=148",
             "--unk_id=292"])
    m_ora = str(tmp_path / "o.model")
    O.train(text, m_ora, 300, 0.999, 0, 292, 29, 148)
    assert open(model, "rb").read() == open(m_ora, "rb").read()
    out = run_cli(["encode", f"--model={model}", "--output_type=id", "--bos", "--eos"], test)
    want = O.Model(m_ora).encode(test.split(b"\n")[:-1], True, True)
    assert out.decode() == "".join("".join(f"{t} " for t in row) + "\n" for row in want)
    assert out.startswith(b"29 ")
    sub = run_cli(["encode", f"--model={model}", "--output_type=subword", "--stream", "--reverse", "--eos"], test)
    assert sub.startswith(b"<EOS> ")
    dec = run_cli(["decode", f"--model={model}", "--ignore_ids=29,148"], out)
    import re
    want_dec = [" ".join(re.sub("e+", "<UNK>", w) for w in ln.split()) for ln in test.decode().split("\n")[:-1]]
    assert dec.decode().split("\n")[:-1] == want_dec  # a run of unknown chars decodes to one <UNK>
    voc = run_cli(["vocab", f"--model={model}", "--verbose"]).decode().split("\n")[:-1]
    assert len(voc) == 300 and voc[292].split("\t")[1] == "<UNK>" and "+" in voc[299]
    # several batches through the two-lane pipeline (reader -> 2 workers -> ordered writer): same bytes as one batch
    many = run_cli(["encode", f"--model={model}", "--output_type=id", "--bos", "--eos"], test, env={"YTTM_CLI_BATCH_BYTES": "150"})
    assert many == out
    many = run_cli(["encode", f"--model={model}", "--output_type=subword"], test, env={"YTTM_CLI_BATCH_BYTES": "61"})
    assert many == run_cli(["encode", f"--model={model}", "--output_type=subword", "--stream"], test)
    # the same through the batch encoder's word cache (batches of 8 MB and more take it by themselves)
    assert run_cli(["encode", f"--model={model}", "--output_type=id", "--bos", "--eos"], test, env={"YTTM_ENCODE_CACHE": "1"}) == out
    assert run_cli(["encode", f"--model={model}", "--output_type=subword"], test, env={"YTTM_ENCODE_CACHE": "1", "YTTM_CLI_BATCH_BYTES": "400"}) == many
    # the loops are bytes end to end: invalid UTF-8, a last line without newline, empty lines, CR
    odd = b"ab\xffcd ef\n\n  \nabc\r\n\xe2\x82 ab\xc0\x80cd\nlast line without newline ab"
    got = run_cli(["encode", f"--model={model}", "--output_type=id"], odd)
    want = O.Model(m_ora).encode(odd.split(b"\n"), False, False)
    assert got.decode() == "".join("".join(f"{t} " for t in row) + "\n" for row in want)
    assert run_cli(["encode", f"--model={model}", "--output_type=id"], odd, env={"YTTM_ENCODE_CACHE": "1"}) == got
    import refbin
    if refbin.available("prod"):  # the unmodified reference's CLI loop output for the same bytes
        lines = str(tmp_path / "odd.txt")
        open(lines, "wb").write(odd)
        ref_ids = refbin.encode(model, lines, n_threads=1)
        assert got.decode() == "".join("".join(f"{t} " for t in row) + "\n" for row in ref_ids)
        ref_sub = refbin.encode(model, lines, n_threads=1, subword=True)
        got_sub = run_cli(["encode", f"--model={model}", "--output_type=subword"], odd)
        assert got_sub == "".join("".join(f"{t} " for t in row) + "\n" for row in ref_sub).encode("utf-8", "surrogateescape")


def test_cli_roundtrip_emulated(tmp_path, sim_lib):
    _roundtrip(tmp_path)


@pytest.mark.gpu
def test_cli_roundtrip_gpu(tmp_path):
    _roundtrip(tmp_path)
