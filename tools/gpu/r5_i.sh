#!/bin/bash
# round 5, call 9: GPU test-suite with the chunked front end; the 1 GB corpora in chunks (time, peak HBM, pins)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r5i_gputest.log
grep -n "passed\|failed\|rror" gpurun_out/r5i_gputest.log | head -5
python - <<'P' 2>&1 | grep -v "^id:\|^number of\|^model saved\|^Training\|^  \|^reading\|^learning\|^$"
import ctypes as C, hashlib, json, os, sys, time
sys.path.insert(0, "tests")
import gen
from youtokentome_amd import _lib
L = _lib.load()
pins = json.load(open("tests/golden/full_size_pins.json"))
for kind, pin in (("abcd", "c2_1gb"), ("zipf", "c3_1gb"), ("cjk", "c6_cjk_1gb")):
    text = {"abcd": lambda: gen.abcd_corpus(1_000_000_000, seed=19, survey_stream=True), "zipf": lambda: gen.zipf_corpus_fast(1_000_000_000, seed=7, vocab=400000), "cjk": lambda: gen.cjk_corpus_fast(1_000_000_000, seed=11)}[kind]()
    open("/tmp/ck.txt", "wb").write(text); del text
    for mb in ("0", "128", "256"):
        os.environ["YTTM_FE_CHUNK_MB"] = mb
        err, rep = C.create_string_buffer(2048), C.create_string_buffer(16384)
        best = 1e9
        for i in range(2):
            t = time.perf_counter()
            rc = L.yttm_train_bpe_ex(b"/tmp/ck.txt", b"/tmp/ck.model", 32000, 1.0, 8, 0, 1, 2, 3, 0, rep, 16384, err, 2048)
            best = min(best, time.perf_counter() - t)
            assert rc == 0, err.value
        r = json.loads(rep.value.decode())
        print(kind, "chunk MB", mb, "seconds %.4f" % best, "chunks", r["front_end_chunks"], "peak GB %.2f" % (r["peak_device_bytes"] / 1e9), "upload %.3f frontend %.3f merge %.3f" % (r["seconds_upload"], r["seconds_frontend"], r["seconds_merge"]),
              "pin", hashlib.md5(open("/tmp/ck.model", "rb").read()).hexdigest() == pins[pin]["model_md5"], flush=True)
P
