#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
YTTM_TRACE=1 python - <<'P' 2>&1 | grep "train_bpe:\|wall\|merge loop wall"
import ctypes as C, os, sys, time, json
R = os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import gen
text = gen.abcd_corpus(1_000_000_000, seed=19, survey_stream=True)
open("/tmp/up.txt", "wb").write(text)
from youtokentome_amd import _lib
L = _lib.load()
err, rep = C.create_string_buffer(2048), C.create_string_buffer(16384)
for i in range(4):
    t = time.perf_counter()
    rc = L.yttm_train_bpe_comm(b"/tmp/up.txt", b"/tmp/up.model", 32000, 1.0, 8, 0, 1, 2, 3, 0, 0, None, rep, 16384, err, 2048)
    w = time.perf_counter() - t
    r = json.loads(rep.value.decode())
    print("wall %.4f total %.4f upload %.4f frontend %.4f merge %.4f io %.4f" % (w, r["seconds_total"], r["seconds_upload"], r["seconds_frontend"], r["seconds_merge"], r["seconds_io"]), flush=True)
P
