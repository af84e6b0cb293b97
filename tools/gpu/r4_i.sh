#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
nproc; grep -m1 "model name" /proc/cpuinfo; free -g | head -2
python - <<'P'
import ctypes as C, os, sys, time, json
R = os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import gen
text = gen.abcd_corpus(1_000_000_000, seed=19, survey_stream=True)
open("/tmp/up.txt", "wb").write(text)
from youtokentome_amd import _lib
L = _lib.load()
err, rep = C.create_string_buffer(2048), C.create_string_buffer(16384)
for thr, mb in (("0", "8"), ("3", "8"), ("4", "8"), ("4", "4"), ("6", "8"), ("8", "8"), ("0", "8"), ("4", "8")):
    os.environ["YTTM_IO_THREADS"] = thr  # (0: the default)
    os.environ["YTTM_IO_CHUNK_MB"] = mb
    ts = []
    for i in range(4):
        t = time.perf_counter()
        rc = L.yttm_train_bpe_comm(b"/tmp/up.txt", b"/tmp/up.model", 32000, 1.0, 8, 0, 1, 2, 3, 0, 0, None, rep, 16384, err, 2048)
        ts.append(time.perf_counter() - t)
        r = json.loads(rep.value.decode())
    print("io threads", thr, "chunk MB", mb, "wall best %.4f" % min(ts), "upload %.4f" % r["seconds_upload"], "total %.4f" % r["seconds_total"], flush=True)
P
