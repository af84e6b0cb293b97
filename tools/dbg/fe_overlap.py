"""GPU tuning aid: file -> model wall of the 1 GB abcd corpus by part size / K2b grid of the front end under the upload."""
import ctypes as C, json, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import gen
from youtokentome_amd import _lib
L = _lib.load()
text = gen.abcd_corpus(1_000_000_000, seed=19, line=100, survey_stream=False)
path = "/tmp/fe_corpus.txt"
open(path, "wb").write(text)
del text
err, rep = C.create_string_buffer(2048), C.create_string_buffer(16384)
CFG = [{"YTTM_FE_NO_OVERLAP": "1"}, {}, {"YTTM_FE_PART_KB": str(32 << 10)}, {"YTTM_FE_PART_KB": str(128 << 10)}, {"YTTM_FE_PART_KB": str(256 << 10)},
       {"YTTM_FE_K2B_BLOCKS": "1024"}, {"YTTM_FE_K2B_BLOCKS": "4096"}, {"YTTM_FE_K2B_BLOCKS": "8192"}, {"YTTM_FE_NO_SPEC": "1"},
       {"YTTM_FE_PART_KB": str(128 << 10), "YTTM_FE_K2B_BLOCKS": "4096"}]
for cfg in CFG:
    for k in ("YTTM_FE_NO_OVERLAP", "YTTM_FE_PART_KB", "YTTM_FE_K2B_BLOCKS", "YTTM_FE_NO_SPEC", "YTTM_TRACE"):
        os.environ.pop(k, None)
    os.environ.update(cfg)
    ts = []
    for i in range(4):
        if i == 3:
            os.environ["YTTM_TRACE"] = "/dev/null"
        t0 = time.perf_counter()
        rc = L.yttm_train_bpe_comm(path.encode(), b"/tmp/fe.model", 32000, 1.0, 8, 0, 1, 2, 3, 0, 0, None, rep, 16384, err, 2048)
        ts.append(time.perf_counter() - t0)
        assert rc == 0, err.value
    r = json.loads(rep.value.decode())
    print("cfg %-70s best %.2f ms of %s  upload %.2f frontend %.2f merge %.2f  overlapped %d" % (
        cfg, min(ts[:3]) * 1e3, ["%.1f" % (t * 1e3) for t in ts], r["seconds_upload"] * 1e3, r["seconds_frontend"] * 1e3, r["seconds_merge"] * 1e3, r["front_end_overlapped"]), flush=True)
