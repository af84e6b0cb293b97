"""GPU tuning aid: one traced training (YTTM_TRACE=1 prints the merge-loop wall split and the fused tail's timing marks)."""
import ctypes as C, os, sys, time, json
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import gen
kind, mb, vocab = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
text = (gen.abcd_corpus(mb * 1_000_000, seed=19, survey_stream=True) if kind == "abcd" else gen.cjk_corpus_fast(mb * 1_000_000, seed=11) if kind == "cjk"
            else gen.zipf_corpus_fast(mb * 1_000_000, seed=7, vocab=400000))
open("/tmp/tt.txt", "wb").write(text)
os.environ["YTTM_TRACE"] = "1"
from youtokentome_amd import _lib
L = _lib.load()
err, rep = C.create_string_buffer(2048), C.create_string_buffer(16384)
for i in range(3):
    t = time.time()
    rc = L.yttm_train_bpe_ex(b"/tmp/tt.txt", b"/tmp/tt.model", vocab, 1.0, 1, 0, 1, 2, 3, 0, rep, 16384, err, 2048)
    assert rc == 0, err.value
    r = json.loads(rep.value.decode())
    print("train wall %.4f  rounds %d  merge loop %.4f  frontend %.4f upload %.4f" % (time.time() - t, r["rounds"], r["seconds_merge"], r["seconds_frontend"], r["seconds_upload"]), flush=True)
