"""GPU tuning aid: front-end kernel times of one profiled training (args: kind mb)."""
import ctypes as C, os, sys, json
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import gen, torch
kind, mb = sys.argv[1], int(sys.argv[2])
text = gen.abcd_corpus(mb * 1_000_000, seed=19, survey_stream=True) if kind == "abcd" else gen.zipf_corpus_fast(mb * 1_000_000, seed=7, vocab=400000)
from youtokentome_amd import _lib
L = _lib.load()
d = torch.frombuffer(bytearray(text), dtype=torch.uint8).cuda()
err, rep = C.create_string_buffer(2048), C.create_string_buffer(16384)
for i in range(2):
    rc = L.yttm_train_bpe_from_device(C.c_void_p(d.data_ptr()), d.numel(), b"/tmp/ft.model", 32000, 1.0, 0, 1, 2, 3, 0, 1, rep, 16384, err, 2048)
    assert rc == 0, err.value
r = json.loads(rep.value.decode())
print(kind, {k: round(v["ms"], 2) for k, v in r["kernels"].items() if v["launches"]}, "retries", r["word_table_retries"])
