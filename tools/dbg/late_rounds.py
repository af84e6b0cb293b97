"""GPU tuning aid: per-launch K4 times (HIP events) of one training, averaged over ranges of rounds."""
import ctypes as C, os, sys, json
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import gen, torch
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
text = gen.abcd_corpus(mb * 1_000_000, seed=19, survey_stream=True)
from youtokentome_amd import _lib
L = _lib.load()
d = torch.frombuffer(bytearray(text), dtype=torch.uint8).cuda()
err, rep = C.create_string_buffer(2048), C.create_string_buffer(16384)
os.environ["YTTM_TRACE"] = "/tmp/lr.trace"
for i in range(2):
    rc = L.yttm_train_bpe_from_device(C.c_void_p(d.data_ptr()), d.numel(), b"/tmp/lr.model", 32000, 1.0, 0, 1, 2, 3, 0, 1, rep, 16384, err, 2048)
    assert rc == 0, err.value
r = json.loads(rep.value.decode())
k4 = [float(l.split()[1]) for l in open("/tmp/lr.trace") if l.startswith("5 ")]
print("rounds", len(k4), "gathered", r["gathered_rounds"], "index builds", r["index_builds"], "total K4 ms %.1f" % sum(k4))
for a, b in ((0, 11), (11, 46), (46, 100), (100, 200), (200, 300), (300, 400), (400, 500), (500, 600), (600, len(k4))):
    seg = k4[a:b]
    if seg:
        print("rounds %d-%d: avg %.1f us, sum %.1f ms" % (a + 1, b, 1e3 * sum(seg) / len(seg), sum(seg)))
