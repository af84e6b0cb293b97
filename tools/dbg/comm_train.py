"""GPU tuning aid: one training through an RCCL communicator of size 1 (every collective of the N>1 path, no peers)."""
import ctypes as C, os, sys, json, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import gen, torch
kind, mb = sys.argv[1], int(sys.argv[2])
text = gen.abcd_corpus(mb * 1_000_000, seed=19, survey_stream=True) if kind == "abcd" else gen.zipf_corpus_fast(mb * 1_000_000, seed=7, vocab=400000)
os.environ["YTTM_TRACE"] = "1"
from youtokentome_amd import _lib
L = _lib.load()
idbuf = (C.c_uint8 * 128)()
assert L.yttm_comm_rccl_unique_id(idbuf) == 0
comm = C.c_void_p()
assert L.yttm_comm_rccl_create(idbuf, 0, 1, 0, C.byref(comm)) == 0
d = torch.frombuffer(bytearray(text), dtype=torch.uint8).cuda()
err, rep = C.create_string_buffer(2048), C.create_string_buffer(16384)
for i in range(2):
    t = time.time()
    rc = L.yttm_train_bpe_from_device_comm(C.c_void_p(d.data_ptr()), d.numel(), b"/tmp/ct.model", 32000, 1.0, 0, 1, 2, 3, 0, 1, comm, rep, 16384, err, 2048)
    assert rc == 0, err.value
    r = json.loads(rep.value.decode())
    print("wall %.4f rounds %d merge loop %.4f exchange retries %d top refills %d" % (time.time() - t, r["rounds"], r["seconds_merge"], r["exchange_retries"], r["top_refills"]),
          {k: (round(v["ms"], 1), v["launches"]) for k, v in r["kernels"].items() if v["launches"]}, flush=True)
