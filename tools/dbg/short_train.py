"""GPU tuning aid: ONE training from the HBM-resident corpus (args: kind mb vocab [comm]), for counter passes / kernel traces under rocprofv3.
comm: through an RCCL communicator of one rank (the multi-GPU round protocol)."""
import ctypes as C, os, sys, json
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import gen, torch
kind, mb, vocab = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
cache = "/tmp/corpus_%s_%d.bin" % (kind, mb)
if os.path.exists(cache):
    text = open(cache, "rb").read()
else:
    text = (gen.abcd_corpus(mb * 1_000_000, seed=19, survey_stream=True) if kind == "abcd" else gen.cjk_corpus_fast(mb * 1_000_000, seed=11) if kind == "cjk"
            else gen.zipf_corpus_fast(mb * 1_000_000, seed=7, vocab=400000))
    open(cache, "wb").write(text)
from youtokentome_amd import _lib
L = _lib.load()
d = torch.frombuffer(bytearray(text), dtype=torch.uint8).cuda()
err, rep = C.create_string_buffer(2048), C.create_string_buffer(16384)
if len(sys.argv) > 4 and sys.argv[4] == "comm":
    idbuf = (C.c_uint8 * 128)()
    assert L.yttm_comm_rccl_unique_id(idbuf) == 0
    comm = C.c_void_p()
    assert L.yttm_comm_rccl_create(idbuf, 0, 1, 0, C.byref(comm)) == 0
    rc = L.yttm_train_bpe_from_device_comm(C.c_void_p(d.data_ptr()), d.numel(), b"/tmp/st.model", vocab, 1.0, 0, 1, 2, 3, 0, 0, comm, rep, 16384, err, 2048)
else:
    rc = L.yttm_train_bpe_from_device(C.c_void_p(d.data_ptr()), d.numel(), b"/tmp/st.model", vocab, 1.0, 0, 1, 2, 3, 0, 0, rep, 16384, err, 2048)
assert rc == 0, err.value
r = json.loads(rep.value.decode())
print(kind, "rounds", r["rounds"], "merge s", r["seconds_merge"])
