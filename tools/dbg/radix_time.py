"""GPU tuning aid: K3 by radix partition on the CJK-shaped corpus -- N trainings in a row from the HBM-resident text, front-end seconds and the pair
count's kernel time of each (the first pays the scratch buffers' hipMalloc, the later ones take them from the pool)."""
import ctypes as C, json, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import gen, torch
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
text = gen.cjk_corpus_fast(mb * 1_000_000, seed=11)
from youtokentome_amd import _lib
L = _lib.load()
d = torch.frombuffer(bytearray(text), dtype=torch.uint8).cuda()
del text
for env in ({"YTTM_K3_RADIX_MIN": "1000000000000"}, {}, {}, {}, {"YTTM_K3_RADIX_MIN": "1000000000000"}, {}):
    for k in ("YTTM_K3_RADIX_MIN",):
        os.environ.pop(k, None)
    os.environ.update(env)
    err, rep = C.create_string_buffer(2048), C.create_string_buffer(16384)
    t = time.time()
    rc = L.yttm_train_bpe_from_device(C.c_void_p(d.data_ptr()), d.numel(), b"/tmp/rt.model", 32000, 1.0, 0, 1, 2, 3, 0, 1, rep, 16384, err, 2048)
    wall = time.time() - t
    assert rc == 0, err.value
    r = json.loads(rep.value.decode())
    print("radix" if r["k3_radix"] else "general", "wall %.4f" % wall, "frontend %.4f" % r["seconds_frontend"], "merge %.4f" % r["seconds_merge"],
          "pair_count ms", r["kernels"]["pair_count"]["ms"], "build ms", r["kernels"]["build"]["ms"], "peak GB %.2f" % (r["peak_device_bytes"] / 1e9), flush=True)
