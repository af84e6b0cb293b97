"""GPU tuning aid: `yttm encode` (the drop-in's CLI, ids) over the lines of the 1 GB Zipf corpus, stdin -> a file; process start to exit."""
import ctypes as C, os, subprocess, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import gen, torch
from youtokentome_amd import _lib
L = _lib.load()
text = gen.zipf_corpus_fast(1_000_000_000, seed=7, vocab=400000)
open("/tmp/cli_in.txt", "wb").write(text)
d = torch.frombuffer(bytearray(text), dtype=torch.uint8).cuda()
err, rep = C.create_string_buffer(2048), C.create_string_buffer(16384)
assert L.yttm_train_bpe_from_device(C.c_void_p(d.data_ptr()), d.numel(), b"/tmp/cli.model", 32000, 1.0, 0, 1, 2, 3, 0, 0, rep, 16384, err, 2048) == 0
del d
torch.cuda.empty_cache()
code = "import sys; sys.path.insert(0, %r); from youtokentome_amd.yttm_cli import main; sys.argv = ['yttm', 'encode', '--model', '/tmp/cli.model', '--output_type', 'id', '--n_threads', '8']; main()" % R
for i in range(3):
    with open("/tmp/cli_in.txt", "rb") as fi, open("/tmp/cli_out.txt", "wb") as fo:
        t0 = time.perf_counter()
        r = subprocess.run([sys.executable, "-c", code], stdin=fi, stdout=fo, stderr=subprocess.PIPE)
        dt = time.perf_counter() - t0
    assert r.returncode == 0, r.stderr[-500:]
    print("yttm encode: %.3f s, %d output bytes, %.2e lines/s" % (dt, os.path.getsize("/tmp/cli_out.txt"), text.count(b"\n") / dt), flush=True)
