"""GPU tuning aid: per-phase cycle shares of the K4 kernel by ranges of rounds (needs `make -C youtokentome_amd/csrc PROF=2`)."""
import ctypes as C, os, sys, json
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["YTTM_AMD_LIB"] = os.path.join(R, "youtokentome_amd", "libyttm_prof.so")
os.environ["YTTM_TRACE_ROUNDS"] = "/tmp/k4ph.txt"
os.environ["YTTM_NO_FUSE"] = "1"
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import gen, torch
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
kind = sys.argv[2] if len(sys.argv) > 2 else "abcd"
text = gen.cjk_corpus_fast(mb * 1_000_000, seed=11) if kind == "cjk" else gen.abcd_corpus(mb * 1_000_000, seed=19, survey_stream=True)
from youtokentome_amd import _lib
L = _lib.load()
d = torch.frombuffer(bytearray(text), dtype=torch.uint8).cuda()
err, rep = C.create_string_buffer(2048), C.create_string_buffer(16384)
rc = L.yttm_train_bpe_from_device(C.c_void_p(d.data_ptr()), d.numel(), b"/tmp/k4ph.model", 32000, 1.0, 0, 1, 2, 3, 0, 0, rep, 16384, err, 2048)
assert rc == 0, err.value
rows = [[int(x) for x in l.split()] for l in open("/tmp/k4ph.txt")]
names = {0: "find sites (regs)", 1: "single-site / stage", 2: "prefetch issue", 3: "phase 1a", 4: "4", 5: "phase 2 emits", 13: "phase 2 context", 6: "phase 3 (compact; word mode: + write-back, records)", 7: "loop end", 9: "word mode: fetch", 10: "word mode: record flush", 11: "wait for block", 12: "flush"}
prev = [0] * 16
for a, b in ((1, 11), (12, 28), (29, 46), (47, 100), (101, 200), (201, 400), (401, 700), (701, len(rows))):
    cur = rows[min(b, len(rows)) - 1][7:23]
    base = rows[a - 2][7:23] if a > 1 else [0] * 16
    dlt = [c - p for c, p in zip(cur, base)]
    tot = sum(dlt[i] for i in names)
    sites = rows[min(b, len(rows)) - 1][2] - (rows[a - 2][2] if a > 1 else 0)
    tiles = rows[min(b, len(rows)) - 1][3] - (rows[a - 2][3] if a > 1 else 0)
    print("rounds %d-%d: LDS-hash misses %d (%.1f per site), %.0f cycles each, %.1f%% of the marked cycles" % (a, b, dlt[14], dlt[14] / max(sites, 1), dlt[15] / max(dlt[14], 1), 100.0 * dlt[15] / tot))
    print("rounds %d-%d: %d sites, %d dirty tiles, %.0f marked Mcycles; " % (a, b, sites, tiles, tot / 1e6) + ", ".join("%s %.0f%%" % (names[i], 100.0 * dlt[i] / tot) for i in sorted(names) if dlt[i] * 50 > tot)); continue
    print("rounds %d-%d: %d sites, %d dirty tiles; " % (a, b, sites, tiles) + ", ".join("%s %.0f%%" % (names[i], 100.0 * dlt[i] / tot) for i in sorted(names) if dlt[i] * 50 > tot))
