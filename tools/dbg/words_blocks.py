"""GPU tuning aid (needs `make -C youtokentome_amd/csrc PROF=1`): the per-workgroup timeline of k_words in every 50th merge round -- when each workgroup
started, how long its set-up (tables, rule look-ups), its words (gather + merge) and its epilogue (records, count updates) took, how many words it had.
usage: python tools/dbg/words_blocks.py [abcd|cjk] [MB]"""
import ctypes as C, glob, os, sys, statistics as st
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["YTTM_AMD_LIB"] = os.path.join(R, "youtokentome_amd", "libyttm_prof.so")
os.environ["YTTM_TRACE_ROUNDS"] = "/tmp/wb_rounds.txt"
os.environ["YTTM_TRACE_BLOCKS"] = "/tmp/wb_blocks"
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import gen, torch
kind = sys.argv[1] if len(sys.argv) > 1 else "cjk"
mb = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
text = gen.cjk_corpus_fast(mb * 1_000_000, seed=11) if kind == "cjk" else gen.abcd_corpus(mb * 1_000_000, seed=19, survey_stream=True)
from youtokentome_amd import _lib
L = _lib.load()
d = torch.frombuffer(bytearray(text), dtype=torch.uint8).cuda()
for f in glob.glob("/tmp/wb_blocks.*"):
    os.remove(f)
err, rep = C.create_string_buffer(2048), C.create_string_buffer(16384)
rc = L.yttm_train_bpe_from_device(C.c_void_p(d.data_ptr()), d.numel(), b"/tmp/wb.model", 32000, 1.0, 0, 1, 2, 3, 0, 0, rep, 16384, err, 2048)
assert rc == 0, err.value
for f in sorted(glob.glob("/tmp/wb_blocks.*"), key=lambda p: int(p.rsplit(".", 1)[1])):
    rnd = int(f.rsplit(".", 1)[1])
    rows = [[int(x) for x in l.split()] for l in open(f)]
    marks = {r[0] - 512: r[1] for r in rows if 512 <= r[0] < 1024}
    # (rows k_words wrote this round: packed offsets of a few hundred microseconds at most -- rows beyond its grid still hold what a tile round left)
    rows = [r for r in rows if r[1] and 0 < (r[3] & 0xffffffff) < 1000000 and (r[2] >> 32) < 1000000 and (r[2] & 0xffffffff) < 1000000]
    t_med = sorted(r[1] for r in rows)[len(rows) // 2] if rows else 0
    rows = [r for r in rows if abs(r[1] - t_med) < 100000]  # (and started within a millisecond of the others)
    if not rows or rnd < 100 or (rnd % 250 and rnd not in (100, 150, 200)):
        continue
    t0 = min(r[1] for r in rows)
    start = [(r[1] - t0) / 100.0 for r in rows]
    setup = [(r[2] & 0xffffffff) / 100.0 for r in rows]
    words = [((r[2] >> 32) - (r[2] & 0xffffffff)) / 100.0 for r in rows]
    tot = [(r[3] & 0xffffffff) / 100.0 for r in rows]
    epi = [t - ((r[2] >> 32) / 100.0) for t, r in zip(tot, rows)]
    nw = [r[3] >> 32 for r in rows]
    end = [s + t for s, t in zip(start, tot)]
    mk = [marks.get(r[0], 0) for r in rows]
    m1, m2, m3 = [(m & 0xffff) / 100.0 for m in mk], [((m >> 16) & 0xffff) / 100.0 for m in mk], [((m >> 32) & 0xffff) / 100.0 for m in mk]
    q = lambda v: "%.1f / %.1f / %.1f" % (st.median(v), sorted(v)[int(0.9 * (len(v) - 1))], max(v))
    print("round %5d: %3d workgroups, %5d words (max %3d per workgroup); us median / p90 / max: start %s | set-up %s (look-ups issued at %s, tables + barrier %s, runs scanned %s) | words %s | epilogue %s | whole %s | last end %.1f"
          % (rnd, len(rows), sum(nw), max(nw), q(start), q(setup), q(m1), q(m2), q(m3), q(words), q(epi), q(tot), max(end)), flush=True)
