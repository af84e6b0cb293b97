"""CPU soak of the batch encoder: multi-script text with invalid bytes cut at arbitrary byte positions, word cache on == off == oracle.\nusage: python tools/soak_encode.py <seconds> <seed>"""
import os, sys, pathlib, tempfile, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("YTTM_AMD_LIB", os.path.join(R, "tests", "hipsim", "_build", "libyttm_sim.so"))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import stage_checks as S
budget=float(sys.argv[1]); seed=int(sys.argv[2])
devnull = os.open(os.devnull, os.O_WRONLY); os.dup2(devnull, 2)
t0=time.time(); n=0
while time.time()-t0 < budget:
    try:
        S.check_encode_word_cache_fuzz(pathlib.Path(tempfile.mkdtemp()), trials=8, seed=seed*100000+n)
        if n % 5 == 0:
            S.check_encode_mixed_shapes(n_sent=80, seed=seed*1000+n)
            S.check_encode_word_cache(n_sent=60, seed=seed*1000+n)
    except Exception as e:
        print("FAIL", seed*100000+n, repr(e)[:300], flush=True); raise
    n+=1
print("encode soak ok:", n, "rounds in %.0f s" % (time.time()-t0), flush=True)
