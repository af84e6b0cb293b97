"""GPU tuning aid: front end + K3 only, on the 1 GB 'abcd ' corpus (run under rocprofv3 --kernel-trace --stats; YTTM_K3_VARIANT picks an
experimental variant of the dense kernel -- the counts are then wrong)."""
import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import gen
from stage_lib import Ctx
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
text = gen.abcd_corpus(mb * 1_000_000, seed=19, survey_stream=True)
c = Ctx()
c.upload(text)
c.char_hist()
nu, nt = c.build_word_table(np.array([9601, 97, 98, 99, 100], np.uint32), np.array([4, 5, 6, 7, 8], np.uint32), 4, 8192)
for _ in range(3):
    c.pair_count()
print("unique words", nu, "tokens", nt)
c.close()
