#!/bin/bash
# the ordering test's two runs by hand, traces kept: where do the candidate traces part?
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/v
cd $R
python - <<'P'
import sys, os
sys.path.insert(0, "tests")
import gen, json
pin = json.load(open("tests/golden/full_size_pins.json"))["c2_100mb"]
open("/tmp/c2.txt", "wb").write(gen.abcd_corpus(pin["corpus_bytes"] + 1, seed=19, survey_stream=True))
P
export YTTM_NO_REFINE=1 YTTM_WORD_MIN_TOKENS=0
for tag in nofuse fuse0 fuse1 nofuse_noov fuse_noov; do
  unset YTTM_NO_FUSE YTTM_FE_NO_OVERLAP
  case $tag in nofuse*) export YTTM_NO_FUSE=1;; esac
  case $tag in *_noov) export YTTM_FE_NO_OVERLAP=1;; esac
  YTTM_TRACE=/dev/null YTTM_DBG_CAND=gpurun_out/v/$tag.cand python tests/gpu_train_worker.py /tmp/c2.txt /tmp/$tag.model 32000 0 > gpurun_out/v/$tag.out 2> gpurun_out/v/$tag.err
  md5sum /tmp/$tag.model | cut -c1-32; wc -l gpurun_out/v/$tag.cand
  grep -h "front end under\|index build\|merge loop wall" gpurun_out/v/$tag.err | cut -c1-250
done
