#!/bin/bash
# GPU soak of this round's new paths: the batch encoder under random hooks, trainings with the front end under the upload forced on toy files
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
export YTTM_AMD_LIB=$R/youtokentome_amd/libyttm_mi355x.so
S=${SOAK_SECONDS:-170}
python tools/soak_encode.py $S 101 > gpurun_out/t_enc1.log 2>&1 &
python tools/soak_encode.py $S 102 > gpurun_out/t_enc2.log 2>&1 &
YTTM_FE_OVERLAP_MIN=0 YTTM_FE_PART_KB=4 YTTM_IO_CHUNK_KB=4 python tools/soak_sim.py $S 103 > gpurun_out/t_fe1.log 2>&1 &
YTTM_FE_OVERLAP_MIN=0 YTTM_FE_PART_KB=16 YTTM_IO_CHUNK_KB=8 python tools/soak_sim.py $S 104 big > gpurun_out/t_fe2.log 2>&1 &
wait
tail -n 2 gpurun_out/t_enc1.log gpurun_out/t_enc2.log gpurun_out/t_fe1.log gpurun_out/t_fe2.log
