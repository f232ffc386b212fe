#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_run5
mkdir -p $O
cd $R
for j in 0 1 2; do DGSQP_HIP_LIB=dgsqp_amd/csrc/libdgsqp_hip_prof.so timeout 300 python tools/gpu_tail_profile.py dyn_curve_N25 $j 4 >> $O/tail_profile.txt 2>&1; done
timeout 600 python tools/gpu_tail_predictor.py dyn_curve_N25 4096 $O/tail_dyn.npz >> $O/tail_profile.txt 2>&1
cat $O/tail_profile.txt
