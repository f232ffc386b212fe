#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_run3
mkdir -p $O
cd $R
timeout 600 python tools/gpu_coop_debug.py dyn_curve_N25 768 5 > $O/coop_debug_dyn.txt 2>&1
timeout 600 python tools/gpu_coop_debug.py dyn_curve_N25 1024 1 > $O/coop_debug_dyn_1024.txt 2>&1
timeout 600 python tools/gpu_coop_debug.py kb_chicane_N25 1024 1 > $O/coop_debug_chicane.txt 2>&1
cat $O/coop_debug_dyn.txt $O/coop_debug_dyn_1024.txt $O/coop_debug_chicane.txt
