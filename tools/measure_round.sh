#!/bin/bash
# Round measurements on the GPU box: bench lines, rocprofv3 kernel stats, PMC passes, phase cycles.  Run from the repo root.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/round
mkdir -p $O
cd $R
python bench.py > $O/bench_dyn_curve_N25.json 2> $O/bench_dyn.err
python bench.py --pipeline 1 --cpu-sample 0 > $O/bench_dyn_curve_N25_pipeline1.json 2>> $O/bench_dyn.err
python bench.py --workload kb_curve_N25 > $O/bench_kb_curve_N25.json 2> $O/bench_kb.err
python bench.py --workload kb_curve_N25 --pipeline 1 --cpu-sample 0 > $O/bench_kb_curve_N25_pipeline1.json 2>> $O/bench_kb.err
python bench.py --workload kb_chicane_N25 --cpu-sample 0 > $O/bench_kb_chicane_N25.json 2>> $O/bench_kb.err
python bench.py --workload kb_curve_N25 --batch 4096 --cpu-sample 0 > $O/bench_kb_curve_N25_B4096.json 2>> $O/bench_kb.err
python bench.py --workload kb_curve_N25 --batch 16384 --cpu-sample 0 > $O/bench_kb_curve_N25_B16384.json 2>> $O/bench_kb.err
python bench.py --batch 4096 --cpu-sample 0 > $O/bench_dyn_curve_N25_B4096.json 2>> $O/bench_dyn.err
cd $R
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DDG_PROF -o dgsqp_amd/csrc/libdgsqp_hip_prof.so dgsqp_amd/csrc/dgsqp_api.hip   # diagnostic build, always fresh
DGSQP_HIP_LIB=dgsqp_amd/csrc/libdgsqp_hip_prof.so python tools/gpu_time.py dyn 25 1024 > $O/phase_cycles_dyn_curve_N25_B1024.txt 2>&1
DGSQP_HIP_LIB=dgsqp_amd/csrc/libdgsqp_hip_prof.so python tools/gpu_time.py kbcurve 25 1024 > $O/phase_cycles_kb_curve_N25_B1024.txt 2>&1
find $O -name "*.csv" | head -40
tail -c 600 $O/bench_dyn_curve_N25.json
