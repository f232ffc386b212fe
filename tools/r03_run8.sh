#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_run8
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu.py -q -m gpu -k "f1_spline and 50" -s > $O/test_f1.log 2>&1; echo "f1 rc $?" >> $O/summary.txt
grep -E "identical|Error|assert|passed|failed" $O/test_f1.log | cut -c1-300 | tail -12 >> $O/summary.txt
timeout 900 python -m pytest tests/test_gpu.py -q -m gpu -k "device_sampler" -s > $O/test_sampler.log 2>&1; echo "sampler rc $?" >> $O/summary.txt
grep -E "Error|assert|passed|failed" $O/test_sampler.log | cut -c1-300 | tail -12 >> $O/summary.txt
timeout 1200 python -m pytest tests/test_gpu.py -q -m gpu -k "xl_layout or three_agents or six_agent or classical_qp_storage or f1_spline" -s > $O/test_xl.log 2>&1; echo "xl rc $?" >> $O/summary.txt
grep -E "identical|Error|assert|passed|failed" $O/test_xl.log | cut -c1-300 | tail -20 >> $O/summary.txt
DGSQP_XL_NOPACK=1 timeout 600 python bench.py --workload kb_curve3_N25 --steps 24 --cpu-sample 0 --host-steps 0 > $O/bench_kb_curve3_N25_nopack.json 2>> $O/bench.err
timeout 600 python bench.py --workload kb_curve3_N25 --steps 24 --cpu-sample 0 --host-steps 0 > $O/bench_kb_curve3_N25_pack.json 2>> $O/bench.err
for f in $O/bench_*.json; do echo $f; python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d.get('value_single_launch'), d['converged_fraction'], d['mean_iters'])"; done >> $O/summary.txt 2>&1
if [ -f dgsqp_amd/csrc/libdgsqp_hip_prof.so ]; then DGSQP_HIP_LIB=dgsqp_amd/csrc/libdgsqp_hip_prof.so timeout 300 python tools/gpu_time.py agents3 25 512 > $O/phase_agents3_pack.txt 2>&1; grep -E "jacobi|e_tri|qp |q_warm|scen/s" $O/phase_agents3_pack.txt >> $O/summary.txt; fi
cat $O/summary.txt
