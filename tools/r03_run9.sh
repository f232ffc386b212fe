#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_run9
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu.py -q -m gpu -k "xl_layout or three_agents or six_agent or classical_qp_storage or f1_spline or reg0 or big_layout" -s > $O/test_xl.log 2>&1; echo "xl rc $?" >> $O/summary.txt
grep -E "identical|Error|assert|passed|failed" $O/test_xl.log | cut -c1-250 | tail -24 >> $O/summary.txt
timeout 600 python bench.py --workload kb_curve3_N25 --steps 24 --cpu-sample 0 --host-steps 0 > $O/bench_kb_curve3_N25_pack_el.json 2>> $O/bench.err
DGSQP_XL_NOEL=1 timeout 600 python bench.py --workload kb_curve3_N25 --steps 24 --cpu-sample 0 --host-steps 0 > $O/bench_kb_curve3_N25_pack_noel.json 2>> $O/bench.err
timeout 600 python bench.py --workload kb_barc3_N25 --batch 512 --steps 16 --pipeline 2 --group 4 --single-steps 0 --host-steps 0 --cpu-sample 0 > $O/bench_kb_barc3_N25.json 2>> $O/bench.err
for f in $O/bench_*.json; do echo $f; python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d.get('value_single_launch'), d['converged_fraction'], d['mean_iters'])"; done >> $O/summary.txt 2>&1
DGSQP_HIP_LIB=dgsqp_amd/csrc/libdgsqp_hip_prof.so timeout 300 python tools/gpu_time.py agents3 25 512 > $O/phase_agents3.txt 2>&1; grep -E "jacobi|e_tri|qp |q_warm|q_dir|q_upd|scen/s|dp " $O/phase_agents3.txt >> $O/summary.txt
timeout 600 python tools/gpu_qp_warm_vs_cold.py dyn_curve_N25 >> $O/summary.txt 2>&1
cat $O/summary.txt
