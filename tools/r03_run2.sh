#!/bin/bash
# round 3, GPU call 2: cooperative line search (bit-identity, timing), KKT polish on the reg = 0 games, full suite, bench lines, reg0 study data
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_run2
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu.py -x -q -m gpu -k "cooperative" -s > $O/test_coop.log 2>&1
echo "coop rc $?" >> $O/summary.txt
grep -E "kernel ms|passed|failed|Error|error" $O/test_coop.log | cut -c1-400 >> $O/summary.txt
timeout 900 python -m pytest tests/test_gpu.py -q -m gpu -k "reg0 or golden or big_layout or six_agent" -s > $O/test_reg0.log 2>&1
echo "reg0 rc $?" >> $O/summary.txt
grep -E "identical|passed|failed" $O/test_reg0.log | cut -c1-600 >> $O/summary.txt
timeout 1800 python -m pytest tests -q -m gpu > $O/test_gpu_all.log 2>&1
echo "gpu suite rc $?" >> $O/summary.txt
tail -5 $O/test_gpu_all.log >> $O/summary.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --coop off --cpu-sample 0 > $O/bench_driver_style_coop_off.json 2>> $O/bench.err
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --group 20 --pipeline 1 --cpu-sample 0 --single-steps 0 --host-steps 0 > $O/bench_group20.json 2>> $O/bench.err
timeout 600 python bench.py --workload kb_curve_N25 --steps 20 --warmup 5 --cpu-sample 0 > $O/bench_kb_curve_N25.json 2>> $O/bench.err
for f in $O/bench_*.json; do echo $f; python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d.get('value_single_launch'), d.get('value_host_inclusive'), d['converged_fraction'], d['mean_iters'])"; done >> $O/summary.txt 2>&1
timeout 600 python tools/reg0_qp_study.py gpu $O/reg0_qp_kb_curve_N20.npz kb_curve_reg0_N20 48 10 >> $O/summary.txt 2>&1
timeout 600 python tools/reg0_qp_study.py gpu $O/reg0_qp_merge_N20.npz merge_N20 16 8 >> $O/summary.txt 2>&1
cat $O/summary.txt
