#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_run4
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu.py -x -q -m gpu -k "cooperative" -s > $O/test_coop.log 2>&1
echo "coop rc $?" >> $O/summary.txt
grep -E "kernel ms|passed|failed|Error|error" $O/test_coop.log | cut -c1-600 >> $O/summary.txt
for st in 1 2; do DGSQP_COOP_START=$st timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 > $O/bench_driver_style_start$st.json 2>> $O/bench.err; done
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --group 20 --pipeline 1 --cpu-sample 0 --single-steps 0 --host-steps 0 > $O/bench_group20.json 2>> $O/bench.err
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --group 20 --pipeline 1 --coop off --cpu-sample 0 --single-steps 0 --host-steps 0 > $O/bench_group20_coop_off.json 2>> $O/bench.err
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --group 10 --pipeline 2 --cpu-sample 0 --single-steps 0 --host-steps 0 > $O/bench_group10.json 2>> $O/bench.err
timeout 600 python bench.py --gpus 1 --steps 120 --warmup 5 --cpu-sample 0 --single-steps 0 --host-steps 0 > $O/bench_steps120.json 2>> $O/bench.err
for f in $O/bench_*.json; do echo $f; python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d.get('value_single_launch'), d.get('value_host_inclusive'), d['converged_fraction'], d['mean_iters'])"; done >> $O/summary.txt 2>&1
timeout 600 python tools/gpu_tail_predictor.py dyn_curve_N25 4096 $O/tail_dyn.npz >> $O/summary.txt 2>&1
cat $O/summary.txt
