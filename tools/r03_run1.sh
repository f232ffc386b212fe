#!/bin/bash
# round 3, GPU call 1: n = 300 test, full gpu suite, driver-style bench line, grouping experiments
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_run1
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu.py -x -q -m gpu -k "six_agent_merge" -s > $O/test_merge6.log 2>&1
echo "merge6 rc $?" >> $O/summary.txt
timeout 1500 python -m pytest tests -x -q -m gpu > $O/test_gpu_all.log 2>&1
echo "gpu suite rc $?" >> $O/summary.txt
tail -3 $O/test_gpu_all.log >> $O/summary.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err
python bench.py --gpus 1 --steps 20 --warmup 5 --group 20 --pipeline 1 --cpu-sample 0 --single-steps 0 --host-steps 0 > $O/bench_group20.json 2>> $O/bench.err
python bench.py --gpus 1 --steps 20 --warmup 5 --group 10 --pipeline 2 --cpu-sample 0 --single-steps 0 --host-steps 0 > $O/bench_group10.json 2>> $O/bench.err
python bench.py --gpus 1 --steps 20 --warmup 5 --group 7 --pipeline 3 --cpu-sample 0 --single-steps 0 --host-steps 0 > $O/bench_group7.json 2>> $O/bench.err
timeout 600 python bench.py --workload merge6_N25 --batch 256 --steps 8 --warmup 1 --group 4 --pipeline 2 --cpu-sample 0 --single-steps 1 --host-steps 0 > $O/bench_merge6.json 2>> $O/bench.err
for f in $O/bench_*.json; do echo $f; python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d.get('value_single_launch'), d.get('value_host_inclusive'), d['converged_fraction'], d['mean_iters'])"; done >> $O/summary.txt 2>&1
cat $O/summary.txt
