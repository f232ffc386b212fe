#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_run12
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu.py -q -m gpu -k "cooperative" -s > $O/test.log 2>&1; echo "coop test rc $?" >> $O/summary.txt
grep -E "kernel ms alone|passed|failed|Error" $O/test.log | cut -c1-400 >> $O/summary.txt
for cfg in "off 2" "auto 2" "auto 1" "auto 4"; do
  set -- $cfg
  DGSQP_COOP_START=$2 timeout 600 python bench.py --steps 1 --warmup 0 --single-steps 12 --host-steps 0 --cpu-sample 0 --coop $1 > $O/bench_single12_coop_$1_s$2.json 2>> $O/bench.err
  python -c "import json; d=json.load(open('$O/bench_single12_coop_$1_s$2.json')); print('coop $1 start $2: single-launch', round(d['value_single_launch']), 'kernel ms', round(d['roofline']['kernel_ms'],1))" >> $O/summary.txt
done
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --single-steps 0 --host-steps 0 --cpu-sample 0 > $O/bench_driver.json 2>> $O/bench.err
python -c "import json; d=json.load(open('$O/bench_driver.json')); print('driver-style:', round(d['value']))" >> $O/summary.txt
timeout 600 python tools/gpu_coop_debug.py dyn_curve_N25 1024 1 2>&1 | grep -E "kernel|plain" | cut -c1-200 >> $O/summary.txt
# XL: blocked warm start
timeout 1500 python -m pytest tests/test_gpu.py -q -m gpu -k "xl_layout or three_agents or six_agent or classical_qp_storage or f1_spline" -s > $O/test_xl.log 2>&1; echo "xl rc $?" >> $O/summary.txt
grep -E "identical|Error|assert|passed|failed|iterate differences" $O/test_xl.log | cut -c1-250 | tail -16 >> $O/summary.txt
for nb in 0 1; do
  if [ $nb = 1 ]; then export DGSQP_XL_NOBLOCK=1; fi
  timeout 600 python bench.py --workload kb_curve3_N25 --steps 24 --cpu-sample 0 --host-steps 0 > $O/bench_kb_curve3_N25_noblock$nb.json 2>> $O/bench.err
  timeout 600 python bench.py --workload kb_curve_N50 --batch 512 --steps 16 --pipeline 2 --group 4 --single-steps 0 --host-steps 0 --cpu-sample 0 > $O/bench_kb_curve_N50_noblock$nb.json 2>> $O/bench.err
  unset DGSQP_XL_NOBLOCK
done
for f in $O/bench_kb_*.json; do echo $f; python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d.get('value_single_launch'), d['converged_fraction'], d['mean_iters'])"; done >> $O/summary.txt 2>&1
DGSQP_HIP_LIB=dgsqp_amd/csrc/libdgsqp_hip_prof.so timeout 300 python tools/gpu_time.py agents3 25 512 > $O/phase_agents3.txt 2>&1; grep -E "jacobi|e_tri|qp |q_warm|scen/s" $O/phase_agents3.txt >> $O/summary.txt
cat $O/summary.txt
