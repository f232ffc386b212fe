#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_run13
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu.py -q -m gpu -k "cooperative" -s > $O/test.log 2>&1; echo "coop test rc $?" >> $O/summary.txt
grep -E "kernel ms alone|passed|failed|Error" $O/test.log | cut -c1-400 >> $O/summary.txt
for hp in 16 32 64 128; do
  DGSQP_COOP_HELPERS=$hp timeout 600 python bench.py --steps 1 --warmup 0 --single-steps 12 --host-steps 0 --cpu-sample 0 > $O/bench_single12_h$hp.json 2>> $O/bench.err
  python -c "import json; d=json.load(open('$O/bench_single12_h$hp.json')); print('helpers $hp: single-launch', round(d['value_single_launch']), 'kernel ms', round(d['roofline']['kernel_ms'],1))" >> $O/summary.txt
  DGSQP_COOP_HELPERS=$hp timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --single-steps 0 --host-steps 0 --cpu-sample 0 > $O/bench_driver_h$hp.json 2>> $O/bench.err
  python -c "import json; d=json.load(open('$O/bench_driver_h$hp.json')); print('helpers $hp driver-style:', round(d['value']))" >> $O/summary.txt
  DGSQP_COOP_HELPERS=$hp timeout 600 python tools/gpu_coop_debug.py dyn_curve_N25 1024 1 2>&1 | grep -E "start 2: kernel|plain launch" | cut -c1-160 >> $O/summary.txt
done
cat $O/summary.txt
