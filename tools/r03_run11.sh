#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_run11
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu.py -q -m gpu -k "monte_carlo_example or cooperative" -s > $O/test.log 2>&1; echo "tests rc $?" >> $O/summary.txt
grep -E "kernel ms alone|passed|failed|Error" $O/test.log | cut -c1-400 >> $O/summary.txt
# single-launch average over 12 batches: plain, cooperative (start 2), with windows of 8 and 16 trials
for cfg in "off 64" "auto 64" "auto 16" "auto 8"; do
  set -- $cfg
  DGSQP_COOP_WINDOW=$2 timeout 600 python bench.py --steps 1 --warmup 0 --single-steps 12 --host-steps 0 --cpu-sample 0 --coop $1 > $O/bench_single12_coop_$1_w$2.json 2>> $O/bench.err
  python -c "import json; d=json.load(open('$O/bench_single12_coop_$1_w$2.json')); print('coop $1 window $2: single-launch', round(d['value_single_launch']), 'kernel ms', round(d['roofline']['kernel_ms'],1))" >> $O/summary.txt
done
for cfg in "auto 64" "auto 16"; do
  set -- $cfg
  DGSQP_COOP_WINDOW=$2 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --single-steps 0 --host-steps 0 --cpu-sample 0 --coop $1 > $O/bench_driver_coop_$1_w$2.json 2>> $O/bench.err
  python -c "import json; d=json.load(open('$O/bench_driver_coop_$1_w$2.json')); print('driver-style coop $1 window $2:', round(d['value']))" >> $O/summary.txt
done
cat $O/summary.txt
