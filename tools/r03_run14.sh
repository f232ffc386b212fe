#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_run14
mkdir -p $O
cd $R
for w in kb_curve_N25 kb_chicane_N25 merge_N20 kb_barc2_N15; do
  for cp in off auto; do
    timeout 600 python bench.py --workload $w --steps 20 --warmup 5 --single-steps 12 --host-steps 0 --cpu-sample 0 --coop $cp > $O/bench_${w}_$cp.json 2>> $O/bench.err
    python -c "import json; d=json.load(open('$O/bench_${w}_$cp.json')); print('$w coop $cp: 20 steps', round(d['value']), 'single-launch (12 batches)', round(d['value_single_launch']), 'kernel ms', round(d['roofline']['kernel_ms'],1))" >> $O/summary.txt
  done
done
cat $O/summary.txt
