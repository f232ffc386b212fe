#!/bin/bash
# XL bench lines issued as ONE cooperative launch over all 16 steps (deferral of long scenarios) vs 4 launches of 4 (GPU box)
O=gpurun_out/grid; mkdir -p $O
run() { tag=$1; shift; "$@" > $O/$tag.json 2> $O/$tag.err; python3 -c "import json,sys; d=json.load(open('$O/$tag.json')); print('$tag', round(d['value'],1), d['ms_per_step'], d['converged_fraction'])" || tail -3 $O/$tag.err; }
X="--single-steps 0 --host-steps 0 --cpu-sample 0 --steps 16 --warmup 1"
run f1_one python bench.py --workload kb_f1_N50 --batch 256 $X
run f1_4x4 python bench.py --workload kb_f1_N50 --batch 256 $X --group 4 --pipeline 2
run n200_one python bench.py --workload kb_curve_N50 --batch 512 $X
run merge6_one python bench.py --workload merge6_N25 --batch 256 $X
run barc3_one python bench.py --workload kb_barc3_N25 --batch 512 $X
run curve3_one python bench.py --workload kb_curve3_N25 --steps 16 --warmup 1 --single-steps 0 --host-steps 0 --cpu-sample 0
