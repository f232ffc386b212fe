#!/bin/bash
# bench.py driver command under a few deferral settings (GPU box)
O=gpurun_out/grid; mkdir -p $O
run() { tag=$1; shift; "$@" > $O/$tag.json 2> $O/$tag.err; python3 -c "import json,sys; d=json.load(open('$O/$tag.json')); print('$tag', round(d['value']), d['ms_per_step'])" || tail -3 $O/$tag.err; }
B="python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --single-steps 0 --host-steps 0"
run k8f2 $B
run k6f15 env DGSQP_DEFER_MIN_IT=6 DGSQP_DEFER_FACTOR=1.5 DGSQP_DEFER_CAP_FRAC=0.5 $B
run k6f2 env DGSQP_DEFER_MIN_IT=6 DGSQP_DEFER_FACTOR=2.0 $B
run k10f2 env DGSQP_DEFER_MIN_IT=10 DGSQP_DEFER_FACTOR=2.0 $B
run k8f15 env DGSQP_DEFER_FACTOR=1.5 DGSQP_DEFER_CAP_FRAC=0.5 $B
run k5f1 env DGSQP_DEFER_MIN_IT=5 DGSQP_DEFER_FACTOR=1.0 DGSQP_DEFER_CAP_FRAC=0.6 $B
run k8f25 env DGSQP_DEFER_FACTOR=2.5 $B
run k8f2_h128 env DGSQP_COOP_HELPERS=128 $B
run k8f2_h32 env DGSQP_COOP_HELPERS=32 $B
