#!/bin/bash
# bench.py driver command under a few deferral settings, three repetitions each (GPU box)
O=gpurun_out/grid; mkdir -p $O
run() { tag=$1; shift; "$@" > $O/$tag.json 2> $O/$tag.err; python3 -c "import json,sys; d=json.load(open('$O/$tag.json')); print('$tag', round(d['value']), d.get('value_single_launch') and round(d['value_single_launch']), d.get('value_host_inclusive') and round(d['value_host_inclusive']), d.get('value_host_inclusive_grouped') and round(d['value_host_inclusive_grouped']), d['ms_per_step'])" || tail -3 $O/$tag.err; }
B="python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --single-steps 0 --host-steps 0"
for rep in 1 2 3; do
run drv_auto_$rep $B
run drv_f3_$rep env DGSQP_DEFER_FACTOR=3.0 $B
run drv_f4_$rep env DGSQP_DEFER_FACTOR=4.0 $B
run drv_k12f3_$rep env DGSQP_DEFER_MIN_IT=12 DGSQP_DEFER_FACTOR=3.0 $B
run drv_off_$rep env DGSQP_DEFER=0 $B
done
