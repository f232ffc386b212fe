O=gpurun_out/grid; mkdir -p $O
run() { tag=$1; shift; "$@" > $O/$tag.json 2> $O/$tag.err; python3 -c "import json,sys; d=json.load(open('$O/$tag.json')); print('$tag', round(d['value']), d.get('value_single_launch') and round(d['value_single_launch']), d.get('value_host_inclusive') and round(d['value_host_inclusive']), d.get('value_host_inclusive_grouped') and round(d['value_host_inclusive_grouped']))" || tail -3 $O/$tag.err; }
run drv python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0
run merge python bench.py --workload merge_N20 --cpu-sample 0
run merge_nodefer env DGSQP_DEFER=0 python bench.py --workload merge_N20 --cpu-sample 0
run kbcurve python bench.py --workload kb_curve_N25 --cpu-sample 0
run chicane python bench.py --workload kb_chicane_N25 --cpu-sample 0
