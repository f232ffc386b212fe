R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/round
mkdir -p $O; cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_dyn $O/prof_kb
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_dyn -- python3 $R/bench.py --warmup 0 --cpu-sample 0 > $O/prof_dyn_bench.json 2> $O/prof_dyn.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kb -- python3 $R/bench.py --workload kb_curve_N25 --warmup 0 --cpu-sample 0 > $O/prof_kb_bench.json 2> $O/prof_kb.err
sed -n 2p $O/prof_dyn/*/*_kernel_stats.csv | cut -c1-30,130-230; sed -n 2p $O/prof_kb/*/*_kernel_stats.csv | cut -c1-30,130-230
python3 -c "
import json
for f in ('prof_dyn_bench','prof_kb_bench'):
    d=json.load(open('$O/'+f+'.json')); print(f, round(d['value'],1), d['roofline']['kernel_ms'])"
