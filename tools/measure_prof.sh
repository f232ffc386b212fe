#!/bin/bash
# rocprofv3 passes of the round (kernel trace + stats; FETCH_SIZE and WRITE_SIZE in their own runs), CSV output.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/round
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_dyn -- python3 $R/bench.py --cpu-sample 0 > $O/prof_dyn_bench.json 2> $O/prof_dyn.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kb -- python3 $R/bench.py --workload kb_curve_N25 --cpu-sample 0 > $O/prof_kb_bench.json 2> $O/prof_kb.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_dyn -- python3 $R/bench.py --steps 1 --warmup 0 --pipeline 1 --cpu-sample 0 > $O/pmc_fetch_dyn.json 2> $O/pmc_fetch_dyn.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_dyn -- python3 $R/bench.py --steps 1 --warmup 0 --pipeline 1 --cpu-sample 0 > $O/pmc_write_dyn.json 2> $O/pmc_write_dyn.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_kb -- python3 $R/bench.py --workload kb_curve_N25 --steps 1 --warmup 0 --pipeline 1 --cpu-sample 0 > $O/pmc_fetch_kb.json 2> $O/pmc_fetch_kb.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_kb -- python3 $R/bench.py --workload kb_curve_N25 --steps 1 --warmup 0 --pipeline 1 --cpu-sample 0 > $O/pmc_write_kb.json 2> $O/pmc_write_kb.err
find $O -name "*.csv" -size -2000k | head -40
