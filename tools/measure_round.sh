#!/bin/bash
# Round measurements on the GPU box: bench lines, rocprofv3 kernel stats, PMC passes, phase cycles.  Run from the repo root.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/round
mkdir -p $O
cd $R
python bench.py > $O/bench_dyn_curve_N25.json 2> $O/bench_dyn.err
python bench.py --workload kb_curve_N25 > $O/bench_kb_curve_N25.json 2> $O/bench_kb.err
python bench.py --workload kb_chicane_N25 --cpu-sample 0 > $O/bench_kb_chicane_N25.json 2>> $O/bench_kb.err
python bench.py --workload kb_curve_N25 --batch 4096 --cpu-sample 0 > $O/bench_kb_curve_N25_B4096.json 2>> $O/bench_kb.err
python bench.py --workload kb_curve_N25 --batch 16384 --cpu-sample 0 > $O/bench_kb_curve_N25_B16384.json 2>> $O/bench_kb.err
python bench.py --batch 4096 --cpu-sample 0 > $O/bench_dyn_curve_N25_B4096.json 2>> $O/bench_dyn.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof_dyn -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 > $O/prof_dyn_bench.json 2> $O/prof_dyn.err
rocprofv3 --kernel-trace --stats -d $O/prof_kb -- python3 $R/bench.py --workload kb_curve_N25 --steps 2 --warmup 1 --cpu-sample 0 > $O/prof_kb_bench.json 2> $O/prof_kb.err
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch_dyn -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 > $O/pmc_fetch_dyn.json 2> $O/pmc_fetch_dyn.err
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write_dyn -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 > $O/pmc_write_dyn.json 2> $O/pmc_write_dyn.err
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch_kb -- python3 $R/bench.py --workload kb_curve_N25 --steps 1 --warmup 0 --cpu-sample 0 > $O/pmc_fetch_kb.json 2> $O/pmc_fetch_kb.err
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write_kb -- python3 $R/bench.py --workload kb_curve_N25 --steps 1 --warmup 0 --cpu-sample 0 > $O/pmc_write_kb.json 2> $O/pmc_write_kb.err
cd $R
DGSQP_HIP_LIB=dgsqp_amd/csrc/libdgsqp_hip_prof.so python tools/gpu_time.py dyn 25 1024 > $O/phase_cycles_dyn_curve_N25_B1024.txt 2>&1
DGSQP_HIP_LIB=dgsqp_amd/csrc/libdgsqp_hip_prof.so python tools/gpu_time.py kbcurve 25 1024 > $O/phase_cycles_kb_curve_N25_B1024.txt 2>&1
find $O -name "*.csv" | head -40
tail -c 600 $O/bench_dyn_curve_N25.json
