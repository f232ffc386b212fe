#!/bin/bash
# Round-2 measurement artefacts -> gpurun_out/round2/ (copied into profiles/ by tools/collect_profiles2.sh).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/round2
mkdir -p $O
cd $R
nproc > $O/nproc.txt
# bench lines
python bench.py > $O/bench_dyn_curve_N25.json 2> $O/bench.err
python bench.py --steps 240 --cpu-sample 0 > $O/bench_dyn_curve_N25_steps240.json 2>> $O/bench.err
python bench.py --group 1 --pipeline 12 --cpu-sample 0 > $O/bench_dyn_curve_N25_group1_pipeline12.json 2>> $O/bench.err
python bench.py --group 1 --pipeline 12 --steps 40 --batches 24 --cpu-sample 0 > $O/bench_dyn_curve_N25_group1_steps40.json 2>> $O/bench.err
python bench.py --group 1 --steps 10 --pipeline 5 --cpu-sample 0 > $O/bench_dyn_curve_N25_steps10_pipeline5.json 2>> $O/bench.err
python bench.py --batch 4096 --steps 32 --group 4 --pipeline 4 --cpu-sample 0 > $O/bench_dyn_curve_N25_B4096.json 2>> $O/bench.err
python bench.py --workload dyn_curve_N25_stress --cpu-sample 0 > $O/bench_dyn_curve_N25_stress.json 2>> $O/bench.err
python bench.py --workload kb_curve_N25 --cpu-sample 64 > $O/bench_kb_curve_N25.json 2>> $O/bench.err
python bench.py --workload kb_curve_N25 --steps 240 --cpu-sample 0 --single-steps 0 --host-steps 0 > $O/bench_kb_curve_N25_steps240.json 2>> $O/bench.err
python bench.py --workload kb_curve_N25 --cpu-sample 0 --eig-floor 1e-6 --snap-active-bounds > $O/bench_kb_curve_N25_floor1e-6_snap.json 2>> $O/bench.err
python bench.py --workload kb_chicane_N25 --cpu-sample 0 > $O/bench_kb_chicane_N25.json 2>> $O/bench.err
python bench.py --workload kb_barc2_N15 --cpu-sample 0 > $O/bench_kb_barc2_N15.json 2>> $O/bench.err
python bench.py --workload merge_N20 --cpu-sample 0 > $O/bench_merge_N20.json 2>> $O/bench.err
python bench.py --workload kb_curve3_N25 --steps 48 --cpu-sample 0 > $O/bench_kb_curve3_N25.json 2>> $O/bench.err
python bench.py --workload kb_f1_N50 --batch 256 --steps 16 --pipeline 2 --group 4 --single-steps 0 --host-steps 0 --cpu-sample 0 > $O/bench_kb_f1_N50_B256.json 2>> $O/bench.err
python bench.py --workload kb_barc3_N25 --batch 512 --steps 16 --pipeline 2 --group 4 --single-steps 0 --host-steps 0 --cpu-sample 0 > $O/bench_kb_barc3_N25_B512.json 2>> $O/bench.err
python bench.py --workload kb_curve_N50 --batch 512 --steps 16 --pipeline 2 --group 4 --single-steps 0 --host-steps 0 --cpu-sample 0 > $O/bench_kb_curve_N50_B512.json 2>> $O/bench.err
python bench.py --workload dyn_curve_N25_v2 --batch 512 --steps 8 --group 4 --pipeline 2 --single-steps 1 --host-steps 0 --cpu-sample 0 > $O/bench_dyn_curve_N25_v2_B512.json 2>> $O/bench.err
python bench.py --workload dyn_barc_N25_v2 --batch 512 --steps 8 --group 4 --pipeline 2 --single-steps 1 --host-steps 0 --cpu-sample 0 > $O/bench_dyn_barc_N25_v2_B512.json 2>> $O/bench.err
# parity tables
python -m pytest tests -m gpu -q -s 2>&1 | grep -E "identical|passed|failed" | cut -c1-2000 > $O/gpu_tests_parity_lines.txt
python tools/gpu_forks.py dyn_curve_N25 > $O/forks_dyn_curve_N25.txt 2>&1
python tools/gpu_forks.py kb_chicane_N15 > $O/forks_kb_chicane_N15.txt 2>&1
python tools/gpu_forks.py kb_barc2_N15 > $O/forks_kb_barc2_N15.txt 2>&1
FORKS_B=192 python tools/gpu_forks.py kb_curve_reg0_N20 > $O/forks_kb_curve_reg0_N20.txt 2>&1
# phase cycles (diagnostic build)
if [ -f dgsqp_amd/csrc/libdgsqp_hip_prof.so ]; then
  DGSQP_HIP_LIB=dgsqp_amd/csrc/libdgsqp_hip_prof.so python tools/gpu_time.py dyn 25 1024 > $O/phase_cycles_dyn_curve_N25_B1024.txt 2>&1
  DGSQP_HIP_LIB=dgsqp_amd/csrc/libdgsqp_hip_prof.so python tools/gpu_time.py kbcurve 25 1024 > $O/phase_cycles_kb_curve_N25_B1024.txt 2>&1
  DGSQP_HIP_LIB=dgsqp_amd/csrc/libdgsqp_hip_prof.so python tools/gpu_time.py agents3 25 512 > $O/phase_cycles_kb_curve3_N25_B512.txt 2>&1
fi
# rocprofv3: kernel trace + stats on launches issued one at a time (the HIP-event kernel_ms of the same run must agree)
cd /tmp && export TMPDIR=/tmp
for w in dyn_curve_N25 kb_curve_N25; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$w -- python3 $R/bench.py --workload $w --steps 6 --warmup 0 --group 1 --pipeline 1 --single-steps 0 --host-steps 0 --cpu-sample 0 > $O/prof_${w}_bench.json 2> $O/prof_$w.err
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$w -- python3 $R/bench.py --workload $w --steps 2 --warmup 0 --group 1 --pipeline 1 --single-steps 0 --host-steps 0 --cpu-sample 0 > $O/pmc_fetch_$w.json 2> $O/pmc_fetch_$w.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$w -- python3 $R/bench.py --workload $w --steps 2 --warmup 0 --group 1 --pipeline 1 --single-steps 0 --host-steps 0 --cpu-sample 0 > $O/pmc_write_$w.json 2> $O/pmc_write_$w.err
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_sq_$w -- python3 $R/bench.py --workload $w --steps 2 --warmup 0 --group 1 --pipeline 1 --single-steps 0 --host-steps 0 --cpu-sample 0 > $O/pmc_sq_$w.json 2> $O/pmc_sq_$w.err
  rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU --output-format csv -d $O/pmc_f64_$w -- python3 $R/bench.py --workload $w --steps 2 --warmup 0 --group 1 --pipeline 1 --single-steps 0 --host-steps 0 --cpu-sample 0 > $O/pmc_f64_$w.json 2> $O/pmc_f64_$w.err
done
find $O -name "*.csv" | head -40
