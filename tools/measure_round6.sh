#!/bin/bash
# Round-6 measurement artefacts -> gpurun_out/round6/ (copied into profiles/ by tools/collect_profiles6.sh).  EVERY step goes through
# step(): the exit code is captured before anything else runs and recorded in $O/steps.txt (collect_profiles6.sh refuses to copy the
# output of a step that failed).  PART selects: 1 bench lines, 2 parity tables, 3 rocprofv3 kernel stats + PMC passes, 4 phase cycles
# (diagnostic build, tools/build_prof.sh beforehand).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/round6
PART=${PART:-1234}
mkdir -p $O
cd $R
touch $O/steps.txt
step() {   # step <name> <stdout file> <command...>: run, record the exit code next to the name (stderr -> $O/stderr_<name>.txt)
  local name=$1 out=$2; shift 2
  "$@" > "$out" 2> "$O/stderr_$name.txt"
  local rc=$?
  sed -i "/^$name /d" $O/steps.txt
  echo "$name $rc" >> $O/steps.txt
  [ $rc -ne 0 ] && echo "FAILED ($rc): $name" >&2
  return 0
}
nproc > $O/nproc.txt
# the fingerprint of the kernel sources THESE measurements run on, taken now (collect_profiles6.sh hands it to tools/pmc_summary.py and
# refuses to collect when the tree has changed since: a PMC summary never carries the fingerprint of sources it was not measured on)
python3 -c 'import bench; print(bench.source_fingerprint())' > $O/source_sha256.txt
if [[ $PART == *1* ]]; then
# the driver's own command: the compact headline line; --qp osqp + configs[2], [3], [4] at BASELINE's batch sizes run as child processes (full records: bench_workloads.json)
step bench_driver $O/bench_driver_steps20_warmup5.json python bench.py --gpus 1 --steps 20 --warmup 5
cp bench_workloads.json $O/bench_workloads.json 2>/dev/null
step bench_steady $O/bench_dyn_curve_N25_steps120.json python bench.py --cpu-sample 0 --extras off --line full
step bench_kb_curve_N25 $O/bench_kb_curve_N25.json python bench.py --workload kb_curve_N25 --cpu-sample 0 --line full
# OSQP's arithmetic on the XL layout (round 5): configs[2], [3], [4] with qp_method = osqp
step bench_barc3_osqp $O/bench_kb_barc3_N25_B4096_qp_osqp.json python bench.py --workload kb_barc3_N25 --qp osqp --batch 4096 --steps 1 --warmup 0 --pipeline 1 --batches 1 --single-steps 0 --host-steps 0 --cpu-sample 0 --line full
step bench_curve3_osqp $O/bench_kb_curve3_N25_B4096_qp_osqp.json python bench.py --workload kb_curve3_N25 --qp osqp --batch 4096 --steps 1 --warmup 0 --pipeline 1 --batches 1 --single-steps 0 --host-steps 0 --cpu-sample 0 --line full
step bench_curve3_osqp_mixed $O/bench_kb_curve3_N25_B4096_qp_osqp_mixed.json python bench.py --workload kb_curve3_N25 --qp osqp --mixed-precision --batch 4096 --steps 1 --warmup 0 --pipeline 1 --batches 1 --single-steps 0 --host-steps 0 --cpu-sample 0 --line full
step bench_f1_osqp_mixed $O/bench_kb_f1_N50_B4096_qp_osqp_mixed.json python bench.py --workload kb_f1_N50 --qp osqp --mixed-precision --batch 4096 --steps 1 --warmup 0 --pipeline 1 --batches 1 --single-steps 0 --host-steps 0 --cpu-sample 0 --line full
step bench_f1_osqp $O/bench_kb_f1_N50_B4096_qp_osqp.json python bench.py --workload kb_f1_N50 --qp osqp --batch 4096 --steps 1 --warmup 0 --pipeline 1 --batches 1 --single-steps 0 --host-steps 0 --cpu-sample 0 --line full
step bench_merge6_osqp $O/bench_merge6_N25_B4096_qp_osqp.json python bench.py --workload merge6_N25 --qp osqp --batch 4096 --steps 1 --warmup 0 --pipeline 1 --batches 1 --single-steps 0 --host-steps 0 --cpu-sample 0 --line full
fi
if [[ $PART == *2* ]]; then
step gpu_tests $O/gpu_tests_full.txt python -m pytest tests -m gpu -q -s
grep -E "identical|largest relative|converged device|kernel ms alone|passed|failed|OSQP on the device|qp_method osqp|event traces|iterate difference|restated-OSQP" $O/gpu_tests_full.txt | cut -c1-2000 > $O/gpu_tests_parity_lines.txt
step osqp_vs_pyref $O/osqp_vs_pyref.txt python tools/gpu_osqp_vs_pyref.py
fi
if [[ $PART == *3* ]]; then
# rocprofv3 (the program directly after --): (a) the driver's command without the extra legs = the GROUPED schedule of the timed region,
# (b) launches one at a time (the HIP-event kernel_ms of the same run must agree with the stats file); PMC counters in passes of their own
cd /tmp && export TMPDIR=/tmp
step rocprof_grouped $O/prof_grouped_bench.json rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_grouped -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --extras off --single-steps 0 --host-steps 0 --cpu-sample 0 --line full
for w in dyn_curve_N25; do
  step rocprof_single_$w $O/prof_${w}_bench.json rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$w -- python3 $R/bench.py --workload $w --steps 6 --warmup 0 --group 1 --pipeline 1 --extras off --single-steps 0 --host-steps 0 --cpu-sample 0 --line full
  for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "f64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU"; do
    set -- $pass; tag=$1; shift
    step pmc_${tag}_$w $O/pmc_${tag}_$w.json rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_${tag}_$w -- python3 $R/bench.py --workload $w --steps 2 --warmup 0 --group 1 --pipeline 1 --coop off --extras off --single-steps 0 --host-steps 0 --cpu-sample 0 --line full
  done
done
w=merge6_N25       # the XL layout keeps its matrices in the L2 / MALL scratch: memory-side traffic of one launch of 256 six-car merges
step rocprof_single_$w $O/prof_${w}_bench.json rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$w -- python3 $R/bench.py --workload $w --batch 256 --steps 2 --warmup 0 --group 1 --pipeline 1 --coop off --extras off --single-steps 0 --host-steps 0 --cpu-sample 0 --line full
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE"; do
  set -- $pass; tag=$1; shift
  step pmc_${tag}_$w $O/pmc_${tag}_$w.json rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_${tag}_$w -- python3 $R/bench.py --workload $w --batch 256 --steps 2 --warmup 0 --group 1 --pipeline 1 --coop off --extras off --single-steps 0 --host-steps 0 --cpu-sample 0 --line full
done
cd $R
fi
if [[ $PART == *4* ]]; then
if [ -f dgsqp_amd/csrc/libdgsqp_hip_prof.so ]; then
  export DGSQP_HIP_LIB=$R/dgsqp_amd/csrc/libdgsqp_hip_prof.so
  step phase_dyn $O/phase_cycles_dyn_curve_N25_B1024.txt python tools/gpu_time.py dyn 25 1024
  DGSQP_QP_METHOD=osqp step phase_dyn_osqp $O/phase_cycles_dyn_curve_N25_B1024_qp_osqp.txt python tools/gpu_time.py dyn 25 1024
  DGSQP_QP_METHOD=osqp step phase_merge6_osqp $O/phase_cycles_merge6_N25_B256_qp_osqp.txt python tools/gpu_time.py merge6_N25 0 256
  DGSQP_QP_METHOD=osqp step phase_f1_osqp $O/phase_cycles_kb_f1_N50_B256_qp_osqp.txt python tools/gpu_time.py kb_f1_N50 0 256
  step phase_merge6 $O/phase_cycles_merge6_N25_B256.txt python tools/gpu_time.py merge6_N25 0 256
  unset DGSQP_HIP_LIB
else
  echo "prof_library_missing 1" >> $O/steps.txt
fi
fi
cat $O/steps.txt
