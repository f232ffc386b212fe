#!/bin/bash
# Round-4 measurement artefacts -> gpurun_out/round4/ (copied into profiles/ by tools/collect_profiles4.sh).  Every step's exit code is
# recorded in $O/steps.txt; collect_profiles4.sh refuses to copy the output of a step that failed.  PART=1|2|3 runs a third of it (the
# whole set takes ~45 GPU-minutes); PART=5 = the XL bench lines of part 1 alone, 7 = its other lines, 6 = the LDS-path phase files.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/round4
PART=${PART:-123}
mkdir -p $O
cd $R
touch $O/steps.txt
step() {   # step <name> <outfile> <command...>: run, record the exit code next to the name
  local name=$1 out=$2; shift 2
  "$@" > $out 2>> $O/stderr_$name.txt
  local rc=$?
  sed -i "/^$name /d" $O/steps.txt
  echo "$name $rc" >> $O/steps.txt
  [ $rc -ne 0 ] && echo "FAILED ($rc): $name" >&2
  return 0
}
nproc > $O/nproc.txt
if [[ $PART == *1* || $PART == *7* ]]; then
# ---- bench lines (PART=7: the LDS / big-layout lines alone).  The first one is the driver's own command; --qp osqp = the reference's own QP arithmetic
step bench_dyn_curve_N25_driver $O/bench_dyn_curve_N25_driver_steps20_warmup5.json python bench.py --gpus 1 --steps 20 --warmup 5
step bench_dyn_curve_N25_default $O/bench_dyn_curve_N25.json python bench.py --cpu-sample 0
step bench_dyn_curve_N25_osqp $O/bench_dyn_curve_N25_qp_osqp.json python bench.py --gpus 1 --steps 20 --warmup 5 --qp osqp
step bench_dyn_curve_N25_osqp_steady $O/bench_dyn_curve_N25_qp_osqp_steps120.json python bench.py --qp osqp --cpu-sample 0
step bench_kb_curve_N25 $O/bench_kb_curve_N25.json python bench.py --workload kb_curve_N25
for w in kb_chicane_N25 kb_barc2_N15 merge_N20; do step bench_$w $O/bench_$w.json python bench.py --workload $w --cpu-sample 0; done
for w in kb_curve_N25 kb_chicane_N25 kb_barc2_N15 merge_N20; do step bench_${w}_osqp $O/bench_${w}_qp_osqp.json python bench.py --workload $w --qp osqp --cpu-sample 0 --steps 24; done
fi
if [[ $PART == *1* || $PART == *5* ]]; then
# ---- the XL games (PART=5: these lines alone, after a change that only touches dgsqp_xl.h)
step bench_kb_curve3_N25 $O/bench_kb_curve3_N25.json python bench.py --workload kb_curve3_N25 --steps 48 --cpu-sample 0
step bench_kb_barc3_N25 $O/bench_kb_barc3_N25_B512.json python bench.py --workload kb_barc3_N25 --batch 512 --steps 16 --single-steps 0 --host-steps 0 --cpu-sample 0
step bench_kb_f1_N50 $O/bench_kb_f1_N50_B256.json python bench.py --workload kb_f1_N50 --batch 256 --steps 16 --single-steps 0 --host-steps 0 --cpu-sample 0
step bench_kb_curve_N50 $O/bench_kb_curve_N50_B512.json python bench.py --workload kb_curve_N50 --batch 512 --steps 16 --single-steps 0 --host-steps 0 --cpu-sample 0
step bench_merge6_N25 $O/bench_merge6_N25_B256.json python bench.py --workload merge6_N25 --batch 256 --steps 16 --pipeline 2 --group 4 --single-steps 1 --host-steps 0 --cpu-sample 16
step bench_dyn_curve_N25_v2_steps48 $O/bench_dyn_curve_N25_v2_B512_steps48.json python bench.py --workload dyn_curve_N25_v2 --batch 512 --steps 48 --group 8 --pipeline 3 --single-steps 0 --host-steps 0 --cpu-sample 0
fi
if [[ $PART == *2* ]]; then
# ---- BASELINE configs[2..4] at the batch sizes BASELINE.json names: ONE cooperative launch of the whole batch
step bench_kb_barc3_N25_B4096 $O/bench_kb_barc3_N25_B4096.json python bench.py --workload kb_barc3_N25 --batch 4096 --steps 1 --warmup 0 --pipeline 1 --batches 1 --single-steps 0 --host-steps 0 --cpu-sample 0
step bench_kb_f1_N50_B16384 $O/bench_kb_f1_N50_B16384.json python bench.py --workload kb_f1_N50 --batch 16384 --steps 1 --warmup 0 --pipeline 1 --batches 1 --single-steps 0 --host-steps 0 --cpu-sample 0
step bench_merge6_N25_B65536 $O/bench_merge6_N25_B65536.json python bench.py --workload merge6_N25 --batch 65536 --steps 1 --warmup 0 --pipeline 1 --batches 1 --single-steps 0 --host-steps 0 --cpu-sample 0
# ---- parity tables
python -m pytest tests -m gpu -q -s 2>&1 | grep -E "identical|largest relative|converged device|kernel ms alone|passed|failed|OSQP on the device|qp_method osqp|event traces|iterate difference" | cut -c1-2000 > $O/gpu_tests_parity_lines.txt
sed -i "/^gpu_tests /d" $O/steps.txt; echo "gpu_tests ${PIPESTATUS[0]}" >> $O/steps.txt
step osqp_vs_pyref $O/osqp_vs_pyref.txt python tools/gpu_osqp_vs_pyref.py
# ---- phase cycles (diagnostic build, built beforehand in the build container: tools/build_prof.sh)
if [ -f dgsqp_amd/csrc/libdgsqp_hip_prof.so ]; then
  export DGSQP_HIP_LIB=$R/dgsqp_amd/csrc/libdgsqp_hip_prof.so
  step phase_dyn $O/phase_cycles_dyn_curve_N25_B1024.txt python tools/gpu_time.py dyn 25 1024
  DGSQP_QP_METHOD=osqp step phase_dyn_osqp $O/phase_cycles_dyn_curve_N25_B1024_qp_osqp.txt python tools/gpu_time.py dyn 25 1024
  step phase_kb $O/phase_cycles_kb_curve_N25_B1024.txt python tools/gpu_time.py kbcurve 25 1024
  step phase_agents3 $O/phase_cycles_kb_curve3_N25_B512.txt python tools/gpu_time.py agents3 25 512
  step phase_f1 $O/phase_cycles_kb_f1_N50_B256.txt python tools/gpu_time.py kb_f1_N50 0 256
  step phase_merge6 $O/phase_cycles_merge6_N25_B256.txt python tools/gpu_time.py merge6_N25 0 256
  step phase_barc3 $O/phase_cycles_kb_barc3_N25_B512.txt python tools/gpu_time.py kb_barc3_N25 0 512
  unset DGSQP_HIP_LIB
else
  echo "prof_library_missing 1" >> $O/steps.txt
fi
fi
if [[ $PART == *6* ]]; then      # the phase cycles of the LDS-path workloads alone (after a change that does not touch the XL kernels)
if [ -f dgsqp_amd/csrc/libdgsqp_hip_prof.so ]; then
  export DGSQP_HIP_LIB=$R/dgsqp_amd/csrc/libdgsqp_hip_prof.so
  step phase_dyn $O/phase_cycles_dyn_curve_N25_B1024.txt python tools/gpu_time.py dyn 25 1024
  DGSQP_QP_METHOD=osqp step phase_dyn_osqp $O/phase_cycles_dyn_curve_N25_B1024_qp_osqp.txt python tools/gpu_time.py dyn 25 1024
  step phase_kb $O/phase_cycles_kb_curve_N25_B1024.txt python tools/gpu_time.py kbcurve 25 1024
  unset DGSQP_HIP_LIB
fi
fi
if [[ $PART == *3* ]]; then
# ---- rocprofv3 (program directly after --): (a) the driver's own command = the GROUPED schedule of the timed region, (b) launches one at a
#      time (the HIP-event kernel_ms of the same run must agree with the stats file), (c) the same for --qp osqp
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_grouped -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --single-steps 0 --host-steps 0 --cpu-sample 0 > $O/prof_grouped_bench.json 2> $O/prof_grouped.err
sed -i "/^rocprof_grouped /d" $O/steps.txt; echo "rocprof_grouped $?" >> $O/steps.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_grouped_osqp -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --qp osqp --single-steps 0 --host-steps 0 --cpu-sample 0 > $O/prof_grouped_osqp_bench.json 2> $O/prof_grouped_osqp.err
sed -i "/^rocprof_grouped_osqp /d" $O/steps.txt; echo "rocprof_grouped_osqp $?" >> $O/steps.txt
for w in dyn_curve_N25 kb_curve_N25; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$w -- python3 $R/bench.py --workload $w --steps 6 --warmup 0 --group 1 --pipeline 1 --single-steps 0 --host-steps 0 --cpu-sample 0 > $O/prof_${w}_bench.json 2> $O/prof_$w.err
  sed -i "/^rocprof_single_$w /d" $O/steps.txt; echo "rocprof_single_$w $?" >> $O/steps.txt
  # counters in their own passes (gpurun refuses --pmc together with trace domains); plain launches: the kernel itself, no helper work
  for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "f64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU"; do
    set -- $pass; tag=$1; shift
    rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_${tag}_$w -- python3 $R/bench.py --workload $w --steps 2 --warmup 0 --group 1 --pipeline 1 --coop off --single-steps 0 --host-steps 0 --cpu-sample 0 > $O/pmc_${tag}_$w.json 2> $O/pmc_${tag}_$w.err
    sed -i "/^pmc_${tag}_$w /d" $O/steps.txt; echo "pmc_${tag}_$w $?" >> $O/steps.txt
  done
done
# the XL layout keeps its matrices in the L2 / MALL scratch: HBM-side traffic of one launch of 256 six-car merges (n = 300)
w=merge6_N25
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE"; do
  set -- $pass; tag=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_${tag}_$w -- python3 $R/bench.py --workload $w --batch 256 --steps 2 --warmup 0 --group 1 --pipeline 1 --coop off --single-steps 0 --host-steps 0 --cpu-sample 0 > $O/pmc_${tag}_$w.json 2> $O/pmc_${tag}_$w.err
  sed -i "/^pmc_${tag}_$w /d" $O/steps.txt; echo "pmc_${tag}_$w $?" >> $O/steps.txt
done
fi
cat $O/steps.txt
