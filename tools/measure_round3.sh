#!/bin/bash
# Round-3 measurement artefacts -> gpurun_out/round3/ (copied into profiles/ by tools/collect_profiles3.sh).  Every step's exit code
# is recorded in $O/steps.txt; collect_profiles3.sh refuses to copy the output of a step that failed.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/round3
mkdir -p $O
cd $R
: > $O/steps.txt
step() {   # step <name> <outfile> <command...>: run, record the exit code next to the name
  local name=$1 out=$2; shift 2
  "$@" > $out 2>> $O/stderr_$name.txt
  local rc=$?
  echo "$name $rc" >> $O/steps.txt
  [ $rc -ne 0 ] && echo "FAILED ($rc): $name" >&2
  return 0
}
nproc > $O/nproc.txt
# ---- bench lines.  The first one is the driver's own command.
step bench_dyn_curve_N25_driver $O/bench_dyn_curve_N25_driver_steps20_warmup5.json python bench.py --gpus 1 --steps 20 --warmup 5
step bench_dyn_curve_N25_default $O/bench_dyn_curve_N25.json python bench.py --cpu-sample 0
step bench_dyn_curve_N25_coop_off $O/bench_dyn_curve_N25_driver_coop_off.json python bench.py --gpus 1 --steps 20 --warmup 5 --coop off --cpu-sample 0
step bench_dyn_single12_coop $O/bench_dyn_curve_N25_single12.json python bench.py --steps 1 --warmup 0 --single-steps 12 --host-steps 0 --cpu-sample 0
step bench_dyn_single12_plain $O/bench_dyn_curve_N25_single12_coop_off.json python bench.py --steps 1 --warmup 0 --single-steps 12 --host-steps 0 --cpu-sample 0 --coop off
step bench_dyn_group12 $O/bench_dyn_curve_N25_driver_group12.json python bench.py --gpus 1 --steps 20 --warmup 5 --group 12 --pipeline 5 --cpu-sample 0 --single-steps 0 --host-steps 0
step bench_dyn_defer_off $O/bench_dyn_curve_N25_driver_deferral_off.json env DGSQP_DEFER=0 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --single-steps 0 --host-steps 0
step bench_dyn_B4096 $O/bench_dyn_curve_N25_B4096.json python bench.py --batch 4096 --steps 32 --group 4 --pipeline 4 --cpu-sample 0
step bench_kb_curve_N25 $O/bench_kb_curve_N25.json python bench.py --workload kb_curve_N25
step bench_kb_chicane_N25 $O/bench_kb_chicane_N25.json python bench.py --workload kb_chicane_N25 --cpu-sample 0
step bench_kb_barc2_N15 $O/bench_kb_barc2_N15.json python bench.py --workload kb_barc2_N15 --cpu-sample 0
step bench_merge_N20 $O/bench_merge_N20.json python bench.py --workload merge_N20 --cpu-sample 0
step bench_kb_curve3_N25 $O/bench_kb_curve3_N25.json python bench.py --workload kb_curve3_N25 --steps 48 --cpu-sample 0
step bench_kb_barc3_N25 $O/bench_kb_barc3_N25_B512.json python bench.py --workload kb_barc3_N25 --batch 512 --steps 16 --single-steps 0 --host-steps 0 --cpu-sample 0
step bench_kb_f1_N50 $O/bench_kb_f1_N50_B256.json python bench.py --workload kb_f1_N50 --batch 256 --steps 16 --single-steps 0 --host-steps 0 --cpu-sample 0
step bench_kb_curve_N50 $O/bench_kb_curve_N50_B512.json python bench.py --workload kb_curve_N50 --batch 512 --steps 16 --single-steps 0 --host-steps 0 --cpu-sample 0
step bench_merge6_N25 $O/bench_merge6_N25_B256.json python bench.py --workload merge6_N25 --batch 256 --steps 16 --pipeline 2 --group 4 --single-steps 1 --host-steps 0 --cpu-sample 16
step bench_dyn_curve_N25_v2_steps48 $O/bench_dyn_curve_N25_v2_B512_steps48.json python bench.py --workload dyn_curve_N25_v2 --batch 512 --steps 48 --group 8 --pipeline 3 --single-steps 0 --host-steps 0 --cpu-sample 0
step bench_dyn_curve_N25_v2 $O/bench_dyn_curve_N25_v2_B512.json python bench.py --workload dyn_curve_N25_v2 --batch 512 --steps 8 --group 4 --pipeline 2 --single-steps 1 --host-steps 0 --cpu-sample 0
# ---- parity tables
python -m pytest tests -m gpu -q -s 2>&1 | grep -E "identical|largest relative|converged device|kernel ms alone|passed|failed" | cut -c1-2000 > $O/gpu_tests_parity_lines.txt
echo "gpu_tests ${PIPESTATUS[0]}" >> $O/steps.txt
TIGHT=1 step vs_oracle_dyn $O/device_vs_oracle_dyn_curve_N25_B512.txt python tools/gpu_vs_oracle.py dyn 25 512
TIGHT=1 step vs_oracle_chicane $O/device_vs_oracle_kb_chicane_N25_B512.txt python tools/gpu_vs_oracle.py kbchicane 25 512
step coop_debug $O/coop_line_search_dyn_curve_N25_B1024.txt python tools/gpu_coop_debug.py dyn_curve_N25 1024 1
step defer_debug $O/deferral_dyn_curve_N25_20x1024.txt python tools/gpu_defer_debug.py dyn_curve_N25 1024 20
step defer_timeline $O/deferral_timeline_dyn_curve_N25_20x1024.txt python tools/gpu_defer_timeline.py dyn_curve_N25 1024 20 8 2.0
step forks_dyn $O/forks_dyn_curve_N25.txt python tools/gpu_forks.py dyn_curve_N25
FORKS_B=192 step forks_kb_curve_reg0 $O/forks_kb_curve_reg0_N20.txt python tools/gpu_forks.py kb_curve_reg0_N20
# ---- phase cycles, tail composition (diagnostic build, built beforehand in the build container: tools/build_prof.sh)
if [ -f dgsqp_amd/csrc/libdgsqp_hip_prof.so ]; then
  export DGSQP_HIP_LIB=dgsqp_amd/csrc/libdgsqp_hip_prof.so
  step phase_dyn $O/phase_cycles_dyn_curve_N25_B1024.txt python tools/gpu_time.py dyn 25 1024
  step phase_kb $O/phase_cycles_kb_curve_N25_B1024.txt python tools/gpu_time.py kbcurve 25 1024
  step phase_agents3 $O/phase_cycles_kb_curve3_N25_B512.txt python tools/gpu_time.py agents3 25 512
  step tail_dyn $O/tail_composition_dyn_curve_N25.txt python tools/gpu_tail_profile.py dyn_curve_N25 1 4
  step slowest_dyn $O/slowest_scenarios_dyn_curve_N25.txt python tools/gpu_scn_profile.py dyn_curve_N25 23:739 9:596 10:339
  unset DGSQP_HIP_LIB
else
  echo "prof_library_missing 1" >> $O/steps.txt
fi
# ---- rocprofv3 (program directly after --): (a) the driver's own command = the GROUPED schedule of the timed region, (b) launches one at a
#      time (the HIP-event kernel_ms of the same run must agree with the stats file)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_grouped -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --single-steps 0 --host-steps 0 --cpu-sample 0 > $O/prof_grouped_bench.json 2> $O/prof_grouped.err
echo "rocprof_grouped $?" >> $O/steps.txt
for w in dyn_curve_N25 kb_curve_N25; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$w -- python3 $R/bench.py --workload $w --steps 6 --warmup 0 --group 1 --pipeline 1 --single-steps 0 --host-steps 0 --cpu-sample 0 > $O/prof_${w}_bench.json 2> $O/prof_$w.err
  echo "rocprof_single_$w $?" >> $O/steps.txt
  # counters in their own passes (gpurun refuses --pmc together with trace domains); plain launches: the kernel itself, no helper work
  for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "f64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU"; do
    set -- $pass; tag=$1; shift
    rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_${tag}_$w -- python3 $R/bench.py --workload $w --steps 2 --warmup 0 --group 1 --pipeline 1 --coop off --single-steps 0 --host-steps 0 --cpu-sample 0 > $O/pmc_${tag}_$w.json 2> $O/pmc_${tag}_$w.err
    echo "pmc_${tag}_$w $?" >> $O/steps.txt
  done
done
cat $O/steps.txt
