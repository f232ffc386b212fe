mkdir -p gpurun_out/round6
O=gpurun_out/round6
timeout 900 python3 -m pytest tests/test_gpu.py -x -q -k "xl_device_osqp_matches or xl_osqp_sizes or xl_event_trace_parity_with_osqp" 2>&1 | tail -3 > $O/t10_osqp_parity.txt
cat $O/t10_osqp_parity.txt
B="--steps 1 --warmup 0 --pipeline 1 --batches 1 --group 1 --single-steps 0 --host-steps 0 --cpu-sample 0 --extras off"
val() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['converged_fraction'], d['roofline']['kernel_ms'])"; }
echo "curve3 osqp B4096: $(timeout 300 python3 bench.py --workload kb_curve3_N25 --qp osqp --batch 4096 $B 2>/dev/null | val)" | tee $O/t10_xl_osqp_bench.txt
echo "f1 osqp B1024: $(timeout 300 python3 bench.py --workload kb_f1_N50 --qp osqp --batch 1024 $B 2>/dev/null | val)" | tee -a $O/t10_xl_osqp_bench.txt
echo "merge6 osqp B1024: $(timeout 400 python3 bench.py --workload merge6_N25 --qp osqp --batch 1024 $B 2>/dev/null | val)" | tee -a $O/t10_xl_osqp_bench.txt
export DGSQP_HIP_LIB=$PWD/dgsqp_amd/csrc/libdgsqp_hip_prof.so
DGSQP_QP_METHOD=osqp timeout 600 python3 tools/gpu_time.py merge6_N25 0 256 > $O/t10_phase_merge6_osqp.txt 2>&1
DGSQP_QP_METHOD=osqp timeout 600 python3 tools/gpu_time.py kb_curve3_N25 0 512 > $O/t10_phase_curve3_osqp.txt 2>&1
unset DGSQP_HIP_LIB
grep -E "o_gt|o_pmul|o_gs |o_upd|o_admm|o_iters|scen/s" $O/t10_phase_merge6_osqp.txt $O/t10_phase_curve3_osqp.txt
timeout 1500 python3 -m pytest tests/test_gpu.py -x -q -s -k "prefix_parity" 2>&1 | tail -40 > $O/t10_new_tests.txt
grep -E "stable prefixes|passed|failed|K\^-1 fp32|Error|assert" $O/t10_new_tests.txt | cut -c1-900
