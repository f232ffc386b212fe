mkdir -p gpurun_out/round6
O=gpurun_out/round6
timeout 900 python3 -m pytest tests/test_gpu.py -x -q -k "xl_device_osqp_matches or xl_osqp_sizes or xl_event_trace_parity_with_osqp or test_device_osqp_matches" 2>&1 | tail -5 > $O/t6_osqp_parity.txt
cat $O/t6_osqp_parity.txt
B="--steps 1 --warmup 0 --pipeline 1 --batches 1 --group 1 --single-steps 0 --host-steps 0 --cpu-sample 0 --extras off"
val() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['converged_fraction'], d['roofline']['kernel_ms'])"; }
echo "curve3 osqp B4096: $(timeout 300 python3 bench.py --workload kb_curve3_N25 --qp osqp --batch 4096 $B 2>/dev/null | val)" | tee -a $O/t6_xl_osqp_bench.txt
echo "f1 osqp B1024: $(timeout 300 python3 bench.py --workload kb_f1_N50 --qp osqp --batch 1024 $B 2>/dev/null | val)" | tee -a $O/t6_xl_osqp_bench.txt
echo "merge6 osqp B1024: $(timeout 400 python3 bench.py --workload merge6_N25 --qp osqp --batch 1024 $B 2>/dev/null | val)" | tee -a $O/t6_xl_osqp_bench.txt
S="--qp osqp --steps 20 --warmup 5 --single-steps 0 --host-steps 0 --cpu-sample 0 --extras off"
echo "osqp default (iterations, factor 4): $(timeout 300 python3 bench.py $S 2>/dev/null | val)" | tee -a $O/t6_defer_time.txt
for f in 1.5 2.0 3.0 4.0 6.0; do echo "osqp time mode factor $f: $(DGSQP_DEFER_TIME=1 DGSQP_DEFER_FACTOR=$f timeout 300 python3 bench.py $S 2>/dev/null | val)" | tee -a $O/t6_defer_time.txt; done
for f in 1.5 2.0 3.0; do echo "active_set time mode factor $f: $(DGSQP_DEFER_TIME=1 DGSQP_DEFER_FACTOR=$f timeout 300 python3 bench.py --steps 20 --warmup 5 --single-steps 0 --host-steps 0 --cpu-sample 0 --extras off 2>/dev/null | val)" | tee -a $O/t6_defer_time.txt; done
timeout 1500 python3 -m pytest tests/test_gpu.py -x -q -s -k "prefix_parity or example_drivers or bench_line_contract or reference_lsqr_setting or mixed_precision" 2>&1 | tail -40 > $O/t6_new_tests.txt
tail -30 $O/t6_new_tests.txt
