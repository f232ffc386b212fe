#!/bin/bash
# Copy the artefacts of tools/measure_*.sh (merged into gpurun_out/round by gpurun) into profiles/ under the round's prefix.
# usage: tools/collect_profiles.sh [r01]
P=${1:-r01}
R=$(cd "$(dirname "$0")/.." && pwd)
G=$R/gpurun_out/round
D=$R/profiles
rm -f $D/${P}_bench_* $D/${P}_*kernel_stats.csv $D/${P}_*kernel_trace.csv $D/${P}_*_bench_under_rocprof.json $D/${P}_pmc_* $D/${P}_phase_cycles_*
for f in $G/bench_*.json; do cp $f $D/${P}_$(basename $f); done
for w in dyn kb; do
  W=${w}_curve_N25
  cp $(ls -t $G/prof_$w/runc/*_kernel_stats.csv | head -1) $D/${P}_${W}_kernel_stats.csv
  cp $(ls -t $G/prof_$w/runc/*_kernel_trace.csv | head -1) $D/${P}_${W}_kernel_trace.csv
  cp $G/prof_${w}_bench.json $D/${P}_${W}_bench_under_rocprof.json
  cp $(ls -t $G/pmc_fetch_$w/runc/*_counter_collection.csv | head -1) $D/${P}_pmc_fetch_$W.csv
  cp $(ls -t $G/pmc_write_$w/runc/*_counter_collection.csv | head -1) $D/${P}_pmc_write_$W.csv
  cp $(ls -t $G/pmc_sq_$w/runc/*_counter_collection.csv | head -1) $D/${P}_pmc_sq_pass1_$w.csv
  cp $(ls -t $G/pmc_sq2_$w/runc/*_counter_collection.csv | head -1) $D/${P}_pmc_sq_pass2_$w.csv
  cp $(ls -t $G/pmc_f64_$W/runc/*_counter_collection.csv | head -1) $D/${P}_pmc_sq_pass3_$w.csv
  cp $G/phase_cycles_${W}_B1024.txt $D/${P}_phase_cycles_${W}_B1024.txt
  python $R/tools/pmc_summary.py $D/${P}_pmc_fetch_$W.csv $D/${P}_pmc_write_$W.csv $W 1024 $D/${P}_pmc_$W.json > /dev/null
  python $R/tools/pmc_sq_summary.py $W 1024 $D/${P}_pmc_sq_$W.json $D/${P}_pmc_sq_pass1_$w.csv $D/${P}_pmc_sq_pass2_$w.csv $D/${P}_pmc_sq_pass3_$w.csv > /dev/null
done
python - $D $P <<'PY'
import json, sys, glob, os, re
D, P = sys.argv[1], sys.argv[2]
for f in sorted(glob.glob(f'{D}/{P}_bench_*.json')):
    d = json.load(open(f))
    print(os.path.basename(f), round(d['value'], 1), 'scen/s', round(d['ms_per_step'], 1), 'ms/step conv', round(d['converged_fraction'], 3), 'iters', round(d['mean_iters'], 2),
          'kernel_ms', round(d['roofline']['kernel_ms'], 1), 'cpu', round(d['cpu_baseline']['value'], 2) if 'cpu_baseline' in d else None)
for w in ('dyn_curve_N25', 'kb_curve_N25'):
    for line in open(f'{D}/{P}_{w}_kernel_stats.csv'):
        if 'dg_solve_kernel' in line:
            print(w, 'rocprof avg ms', float(line.split('",')[1].split(',')[2]) / 1e6, '| bench under rocprof kernel_ms', json.load(open(f'{D}/{P}_{w}_bench_under_rocprof.json'))['roofline']['kernel_ms'])
    t = json.load(open(f'{D}/{P}_pmc_{w}.json')); q = json.load(open(f'{D}/{P}_pmc_sq_{w}.json'))
    print(w, 'traffic GB', t['traffic_bytes_per_launch'] / 1e9, 'write GB', t['WRITE_SIZE_KB'] * 1024 / 1e9, 'Gflop/solve', q['fp64_flop_per_solve_upper_bound'] / 1e9, q['wave_cycle_shares'])
    rows = {}
    for line in open(f'{D}/{P}_phase_cycles_{w}_B1024.txt'):
        m = re.match(r'\s+(\S+)\s+cycles\s+(\d+)\s+calls\s+(\d+)', line)
        if m: rows[m.group(1)] = (int(m.group(2)), int(m.group(3)))
    nqp = rows['qp'][1]
    print(w, 'Mcycles per QP:', ' '.join(f"{k}={rows[k][0]/nqp/1e6:.2f}" for k in ('rollout','deriv1','deriv2','chains','dp','jacobi','pform','qp','merit','e_tri','e_bis','q_warm') if k in rows),
          'sum', round(sum(rows[k][0] for k in ('rollout','deriv1','deriv2','chains','dp','jacobi','pform','qp','merit','lsqr','qtmul')) / nqp / 1e6, 2))
PY
