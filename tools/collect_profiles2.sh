#!/bin/bash
# Copy the artefacts of tools/measure_round2.sh (merged into gpurun_out/round2 by gpurun) into profiles/ under the prefix r02.
R=$(cd "$(dirname "$0")/.." && pwd)
G=$R/gpurun_out/round2
D=$R/profiles
P=r02
rm -f $D/${P}_bench_* $D/${P}_*kernel_stats.csv $D/${P}_*kernel_trace.csv $D/${P}_*_bench_under_rocprof.json $D/${P}_pmc_* $D/${P}_phase_cycles_* $D/${P}_forks_* $D/${P}_gpu_tests_parity_lines.txt
for f in $G/bench_*.json; do cp $f $D/${P}_$(basename $f); done
for f in $G/forks_*.txt $G/phase_cycles_*.txt $G/gpu_tests_parity_lines.txt; do [ -f $f ] && cp $f $D/${P}_$(basename $f); done
for W in dyn_curve_N25 kb_curve_N25; do
  cp $(ls -t $G/prof_$W/runc/*_kernel_stats.csv | head -1) $D/${P}_${W}_kernel_stats.csv
  cp $(ls -t $G/prof_$W/runc/*_kernel_trace.csv | head -1) $D/${P}_${W}_kernel_trace.csv
  cp $G/prof_${W}_bench.json $D/${P}_${W}_bench_under_rocprof.json
  cp $(ls -t $G/pmc_fetch_$W/runc/*_counter_collection.csv | head -1) $D/${P}_pmc_fetch_$W.csv
  cp $(ls -t $G/pmc_write_$W/runc/*_counter_collection.csv | head -1) $D/${P}_pmc_write_$W.csv
  cp $(ls -t $G/pmc_sq_$W/runc/*_counter_collection.csv | head -1) $D/${P}_pmc_sq_$W.csv
  [ -d $G/pmc_f64_$W ] && cp $(ls -t $G/pmc_f64_$W/runc/*_counter_collection.csv | head -1) $D/${P}_pmc_f64_$W.csv
  python $R/tools/pmc_summary.py $D/${P}_pmc_fetch_$W.csv $D/${P}_pmc_write_$W.csv $W 1024 $D/${P}_pmc_$W.json > /dev/null
  python - $D/${P}_pmc_$W.json $D/${P}_pmc_sq_$W.csv <<'PY'
import csv, json, sys, collections
d = json.load(open(sys.argv[1]))
t = collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[2])):
    if r['Kernel_Name'].startswith('dg_solve_kernel'): t[r['Counter_Name']] += float(r['Counter_Value'])
wc = t['SQ_WAVE_CYCLES'] or 1.0
d['game_def'] = 'round2'
import os
f64 = sys.argv[2].replace('_pmc_sq_', '_pmc_f64_')
if os.path.exists(f64):
    u = collections.defaultdict(float); launches = set()
    for r in csv.DictReader(open(f64)):
        if r['Kernel_Name'].startswith('dg_solve_kernel'):
            u[r['Counter_Name']] += float(r['Counter_Value']); launches.add(r['Dispatch_Id'])
    nl = max(1, len(launches))
    flop = 64.0 * (2 * u['SQ_INSTS_VALU_FMA_F64'] + u['SQ_INSTS_VALU_MUL_F64'] + u['SQ_INSTS_VALU_ADD_F64'] + u['SQ_INSTS_VALU_TRANS_F64']) / nl
    d['fp64_flop_per_launch_upper_bound'] = flop
    d['fp64_flop_per_solve_upper_bound'] = flop / d.get('batch_per_gpu', 1024)
    d['fp64_share_of_valu_instructions'] = (u['SQ_INSTS_VALU_FMA_F64'] + u['SQ_INSTS_VALU_MUL_F64'] + u['SQ_INSTS_VALU_ADD_F64'] + u['SQ_INSTS_VALU_TRANS_F64']) / max(u['SQ_INSTS_VALU'], 1.0)
d['sq_wave_cycle_shares'] = {'waiting (SQ_WAIT_ANY)': t['SQ_WAIT_ANY'] / wc, 'issue stalls (SQ_WAIT_INST_ANY)': t['SQ_WAIT_INST_ANY'] / wc, 'issuing (SQ_ACTIVE_INST_ANY)': t['SQ_ACTIVE_INST_ANY'] / wc}
json.dump(d, open(sys.argv[1], 'w'), indent=1)
PY
done
python - $D $P <<'PY'
import json, sys, glob, os
D, P = sys.argv[1], sys.argv[2]
rows = []
for f in sorted(glob.glob(f'{D}/{P}_bench_*.json')):
    try:
        d = json.load(open(f))
    except Exception:
        continue
    c = d['config']
    fmt = lambda v: '—' if v is None else f'{v:,.0f}'
    rows.append(f"| `{os.path.basename(f)}` | {c['workload']} | {c['layout']} | {c['batch_per_gpu']} | {d['steps']} / {c.get('batches_per_launch', 1)} x {c.get('launches_in_flight', c['batches_in_flight'])} | {fmt(d['value'])} | {fmt(d.get('value_single_launch'))} | "
                f"{fmt(d.get('value_host_inclusive'))} | {d['converged_fraction']:.3f} | {d['mean_iters']:.1f} | {d['mean_qp_solves']:.1f} | {d['roofline']['kernel_ms']:.0f} | "
                f"{('%.1f (%d threads)' % (d['cpu_baseline']['value'], d['cpu_baseline']['cores'])) if 'cpu_baseline' in d else '—'} |")
print('| file | workload | layout | B per GPU | steps / batches per launch x launches in flight | scen/s | one launch at a time | host-inclusive | converged | mean iters (conv.) | mean QPs | kernel ms (one at a time) | CPU oracle scen/s |')
print('|---|---|---|---|---|---|---|---|---|---|---|---|---|')
print('\n'.join(rows))
for w in ('dyn_curve_N25', 'kb_curve_N25'):
    for line in open(f'{D}/{P}_{w}_kernel_stats.csv'):
        if 'dg_solve_kernel' in line:
            print(w, 'rocprof avg ms', float(line.split('",')[1].split(',')[2]) / 1e6, '| bench under rocprof kernel_ms', json.load(open(f'{D}/{P}_{w}_bench_under_rocprof.json'))['roofline']['kernel_ms'])
    t = json.load(open(f'{D}/{P}_pmc_{w}.json'))
    print(w, 'traffic GB per launch', t['traffic_bytes_per_launch'] / 1e9, 'of which writes', t['WRITE_SIZE_KB'] * 1024 / 1e9, t['sq_wave_cycle_shares'])
PY
