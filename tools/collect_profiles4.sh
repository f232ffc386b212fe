#!/bin/bash
# Copy the artefacts of tools/measure_round4.sh (merged into gpurun_out/round4 by gpurun) into profiles/ under the prefix r03.
# Refuses (exit 1) when a measurement step failed: a traceback must never be committed as a profile again.
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
G=$R/gpurun_out/round4
D=$R/profiles
P=r04
if [ ! -f $G/steps.txt ]; then echo "no $G/steps.txt: run tools/measure_round4.sh on the GPU box first" >&2; exit 1; fi
BAD=$(awk '$2 != 0 {print $1}' $G/steps.txt)
if [ -n "$BAD" ]; then
  if [ "${ALLOW_FAILED:-0}" = 1 ]; then echo "measurement steps failed: $BAD -- copying the others (ALLOW_FAILED=1)" >&2; else echo "measurement steps failed: $BAD -- nothing copied" >&2; exit 1; fi
fi
ok() { grep -q "^$1 0$" $G/steps.txt; }
cp $G/steps.txt $D/${P}_measure_steps.txt
for f in $G/bench_*.json; do python3 -c "import json,sys; json.load(open('$f'))" && cp $f $D/${P}_$(basename $f); done
for f in $G/phase_cycles_*.txt $G/gpu_tests_parity_lines.txt; do
  [ -f $f ] || continue
  if grep -q "Traceback" $f; then echo "traceback in $f -- not copied" >&2; exit 1; fi
  cp $f $D/${P}_$(basename $f)
done
cp $(ls -t $G/prof_grouped/*/*_kernel_stats.csv | head -1) $D/${P}_dyn_curve_N25_grouped_kernel_stats.csv
cp $(ls -t $G/prof_grouped/*/*_kernel_trace.csv | head -1) $D/${P}_dyn_curve_N25_grouped_kernel_trace.csv
cp $G/prof_grouped_bench.json $D/${P}_dyn_curve_N25_grouped_bench_under_rocprof.json
cp $(ls -t $G/prof_grouped_osqp/*/*_kernel_stats.csv | head -1) $D/${P}_dyn_curve_N25_qp_osqp_grouped_kernel_stats.csv
cp $(ls -t $G/prof_grouped_osqp/*/*_kernel_trace.csv | head -1) $D/${P}_dyn_curve_N25_qp_osqp_grouped_kernel_trace.csv
cp $G/prof_grouped_osqp_bench.json $D/${P}_dyn_curve_N25_qp_osqp_grouped_bench_under_rocprof.json
{ head -7 $D/${P}_osqp_vs_pyref.txt 2>/dev/null | grep '^#'; grep qp_method $G/osqp_vs_pyref.txt; } > $D/${P}_osqp_vs_pyref.txt.new && mv $D/${P}_osqp_vs_pyref.txt.new $D/${P}_osqp_vs_pyref.txt
for W in dyn_curve_N25 kb_curve_N25; do
  cp $(ls -t $G/prof_$W/*/*_kernel_stats.csv | head -1) $D/${P}_${W}_kernel_stats.csv
  cp $(ls -t $G/prof_$W/*/*_kernel_trace.csv | head -1) $D/${P}_${W}_kernel_trace.csv
  cp $G/prof_${W}_bench.json $D/${P}_${W}_bench_under_rocprof.json
  for tag in fetch write sq f64; do cp $(ls -t $G/pmc_${tag}_$W/*/*_counter_collection.csv | head -1) $D/${P}_pmc_${tag}_$W.csv; done
  python3 $R/tools/pmc_summary.py $D/${P}_pmc_fetch_$W.csv $D/${P}_pmc_write_$W.csv $W 1024 $D/${P}_pmc_$W.json > /dev/null
  python3 - $D/${P}_pmc_$W.json $D/${P}_pmc_sq_$W.csv $D/${P}_pmc_f64_$W.csv <<'PY'
import csv, json, sys, collections
d = json.load(open(sys.argv[1]))
t = collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[2])):
    if r['Kernel_Name'].startswith('dg_solve_kernel'): t[r['Counter_Name']] += float(r['Counter_Value'])
wc = t['SQ_WAVE_CYCLES'] or 1.0
u = collections.defaultdict(float); launches = set()
for r in csv.DictReader(open(sys.argv[3])):
    if r['Kernel_Name'].startswith('dg_solve_kernel'):
        u[r['Counter_Name']] += float(r['Counter_Value']); launches.add(r['Dispatch_Id'])
nl = max(1, len(launches))
flop = 64.0 * (2 * u['SQ_INSTS_VALU_FMA_F64'] + u['SQ_INSTS_VALU_MUL_F64'] + u['SQ_INSTS_VALU_ADD_F64'] + u['SQ_INSTS_VALU_TRANS_F64']) / nl
d['round'] = 'r04'
d['fp64_flop_per_launch_upper_bound'] = flop
d['fp64_flop_per_solve_upper_bound'] = flop / d.get('batch_per_gpu', 1024)
d['fp64_share_of_valu_instructions'] = (u['SQ_INSTS_VALU_FMA_F64'] + u['SQ_INSTS_VALU_MUL_F64'] + u['SQ_INSTS_VALU_ADD_F64'] + u['SQ_INSTS_VALU_TRANS_F64']) / max(u['SQ_INSTS_VALU'], 1.0)
d['sq_wave_cycle_shares'] = {'waiting (SQ_WAIT_ANY)': t['SQ_WAIT_ANY'] / wc, 'issue stalls (SQ_WAIT_INST_ANY)': t['SQ_WAIT_INST_ANY'] / wc, 'issuing (SQ_ACTIVE_INST_ANY)': t['SQ_ACTIVE_INST_ANY'] / wc}
json.dump(d, open(sys.argv[1], 'w'), indent=1)
PY
done
if [ -d $G/pmc_fetch_merge6_N25 ]; then
  for tag in fetch write; do cp $(ls -t $G/pmc_${tag}_merge6_N25/*/*_counter_collection.csv | head -1) $D/${P}_pmc_${tag}_merge6_N25.csv; done
  python3 $R/tools/pmc_summary.py $D/${P}_pmc_fetch_merge6_N25.csv $D/${P}_pmc_write_merge6_N25.csv merge6_N25 256 $D/${P}_pmc_merge6_N25.json > /dev/null
fi
python3 - $D $P <<'PY'
import json, sys, glob, os
D, P = sys.argv[1], sys.argv[2]
rows = []
fmt = lambda v: '—' if v is None else f'{v:,.0f}'
for f in sorted(glob.glob(f'{D}/{P}_bench_*.json')):
    d = json.load(open(f)); c = d['config']
    rows.append(f"| `{os.path.basename(f)}` | {c['workload']}{' (qp osqp)' if c.get('qp_method') == 'osqp' else ''} | {c['layout']} | {c['batch_per_gpu']} | {d['steps']} / {c.get('batches_per_launch', 1)} x {c.get('launches_in_flight', 1)} | {fmt(d['value'])} | {fmt(d.get('value_single_launch'))} | "
                f"{fmt(d.get('value_host_inclusive'))} | {fmt(d.get('value_host_inclusive_grouped'))} | {d['converged_fraction']:.3f} | {d['mean_iters']:.1f} | {d['mean_qp_solves']:.1f} | {d['roofline'].get('single_launch', d['roofline'])['kernel_ms']:.0f} | "
                f"{('%.1f on %d threads, %.2f on one core' % (d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['cpu_baseline'].get('value_one_core', float('nan')))) if 'cpu_baseline' in d else '—'} |")
print('| file | workload | layout | B per GPU | steps / batches per launch x launches in flight | scen/s | one launch at a time | host-inclusive (1 batch) | host-inclusive (group) | converged | mean iters (conv.) | mean QPs | kernel ms (one at a time) | CPU oracle scen/s |')
print('|---|---|---|---|---|---|---|---|---|---|---|---|---|---|')
print('\n'.join(rows))
def avg_ms(path):
    for line in open(path):
        if 'dg_solve_kernel' in line:
            return float(line.split('",')[1].split(',')[2]) / 1e6, int(line.split('",')[1].split(',')[0])
    return float('nan'), 0
a, n = avg_ms(f'{D}/{P}_dyn_curve_N25_grouped_kernel_stats.csv')
g = json.load(open(f'{D}/{P}_dyn_curve_N25_grouped_bench_under_rocprof.json'))
print(f'grouped schedule (driver command): rocprof {n} launches of dg_solve_kernel, average {a:.1f} ms; bench under rocprof: value {g["value"]:.0f} scen/s, ms_per_step {g["ms_per_step"]:.1f}')
for w in ('dyn_curve_N25', 'kb_curve_N25'):
    a, n = avg_ms(f'{D}/{P}_{w}_kernel_stats.csv')
    print(w, f'one at a time: rocprof avg {a:.1f} ms over {n} launches | HIP events of the same run', json.load(open(f'{D}/{P}_{w}_bench_under_rocprof.json'))['roofline']['kernel_ms'], '(launches of the timed region)')
    t = json.load(open(f'{D}/{P}_pmc_{w}.json'))
    print(w, 'traffic GB per launch (FETCH x2 + WRITE)', t['traffic_bytes_per_launch'] / 1e9, 'uncorrected', t['traffic_bytes_per_launch_fetch_uncorrected'] / 1e9, 'writes', t['WRITE_SIZE_KB'] * 1024 / 1e9, t['sq_wave_cycle_shares'], 'fp64 Gflop per solve (upper bound)', t['fp64_flop_per_solve_upper_bound'] / 1e9)
PY
