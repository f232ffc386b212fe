#!/bin/bash
# Copy the artefacts of tools/measure_round6.sh (merged into gpurun_out/round6 by gpurun) into profiles/ under the prefix r06.
# Refuses (exit 1) when a measurement step failed (steps.txt holds every step's own exit code); ALLOW_FAILED=1 copies the others.
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
G=$R/gpurun_out/round6
D=$R/profiles
P=r06
if [ ! -f $G/steps.txt ]; then echo "no $G/steps.txt: run tools/measure_round6.sh on the GPU box first" >&2; exit 1; fi
BAD=$(awk '$2 != 0 {print $1}' $G/steps.txt)
if [ -n "$BAD" ]; then
  if [ "${ALLOW_FAILED:-0}" = 1 ]; then echo "measurement steps failed: $BAD -- copying the others (ALLOW_FAILED=1)" >&2; else echo "measurement steps failed: $BAD -- nothing copied" >&2; exit 1; fi
fi
ok() { grep -q "^$1 0$" $G/steps.txt; }
# the kernel sources the measurements ran on (written by measure_round6.sh BEFORE its passes) must be the sources of this tree
if [ ! -f $G/source_sha256.txt ]; then echo "no $G/source_sha256.txt: measure_round6.sh did not record the fingerprint of its sources" >&2; exit 1; fi
SHA=$(cat $G/source_sha256.txt)
NOW=$(cd $R && python3 -c 'import bench; print(bench.source_fingerprint())')
if [ "$SHA" != "$NOW" ]; then echo "the kernel sources changed since the measurement ($SHA measured, $NOW now): nothing collected -- measure again" >&2; exit 1; fi
cp $G/source_sha256.txt $D/${P}_measured_source_sha256.txt
[ -f $G/bench_workloads.json ] && cp $G/bench_workloads.json $D/${P}_bench_workloads.json
cp $G/steps.txt $D/${P}_measure_steps.txt
for f in $G/bench_*.json; do [ -f $f ] && python3 -c "import json,sys; json.load(open('$f'))" && cp $f $D/${P}_$(basename $f); done
for f in $G/phase_cycles_*.txt $G/gpu_tests_parity_lines.txt $G/osqp_vs_pyref.txt; do
  [ -f $f ] || continue
  if grep -q "Traceback" $f; then echo "traceback in $f -- not copied" >&2; continue; fi
  cp $f $D/${P}_$(basename $f)
done
if ok rocprof_grouped; then
  cp $(ls -t $G/prof_grouped/*/*_kernel_stats.csv | head -1) $D/${P}_dyn_curve_N25_grouped_kernel_stats.csv
  cp $(ls -t $G/prof_grouped/*/*_kernel_trace.csv | head -1) $D/${P}_dyn_curve_N25_grouped_kernel_trace.csv
  cp $G/prof_grouped_bench.json $D/${P}_dyn_curve_N25_grouped_bench_under_rocprof.json
fi
for W in dyn_curve_N25 merge6_N25; do
  if ok rocprof_single_$W; then
    cp $(ls -t $G/prof_$W/*/*_kernel_stats.csv | head -1) $D/${P}_${W}_kernel_stats.csv
    cp $(ls -t $G/prof_$W/*/*_kernel_trace.csv | head -1) $D/${P}_${W}_kernel_trace.csv
    cp $G/prof_${W}_bench.json $D/${P}_${W}_bench_under_rocprof.json
  fi
  B=1024; [ $W = merge6_N25 ] && B=256
  if ok pmc_fetch_$W && ok pmc_write_$W; then
    for tag in fetch write; do cp $(ls -t $G/pmc_${tag}_$W/*/*_counter_collection.csv | head -1) $D/${P}_pmc_${tag}_$W.csv; done
    python3 $R/tools/pmc_summary.py $D/${P}_pmc_fetch_$W.csv $D/${P}_pmc_write_$W.csv $W $B $D/${P}_pmc_$W.json active_set $SHA > /dev/null
  fi
  if ok pmc_sq_$W && ok pmc_f64_$W; then
    for tag in sq f64; do cp $(ls -t $G/pmc_${tag}_$W/*/*_counter_collection.csv | head -1) $D/${P}_pmc_${tag}_$W.csv; done
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
d['round'] = 'r06'
d['fp64_flop_per_launch_upper_bound'] = flop
d['fp64_flop_per_solve_upper_bound'] = flop / d.get('batch_per_gpu', 1024)
d['fp64_share_of_valu_instructions'] = (u['SQ_INSTS_VALU_FMA_F64'] + u['SQ_INSTS_VALU_MUL_F64'] + u['SQ_INSTS_VALU_ADD_F64'] + u['SQ_INSTS_VALU_TRANS_F64']) / max(u['SQ_INSTS_VALU'], 1.0)
d['sq_wave_cycle_shares'] = {'waiting (SQ_WAIT_ANY)': t['SQ_WAIT_ANY'] / wc, 'issue stalls (SQ_WAIT_INST_ANY)': t['SQ_WAIT_INST_ANY'] / wc, 'issuing (SQ_ACTIVE_INST_ANY)': t['SQ_ACTIVE_INST_ANY'] / wc}
json.dump(d, open(sys.argv[1], 'w'), indent=1)
PY
  fi
done
python3 - $D $P <<'PY'
import json, sys, glob, os
D, P = sys.argv[1], sys.argv[2]
def avg_ms(path):
    for line in open(path):
        if 'dg_solve_kernel' in line:
            return float(line.split('",')[1].split(',')[2]) / 1e6, int(line.split('",')[1].split(',')[0])
    return float('nan'), 0
for w, tag in (('dyn_curve_N25_grouped', 'grouped schedule (driver command without the extra legs)'), ('dyn_curve_N25', 'one launch at a time'), ('merge6_N25', 'one launch of 256 merges at a time')):
    f = f'{D}/{P}_{w}_kernel_stats.csv'
    if not os.path.exists(f): continue
    a, n = avg_ms(f)
    b = json.loads([ln for ln in open(f'{D}/{P}_{w}_bench_under_rocprof.json') if ln.startswith('{')][-1])
    print(f'{w}: {tag}: rocprof {n} launches of dg_solve_kernel, average {a:.1f} ms | HIP events of the same run: {b["roofline"]["kernel_ms"]:.1f} ms ({b["roofline"]["launches_timed"]} launches), value {b["value"]:.0f} scen/s')
for w in ('dyn_curve_N25', 'merge6_N25'):
    f = f'{D}/{P}_pmc_{w}.json'
    if os.path.exists(f):
        t = json.load(open(f))
        print(w, 'traffic GB per launch (FETCH x2 + WRITE)', t['traffic_bytes_per_launch'] / 1e9, 'uncorrected', t['traffic_bytes_per_launch_fetch_uncorrected'] / 1e9, 'writes', t['WRITE_SIZE_KB'] * 1024 / 1e9, t.get('sq_wave_cycle_shares'), 'fp64 Gflop per solve (upper bound)', t.get('fp64_flop_per_solve_upper_bound', 0) / 1e9)
PY
