"""Summarise the SQ counter passes of dg_solve_kernel (tools/measure_sq.sh, tools/measure_f64.sh) into one JSON.
usage: pmc_sq_summary.py <workload> <batch> <out.json> <counter_collection.csv>..."""
import csv, json, sys, collections

wl, batch, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
t = collections.defaultdict(float)
for path in sys.argv[4:]:
    seen = set()
    for r in csv.DictReader(open(path)):
        if r['Kernel_Name'].startswith('dg_solve_kernel'):
            key = (path, r['Counter_Name'])
            t[r['Counter_Name']] = (t[r['Counter_Name']] if key in seen else 0.0) + float(r['Counter_Value'])
            seen.add(key)
wc = t['SQ_WAVE_CYCLES']
f64 = t['SQ_INSTS_VALU_FMA_F64'] + t['SQ_INSTS_VALU_MUL_F64'] + t['SQ_INSTS_VALU_ADD_F64'] + t['SQ_INSTS_VALU_TRANS_F64']
flop_ub = 64.0 * (2 * t['SQ_INSTS_VALU_FMA_F64'] + t['SQ_INSTS_VALU_MUL_F64'] + t['SQ_INSTS_VALU_ADD_F64'] + t['SQ_INSTS_VALU_TRANS_F64'])
res = {
    'workload': wl, 'batch_per_gpu': batch, 'kernel': 'dg_solve_kernel', 'launches': 1,
    'counters': dict(t),
    'wave_cycle_shares': {'waiting (s_waitcnt / barrier), SQ_WAIT_ANY': t['SQ_WAIT_ANY'] / wc,
                          'issue stall, SQ_WAIT_INST_ANY': t['SQ_WAIT_INST_ANY'] / wc,
                          'issuing, SQ_ACTIVE_INST_ANY': t['SQ_ACTIVE_INST_ANY'] / wc},
    'valu_share_of_issue': t['SQ_ACTIVE_INST_VALU'] / t['SQ_ACTIVE_INST_ANY'],
    'fp64_share_of_valu_instructions': f64 / t['SQ_INSTS_VALU'],
    'fp64_flop_per_launch_upper_bound': flop_ub,
    'fp64_flop_per_solve_upper_bound': flop_ub / batch,
    'note': 'one launch of one batch (bench.py --steps 1 --warmup 0 --pipeline 1) per pass; wave-instruction counts x 64 lanes '
            '(fma = 2 flop) bound the flop count from above because partially masked wavefronts are counted in full',
}
json.dump(res, open(out, 'w'), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != 'counters'}, indent=1))
