"""Turn the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, see MI355X_MICROARCH.md) into the JSON
bench.py reads for roofline.traffic.  usage: pmc_summary.py <fetch_counter_csv> <write_counter_csv> <workload> <batch> <out.json> [qp_method [measured_source_sha256]]
The summary records the fingerprint of the kernel sources it was MEASURED on -- the value tools/measure_round6.sh wrote on the GPU box
before its passes (argument 7); a summary is refused here when that value is not the fingerprint of the present tree, and by bench.py
once dgsqp_amd/csrc has changed.  (Without the argument the present tree's fingerprint is recorded, as rounds 4-5 did.)"""
import csv, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from bench import source_fingerprint


def per_launch(path, counter):
    tot, n = 0.0, 0
    for row in csv.DictReader(open(path)):
        if row['Kernel_Name'].startswith('dg_solve_kernel') and row['Counter_Name'] == counter:
            tot += float(row['Counter_Value']); n += 1
    return tot / max(n, 1), n


measured_sha = sys.argv[7] if len(sys.argv) > 7 else source_fingerprint()
if measured_sha != source_fingerprint():
    sys.exit(f'pmc_summary.py: the counters were measured on kernel sources {measured_sha[:12]}..., the tree is at {source_fingerprint()[:12]}...: refused')
fetch_kb, nf = per_launch(sys.argv[1], 'FETCH_SIZE')
write_kb, nw = per_launch(sys.argv[2], 'WRITE_SIZE')
out = {
    'workload': sys.argv[3], 'batch_per_gpu': int(sys.argv[4]), 'kernel': 'dg_solve_kernel', 'qp_method': sys.argv[6] if len(sys.argv) > 6 else 'active_set',
    'source_sha256': measured_sha,
    'launches_averaged': [nf, nw],
    'FETCH_SIZE_KB': fetch_kb, 'WRITE_SIZE_KB': write_kb,
    # gfx950 (MI355X_MICROARCH.md, section HBM): FETCH_SIZE tallies 128-B requests at 64 B -- doubled before it is compared with a
    # byte count, as the guide prescribes; WRITE_SIZE is exact.  (The factor is calibrated on 16-B-per-lane streaming reads; this
    # kernel reads 8 B per lane -- Taylor tensor, raw Q, y_j columns, spills -- so the doubled figure is an upper estimate of the read
    # side and the raw one a lower bound; both are kept.)
    'traffic_bytes_per_launch': (2.0 * fetch_kb + write_kb) * 1024.0,
    'traffic_bytes_per_launch_fetch_uncorrected': (fetch_kb + write_kb) * 1024.0,
    'note': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; values are KB per launch of one batch. '
            'Traffic is per-workgroup scratch (Taylor tensor, raw Q, active-row products, register spills), not the algorithmic I/O.',
}
json.dump(out, open(sys.argv[5], 'w'), indent=1)
print(json.dumps(out))
