"""Regenerate the measured table of DESIGN.md section 5 (round 5) from the bench lines under profiles/ (tools/collect_profiles5.sh copies
them there).  The table sits between <!-- r05:bench begin --> / <!-- r05:bench end -->; prose is written by hand."""
import json, pathlib
ROOT = pathlib.Path(__file__).resolve().parent.parent
P = ROOT / 'profiles'
f = lambda v: '—' if v is None else f'{v:,.0f}'
drv = json.load(open(P / 'r05_bench_driver_steps20_warmup5.json'))


def row(label, layout, d, timed_by):
    sf = d['status_fractions']
    fail = f"{100 * sf['qp_fail']:.1f} % / {100 * sf['max_it']:.1f} %"
    r = d['roofline']
    return (f"| {label} | {layout} | **{f(d['value'])}** | {d['config']['batch_per_gpu']:,} x {d['steps']} | {100 * d['converged_fraction']:.1f} % | {fail} | {d['mean_iters']:.1f} / {d['mean_qp_solves']:.1f} | "
            f"{r['flop_model']['flop_per_solve'] / 1e9:.2f} | {100 * r['frac']:.1f} % | {r['hbm']['frac']:.1e} | {timed_by} |")


W = {w['tag']: w for w in drv['workloads']}
rows = ['| workload (fp64) | layout / QP | scen/s (`value`) | batch x steps | converged | `qp_fail` / `max_it` | mean iters (conv.) / QPs | Gflop per solve (§8d model) | fp64 vector roof | HBM roof | timed by |',
        '|---|---|---|---|---|---|---|---|---|---|---|']
lab = {'configs[1] (the headline)': ('**configs[1]** 2-agent dynamic bicycle curve N=25, rk4 M=10, reg 1e-3', 'LDS, exact QP'),
       'configs[1] --qp osqp': ("same game, **`--qp osqp`** (the reference's own QP arithmetic)", 'LDS, OSQP'),
       'configs[2] B=4096': ('**configs[2]** 3-car BARC circuit N=25 (n = 150), reg 0', 'XL packed, exact QP'),
       'configs[2] size, solvable game, B=4096': ('3-car curve-track race N=25 (`agents.py`, M = 3): the solvable game of configs[2]\'s size', 'XL packed, exact QP'),
       'configs[3] B=16384': ('**configs[3]** 2-car F1 track N=50 (n = 200), reg 1e-3', 'XL, exact QP'),
       'configs[4] B=65536': ('**configs[4]** 6-car merge N=25 (n = 300, 1,587 rows), reg 0', 'XL, exact QP'),
       'configs[2] --qp osqp, B=4096': ('configs[2], `--qp osqp`', 'XL, OSQP'),
       'configs[3] --qp osqp, reduced batch B=2048': ('configs[3], `--qp osqp`, reduced batch', 'XL, OSQP'),
       'configs[4] --qp osqp, reduced batch B=2048': ('configs[4], `--qp osqp`, reduced batch', 'XL, OSQP')}
for tag, (label, layout) in lab.items():
    rows.append(row(label, layout, W[tag], 'the driver\'s command (one run, one JSON line)'))
for name, label, layout in (('dyn_curve_N25_steps120', 'configs[1], 120 steps (steady state: 12 batches per launch, 5 launches in flight)', 'LDS, exact QP'),
                            ('kb_curve_N25', '2-agent KB curve N=25, reg=0 (`curve.py`), 120 steps', 'LDS, classical QP'),
                            ('kb_curve3_N25_B4096_qp_osqp', '3-car curve-track race N=25, `--qp osqp`', 'XL packed, OSQP'),
                            ('kb_f1_N50_B4096_qp_osqp', 'configs[3], `--qp osqp`, B = 4,096', 'XL, OSQP'),
                            ('merge6_N25_B8192_qp_osqp', 'configs[4], `--qp osqp`, B = 8,192', 'XL, OSQP')):
    rows.append(row(label, layout, json.load(open(P / f'r05_bench_{name}.json')), '`tools/measure_round5.sh`'))
text = '\n'.join(rows)
path = ROOT / 'DESIGN.md'
s = path.read_text()
a, b = '<!-- r05:bench begin -->', '<!-- r05:bench end -->'
assert a in s and b in s
s = s[:s.index(a) + len(a)] + '\n' + text + '\n' + s[s.index(b):]
path.write_text(s)
c = drv['cpu_baseline']
print(text)
print('headline extras:', drv['value_single_launch'], drv['value_host_inclusive'], drv['value_host_inclusive_grouped'], c['value'], c['cores'], c['value_wall'], c['value_one_core'])
