"""Regenerate the measured tables of DESIGN.md (section 5) and profiles/README.md (round 4) from the bench lines under profiles/.
The tables sit between <!-- r04:NAME begin --> / <!-- r04:NAME end --> markers; prose is written by hand."""
import json, pathlib, subprocess
ROOT = pathlib.Path(__file__).resolve().parent.parent
P = ROOT / 'profiles'
f = lambda v: '—' if v is None else f'{v:,.0f}'
out = subprocess.run(['bash', str(ROOT / 'tools' / 'collect_profiles4.sh')], capture_output=True, text=True)      # (first: the rows below read what it copies)
assert out.returncode == 0, out.stderr
def L(name):
    return json.load(open(P / f'r04_bench_{name}.json'))
def cpu(d):
    c = d.get('cpu_baseline')
    return '—' if not c else f"{c['value']:.1f} sustained / {c['value_wall']:.1f} wall ({c['cores']} threads) / {c.get('value_one_core', float('nan')):.2f}"
def row(label, layout, d, note='', bold=False):
    hi = d.get('value_host_inclusive'); hg = d.get('value_host_inclusive_grouped')
    host = '—' if hi is None else f'{f(hi)} / {f(hg)}'
    v = f"**{f(d['value'])}**" if bold else f(d['value'])
    return f"| {label} | {layout} | {v}{note} | {f(d.get('value_single_launch'))} | {host} | {100 * d['converged_fraction']:.1f} % | {d['mean_iters']:.1f} | {cpu(d)} |"
rows = ['| workload (fp64) | layout / QP | scen/s (`value`) | one launch at a time | host-inclusive: one batch / one group | converged | mean iters (conv.) | CPU oracle: sustained / wall (threads) / one core |',
        '|---|---|---|---|---|---|---|---|',
        row('**configs[1]** 2-agent dynamic bicycle curve N=25, rk4 M=10, reg 1e-3 — **driver command, 20 steps**', 'LDS, exact QP', L('dyn_curve_N25_driver_steps20_warmup5'), ' (round 3: 11,119)', True),
        row('same, 120 steps (steady state: 12 batches per launch, 5 launches in flight)', 'LDS, exact QP', L('dyn_curve_N25'), ' (round 3: 12,209)'),
        row("same game, **`--qp osqp`** (the reference's own QP arithmetic) — driver command, 20 steps", 'LDS, OSQP', L('dyn_curve_N25_qp_osqp'), '', True),
        row('same, `--qp osqp`, 120 steps', 'LDS, OSQP', L('dyn_curve_N25_qp_osqp_steps120')),
        row('2-agent KB curve N=25, reg=0 (`curve.py`), literal floor', 'LDS, classical QP', L('kb_curve_N25')),
        row('same, `--qp osqp`', 'LDS, OSQP', L('kb_curve_N25_qp_osqp')),
        row('2-agent KB chicane N=25, reg=1e-3', 'LDS, exact QP', L('kb_chicane_N25')),
        row('same, `--qp osqp`', 'LDS, OSQP', L('kb_chicane_N25_qp_osqp')),
        row('KB race BARC circuit N=15, reg=0', 'LDS, classical', L('kb_barc2_N15')),
        row('same, `--qp osqp`', 'LDS, OSQP', L('kb_barc2_N15_qp_osqp')),
        row('3-car merge N=20, reg=0', 'big, classical', L('merge_N20')),
        row('same, `--qp osqp` (K⁻¹ in the L2 scratch, 246 dense gradients)', 'big, OSQP', L('merge_N20_qp_osqp')),
        row('3-agent KB curve N=25 (configs[2] size)', 'XL, packed LDS matrices', L('kb_curve3_N25')),
        row('3-agent BARC circuit N=25 (**configs[2]** game), B=512, 16 steps', 'XL, packed', L('kb_barc3_N25_B512')),
        row('**configs[2] at its own batch: B = 4,096** in one cooperative launch', 'XL, packed', L('kb_barc3_N25_B4096')),
        row('2-agent F1 track N=50 (**configs[3]** game), B=256, 16 steps', 'XL', L('kb_f1_N50_B256')),
        row('**configs[3] at its own batch: B = 16,384** in one cooperative launch', 'XL', L('kb_f1_N50_B16384')),
        row('2-agent KB curve N=50 (n = 200), B=512', 'XL', L('kb_curve_N50_B512')),
        row('**configs[4]** 6-car merge N=25 (n = 300, 1,587 rows), B=256, 16 steps', 'XL, tables in constant memory', L('merge6_N25_B256')),
        row('**configs[4] at its own batch: B = 65,536** in one cooperative launch', 'XL, tables in constant memory', L('merge6_N25_B65536')),
        row('dynamic bicycle curve N=25, DG-SQP v2 (study parameters), B=512, 48 steps (8 batches per launch, 3 launches in flight)', 'LDS', L('dyn_curve_N25_v2_B512_steps48'))]
def put(path, name, text):
    s = path.read_text()
    a, b = f'<!-- r04:{name} begin -->', f'<!-- r04:{name} end -->'
    assert a in s and b in s, (path, name)
    s = s[:s.index(a) + len(a)] + '\n' + text + '\n' + s[s.index(b):]
    path.write_text(s)
put(ROOT / 'DESIGN.md', 'bench', '\n'.join(rows))
lines = out.stdout.splitlines()
i = [k for k, l in enumerate(lines) if l.startswith('| file |')][0]
j = [k for k, l in enumerate(lines) if l.startswith('grouped schedule')][0]
put(P / 'README.md', 'bench', '\n'.join(lines[i:j]))
put(P / 'README.md', 'rocprof', '```\n' + '\n'.join(lines[j:]) + '\n```')
print('\n'.join(rows[2:8]))
