"""Regenerate the measured tables of DESIGN.md (section 5) and profiles/README.md (round 3) from the bench lines under profiles/.
The tables sit between <!-- r03:NAME begin --> / <!-- r03:NAME end --> markers; prose is written by hand."""
import json, pathlib, re, subprocess, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent
P = ROOT / 'profiles'
f = lambda v: '—' if v is None else f'{v:,.0f}'
out = subprocess.run(['bash', str(ROOT / 'tools' / 'collect_profiles3.sh')], capture_output=True, text=True)      # (first: the rows below read what it copies)
assert out.returncode == 0, out.stderr
def L(name):
    return json.load(open(P / f'r03_bench_{name}.json'))
def cpu(d):
    c = d.get('cpu_baseline')
    return '—' if not c else f"{c['value']:.1f} ({c['cores']} threads) / {c.get('value_one_core', float('nan')):.2f}"
def row(label, layout, d, note=''):
    hi = d.get('value_host_inclusive'); hg = d.get('value_host_inclusive_grouped')
    host = '—' if hi is None else f'{f(hi)} / {f(hg)}'
    return f"| {label} | {layout} | {f(d['value'])}{note} | {f(d.get('value_single_launch'))} | {host} | {100 * d['converged_fraction']:.1f} % | {d['mean_iters']:.1f} | {cpu(d)} |"
rows = ['| workload (fp64) | layout | scen/s (`value`) | one launch at a time | host-inclusive: one batch / one group | converged | mean iters (conv.) | CPU oracle: all threads / one core |',
        '|---|---|---|---|---|---|---|---|',
        row('**configs[1]** 2-agent dynamic bicycle curve N=25, rk4 M=10, reg 1e-3 — **driver command, 20 steps**', 'LDS', L('dyn_curve_N25_driver_steps20_warmup5')).replace('| LDS | ', '| LDS | **', 1).replace(' | ', '** | ', 3).replace('** | LDS** | **', ' | LDS | **'),
        row('same, 120 steps (steady state: 12 batches per launch, 5 launches in flight)', 'LDS', L('dyn_curve_N25')),
        row('same, B = 4,096 per step (32 steps, 4 x 4)', 'LDS', L('dyn_curve_N25_B4096')),
        row('2-agent KB curve N=25, reg=0 (`curve.py`), literal floor, polished QP', 'LDS, classical QP', L('kb_curve_N25')),
        row('2-agent KB chicane N=25, reg=1e-3', 'LDS', L('kb_chicane_N25')),
        row('KB race BARC circuit N=15, reg=0', 'LDS, classical', L('kb_barc2_N15')),
        row('3-car merge N=20, reg=0', 'big, classical', L('merge_N20')),
        row('3-agent KB curve N=25 (configs[2] size)', 'XL, packed LDS matrices, blocked warm start', L('kb_curve3_N25'), ' (round 2: 2,622)'),
        row('3-agent BARC circuit N=25 (**configs[2]** game), B=512', 'XL, packed', L('kb_barc3_N25_B512'), ' (round 2: 474; as 4 launches of 4 batches: 460–590 between runs)'),
        row('2-agent F1 track N=50 (**configs[3]** game), B=256', 'XL', L('kb_f1_N50_B256'), ' (as 4 launches of 4 batches: 109)'),
        row('2-agent KB curve N=50 (n = 200), B=512', 'XL', L('kb_curve_N50_B512'), ' (as 4 launches of 4 batches: 459)'),
        row('**configs[4]** 6-car merge N=25 (n = 300, 1,587 rows), B=256', 'XL, tables in constant memory', L('merge6_N25_B256')),
        row('dynamic bicycle curve N=25, DG-SQP v2 (study parameters), B=512, 8 steps (bound by single solves of ~10 s)', 'LDS', L('dyn_curve_N25_v2_B512')),
        row('same, 48 steps (8 batches per launch, 3 launches in flight)', 'LDS', L('dyn_curve_N25_v2_B512_steps48'))]
# the generic row() bolds nothing; fix the first data row by hand
d0 = L('dyn_curve_N25_driver_steps20_warmup5')
rows[2] = row('**configs[1]** 2-agent dynamic bicycle curve N=25, rk4 M=10, reg 1e-3 — **driver command, 20 steps**', 'LDS', d0).replace(f"| LDS | {f(d0['value'])} |", f"| LDS | **{f(d0['value'])}** |")
def put(path, name, text):
    s = path.read_text()
    a, b = f'<!-- r03:{name} begin -->', f'<!-- r03:{name} end -->'
    assert a in s and b in s, (path, name)
    s = s[:s.index(a) + len(a)] + '\n' + text + '\n' + s[s.index(b):]
    path.write_text(s)
put(ROOT / 'DESIGN.md', 'bench', '\n'.join(rows))
lines = out.stdout.splitlines()
i = [k for k, l in enumerate(lines) if l.startswith('| file |')][0]
j = [k for k, l in enumerate(lines) if l.startswith('grouped schedule')][0]
put(P / 'README.md', 'bench', '\n'.join(lines[i:j]))
put(P / 'README.md', 'rocprof', '```\n' + '\n'.join(lines[j:]) + '\n```')
print('\n'.join(rows[2:6]))
