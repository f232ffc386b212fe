"""Fill the @PLACEHOLDER@ numbers of BASELINE.md from the bench lines under profiles/ (round 3)."""
import json, pathlib, re, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent
P = ROOT / 'profiles'
def L(name):
    return json.load(open(P / f'r03_bench_{name}.json'))
f = lambda v: f'{v:,.0f}'
drv, steady = L('dyn_curve_N25_driver_steps20_warmup5'), L('dyn_curve_N25')
kb = L('kb_curve_N25')
vals = {'DRV': f(drv['value']), 'STEADY': f(steady['value']), 'SINGLE': f(drv['value_single_launch']), 'HOST1': f(drv['value_host_inclusive']),
        'HOSTG': f(drv['value_host_inclusive_grouped']), 'BARC3': f(L('kb_barc3_N25_B512')['value']), 'CURVE3': f(L('kb_curve3_N25')['value']),
        'F1': f(L('kb_f1_N50_B256')['value']), 'MERGE6': f(L('merge6_N25_B256')['value']), 'MERGE3': f(L('merge_N20')['value']),
        'KBCURVE': f(kb['value']), 'KBSINGLE': f(kb['value_single_launch']), 'KBHOST': f'{f(kb["value_host_inclusive"])} / {f(kb["value_host_inclusive_grouped"])}'}
p = ROOT / 'BASELINE.md'
s = p.read_text()
for k, v in vals.items():
    s = s.replace(f'@{k}@', v)
left = re.findall(r'@[A-Z0-9]+@', s)
assert not left, left
p.write_text(s)
print(vals)
