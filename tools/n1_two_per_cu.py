#!/usr/bin/env python3
"""Row N1, step (a): what do two scenarios per CU buy?  The same games (n = 60: arena under half a CU's LDS) through the product
library (one 512-thread workgroup per CU) and through the -DDG_BLOCK=256 build (libdgsqp_hip_b256.so: 256-thread workgroups, half
the arena limit, two workgroups per CU), each in its own process: (1) results -- status / iterations / QP solves identical, iterates
compared; (2) throughput of bench.py's timed region (20 batches of 1,024, one cooperative launch) and of a 120-step steady state.

    python tools/n1_two_per_cu.py [--workloads kb_chicane_N15 dyn_curve_N15] [--out profiles/r06_n1_two_per_cu.txt]
"""
import argparse
import json
import os
import pathlib
import subprocess
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
LIBS = {'b512 (product: 1 workgroup of 512 per CU)': ROOT / 'dgsqp_amd' / 'csrc' / 'libdgsqp_hip.so',
        'b256 (2 workgroups of 256 per CU)': ROOT / 'dgsqp_amd' / 'csrc' / 'libdgsqp_hip_b256.so'}

SOLVE = r'''
import sys, numpy as np
sys.path.insert(0, %r)
import bench
from dgsqp_amd.montecarlo import sample_scenarios
from dgsqp_amd.solver import DGSQP
g = bench.make_game(sys.argv[1])
sv = DGSQP(*g.solver_args(), print_method=None)
x0, u = sample_scenarios(g, 512, seed=7)
r = sv.solve_batch(x0, u)
d = sv.dims
np.savez(sys.argv[2], u=r['u'], l=r['l'], status=r['status'], it=r['num_iters'], qp=r['qp_solves'], lds=np.array([d.lds_bytes]), layout=np.array([d.layout]))
'''


def run(cmd, lib, timeout=420):
    env = dict(os.environ, DGSQP_HIP_LIB=str(lib))
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=str(ROOT))
    if out.returncode != 0:
        raise RuntimeError(out.stderr[-1500:])
    return out.stdout


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workloads', nargs='+', default=['kb_chicane_N15', 'dyn_curve_N15', 'dyn_curve_N25'])
    ap.add_argument('--out', default=str(ROOT / 'profiles' / 'r06_n1_two_per_cu.txt'))
    ap.add_argument('--tmp', default='/tmp')
    args = ap.parse_args()
    lines = ['Row N1 step (a): one 512-thread workgroup per CU (product build) against two 256-thread workgroups per CU (-DDG_BLOCK=256), same sources',
             'values: bench.py scenarios/s, batch 1,024; "20 steps" = the driver-sized run (one cooperative launch), "120 steps" = steady state (12 batches per launch, 5 in flight)', '']
    for w in args.workloads:
        res, rate = {}, {}
        for name, lib in LIBS.items():
            f = os.path.join(args.tmp, f'n1_{w}_{lib.stem}.npz')
            try:
                run([sys.executable, '-c', SOLVE % str(ROOT), w, f], lib)
                res[name] = dict(np.load(f))
                for steps in (20, 120):
                    o = run([sys.executable, str(ROOT / 'bench.py'), '--workload', w, '--steps', str(steps), '--warmup', '2', '--single-steps', '0', '--host-steps', '0',
                             '--cpu-sample', '0', '--extras', 'off', '--line', 'full'], lib)
                    d = json.loads([ln for ln in o.splitlines() if ln.startswith('{')][-1])
                    rate[(name, steps)] = (d['value'], d['converged_fraction'], d['mean_qp_solves'])
            except Exception as e:
                lines.append(f'{w}: {name}: FAILED: {e}')
        names = [n for n in LIBS if n in res]
        for n in names:
            r = res[n]
            lines.append(f'{w}: {n}: LDS arena {int(r["lds"][0])} B, layout {int(r["layout"][0])}; ' +
                         '; '.join(f'{steps} steps: {rate[(n, steps)][0]:.0f} scen/s (converged {rate[(n, steps)][1]:.3f}, {rate[(n, steps)][2]:.2f} QPs per solve)' for steps in (20, 120) if (n, steps) in rate))
        if len(names) == 2:
            a, b = res[names[0]], res[names[1]]
            same = (a['status'] == b['status']) & (a['it'] == b['it']) & (a['qp'] == b['qp'])
            conv = same & (a['status'] <= 1)
            du = np.abs(a['u'] - b['u']).max(axis=1) / np.maximum(1.0, np.abs(a['u']).max(axis=1))
            lines.append(f'{w}: identical (status, iterations, QP solves) on {int(same.sum())} of {len(same)} scenarios; iterates of the identical converged ones: max rel diff {du[conv].max() if conv.any() else float("nan"):.2e}, median {np.median(du[conv]) if conv.any() else float("nan"):.2e}')
            for steps in (20, 120):
                if all((n, steps) in rate for n in names):
                    lines.append(f'{w}: {steps} steps: b256 / b512 = {rate[(names[1], steps)][0] / rate[(names[0], steps)][0]:.3f}')
        lines.append('')
    text = '\n'.join(lines)
    print(text)
    pathlib.Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    pathlib.Path(args.out).write_text(text + '\n')


if __name__ == '__main__':
    main()
