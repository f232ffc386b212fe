#!/usr/bin/env python3
"""The same scenarios through layout / block-size variants of the library (development: row N1): the product build, the product build
with the big layout forced (DGSQP_FORCE_BIG) and with the gradients in the scratch as well (DGSQP_FORCE_GD_GLOBAL), and the
-DDG_BLOCK=256 build; every variant in its own process, compared with the first.
usage: python tools/debug/gpu_layout_variants.py [workload] [B]"""
import os
import pathlib
import subprocess
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
CS = ROOT / 'dgsqp_amd' / 'csrc'
SOLVE = r'''
import sys, numpy as np
sys.path.insert(0, %r)
import bench
from dgsqp_amd.montecarlo import sample_scenarios
from dgsqp_amd.solver import DGSQP
g = bench.make_game(sys.argv[1])
sv = DGSQP(*g.solver_args(), print_method=None)
x0, u = sample_scenarios(g, int(sys.argv[3]), seed=7)
r = sv.solve_batch(x0, u)
d = sv.dims
np.savez(sys.argv[2], u=r['u'], l=r['l'], status=r['status'], it=r['num_iters'], qp=r['qp_solves'], lds=np.array([d.lds_bytes]), layout=np.array([d.layout]), ms=np.array([r.get('kernel_ms', 0.0)]))
'''
w = sys.argv[1] if len(sys.argv) > 1 else 'dyn_curve_N25'
B = sys.argv[2] if len(sys.argv) > 2 else '256'
variants = [('b512', 'libdgsqp_hip.so', {}), ('b512 big', 'libdgsqp_hip.so', {'DGSQP_FORCE_BIG': '1'}),
            ('b512 big gd_global', 'libdgsqp_hip.so', {'DGSQP_FORCE_BIG': '1', 'DGSQP_FORCE_GD_GLOBAL': '1'}),
            ('b256', 'libdgsqp_hip_b256.so', {}), ('b256 big', 'libdgsqp_hip_b256.so', {'DGSQP_FORCE_BIG': '1'}),
            ('b256 big gd_global', 'libdgsqp_hip_b256.so', {'DGSQP_FORCE_BIG': '1', 'DGSQP_FORCE_GD_GLOBAL': '1'})]
base = None
for name, lib, env in variants:
    f = f'/tmp/lv_{w}_{name.replace(" ", "_")}.npz'
    out = subprocess.run([sys.executable, '-c', SOLVE % str(ROOT), w, f, B], env=dict(os.environ, DGSQP_HIP_LIB=str(CS / lib), **env), capture_output=True, text=True, timeout=600)
    if out.returncode != 0:
        print(f'{w} {name}: FAILED: {out.stderr.strip().splitlines()[-1] if out.stderr.strip() else out.returncode}')
        continue
    r = dict(np.load(f))
    if base is None:
        base = r
    same = (r['status'] == base['status']) & (r['it'] == base['it']) & (r['qp'] == base['qp'])
    conv = same & (base['status'] <= 1)
    du = np.abs(r['u'] - base['u']).max(axis=1) / np.maximum(1.0, np.abs(base['u']).max(axis=1))
    print(f'{w} {name}: layout {int(r["layout"][0])}, arena {int(r["lds"][0])} B, kernel {float(r["ms"][0]):.1f} ms; converged {np.mean(r["status"] <= 1):.3f}, QPs per solve {r["qp"].mean():.2f}; '
          f'identical to the first variant on {int(same.sum())}/{len(same)}, iterates max rel diff {du[conv].max() if conv.any() else float("nan"):.2e}; status histogram {np.bincount(r["status"], minlength=6).tolist()}')
