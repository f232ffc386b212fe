"""GPU box: full solves with the device's OSQP restatement (qp_method='osqp') against the numpy loop with the restated OSQP
(tests/golden/pyref_osqp_<game>.npz, written by tools/ref_stats.py) -- and, for comparison, the default exact active-set QP.
Prints, per game: identical (status, iterations, QP solves), same converged flag, converged fractions, iterate differences of the
identical converged scenarios.   usage: gpu_osqp_vs_pyref.py [game ...]"""
import pathlib
import sys
import time

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tools'))
from dgsqp_amd.solver import DGSQP  # noqa: E402
from ref_stats import GAMES  # noqa: E402

names = sys.argv[1:] or ['dyn_curve_N25', 'kb_curve_N25', 'kb_chicane_N25', 'kb_barc2_N15', 'merge_N20']
for name in names:
    ref = np.load(ROOT / 'tests' / 'golden' / f'pyref_osqp_{name}.npz')
    g = GAMES[name][0]()
    for method in ('osqp', 'active_set'):
        s = DGSQP(*g.solver_args(), print_method=None, qp_method=method)
        t = time.time()
        res = s.solve_batch(ref['x0'], ref['u_ws'])
        dt = time.time() - t
        st = np.where(res['status'] == 4, 4, res['status'])
        ident = ((st == ref['status']) & (res['num_iters'] == ref['num_iters']) & (res['qp_solves'] == ref['qp_solves'])) | ((st == 4) & (ref['status'] == 4))   # (a solve that raises in the reference has no counts)
        stable = ref['stable']
        rest = ident[~stable].mean() if (~stable).any() else float('nan')
        cd, cr = res['status'] <= 1, ref['status'] <= 1
        idc = ident & cd & cr
        err = np.array([np.abs(res['u'][b] - ref['u'][b]).max() / max(1.0, np.abs(ref['u'][b]).max()) for b in np.nonzero(idc)[0]])
        errl = np.array([np.abs(res['l'][b] - ref['l'][b]).max() / max(1.0, np.abs(ref['l'][b]).max()) for b in np.nonzero(idc)[0]])
        print(f'{name:16s} qp_method {method:10s}: {len(st)} scenarios in {dt:6.2f} s | identical (status, iters, QPs) {ident.mean():.3f}, on the {int(stable.sum())} scenarios the numpy loop itself reproduces under 1e-13 perturbations {ident[stable].mean():.3f}, on the other {int((~stable).sum())} {rest:.3f} | same converged flag {np.mean(cd == cr):.3f} | '
              f'converged device {cd.mean():.3f} numpy+OSQP {cr.mean():.3f} | mean iters (commonly converged) {res["num_iters"][cd & cr].mean():.2f} vs {ref["num_iters"][cd & cr].mean():.2f} | '
              f'identical converged: u median {np.median(err) if len(err) else float("nan"):.1e} max {err.max() if len(err) else float("nan"):.1e} (> 1e-5: {int((err > 1e-5).sum())}), '
              f'l max {errl.max() if len(errl) else float("nan"):.1e}', flush=True)
        if method == 'osqp':
            out = ROOT / 'gpurun_out' / 'osqp_vs_pyref'
            out.mkdir(parents=True, exist_ok=True)
            np.savez_compressed(out / f'device_osqp_{name}.npz', status=res['status'], num_iters=res['num_iters'], qp_solves=res['qp_solves'], u=res['u'], l=res['l'])
            bad = np.nonzero(~ident)[0]
            for b in bad[:8]:
                print(f'      scn {b}: device {int(res["status"][b]), int(res["num_iters"][b]), int(res["qp_solves"][b])} numpy+OSQP {int(ref["status"][b]), int(ref["num_iters"][b]), int(ref["qp_solves"][b])}')
