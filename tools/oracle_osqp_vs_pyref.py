"""Build container / any CPU: the C++ oracle with its OSQP restatement (oracle/osqp.hpp, qp_method = 1) against the numpy loop with the
numpy restatement (tests/golden/pyref_osqp_<game>.npz) -- two CPU implementations of the same algorithm that differ in their eigen
solver (Jacobi vs numpy.linalg.eigh), LSQR (own restatement vs scipy) and dense LU: how often rounding-level differences alone change
a scenario's path.  Also the oracle's own reproducibility under 1e-13 input perturbations (the `stable` mask of the parity tests).
usage: oracle_osqp_vs_pyref.py [game ...]"""
import pathlib
import sys
import time

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tools')); sys.path.insert(0, str(ROOT / 'tests'))
from dgsqp_amd.solver import build_problem, build_params  # noqa: E402
from oracle import oracle  # noqa: E402
from ref_stats import GAMES  # noqa: E402
from conftest import agent_major, stable_mask  # noqa: E402

names = sys.argv[1:] or ['dyn_curve_N25', 'kb_curve_N25', 'kb_chicane_N25', 'kb_barc2_N15', 'merge_N20']
for name in names:
    ref = np.load(ROOT / 'tests' / 'golden' / f'pyref_osqp_{name}.npz')
    g = GAMES[name][0]()
    P, par = build_problem(*g.solver_args()), build_params(g.params, qp_method='osqp')
    u = agent_major(ref['u_ws'])
    t = time.time()
    o = oracle.solve_batch(P, par, ref['x0'], u, nthreads=8)
    dt = time.time() - t
    ident = (o['status'] == ref['status']) & (o['num_iters'] == ref['num_iters']) & (o['qp_solves'] == ref['qp_solves'])
    ident_x = ident | ((o['status'] == 4) & (ref['status'] == 4))
    stable = stable_mask(oracle, P, par, ref["x0"], u, o)
    cd, cr = o['status'] <= 1, ref['status'] <= 1
    print(f'{name:16s}: C++ oracle (OSQP) {len(u)} scenarios in {dt:.0f} s | identical to the numpy loop {ident.mean():.3f} (exceptions on both sides counted as identical: {ident_x.mean():.3f}) | '
          f'oracle reproduces itself under 1e-13 perturbations {stable.mean():.3f} | identical on the oracle-stable {ident_x[stable].mean():.3f} | converged {cd.mean():.3f} vs {cr.mean():.3f}', flush=True)
    np.savez_compressed(f'/tmp/oracle_osqp_{name}.npz', status=o['status'], num_iters=o['num_iters'], qp_solves=o['qp_solves'], u=o['u'], l=o['l'], stable=stable)
