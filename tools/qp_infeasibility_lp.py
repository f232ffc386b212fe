"""Is every QP that the active-set method (oracle and device share it) calls INFEASIBLE really infeasible?

For the first B sampled scenarios of a game the numpy loop (oracle/pyref.py) is run with the C++ oracle's Goldfarb-Idnani QP
(qp='gi': the path the device follows); every QP it declares infeasible is handed to an LP that knows nothing of the QP solver:

        min t   s.t.   G du - t 1 <= -g,   t >= -1           (scipy.optimize.linprog, HiGHS)

The linearised constraints G du <= -g have a solution iff t* <= 0; t* > 0 is the smallest uniform relaxation that makes them
feasible -- an infeasibility certificate independent of the condition number of the projected Hessian (which does not enter).
Also printed: where in the horizon the violated rows sit at the START point (which rows make the first QP infeasible).

    usage: python tools/qp_infeasibility_lp.py <game> <B> [nproc]      games: see tools/ref_stats.py
Writes tests/golden/qp_infeasible_<game>.npz: the (Q-free) data G, g of up to 12 infeasible QPs with their t*, and of up to 4
feasible ones, for tests/test_oracle.py::test_infeasible_verdicts_are_backed_by_an_lp.
"""
import os, sys, pathlib
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "1")
import numpy as np
import scipy.optimize
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tools')); sys.path.insert(0, str(ROOT / 'tests'))
import multiprocessing as mp
import dgsqp_amd.montecarlo as mc
from dgsqp_amd.solver import build_problem, build_params
from oracle import oracle, pyref
from ref_stats import GAMES


def lp_relaxation(G, g):
    """t* of  min t s.t. G du - t <= -g, t >= -1  (du free).  > 0: no du satisfies the linearised constraints."""
    nc, n = G.shape
    if not (np.isfinite(G).all() and np.isfinite(g).all()):
        return np.inf, -1          # non-finite data (a diverged rollout): no QP to speak of
    A = np.hstack([G, -np.ones((nc, 1))])
    c = np.zeros(n + 1); c[-1] = 1.0
    r = scipy.optimize.linprog(c, A_ub=A, b_ub=-g, bounds=[(None, None)] * n + [(-1.0, None)], method='highs')
    return (r.x[-1] if r.status == 0 else np.nan), r.status


class Harvest(pyref.PyRef):
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.qps = []          # (G, g, flag)

    def solve_qp(self, Q, q, G, g):
        du, lhat = super().solve_qp(Q, q, G, g)
        self.qps.append((G.copy(), g.copy(), bool(np.isnan(du).any())))
        return du, lhat


def one(args):
    name, b, B = args
    g = GAMES[name][0]()
    P, par = build_problem(*g.solver_args()), build_params(g.params, eig_floor=1e-10)
    x0, uws = mc.sample_scenarios(g, B, seed=GAMES[name][1])
    nua = 2
    u = np.concatenate([uws[:, :, nua * a:nua * a + nua].reshape(B, -1) for a in range(uws.shape[2] // nua)], axis=1)
    r = Harvest(P, par, qp='gi')
    try:
        with np.errstate(all='ignore'):
            s = r.solve(x0[b], u[b])
        msg = s['msg']
    except (ValueError, FloatingPointError, np.linalg.LinAlgError):
        msg = 'exception'
    out = []
    for k, (G, gg, bad) in enumerate(r.qps):
        if bad or k == 0:
            t, st = lp_relaxation(G, gg)
            out.append((k, bad, t, st, G if (bad or k == 0) else None, gg))
    rows = oracle.rows(P)
    ev0 = oracle.evaluate(P, x0[b], u[b], None, 0)
    viol = np.nonzero(ev0['g'] > 1e-9)[0]
    return b, msg, len(r.qps), out, [(int(rows[i, 0]), int(rows[i, 1]), float(ev0['g'][i])) for i in viol]


if __name__ == '__main__':
    name, B = sys.argv[1], int(sys.argv[2])
    nproc = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    with mp.Pool(nproc) as pool:
        res = pool.map(one, [(name, b, B) for b in range(B)], chunksize=1)
    n_bad = n_bad_true = n_first = n_first_inf = n_nonfinite = 0
    keepG, keepg, keept, keepflag = [], [], [], []
    tvals = []
    for b, msg, nqp, out, viol in res:
        for k, bad, t, st, G, gg in out:
            if bad:
                n_bad += 1
                n_nonfinite += bool(st == -1)
                n_bad_true += bool(t > 1e-9 and st == 0)
                if st == 0:
                    tvals.append(t)
                if st == 0 and len([f for f in keepflag if f]) < 12:
                    keepG.append(G); keepg.append(gg); keept.append(t); keepflag.append(True)
            elif k == 0:
                n_first += 1
                n_first_inf += bool(t > 1e-9)
                if len([f for f in keepflag if not f]) < 4:
                    keepG.append(G); keepg.append(gg); keept.append(t); keepflag.append(False)
    msgs = [m for _, m, _, _, _ in res]
    print(f'# {name}: first {B} scenarios; numpy loop with the oracle\'s active-set QP: ' + ', '.join(f'{m} {msgs.count(m)}' for m in sorted(set(msgs))))
    print(f'QPs declared failed by the active-set method: {n_bad}, of which {n_nonfinite} have non-finite data (diverged rollout); LP certificate t* > 1e-9 (really infeasible): {n_bad_true}; '
          f't* min / median / max: {np.min(tvals) if tvals else float("nan"):.3g} / {np.median(tvals) if tvals else float("nan"):.3g} / {np.max(tvals) if tvals else float("nan"):.3g}')
    print(f'first QPs the active-set method SOLVED: {n_first}; of those the LP calls infeasible: {n_first_inf}')
    kinds = {}
    for b, msg, nqp, out, viol in res:
        for typ, k, v in viol:
            kinds.setdefault((typ, 'k=0' if k == 0 else 'k>0'), []).append(v)
    print('rows violated at the START point (row type of oracle.rows: 0 obstacle, 1/2 rate, 3/4 input box, 5/6 state box; stage): ' +
          ', '.join(f'type {t} {kk}: {len(v)} rows, max {max(v):.3g}' for (t, kk), v in sorted(kinds.items())))
    if keepG:
        np.savez_compressed(ROOT / 'tests' / 'golden' / f'qp_infeasible_{name}.npz', G=np.array(keepG), g=np.array(keepg), t=np.array(keept), infeasible=np.array(keepflag))
