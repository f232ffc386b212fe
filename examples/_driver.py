"""What the five Monte-Carlo drivers of examples/ share: sample -> batches -> grouped launches -> per-sample records in the layout the
reference's scripts pickle (``dict(solve_info=<DGSQP.solve() return value>, params=<DGSQPParams>, init=<joint VehicleStates>)``,
e.g. scripts/DGSQP_ALGAMES_monte_carlo_chicane.py:487-511).  Every sample of the reference's ``for i in range(samples)`` loop is one
scenario of a batch here; equal-sized batches share a launch (``solve_batches`` -> ``dgsqp_launch_staged_group``)."""
import copy
import pathlib
import pickle
import sys
import time

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
from dgsqp_amd.montecarlo import sample_scenarios                                  # noqa: E402
from dgsqp_amd.results import solve_infos, summarize_like_process_data            # noqa: E402
from dgsqp_amd.solver import DGSQP, solve_batches                                  # noqa: E402


def add_common_arguments(ap, num_mc=1000, batch=1024):
    ap.add_argument('--num-mc', type=int, default=num_mc, help='samples (the reference scripts: 1000 / 500 / 100)')
    ap.add_argument('--batch', type=int, default=batch, help='scenarios per batch; up to 8 equal-sized batches share one launch')
    ap.add_argument('--seed', type=int, default=None, help="seed of np.random.default_rng (default: the script's own)")
    ap.add_argument('--out', default=None, help='where the pickle(s) go (a file, or a directory for the per-sample scripts)')
    ap.add_argument('--qp', choices=('active_set', 'osqp'), default='active_set',
                    help="how _solve_qp is computed: the exact KKT point (default) or OSQP's own ADMM + polish arithmetic (the reference's conic('qp', 'osqp'))")
    return ap


def monte_carlo(game, num_mc, batch, seed, qp_method='active_set', solver_kw=None):
    """``num_mc`` samples of ``game`` (the script's sequential ``np.random.default_rng(seed)`` draws, rejection sampling and warm
    starts: ``montecarlo.sample_scenarios``), solved in batches.  Returns (res, x0, u_ws, wall seconds): ``res`` = the arrays of
    ``solve_batch`` concatenated in sample order, plus ``msg``."""
    x0, u_ws = sample_scenarios(game, num_mc, seed=seed)
    bounds = list(range(0, num_mc, batch)) + [num_mc]
    chunks = [(x0[a:b], u_ws[a:b]) for a, b in zip(bounds[:-1], bounds[1:])]
    nfull = sum(1 for c in chunks if len(c[0]) == batch)
    solvers = [DGSQP(*game.solver_args(), print_method=None, qp_method=qp_method, **(solver_kw or {})) for _ in range(max(1, min(8, nfull)))]
    t0 = time.time()
    results = []
    for i in range(0, nfull, len(solvers)):                 # groups of equal-sized batches: one launch each
        grp = chunks[i:min(i + len(solvers), nfull)]
        results += solve_batches(solvers[:len(grp)], grp)
    for c in chunks[nfull:]:                                # the ragged remainder on its own
        results.append(solvers[0].solve_batch(*c))
    wall = time.time() - t0
    res = {k: np.concatenate([r[k] for r in results]) for k in ('u', 'l', 'x', 'status', 'num_iters', 'qp_solves', 'cond', 'cost')}
    res['msg'] = sum((list(r['msg']) for r in results), [])
    return res, x0, u_ws, wall


def records(game, res, x0, wall, params=None):
    """One ``dict(solve_info, params, init)`` per sample; ``init`` = the joint VehicleState list the scripts store (inputs zero, as at
    chicane.py:476-478)."""
    infos = solve_infos(res, wall)
    out = []
    for b, si in enumerate(infos):
        init = game.joint_model.qu2state(None, x0[b], np.zeros(game.joint_model.n_u))
        out.append(dict(solve_info=si, params=copy.deepcopy(params if params is not None else game.params), init=init))
    return out


def dump(path, obj):
    path = pathlib.Path(path)
    path.parent.mkdir(parents=True, exist_ok=True)
    with open(path, 'wb') as f:
        pickle.dump(obj, f)


def report(name, recs, wall):
    n = len(recs)
    print(f'{name}: {n} samples in {wall:.2f} s ({n / max(wall, 1e-9):.0f} scenarios/s); process_data table: {summarize_like_process_data(recs)}')
