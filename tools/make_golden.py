"""Regenerates tests/golden/*.  Run in the BUILD container (needs /root/reference for the parameter
defaults; everything else comes from the repo's own CPU oracle, since the reference solver cannot run
without CasADi+OSQP -- parity is unpinned, see oracle/dgsqp_oracle.cpp)."""
import dataclasses
import json
import os
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
GOLD = ROOT / 'tests' / 'golden'
GOLD.mkdir(parents=True, exist_ok=True)


def defaults(cls):
    return {f.name: (None if f.default is dataclasses.MISSING else f.default) for f in dataclasses.fields(cls)}


def param_defaults():
    ref = pathlib.Path('/root/reference')
    sys.path.insert(0, str(ref))
    os.environ.setdefault('MPLBACKEND', 'Agg')
    from DGSQP.solvers import solver_types as rst
    from DGSQP.dynamics import model_types as rmt
    sys.path.remove(str(ref))
    out = {'DGSQPParams': defaults(rst.DGSQPParams), 'DGSQPV2Params': defaults(rst.DGSQPV2Params), 'PIDParams': defaults(rst.PIDParams),
           'KinematicBicycleConfig': defaults(rmt.KinematicBicycleConfig),
           'DynamicBicycleConfig': defaults(rmt.DynamicBicycleConfig)}
    (GOLD / 'param_defaults.json').write_text(json.dumps(out, indent=1, sort_keys=True))
    for m in [k for k in sys.modules if k == 'DGSQP' or k.startswith('DGSQP.')]:
        del sys.modules[m]


def solve_fixtures(only=None):
    import copy
    from dgsqp_amd.montecarlo import (ablation_racing_game, barc_racing_game, kinematic_racing_game, dynamic_racing_game, merge_game,
                                      sample_scenarios)
    from dgsqp_amd.solver import build_problem, build_params
    from oracle import oracle
    for name, game, B, seed in (('kb_chicane_N15', kinematic_racing_game('chicane', N=15), 32, 11),
                                ('kb_curve_N10', kinematic_racing_game('curve', N=10), 32, 12),
                                ('dyn_curve_N15', dynamic_racing_game(N=15, rk4_substeps=4, game_def='curve'), 16, 13),
                                ('dyn_curve_N25', dynamic_racing_game(N=25, rk4_substeps=10), 64, 1),     # BASELINE configs[1]
                                ('kb_barc2_N15', barc_racing_game(N=15, M=2), 32, 0),        # reg = 0 (comp.py:169)
                                ('merge_N8', merge_game(N=8), 16, 1)) + tuple(                # reg = 0 (merge.py:182)
            # the ablation study (scripts/DGSQP_monte_carlo_ablation.py:166-197): nonmono_ls x merit_function at theta = 90 degrees
            (f'ablation_N{N}_{"nms" if nm else "ls"}_{mf}', ablation_racing_game(N=N, nonmono_ls=nm, merit_function=mf), 32, 1)
            for N in (15, 25) for nm in (True, False) for mf in ('stat_l1', 'stat')):
        if only and name not in only:
            continue
        P = build_problem(*game.solver_args())
        par = build_params(game.params)
        par.lsqr_atol = par.lsqr_btol = 1e-13
        x0, u_tm = sample_scenarios(game, B, seed=seed)
        u_am = np.concatenate([u_tm[:, :, 2 * a:2 * a + 2].reshape(B, -1) for a in range(u_tm.shape[2] // 2)], axis=1)
        out = oracle.solve_batch(P, par, x0, u_am, nthreads=8)
        # ``stable``: the oracle's own (status, iterations, QP solves) survive K re-runs from inputs perturbed by 1e-13 relative
        # (the size of the rounding differences between two correct implementations).  Scenarios that fail this are decided by
        # rounding noise -- e.g. _get_mu's test sum(g - s) > 0 on a sum of +-1e-16 (DGSQP.py:566-585) -- in the reference too;
        # the GPU tests demand identical control flow on the stable ones and report the rest.
        rng = np.random.default_rng(12345)
        stable = np.ones(B, bool)
        for _ in range(4):
            o2 = oracle.solve_batch(P, par, x0 * (1 + 1e-13 * rng.standard_normal(x0.shape)),
                                    u_am * (1 + 1e-13 * rng.standard_normal(u_am.shape)), nthreads=8)
            stable &= (o2['status'] == out['status']) & (o2['num_iters'] == out['num_iters']) & (o2['qp_solves'] == out['qp_solves'])
        ev0 = [oracle.evaluate(P, x0[b], u_am[b], out['l_init'][b], 1) for b in range(4)]
        np.savez_compressed(GOLD / f'{name}.npz', x0=x0, u_ws=u_tm, u=out['u'], l=out['l'], status=out['status'],
                            num_iters=out['num_iters'], qp_solves=out['qp_solves'], cond=out['cond'], cost=out['cost'],
                            l_init=out['l_init'], stable=stable, q0=np.array([e['q'] for e in ev0]), g0=np.array([e['g'] for e in ev0]),
                            Q0=np.array([e['Q'] for e in ev0]))
        print(name, 'status', np.bincount(out['status'], minlength=5), 'mean iters', out['num_iters'].mean(), 'stable', stable.sum(), '/', B)


if __name__ == '__main__':
    only = sys.argv[1:]
    if not only:
        param_defaults()
    solve_fixtures(only)
