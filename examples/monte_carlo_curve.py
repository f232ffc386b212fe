#!/usr/bin/env python3
"""The DG-SQP leg of the reference's Monte-Carlo study on the curve track (scripts/DGSQP_ALGAMES_monte_carlo_curve.py: game
:161-330, sampler and PID warm start :384-467, solve loop :482-485, pickle :486-500, table scripts/process_data_curve.py:37-110) on
the MI355X library: every sample is one scenario of a batch, the batches share launches (dgsqp_launch_staged_group).

    python examples/monte_carlo_curve.py --num-mc 8192 --N 25 --out /tmp/data_curve_N_25.pkl
"""
import argparse
import pathlib
import sys
import time

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from dgsqp_amd.montecarlo import kinematic_racing_game, sample_scenarios          # noqa: E402
from dgsqp_amd.results import save_monte_carlo, summarize_like_process_data     # noqa: E402
from dgsqp_amd.solver import DGSQP, solve_batches                                # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--num-mc', type=int, default=4096, help='samples (the reference script: 1000)')
    ap.add_argument('--N', type=int, default=25, help='horizon (curve.py: 25)')
    ap.add_argument('--batch', type=int, default=1024, help='scenarios per batch; up to 8 batches share one launch')
    ap.add_argument('--seed', type=int, default=1)
    ap.add_argument('--out', default=None, help='pickle in the layout process_data_curve.py reads')
    ap.add_argument('--host-sampler', action='store_true', help="the scripts' sequential numpy draws (np.random.default_rng(seed), curve.py:384-467) "
                    'instead of the counter-based sampler on the device')
    args = ap.parse_args()

    game = kinematic_racing_game('curve', N=args.N, reg=0.0)                       # curve.py:161: reg = 0
    bounds = list(range(0, args.num_mc, args.batch)) + [args.num_mc]
    sizes = [b - a for a, b in zip(bounds[:-1], bounds[1:])]
    nfull = sum(1 for n in sizes if n == args.batch)
    solvers = [DGSQP(*game.solver_args(), print_method=None) for _ in range(max(1, min(8, nfull)))]
    t0 = time.time()
    results = []
    if args.host_sampler:
        x0, u_ws = sample_scenarios(game, args.num_mc, seed=args.seed)             # rejection sampling + PID warm starts on the host
        chunks = [(x0[a:b], u_ws[a:b]) for a, b in zip(bounds[:-1], bounds[1:])]
        for i in range(0, nfull, len(solvers)):                                     # groups of equal-sized batches: one launch each
            grp = chunks[i:min(i + len(solvers), nfull)]
            results += solve_batches(solvers[:len(grp)], grp)
        for c in chunks[nfull:]:                                                     # the ragged remainder on its own
            results.append(solvers[0].solve_batch(*c))
    else:
        # batch j: the device sampler with seed + j (placement, PID warm start, collision rejection, compaction), left staged on its
        # handle; the group is solved by one launch; only the results come back
        for i in range(0, len(sizes), len(solvers)):
            grp = sizes[i:i + len(solvers)]
            for j, (sv, n) in enumerate(zip(solvers, grp)):
                sv.sample_batch(game, n, seed=args.seed + i + j, stage=True, fetch=False)
            same = [k for k, n in enumerate(grp) if n == grp[0]]
            results += solve_batches([solvers[k] for k in same], [grp[k] for k in same])
            for k in range(len(same), len(grp)):                                     # (a ragged last batch)
                results += solve_batches([solvers[k]], [grp[k]])
    wall = time.time() - t0
    res = {k: np.concatenate([r[k] for r in results]) for k in ('u', 'l', 'x', 'status', 'num_iters', 'qp_solves', 'cond', 'cost')}
    res['msg'] = sum((r['msg'] for r in results), [])
    print(f'{args.num_mc} samples in {wall:.2f} s ({args.num_mc / wall:.0f} scenarios/s); converged {np.mean(res["status"] <= 1):.3f}')
    data = save_monte_carlo(args.out, res, game.params, key='sqgames', wall_time=wall) if args.out else None
    if data is not None:
        print('process_data table:', summarize_like_process_data(data['sqgames']))
    return res


if __name__ == '__main__':
    main()
