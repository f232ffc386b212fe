#!/usr/bin/env python3
"""scripts/DGSQP_monte_carlo_agents.py -- the curve-track race with M = 2, 3, 4 cars and horizons 15, 20, 25 (experiment grid
:126-131, game :158-223, sampler + PID warm start :262-308, solve :323-331) -- on the MI355X library.  One pickle per (M, N),
``data_c_<theta>_M_<M>_N_<N>.pkl`` = ``dict(sqgames=[records], track, agent_dyn_configs, joint_model_config)`` (:333-342).

    python examples/monte_carlo_agents.py --num-mc 500 --agents 2 3 --N 15 25 --out /tmp/agents_data
"""
import argparse
import pathlib

from _driver import add_common_arguments, dump, monte_carlo, records, report
from dgsqp_amd.montecarlo import kinematic_racing_game


def main(argv=None):
    ap = add_common_arguments(argparse.ArgumentParser(), num_mc=500)
    ap.add_argument('--N', type=int, nargs='+', default=[15, 20, 25], help='horizons (agents.py:127)')
    ap.add_argument('--agents', type=int, nargs='+', default=[2, 3, 4], help='numbers of cars (agents.py:126)')
    ap.add_argument('--theta', type=float, default=45, help='swept angle of the curve in degrees')
    args = ap.parse_args(argv)
    out = {}
    for M in args.agents:
        for N in args.N:
            game = kinematic_racing_game('curve', theta_deg=args.theta, N=N, M=M, reg=1e-3)
            res, x0, _, wall = monte_carlo(game, args.num_mc, args.batch, 1 if args.seed is None else args.seed, args.qp)
            recs = records(game, res, x0, wall)
            report(f'M={M} N={N}', recs, wall)
            data = dict(sqgames=recs, track=game.track, agent_dyn_configs=[m.model_config for m in game.joint_model.dynamics_models],
                        joint_model_config=game.joint_model.model_config)
            out[(M, N)] = data
            if args.out:
                dump(pathlib.Path(args.out) / f'data_c_{args.theta:g}_M_{M}_N_{N}.pkl', data)
    return out


if __name__ == '__main__':
    main()
