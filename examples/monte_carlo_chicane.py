#!/usr/bin/env python3
"""The DG-SQP leg of scripts/DGSQP_ALGAMES_monte_carlo_chicane.py (game :161-330, sampler + PID warm start :384-467, solve :487-490,
pickle ``data_c_<theta>_N_<N>.pkl`` with the list under ``'dgsqp'`` :501-511) on the MI355X library.  The ALGAMES leg of the script is
out of scope: the pickle carries ``algames=[]``.

    python examples/monte_carlo_chicane.py --num-mc 1000 --N 25 --theta 45 --out /tmp/data_c_45_N_25.pkl
"""
import argparse

from _driver import add_common_arguments, dump, monte_carlo, records, report
from dgsqp_amd.montecarlo import kinematic_racing_game


def main(argv=None):
    ap = add_common_arguments(argparse.ArgumentParser(), num_mc=1000)
    ap.add_argument('--N', type=int, default=25, help='horizon (chicane.py:132-134: 25)')
    ap.add_argument('--theta', type=float, default=45, help='swept angle of the two curves in degrees (chicane.py:130)')
    args = ap.parse_args(argv)
    game = kinematic_racing_game('chicane', theta_deg=args.theta, N=args.N, reg=1e-3)        # chicane.py:164: reg = 1e-3
    res, x0, _, wall = monte_carlo(game, args.num_mc, args.batch, 1 if args.seed is None else args.seed, args.qp)   # chicane.py:136: seed 1
    recs = records(game, res, x0, wall)
    report(game.name, recs, wall)
    data = dict(dgsqp=recs, algames=[], track=game.track, agent_dyn_configs=[m.model_config for m in game.joint_model.dynamics_models],
                joint_model_config=game.joint_model.model_config)
    if args.out:
        dump(args.out, data)
    return data


if __name__ == '__main__':
    main()
