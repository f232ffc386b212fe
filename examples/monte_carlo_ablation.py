#!/usr/bin/env python3
"""scripts/DGSQP_monte_carlo_ablation.py -- the ablation study: the same samples solved by DG-SQP with the watchdog and the
'stat_l1' merit function (``sqgames_all``, :166-180) and by plain backtracking on the 'stat' merit (``sqgames_none``, :183-197), curve
track with a 90 degree turn, horizons 15 / 20 / 25 (:140-153), sampler that also rejects car 2 behind car 1 (:374-396) -- on the MI355X
library.  One pickle per horizon, ``data_c_<theta>_N_<N>.pkl`` = ``dict(sqgames_all=[...], sqgames_none=[...], track, ...)`` (:494-504;
scripts/process_data_ablation.py:25-35 reads both lists).

    python examples/monte_carlo_ablation.py --num-mc 100 --N 15 20 25 --out /tmp/ablation_data
"""
import argparse
import pathlib

from _driver import add_common_arguments, dump, monte_carlo, records, report
from dgsqp_amd.montecarlo import ablation_racing_game


def main(argv=None):
    ap = add_common_arguments(argparse.ArgumentParser(), num_mc=100)
    ap.add_argument('--N', type=int, nargs='+', default=[15, 20, 25], help='horizons (ablation.py:140)')
    ap.add_argument('--theta', type=float, default=90, help='swept angle of the curve in degrees (ablation.py:143)')
    args = ap.parse_args(argv)
    out = {}
    for N in args.N:
        seed = 1 if args.seed is None else args.seed
        data = {}
        for key, nms, merit in (('sqgames_all', True, 'stat_l1'), ('sqgames_none', False, 'stat')):
            game = ablation_racing_game(N=N, nonmono_ls=nms, merit_function=merit, theta_deg=args.theta)
            res, x0, _, wall = monte_carlo(game, args.num_mc, args.batch, seed, args.qp)          # the same seed: the same samples for both solvers
            data[key] = records(game, res, x0, wall)
            report(f'{key} N={N}', data[key], wall)
        data.update(track=game.track, agent_dyn_configs=[m.model_config for m in game.joint_model.dynamics_models],
                    joint_model_config=game.joint_model.model_config)
        out[N] = data
        if args.out:
            dump(pathlib.Path(args.out) / f'data_c_{args.theta:g}_N_{N}.pkl', data)
    return out


if __name__ == '__main__':
    main()
