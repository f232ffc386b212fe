#!/usr/bin/env python3
"""scripts/DGSQP_merge_monte_carlo.py -- three kinematic unicycles (rk3) merging on a highway ramp, N = 20, reg = 0, zero warm start
(game :95-345, sampler :421-480, solve :505-509) -- on the MI355X library.  One pickle per sample, ``sample_<i>.pkl`` =
``dict(dgsqp=<record>, env=<merge environment>)`` (:519-525; scripts/process_data_merge.py:27-40 reads ``data['dgsqp']``); the
environment object of the script is a plotting aid and is stored as None.  ``--agents 6 --N 25`` is BASELINE configs[4].

    python examples/monte_carlo_merge.py --num-mc 1000 --out /tmp/merge_data
"""
import argparse
import pathlib

from _driver import add_common_arguments, dump, monte_carlo, records, report
from dgsqp_amd.montecarlo import merge_game


def main(argv=None):
    ap = add_common_arguments(argparse.ArgumentParser(), num_mc=1000)
    ap.add_argument('--N', type=int, default=20, help='horizon (merge.py:89: 20)')
    ap.add_argument('--agents', type=int, default=3, help='cars (the script: 3; 6 at N = 25 is BASELINE configs[4])')
    args = ap.parse_args(argv)
    game = merge_game(N=args.N, reg=0.0, M=args.agents)                                       # merge.py:182: reg = 0
    res, x0, _, wall = monte_carlo(game, args.num_mc, args.batch, 1 if args.seed is None else args.seed, args.qp)   # merge.py:421: seed 1
    recs = records(game, res, x0, wall)
    report(game.name or 'merge', recs, wall)
    samples = [dict(dgsqp=r, env=None) for r in recs]
    if args.out:
        for i, s in enumerate(samples):
            dump(pathlib.Path(args.out) / f'sample_{i + 1}.pkl', s)
    return samples


if __name__ == '__main__':
    main()
