#!/usr/bin/env python3
"""The DG-SQP leg of scripts/DGSQP_comp_monte_carlo.py -- two cars racing on the L_track_barc circuit, N = 15, reg = 0 (game
:62-345, sampler :360-382, PID warm start :384-445, solve :488-493) -- on the MI355X library.  The script writes ONE pickle per
sample, ``sample_<i>.pkl`` = ``dict(dgsqp=<record>, algames=<record>, env=<track>)`` (:500-506), and so does this driver (the ALGAMES
record is out of scope, SURVEY.md section 2: ``algames=None``; scripts/process_data_comp.py:36-60 reads ``data['dgsqp']``).

    python examples/monte_carlo_comp.py --num-mc 1000 --out /tmp/comp_data
"""
import argparse
import pathlib

from _driver import add_common_arguments, dump, monte_carlo, records, report
from dgsqp_amd.montecarlo import barc_racing_game


def main(argv=None):
    ap = add_common_arguments(argparse.ArgumentParser(), num_mc=1000)
    ap.add_argument('--N', type=int, default=15, help='horizon (comp.py:62: 15)')
    ap.add_argument('--agents', type=int, default=2, help='cars (the script: 2; 3 at N = 25 is BASELINE configs[2])')
    args = ap.parse_args(argv)
    game = barc_racing_game(N=args.N, M=args.agents, reg=0.0)                                 # comp.py:169: reg = 0
    res, x0, _, wall = monte_carlo(game, args.num_mc, args.batch, 0 if args.seed is None else args.seed, args.qp)   # comp.py:360: seed 0
    recs = records(game, res, x0, wall)
    report(game.name or 'kb_barc', recs, wall)
    samples = [dict(dgsqp=r, algames=None, env=game.track) for r in recs]
    if args.out:
        for i, s in enumerate(samples):
            dump(pathlib.Path(args.out) / f'sample_{i + 1}.pkl', s)
    return samples


if __name__ == '__main__':
    main()
