import os
import pathlib
import sys

import numpy as np
import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def oracle():
    from oracle import oracle as orc
    orc.build()
    return orc


def agent_major(u_tm):
    """[B, N, 4] time-major two-agent inputs -> [B, n] agent-major (DGSQP.py:275-280)."""
    B = u_tm.shape[0]
    nu = u_tm.shape[2]
    return np.concatenate([u_tm[:, :, 2 * a:2 * a + 2].reshape(B, -1) for a in range(nu // 2)], axis=1)


@pytest.fixture(scope='session')
def games():
    """Small set of games used across tests: name -> (Game, ProblemT, ParamsT)."""
    from dgsqp_amd.montecarlo import ablation_racing_game, barc_racing_game, kinematic_racing_game, dynamic_racing_game, merge_game
    from dgsqp_amd.solver import build_problem, build_params
    out = {}
    ablation = tuple((f'ablation_N{N}_{"nms" if nm else "ls"}_{mf}', ablation_racing_game(N=N, nonmono_ls=nm, merit_function=mf))
                     for N in (15, 25) for nm in (True, False) for mf in ('stat_l1', 'stat'))      # DGSQP_monte_carlo_ablation.py:166-197
    for name, g in ablation + (('kb_chicane_N15', kinematic_racing_game('chicane', N=15)),
                    ('kb_chicane_N25', kinematic_racing_game('chicane', N=25)),
                    ('kb_curve_N10', kinematic_racing_game('curve', N=10)),
                    ('dyn_curve_N15', dynamic_racing_game(N=15, rk4_substeps=4, game_def='curve')),
                    ('dyn_curve_N25', dynamic_racing_game(N=25, rk4_substeps=10)),      # BASELINE configs[1] itself
                    ('merge_N8', merge_game(N=8)), ('kb_barc2_N15', barc_racing_game(N=15, M=2))):
        out[name] = (g, build_problem(*g.solver_args()), build_params(g.params))
    return out


def tight_lsqr(par):
    """Parity runs use a converged LSQR dual start: at scipy's default 1e-6 tolerance two correct
    implementations stop one Lanczos step apart and differ by ~1e-4 in l0."""
    import copy
    p = copy.copy(par)
    p.lsqr_atol = p.lsqr_btol = 1e-13
    return p


def control_flow(res):
    return np.stack([res['status'], res['num_iters'], res['qp_solves']], axis=1)


def stable_mask(oracle, P, par, x0, u_am, ref, K=3, seed=12345):
    """Scenarios whose (status, iterations, QP solves) the ORACLE ITSELF reproduces from inputs perturbed by 1e-13 relative
    (K re-runs) -- the size of the rounding differences between two correct implementations.  The others are decided by
    rounding noise in the reference's own algorithm (e.g. _get_mu's ``sum(g - s) > 0`` on a sum of +-1e-16,
    DGSQP.py:566-585) and no implementation can be expected to take the same path there."""
    rng = np.random.default_rng(seed)
    ok = np.ones(len(x0), bool)
    for _ in range(K):
        o2 = oracle.solve_batch(P, par, x0 * (1 + 1e-13 * rng.standard_normal(x0.shape)),
                                u_am * (1 + 1e-13 * rng.standard_normal(u_am.shape)), nthreads=min(len(x0), os.cpu_count() or 1))
        ok &= (control_flow(o2) == control_flow(ref)).all(axis=1)
    return ok


def assert_control_flow_parity(res, ref, stable, tag='', min_stable_same=0.95, max_conv_gap=0.05, min_stable_frac=0.5):
    """Device vs oracle: identical (status, iterations, QP solves) on the scenarios the oracle itself reproduces under 1e-13
    perturbations (at most 1 in 20 of them may still fork: K re-runs do not find every fragile decision), converged
    fraction within ``max_conv_gap``.  Returns the mask of identical scenarios; prints the forks (pytest -s / on failure)."""
    same = (control_flow(res) == control_flow(ref)).all(axis=1)
    forks = np.nonzero(~same)[0]
    msg = (f'{tag}: identical {same.sum()}/{len(same)}, oracle-stable {stable.sum()}/{len(stable)}, identical among stable '
           f'{same[stable].sum()}/{stable.sum()}; converged device {np.mean(res["status"] <= 1):.3f} oracle {np.mean(ref["status"] <= 1):.3f}; '
           f'forks (scenario, stable?, device, oracle): ' +
           ', '.join(f'({b}, {bool(stable[b])}, {control_flow(res)[b].tolist()}, {control_flow(ref)[b].tolist()})' for b in forks))
    print(msg)
    assert stable.mean() >= min_stable_frac, msg
    assert same[stable].mean() >= min_stable_same, msg
    assert abs(np.mean(res['status'] <= 1) - np.mean(ref['status'] <= 1)) <= max_conv_gap + 1.0 / len(same), msg
    return same
