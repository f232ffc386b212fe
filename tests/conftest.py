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


def sympy_one_stage_game(kind, method):
    """A ONE-STAGE (N = 1) two-car race on the curve track whose vehicles are the ones of tests/golden/sympy_fd_<kind>.npz (the
    reference's config defaults), with the racing cost of chicane.py:223-277 (atan competition term) and the obstacle row.
    Returns (game, kat).  With N = 1 the game's Hessian is a closed formula in B = fBd and F = fFd (``sympy_one_stage_Q``)."""
    from dgsqp_amd.dynamics import (CasadiDecoupledMultiAgentDynamicsModel, CasadiDynamicBicycleCombined, CasadiKinematicBicycleCombined,
                                    DynamicBicycleConfig, KinematicBicycleConfig, MultiAgentModelConfig)
    from dgsqp_amd.game import CollisionAvoidance, InputRateLimits, RacingCost
    from dgsqp_amd.montecarlo import Game, _bounds, _track
    from dgsqp_amd.solver_types import DGSQPParams
    kat = np.load(ROOT / 'tests' / 'golden' / f'sympy_fd_{kind}.npz')
    M_sub = int(kat[f'{method}_M'])
    track = _track('curve', 45, 1.0)
    if kind == 'kin':
        mk = lambda: CasadiKinematicBicycleCombined(0, KinematicBicycleConfig(dt=0.1, discretization_method=method, M=M_sub, code_gen=False), track=track)
    else:
        mk = lambda: CasadiDynamicBicycleCombined(0, DynamicBicycleConfig(dt=0.1, discretization_method=method, M=M_sub, code_gen=False), track=track)
    joint = CasadiDecoupledMultiAgentDynamicsModel(0, [mk(), mk()], MultiAgentModelConfig(dt=0.1, discretization_method=method, M=M_sub, code_gen=False))
    params = DGSQPParams(dt=0.1, N=1, reg=1e-3, nonmono_ls=True, beta=0.01)
    g = Game(joint, [RacingCost(comp_weights=(10.0, 5.0)) for _ in range(2)], [InputRateLimits((10.0, 4.5), (-10.0, -4.5)) for _ in range(2)],
             CollisionAvoidance([0.2, 0.2]), _bounds(1.0, 2), params, track, 1.0, 0.4, name=f'sympy_{kind}_{method}')
    return g, kat


def sympy_one_stage_Q(kat, method, k1, k2, l_obs, nqa, s_idx, w_in=(1.0, 1.0), w_rate=(1.0, 1.0), w_prog=10.0, w_comp=5.0):
    """Q of the one-stage game at x0 = (point k1, point k2) of the KAT file, u = the points' inputs, multiplier ``l_obs`` on the
    obstacle row, from sympy's EXACT tensors:  rows a of  Duu J^a + B^T (D2 phi^a) B + sum_i (D phi^a)_i F_i  (DGSQP.py:678-727 with N = 1),
    phi^a(x_1) = -w_p s_a + w_c atan(s_b - s_a) + l_obs ((r_1 + r_2)^2 - |p_1 - p_2|^2)."""
    nq = 2 * nqa
    x1 = np.concatenate([kat[f'{method}_fd'][k1], kat[f'{method}_fd'][k2]])
    B = np.zeros((nq, 4))
    F = np.zeros((nq, 4, 4))
    for a, k in enumerate((k1, k2)):
        B[a * nqa:(a + 1) * nqa, 2 * a:2 * a + 2] = kat[f'{method}_jac'][k][:, nqa:]
        F[a * nqa:(a + 1) * nqa, 2 * a:2 * a + 2, 2 * a:2 * a + 2] = kat[f'{method}_hes'][k][:, nqa:, nqa:]
    Q = np.zeros((4, 4))
    for a in range(2):
        b = 1 - a
        sa, sb = a * nqa + s_idx, b * nqa + s_idx
        d = x1[sb] - x1[sa]
        g1, g2 = 1.0 / (1.0 + d * d), -2.0 * d / (1.0 + d * d) ** 2          # atan', atan''
        dphi = np.zeros(nq)
        d2phi = np.zeros((nq, nq))
        dphi[sa] += -w_prog - w_comp * g1
        dphi[sb] += w_comp * g1
        d2phi[sa, sa] += w_comp * g2; d2phi[sb, sb] += w_comp * g2
        d2phi[sa, sb] -= w_comp * g2; d2phi[sb, sa] -= w_comp * g2
        for c in range(2):                                                     # obstacle row: positions are states 0, 1 of each car
            i, j = c, nqa + c
            dphi[i] += l_obs * (-2.0) * (x1[i] - x1[j]); dphi[j] += l_obs * 2.0 * (x1[i] - x1[j])
            d2phi[i, i] += -2.0 * l_obs; d2phi[j, j] += -2.0 * l_obs
            d2phi[i, j] += 2.0 * l_obs; d2phi[j, i] += 2.0 * l_obs
        H = B.T @ d2phi @ B + np.einsum('i,ijk->jk', dphi, F)
        for c in range(2):
            H[2 * a + c, 2 * a + c] += w_in[c] + w_rate[c]
        Q[2 * a:2 * a + 2] = H[2 * a:2 * a + 2]
    return Q
