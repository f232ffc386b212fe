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
    from dgsqp_amd.montecarlo import barc_racing_game, kinematic_racing_game, dynamic_racing_game, merge_game
    from dgsqp_amd.solver import build_problem, build_params
    out = {}
    for name, g in (('kb_chicane_N15', kinematic_racing_game('chicane', N=15)),
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
