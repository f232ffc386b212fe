"""Build container: why BASELINE configs[1] uses the reference's own dynamic-bicycle game (exact_dynamic_game_dynamic.py,
cost_setting 0) instead of round 1's synthetic variant (costs / rate rows of curve.py on the Pacejka vehicle).
(1) Round-1 definition: every QP the oracle reports as infeasible is examined with an LP (min max-violation t subject to
    G du - t <= -g, HiGHS) and with the restated OSQP; |G|max, |Q|max are the largest entries of the constraint Jacobian and of
    the game Hessian at that iterate.
(2) Status fractions of the first 32 sampled scenarios under the candidate definitions (C++ oracle).
usage: python tools/dyn_game_study.py > profiles/r02_dyn_curve_divergence.txt"""
import pathlib, sys, time
import numpy as np
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
from scipy.optimize import linprog
from conftest import agent_major
import dgsqp_amd.montecarlo as mc
from dgsqp_amd.game import InputRateLimits, RacingCost
from dgsqp_amd.solver import build_problem, build_params
from oracle import oracle, osqp_restate, pyref


class Probe(pyref.PyRef):
    def solve_qp(self, Q, q, G, g):
        du, lh = super().solve_qp(Q, q, G, g)
        if np.isnan(du).any() and np.isfinite(G).all():
            n, m = G.shape[1], G.shape[0]
            res = linprog(np.r_[np.zeros(n), 1.0], A_ub=np.hstack([G, -np.ones((m, 1))]), b_ub=-g, bounds=[(None, None)] * (n + 1), method='highs-ds')
            x, y, info = osqp_restate.conic(self.nearest_pd(Q) + self.par.reg * np.eye(n), q, G, -g)
            print(f'   exact QP: infeasible | |G|max {np.abs(G).max():.2e} |Q|max {np.abs(Q).max():.2e} max g {g.max():.2e} | LP: status {res.status} '
                  f'min max-violation {res.fun} | restated OSQP: status {info["status"]} after {info["iters"]} iterations')
        return du, lh


print('== (1) round-1 definition (game_def="curve"), scenarios of the first 16 whose solve ends in an infeasible QP')
g = mc.dynamic_racing_game(N=25, rk4_substeps=10, game_def='curve')
P, par = build_problem(*g.solver_args()), build_params(g.params)
x0, uws = mc.sample_scenarios(g, 16, seed=1)
u = agent_major(uws)
o = oracle.solve_batch(P, par, x0, u, nthreads=8)
for b in np.nonzero(o['status'] == 4)[0]:
    s = Probe(P, par, qp='gi').solve(x0[b], u[b])
    print(f'scenario {b}: numpy loop with the exact QP ends with {s["msg"]} after {s["num_iters"]} iterations / {s["qp_solves"]} QPs')

print('\n== (2) status fractions, first 32 scenarios (C++ oracle)')


def run(tag, g):
    P, par = build_problem(*g.solver_args()), build_params(g.params)
    x0, uws = mc.sample_scenarios(g, 32, seed=1)
    t = time.time()
    o = oracle.solve_batch(P, par, x0, agent_major(uws), nthreads=8)
    st = o['status']
    print(f'{tag:70s} converged {np.mean(st <= 1):.3f} max_it {np.mean(st == 2):.3f} qp_fail {np.mean(st == 4):.3f} mean iters (conv) '
          f'{o["num_iters"][st <= 1].mean():5.2f} mean QPs {o["qp_solves"].mean():5.1f}  ({time.time() - t:.0f} s)')


run('round 1: curve.py costs (atan, weights 10 / 5), rate rows, radii 0.2', mc.dynamic_racing_game(N=25, game_def='curve'))
g = mc.dynamic_racing_game(N=25)
g.agent_constraints = [InputRateLimits((10.0, 4.5), (-10.0, -4.5)) for _ in range(2)]
run('exact_dynamic_game_dynamic.py cost_setting 0 + curve.py rate rows', g)
run('exact_dynamic_game_dynamic.py cost_setting 0 (linear 1 / 5, no agent rows) = configs[1] now', mc.dynamic_racing_game(N=25))
g = mc.dynamic_racing_game(N=25)
g.costs = [RacingCost(input_weight=(.1, .1), input_rate_weight=(.1, .1), comp_weights=(0.0, 1.0), comp_type='linear') for _ in range(2)]
run('exact_dynamic_game_dynamic.py cost_setting 1 (linear 0 / 1, weights 0.1)', g)
