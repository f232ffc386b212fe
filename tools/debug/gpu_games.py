"""Device vs oracle on a game preset that is not one of the bench workloads (run on the GPU box).
usage: gpu_games.py <barc2|barc3|merge> [B] [N]"""
import os, sys, time, pathlib
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
from oracle import oracle
import dgsqp_amd.solver as sv
from dgsqp_amd import montecarlo as mc
from dgsqp_amd.solver import DGSQP, build_problem, build_params

kind = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
if kind == 'barc2':
    g = mc.barc_racing_game(N=int(sys.argv[3]) if len(sys.argv) > 3 else 15, M=2)
elif kind == 'barc3':
    g = mc.barc_racing_game(N=int(sys.argv[3]) if len(sys.argv) > 3 else 15, M=3)
elif kind == 'kbcurve':       # reg = 1e-3
    g = mc.kinematic_racing_game('curve', N=int(sys.argv[3]) if len(sys.argv) > 3 else 25)
elif kind == 'kbcurve_m':      # curve game with DGSQP_M agents (reg = 1e-3)
    g = mc.kinematic_racing_game('curve', N=int(sys.argv[3]) if len(sys.argv) > 3 else 25, M=int(os.environ.get('DGSQP_M', '2')))
elif kind == 'agents3':     # scripts/DGSQP_monte_carlo_agents.py at exp_M = [3], exp_N = [25] (:101-102), reg = 1e-3 (:146)
    g = mc.kinematic_racing_game('curve', N=int(sys.argv[3]) if len(sys.argv) > 3 else 25, M=3)
elif kind in ('kbcurve0', 'kbchicane0'):
    g = mc.kinematic_racing_game('curve' if kind == 'kbcurve0' else 'chicane', N=int(sys.argv[3]) if len(sys.argv) > 3 else 25, reg=0.0)
else:
    g = mc.merge_game(N=int(sys.argv[3]) if len(sys.argv) > 3 else 20, M=int(os.environ.get('DGSQP_M', '3')))
if os.environ.get('DGSQP_BFGS'):
    g.params.hessian_approximation = 'bfgs'
M, N = g.joint_model.n_a, g.params.N


def tight(par):
    par.lsqr_atol = par.lsqr_btol = 1e-13
    par.lsqr_iter_mult = 20
    if os.environ.get('DGSQP_NO_WARM'):
        par.qp_warm_start = 0
    if os.environ.get('DGSQP_EIG_FLOOR'):       # e.g. 1e-10: the literal _nearestPD formula on device and oracle
        par.eig_floor = float(os.environ['DGSQP_EIG_FLOOR'])
    return par


P, par = build_problem(*g.solver_args()), tight(build_params(g.params))
orig = sv.build_params
sv.build_params = lambda p: tight(orig(p))
s = DGSQP(*g.solver_args(), print_method=None)
sv.build_params = orig
x0, u_tm = mc.sample_scenarios(g, B, seed=0 if kind.startswith('barc') else 1)
if kind == 'merge':
    up_noise = 0.05
else:
    up_noise = 0.01
nua = 2
u = np.ascontiguousarray(u_tm.reshape(B, N, M, nua).transpose(0, 2, 1, 3).reshape(B, -1))
rng = np.random.default_rng(1)
up = u + up_noise * rng.standard_normal(u.shape)
l = np.maximum(0, rng.standard_normal((B, s.n_c_total)))
ev = s.evaluate_batch(x0[:8], up[:8], l[:8])
rel = lambda a, b: float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / (1e-300 + np.max(np.abs(b))))
for key in ('x', 'q', 'g', 'G', 'Q'):
    print('evaluate', key, max(rel(ev[key][b], oracle.evaluate(P, x0[b], up[b], l[b], 1)[key]) for b in range(8)))
t = time.time(); res = s.solve_batch(x0, u_tm); tg = time.time() - t
t = time.time(); ref = oracle.solve_batch(P, par, x0, u, nthreads=16); to = time.time() - t
same = (res['status'] == ref['status']) & (res['num_iters'] == ref['num_iters']) & (res['qp_solves'] == ref['qp_solves'])
print(g.name, 'B', B, 'gpu %.3fs oracle %.1fs' % (tg, to), 'identical', same.mean(),
      'conv dev/oracle', (res['status'] <= 1).mean(), (ref['status'] <= 1).mean())
print('status dev', np.bincount(res['status'], minlength=5), 'oracle', np.bincount(ref['status'], minlength=5))
ok = same & (ref['status'] <= 1)
if ok.any():
    du = [rel(res['u'][b], ref['u'][b]) for b in np.where(ok)[0]]
    dl = [rel(res['l'][b], ref['l'][b]) for b in np.where(ok)[0]]
    print('iterate diff on identical converged: u median %.2e max %.2e | l median %.2e max %.2e' % (np.median(du), max(du), np.median(dl), max(dl)))
for b in np.where(~same)[0][:10]:
    print('  differ', b, 'dev', res['status'][b], res['num_iters'][b], res['qp_solves'][b], 'oracle', ref['status'][b], ref['num_iters'][b], ref['qp_solves'][b])
