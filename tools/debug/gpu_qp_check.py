"""Development aid: device _nearestPD + QP against the oracle on one game at a given reg (run on the GPU box).
usage: gpu_qp_check.py <barc2|barc3|kbcurve|kbchicane> <reg> [B]"""
import sys, pathlib
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
from dgsqp_amd import montecarlo as mc
from dgsqp_amd.solver import DGSQP, build_problem, build_params
from oracle import oracle
which, reg = sys.argv[1], float(sys.argv[2])
B = int(sys.argv[3]) if len(sys.argv) > 3 else 12
if which == 'barc2': g = mc.barc_racing_game(N=15, M=2, reg=reg)
elif which == 'barc3': g = mc.barc_racing_game(N=15, M=3, reg=reg)
elif which == 'agents3': g = mc.kinematic_racing_game('curve', N=25, M=3, reg=reg)
else: g = mc.kinematic_racing_game('curve' if which == 'kbcurve' else 'chicane', N=25, reg=reg)
M, N = g.joint_model.n_a, g.params.N
P, par = build_problem(*g.solver_args()), build_params(g.params)
par.lsqr_atol = par.lsqr_btol = 1e-13; par.lsqr_iter_mult = 20
s = DGSQP(*g.solver_args(), print_method=None)
x0, u_tm = mc.sample_scenarios(g, B, seed=0)
u = np.ascontiguousarray(u_tm.reshape(B, N, M, 2).transpose(0, 2, 1, 3).reshape(B, -1))
l = np.array([oracle.dual_init(P, par, x0[b], u[b]) for b in range(B)])
qp = s.qp_batch(x0, u, l)
for b in range(B):
    o = oracle.evaluate(P, x0[b], u[b], l[b], 1)
    Bs = 0.5 * (o['Q'] + o['Q'].T)
    w = np.linalg.eigvalsh(Bs)
    Qpd = oracle.nearest_pd(o['Q'], reg, par.eig_floor)
    du, lam, flag = oracle.qp(Qpd, o['q'], o['G'], o['g'])
    Qd = qp['Qpd'][b]
    wd = np.linalg.eigvalsh(0.5 * (Qd + Qd.T))
    wo = np.linalg.eigvalsh(Qpd)
    d, lh = qp['du'][b], qp['lhat'][b]
    obj = lambda z: 0.5 * z @ Qpd @ z + o['q'] @ z
    kkt = lambda z, m: (np.abs(Qpd @ z + o['q'] + o['G'].T @ m).max(), (o['G'] @ z + o['g']).max(), np.abs(m * (o['G'] @ z + o['g'])).max())
    print(f'scn {b}: kneg {int((w < 0).sum())} |Q| {np.abs(o["Q"]).max():.1e} Qpd err {np.abs(Qd - Qpd).max():.1e} min eig dev {wd[0]:.2e} oracle {wo[0]:.2e} '
          f'| flag dev {qp["flag"][b]} oracle {flag} |du| dev {np.abs(d).max():.2e} oracle {np.abs(du).max():.2e} diff {np.abs(d - du).max():.1e} '
          f'nact dev {int((lh > 0).sum())} oracle {int((lam > 0).sum())} same set {np.array_equal(lh > 0, lam > 0)}')
    print('      obj dev %.9e oracle %.9e | kkt dev (stat %.1e feas %.1e comp %.1e) oracle (stat %.1e feas %.1e comp %.1e)' % ((obj(d), obj(du)) + kkt(d, lh) + kkt(du, lam)))
