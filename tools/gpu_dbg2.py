import sys, pathlib
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent / 'tests'))
from conftest import agent_major, tight_lsqr
from dgsqp_amd.montecarlo import dynamic_racing_game, sample_scenarios
from dgsqp_amd.solver import DGSQP, build_problem, build_params
from oracle import oracle
g = dynamic_racing_game(N=8, rk4_substeps=2)
P, par = build_problem(*g.solver_args()), build_params(g.params)
s = DGSQP(*g.solver_args(), print_method=None)
B = 6
x0, u_tm = sample_scenarios(g, B, seed=22)
u = agent_major(u_tm)
l = np.array([oracle.dual_init(P, tight_lsqr(par), x0[b], u[b]) for b in range(B)])
qp = s.qp_batch(x0, u, l)
ev = s.evaluate_batch(x0, u, l)
for b in range(B):
    o = oracle.evaluate(P, x0[b], u[b], l[b], 1)
    S = 0.5 * (o['Q'] + o['Q'].T)
    w, U = np.linalg.eigh(S)
    Sd = 0.5 * (ev['Q'][b] + ev['Q'][b].T)
    wd = np.linalg.eigvalsh(Sd)
    Qo = oracle.nearest_pd(o['Q'], par.reg)
    w2 = w.copy(); w2[w2 < 0] = 1e-10
    Qn = U @ np.diag(w2) @ U.T + par.reg * np.eye(len(w))
    print(b, 'Qraw diff', np.abs(ev['Q'][b] - o['Q']).max(), 'eig smallest abs', np.sort(np.abs(w))[:3], 'neg', (w < 0).sum(), 'eig diff dev/orc', np.abs(w - wd).max())
    print('   dev vs oracle', np.abs(qp['Qpd'][b] - Qo).max(), 'dev vs numpy', np.abs(qp['Qpd'][b] - Qn).max(), 'oracle vs numpy', np.abs(Qo - Qn).max(), 'max', np.abs(Qn).max())
