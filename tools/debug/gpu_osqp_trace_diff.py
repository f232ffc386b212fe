"""Development aid (GPU box): first diverging SQP event between device and C++ oracle with qp_method='osqp', both with a converged
LSQR dual start.  usage: gpu_osqp_trace_diff.py <game of tools/ref_stats.py> [B] [first scenario]"""
import os, sys, pathlib
import numpy as np
ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tools')); sys.path.insert(0, str(ROOT / 'tests'))
from oracle import oracle
from dgsqp_amd import montecarlo as mc
from dgsqp_amd.solver import DGSQP, build_problem, build_params
from ref_stats import GAMES
name = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
b0 = int(sys.argv[3]) if len(sys.argv) > 3 else 0
qm = os.environ.get('DGSQP_QP_METHOD', 'osqp')
g = GAMES[name][0]()
P, par = build_problem(*g.solver_args()), build_params(g.params, qp_method=qm)
par.lsqr_atol = par.lsqr_btol = 1e-13
par.lsqr_iter_mult = 20
s = DGSQP(*g.solver_args(), print_method=None, lsqr_tol=1e-13, qp_method=qm)
x0, u_tm = mc.sample_scenarios(g, b0 + B, seed=GAMES[name][1])
x0, u_tm = x0[b0:], u_tm[b0:]
u = s._to_agent_major(u_tm)
s.set_trace(20000)
res = s.solve_batch(x0, u_tm)
traces = s.fetch_trace(B)
s.set_trace(0)
names = {40: 'qp solves of the iteration', 1: 'stat', 2: 'p_feas', 3: 'comp', 10: '|du|^2', 11: 'mu', 12: 'phi', 13: 'dphi', 20: 'wd phi1', 21: 'wd phi_n', 22: 'wd phi_n2', 30: 'ls alpha', 31: 'ls phi'}
for b in range(B):
    to = oracle.solve_trace(P, par, x0[b], u[b], max_pairs=20000)
    tg = traces[b]
    k = 0
    while k < min(len(to), len(tg)) and to[k, 0] == tg[k, 0] and abs(tg[k, 1] - to[k, 1]) <= 1e-6 * max(abs(to[k, 1]), 1e-9):
        k += 1
    it = int((to[:k, 0] == 1).sum())
    print(f'scn {b0 + b}: status dev {res["status"][b]} iters {res["num_iters"][b]} qps {res["qp_solves"][b]} | events dev {len(tg)} oracle {len(to)} | agree (1e-6) for {k} events (into SQP iteration {it})')
    for j in range(max(0, k - 4), min(k + 4, len(to), len(tg))):
        print(f'     ev {j}: oracle {names.get(int(to[j, 0]), int(to[j, 0]))} {to[j, 1]:.10e} | dev {names.get(int(tg[j, 0]), int(tg[j, 0]))} {tg[j, 1]:.10e}')
