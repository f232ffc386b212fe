"""Development aid: first diverging SQP event between device and oracle (run on the GPU box).
usage: gpu_trace_diff.py <kbcurve0|kbchicane0|barc2|ablation_<nms|ls>_<stat_l1|stat>> [B] [N]      (environment: DGSQP_SEED)"""
import os, sys, pathlib
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
from oracle import oracle
import dgsqp_amd.solver as sv
from dgsqp_amd import montecarlo as mc
from dgsqp_amd.solver import DGSQP, build_problem, build_params

kind = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
N = int(sys.argv[3]) if len(sys.argv) > 3 else 25
g = mc.ablation_racing_game(N=N, nonmono_ls=kind.split('_')[1] == 'nms', merit_function=kind.split('_', 2)[2]) if kind.startswith('ablation') else mc.kinematic_racing_game('curve', N=N, M=3) if kind == 'agents3' else mc.merge_game(N=N) if kind == 'merge' else (mc.barc_racing_game(N=N, M=2) if kind == 'barc2' else mc.kinematic_racing_game('curve' if kind == 'kbcurve0' else 'chicane', N=N, reg=0.0))
M = g.joint_model.n_a


def tight(par):
    par.lsqr_atol = par.lsqr_btol = 1e-13
    par.lsqr_iter_mult = 20
    if os.environ.get('DGSQP_NO_WARM'):
        par.qp_warm_start = 0
    return par


P, par = build_problem(*g.solver_args()), tight(build_params(g.params))
s = DGSQP(*g.solver_args(), print_method=None, lsqr_tol=1e-13, qp_warm_start=not os.environ.get('DGSQP_NO_WARM'))
x0, u_tm = mc.sample_scenarios(g, B, seed=int(os.environ.get('DGSQP_SEED', 0 if kind == 'barc2' else 1)))
u = np.ascontiguousarray(u_tm.reshape(B, N, M, 2).transpose(0, 2, 1, 3).reshape(B, -1))
s.set_trace(8000)
res = s.solve_batch(x0, u_tm)
traces = s.fetch_trace(B)
s.set_trace(0)
names = {40: 'qp solves of the iteration', 1: 'stat', 2: 'p_feas', 3: 'comp', 10: '|du|^2', 11: 'mu', 12: 'phi', 13: 'dphi', 20: 'wd phi1', 21: 'wd phi_n', 22: 'wd phi_n2', 30: 'ls alpha', 31: 'ls phi'}
for b in range(B):
    to = oracle.solve_trace(P, par, x0[b], u[b])
    tg = traces[b]
    k = 0
    while k < min(len(to), len(tg)) and to[k, 0] == tg[k, 0] and abs(tg[k, 1] - to[k, 1]) <= 1e-3 * max(abs(to[k, 1]), 1e-6):
        k += 1
    it = int((to[:k, 0] == 1).sum())
    print(f'scn {b}: status dev {res["status"][b]} iters {res["num_iters"][b]} | events dev {len(tg)} oracle {len(to)} | agree for {k} events (into SQP iteration {it})')
    for j in range(max(0, k - 3), min(k + 4, len(to), len(tg))):
        print(f'     ev {j}: oracle {names.get(int(to[j, 0]), int(to[j, 0]))} {to[j, 1]:.6e} | dev {names.get(int(tg[j, 0]), int(tg[j, 0]))} {tg[j, 1]:.6e}')


def summarise(t):
    rows, cur = [], None
    for code, v in t:
        code = int(code)
        if code == 1:
            cur = {'stat': v, 'ls': 0, 'wd': 0}
            rows.append(cur)
        elif cur is not None:
            if code in names and code < 20:
                cur[names[code]] = v
            elif code in (20, 21, 22):
                cur['wd'] += 1
            elif code == 30:
                cur['ls'] += 1
    return rows


if os.environ.get('DGSQP_TRACE_SCN'):
    b = int(os.environ['DGSQP_TRACE_SCN'])
    to, tg = summarise(oracle.solve_trace(P, par, x0[b], u[b])), summarise(traces[b])
    keys = ('stat', 'p_feas', 'comp', '|du|^2', 'mu', 'phi', 'dphi', 'wd', 'ls')
    for i in range(max(len(to), len(tg))):
        for tag, rr in (('O', to), ('D', tg)):
            if i < len(rr):
                print(f'{tag} it {i:2d} ' + ' '.join(f'{k} {rr[i].get(k, float("nan")):.3e}' for k in keys))
