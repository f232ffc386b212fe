"""Development aid (GPU box): which scenarios of a golden fixture take a different path on the device than in the
oracle, and the first SQP event at which the two logs differ.
usage: gpu_forks.py <fixture name, e.g. dyn_curve_N25> [rel tol of the event comparison, default 1e-6]"""
import os, sys, pathlib
import numpy as np
ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
from conftest import agent_major, tight_lsqr
from oracle import oracle
import dgsqp_amd.solver as sv
from dgsqp_amd import montecarlo as mc
from dgsqp_amd.solver import DGSQP, build_problem, build_params

name = sys.argv[1]
tol = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-6
games = {'dyn_curve_N25': lambda: mc.dynamic_racing_game(N=25, rk4_substeps=10),
         'dyn_curve_N15': lambda: mc.dynamic_racing_game(N=15, rk4_substeps=4, game_def='curve'),
         'kb_chicane_N15': lambda: mc.kinematic_racing_game('chicane', N=15), 'kb_curve_N10': lambda: mc.kinematic_racing_game('curve', N=10),
         'kb_barc2_N15': lambda: mc.barc_racing_game(N=15, M=2), 'merge_N8': lambda: mc.merge_game(N=8),
         'kb_curve_reg0_N20': lambda: mc.kinematic_racing_game('curve', N=20, reg=0.0), 'merge_N20': lambda: mc.merge_game()}
g = games[name]()
P, par = build_problem(*g.solver_args()), tight_lsqr(build_params(g.params))
gpath = ROOT / 'tests' / 'golden' / f'{name}.npz'
cache = os.environ.get('FORKS_CACHE')
if gpath.exists():
    gold = np.load(gpath)
elif cache and os.path.exists(cache):
    gold = np.load(cache)
else:        # no fixture: the scenarios of the GPU test (seed 1), the oracle run here
    from conftest import stable_mask
    x0, u_tm = mc.sample_scenarios(g, int(os.environ.get('FORKS_B', '48')), seed=1)
    ua = agent_major(u_tm) if u_tm.shape[2] == 4 else np.concatenate([u_tm[:, :, 2 * a:2 * a + 2].reshape(len(x0), -1) for a in range(u_tm.shape[2] // 2)], axis=1)
    ref = oracle.solve_batch(P, par, x0, ua, nthreads=8)
    gold = dict(x0=x0, u_ws=u_tm, status=ref['status'], num_iters=ref['num_iters'], qp_solves=ref['qp_solves'], stable=stable_mask(oracle, P, par, x0, ua, ref))
    if cache:
        np.savez(cache, **gold)
x0, u_tm = gold['x0'], gold['u_ws']
B = len(x0)
s = DGSQP(*g.solver_args(), print_method=None, lsqr_tol=1e-13)
s.set_trace(20000)
res = s.solve_batch(x0, u_tm)
traces = s.fetch_trace(B)
s.set_trace(0)
same = (res['status'] == gold['status']) & (res['num_iters'] == gold['num_iters']) & (res['qp_solves'] == gold['qp_solves'])
stable = gold['stable'] if 'stable' in gold else np.ones(B, bool)
print(f'{name}: identical {same.sum()}/{B}; oracle reproduces itself under 1e-13 input perturbations on {stable.sum()}/{B}; identical among those '
      f'{same[stable].sum()}/{stable.sum()}; converged device {np.mean(res["status"] <= 1):.3f} oracle {np.mean(gold["status"] <= 1):.3f}')
print('fork table: scenario | oracle-stable | device (status, iters, QPs) | oracle | first differing event | cause')
names = {40: 'qp solves of the iteration', 1: 'stat', 2: 'p_feas', 3: 'comp', 10: '|du|^2', 11: 'mu', 12: 'phi', 13: 'dphi', 20: 'wd phi1', 21: 'wd phi_n', 22: 'wd phi_n2', 30: 'ls alpha', 31: 'ls phi'}
u_am = agent_major(u_tm) if u_tm.shape[2] == 4 else np.concatenate([u_tm[:, :, 2 * a:2 * a + 2].reshape(B, -1) for a in range(u_tm.shape[2] // 2)], axis=1)
for b in (np.nonzero(~same)[0] if not os.environ.get('FORKS_QUIET') else []):
    to = oracle.solve_trace(P, par, x0[b], u_am[b], max_pairs=60000)
    tg = traces[b]
    k = 0
    def agree(k):          # p_feas / comp values at rounding level are noise on both sides, not a difference
        if to[k, 0] != tg[k, 0]:
            return False
        if int(to[k, 0]) in (2, 3) and max(abs(to[k, 1]), abs(tg[k, 1])) < 1e-10:
            return True
        return abs(tg[k, 1] - to[k, 1]) <= tol * max(abs(to[k, 1]), 1e-9)
    while k < min(len(to), len(tg)) and agree(k):
        k += 1
    it = int((to[:k, 0] == 1).sum()) - 1
    # classification of the first differing event
    ev = names.get(int(to[k, 0]), '?') if k < len(to) else 'end'
    pf = [to[j, 1] for j in range(k, -1, -1) if j < len(to) and int(to[j, 0]) == 2][:1]
    pfd = [tg[j, 1] for j in range(min(k, len(tg) - 1), -1, -1) if int(tg[j, 0]) == 2][:1]
    if ev == 'mu' and pf and pfd and max(pf[0], pfd[0]) < 1e-12:
        cause = 'mu switch: _get_mu tests sum(g - s) > 0 (threshold 0, DGSQP.py:559-585) on a rounding-level sum (p_feas %.1e / %.1e)' % (pf[0], pfd[0])
    else:
        st = [to[j, 1] for j in range(k, -1, -1) if j < len(to) and int(to[j, 0]) == 1][:1]
        cause = 'rounding amplified over the iterations (values agree to %.0e at the fork; stat %.2g, p_feas %.2g at that iterate)' % (
            abs(tg[k, 1] - to[k, 1]) / max(abs(to[k, 1]), 1e-300) if k < min(len(to), len(tg)) else float('nan'), st[0] if st else float('nan'), pf[0] if pf else float('nan'))
    print(f'| {b} | {bool(stable[b])} | {(int(res["status"][b]), int(res["num_iters"][b]), int(res["qp_solves"][b]))} | {(int(gold["status"][b]), int(gold["num_iters"][b]), int(gold["qp_solves"][b]))} | event {k} = {ev} in SQP iteration {it} | {cause} |')
    print(f'scn {b}: device (status {res["status"][b]}, iters {res["num_iters"][b]}, qps {res["qp_solves"][b]}) oracle ({gold["status"][b]}, {gold["num_iters"][b]}, {gold["qp_solves"][b]}) | events dev {len(tg)} oracle {len(to)} | agree for {k} events (SQP iteration {it})')
    for j in range(max(0, k - 4), min(k + 3, max(len(to), len(tg)))):
        eo = f'{names.get(int(to[j, 0]), int(to[j, 0]))} {to[j, 1]:.10e}' if j < len(to) else '-'
        ed = f'{names.get(int(tg[j, 0]), int(tg[j, 0]))} {tg[j, 1]:.10e}' if j < len(tg) else '-'
        print(f'     ev {j}: oracle {eo} | dev {ed}')
