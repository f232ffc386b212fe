"""Which arithmetic is right on the literal reg = 0 QPs (condition ~1e12)?  

Stage `gpu`  (on the GPU box): solve a batch of the reference's own reg = 0 experiment on the device with the iterate log on, then
             run the device's _nearestPD + QP (test hook dgsqp_qp_batch) at the logged iterates (u_i, l_i) -- the first QP of SQP
             iteration i + 1 -- and save inputs and device answers (du, lhat, projected Hessian M_dev) to an .npz.
Stage `cpu`  (anywhere): the oracle's answer on the same inputs (its own projection M_orc, its own dual active-set QP), and for every
             QP the EXACT minimiser of  min 1/2 x'Mx + q'x  s.t. Gx <= -g  for M = M_dev and for M = M_orc, by a 60-digit KKT solve
             (mpmath) on the active set, accepted only when primal and dual feasibility hold in that arithmetic (strictly convex QP:
             the KKT point is THE minimiser).  Reported per QP:
               dev_vs_orc   |du_dev - du_orc| / |du|                    what the parity tests see
               dev_err      |du_dev - x*(M_dev)| / |x*|                 error of the device's QP solver on ITS matrix
               orc_err      |du_orc - x*(M_orc)| / |x*|                 error of the oracle's QP solver on ITS matrix
               raw_err      the same for the oracle's dual active-set iterate WITHOUT the KKT polish (what rounds 1-2 compared against)
               cross        |x*(M_dev) - x*(M_orc)| / |x*|              what the two fp64 projections alone are responsible for
Usage: python tools/reg0_qp_study.py gpu out.npz [game] [B] [iters]   |   python tools/reg0_qp_study.py cpu out.npz [n_exact]
"""
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'tests'))


def make_game(name):
    from dgsqp_amd.montecarlo import barc_racing_game, kinematic_racing_game, merge_game
    if name == 'kb_curve_reg0_N20':
        return kinematic_racing_game('curve', N=20, reg=0.0)
    if name == 'kb_curve_reg0_N25':
        return kinematic_racing_game('curve', N=25, reg=0.0)
    if name == 'merge_N20':
        return merge_game(N=20)
    if name == 'kb_barc2_N15':
        return barc_racing_game(N=15, M=2)
    raise SystemExit(f'unknown game {name}')


def stage_gpu(out, name='kb_curve_reg0_N20', B=48, iters=10):
    from conftest import agent_major
    from dgsqp_amd.montecarlo import sample_scenarios
    from dgsqp_amd.solver import DGSQP
    g = make_game(name)
    s = DGSQP(*g.solver_args(), print_method=None, lsqr_tol=1e-13)
    x0, u_tm = sample_scenarios(g, B, seed=1)
    s.set_iterate_log(iters + 1)
    res = s.solve_batch(x0, u_tm)
    logs = s.fetch_iterate_log(B)
    s.set_iterate_log(0)
    X0, U, L, SC, IT = [], [], [], [], []
    for b in range(B):
        u_log, l_log = logs[b]
        for i in range(min(len(u_log), iters)):
            X0.append(x0[b]); U.append(u_log[i]); L.append(l_log[i]); SC.append(b); IT.append(i)
    X0, U, L = np.array(X0), np.array(U), np.array(L)
    qp = s.qp_batch(X0, U, L)
    np.savez_compressed(out, game=name, x0=X0, u=U, l=L, scenario=np.array(SC), iteration=np.array(IT), du=qp['du'], lhat=qp['lhat'],
                        Qpd=qp['Qpd'], flag=qp['flag'], status=res['status'], num_iters=res['num_iters'])
    print(f'{name}: {len(U)} QPs at the iterates of {B} device solves -> {out}; flags {np.bincount(qp["flag"])}')


def exact_qp(M, q, G, g, active_sets, dps=60):
    """KKT point on one of the candidate active sets in `dps`-digit arithmetic; None when none of them is primal + dual feasible."""
    import mpmath as mp
    mp.mp.dps = dps
    n = len(q)
    Mm = mp.matrix(M.tolist())
    qm = mp.matrix(q.tolist())
    for A in active_sets:
        A = [int(a) for a in A]
        m = len(A)
        K = mp.zeros(n + m, n + m)
        K[:n, :n] = Mm
        for j, r in enumerate(A):
            for i in range(n):
                if G[r, i] != 0.0:
                    K[n + j, i] = K[i, n + j] = mp.mpf(float(G[r, i]))
        rhs = mp.zeros(n + m, 1)
        for i in range(n):
            rhs[i] = -qm[i]
        for j, r in enumerate(A):
            rhs[n + j] = -mp.mpf(float(g[r]))
        try:
            sol = mp.lu_solve(K, rhs)
        except ZeroDivisionError:
            continue
        x = np.array([float(sol[i]) for i in range(n)])
        lam = np.array([float(sol[n + j]) for j in range(m)])
        slack = G @ x + g           # (fp64 product of the rounded exact point: good to 1e-16 |G||x|)
        if lam.min(initial=0.0) >= -1e-9 * max(1.0, np.abs(lam).max(initial=0.0)) and slack.max() <= 1e-9:
            return x, dict(zip(A, lam))
    return None


def stage_cpu(path, n_exact=40):
    from oracle import oracle
    from dgsqp_amd.solver import build_params, build_problem
    d = np.load(path)
    name = str(d['game'])
    g = make_game(name)
    P, par = build_problem(*g.solver_args()), build_params(g.params)
    nq = len(d['u'])
    rows = []
    for k in range(nq):
        if d['flag'][k] != 0:
            continue
        o = oracle.evaluate(P, d['x0'][k], d['u'][k], d['l'][k], 1)
        M_orc = oracle.nearest_pd(o['Q'], par.reg, par.eig_floor)
        du_o, lam_o, flag = oracle.qp(M_orc, o['q'], o['G'], o['g'])
        if flag != 0:
            continue
        nrm = max(np.linalg.norm(du_o), 1e-300)
        oracle.lib().oracle_set_qp_polish(0)             # the dual active-set iterate as it stands, without the KKT polish (rounds 1-2)
        du_raw = oracle.qp(M_orc, o['q'], o['G'], o['g'])[0]
        oracle.lib().oracle_set_qp_polish(1)
        rows.append(dict(k=k, o=o, M_orc=M_orc, du_o=du_o, lam_o=lam_o, du_raw=du_raw, dev_vs_orc=np.linalg.norm(d['du'][k] - du_o) / nrm,
                         dM=np.abs(d['Qpd'][k] - M_orc).max(), same_set=bool(np.array_equal(d['lhat'][k] > 0, lam_o > 0))))
    dv = np.array([r['dev_vs_orc'] for r in rows])
    print(f'{name}: {len(rows)} feasible QPs; |du_dev - du_orc|/|du|: median {np.median(dv):.2e}, 90 % {np.quantile(dv, 0.9):.2e}, max {dv.max():.2e}; '
          f'same active set on {np.mean([r["same_set"] for r in rows]):.3f}; max |M_dev - M_orc| {max(r["dM"] for r in rows):.2e}')
    order = np.argsort(-dv)
    pick = list(order[:n_exact // 2]) + list(order[len(order) // 2:len(order) // 2 + n_exact - n_exact // 2])     # the worst and a band around the median
    raw = np.array([np.linalg.norm(r['du_raw'] - r['du_o']) / max(np.linalg.norm(r['du_o']), 1e-300) for r in rows])
    print(f'   unpolished oracle iterate vs its polished point, |.|/|du|: median {np.median(raw):.2e}, 90 % {np.quantile(raw, 0.9):.2e}, max {raw.max():.2e} '
          f'(share above 1e-5: {np.mean(raw > 1e-5):.3f})')
    print('   QP  scen iter | dev_vs_orc | dev_err    orc_err    raw_err    cross      | lam_min(M) cond(M)  | sets: dev = orc?  exact(M_dev) = exact(M_orc)?')
    summ = []
    for idx in pick:
        r = rows[idx]
        k, o = r['k'], r['o']
        sets = [np.nonzero(d['lhat'][k] > 0)[0], np.nonzero(r['lam_o'] > 0)[0]]
        ed = exact_qp(d['Qpd'][k], o['q'], o['G'], o['g'], sets)
        eo = exact_qp(r['M_orc'], o['q'], o['G'], o['g'], sets[::-1])
        w = np.linalg.eigvalsh(0.5 * (r['M_orc'] + r['M_orc'].T))
        if ed is None or eo is None:
            print(f'{k:5d} {d["scenario"][k]:5d} {d["iteration"][k]:4d} | {r["dev_vs_orc"]:.2e}   | no candidate active set is optimal in exact arithmetic (dev {ed is not None}, orc {eo is not None})')
            continue
        xd, xo = ed[0], eo[0]
        nx = max(np.linalg.norm(xo), 1e-300)
        e_dev, e_orc, cross = np.linalg.norm(d['du'][k] - xd) / nx, np.linalg.norm(r['du_o'] - xo) / nx, np.linalg.norm(xd - xo) / nx
        e_raw = np.linalg.norm(r['du_raw'] - xo) / nx
        summ.append((r['dev_vs_orc'], e_dev, e_orc, cross))
        print(f'{k:5d} {d["scenario"][k]:5d} {d["iteration"][k]:4d} | {r["dev_vs_orc"]:.2e}   | {e_dev:.2e}   {e_orc:.2e}   {e_raw:.2e}   {cross:.2e}   | {w[0]:.2e}   {w[-1] / w[0]:.1e} | '
              f'{r["same_set"]}  {set(ed[1]) == set(eo[1])}')
    if summ:
        a = np.array(summ)
        print(f'exact-arithmetic sample of {len(a)}: median dev_err {np.median(a[:, 1]):.2e} orc_err {np.median(a[:, 2]):.2e} cross {np.median(a[:, 3]):.2e}; '
              f'max dev_err {a[:, 1].max():.2e} orc_err {a[:, 2].max():.2e} cross {a[:, 3].max():.2e}; '
              f'share of dev_vs_orc explained by the projections alone (cross >= 0.5 dev_vs_orc): {np.mean(a[:, 3] >= 0.5 * a[:, 0]):.2f}')


if __name__ == '__main__':
    if sys.argv[1] == 'gpu':
        stage_gpu(sys.argv[2], *(sys.argv[3:4] or ['kb_curve_reg0_N20']), *[int(v) for v in sys.argv[4:6]])
    else:
        stage_cpu(sys.argv[2], *[int(v) for v in sys.argv[3:4]])
