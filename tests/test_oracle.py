"""The CPU oracle against everything that can pin it without CasADi/OSQP:
finite differences, scipy's own lsqr, numpy's eigh, KKT conditions, scipy.optimize, golden fixtures."""
import pathlib

import numpy as np
import pytest
import scipy.optimize
import scipy.sparse as sp
import scipy.sparse.linalg as spl

from conftest import agent_major, tight_lsqr

GOLD = pathlib.Path(__file__).parent / 'golden'


# ---------------------------------------------------------------------------------------------
# dynamics: f_c vs the independent numpy restatement, f_d derivatives vs central differences
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('name', ['kb_chicane_N15', 'dyn_curve_N15'])
def test_dynamics_and_derivatives(oracle, games, name):
    g, P, _ = games[name]
    mdl = g.joint_model.dynamics_models[0]
    rng = np.random.default_rng(0)
    for _ in range(5):
        q = rng.standard_normal(mdl.n_q) * 0.2
        q[2] = 2.0 + rng.random()                      # forward speed
        q[mdl.s_idx] = rng.random() * 12
        u = np.array([rng.standard_normal() * 0.5, rng.standard_normal() * 0.2])
        dq, qn, J, H = oracle.dynamics(P, 0, q, u)
        np.testing.assert_allclose(dq, mdl.fc(q, u), rtol=1e-12, atol=1e-12)   # dynamics_models.py:1046-1070 / :2008-2062
        z = np.concatenate([q, u])
        nz = len(z)
        eps = 1e-6
        for i in range(nz):
            zp, zm = z.copy(), z.copy()
            zp[i] += eps
            zm[i] -= eps
            _, fp, Jp, _ = oracle.dynamics(P, 0, zp[:mdl.n_q], zp[mdl.n_q:])
            _, fm, Jm, _ = oracle.dynamics(P, 0, zm[:mdl.n_q], zm[mdl.n_q:])
            np.testing.assert_allclose(J[:, i], (fp - fm) / (2 * eps), rtol=1e-6, atol=1e-7)       # fAd, fBd
            np.testing.assert_allclose(H[:, :, i], (Jp - Jm) / (2 * eps), rtol=1e-5, atol=2e-6)    # fEd, fFd, fGd
        assert np.abs(H - H.transpose(0, 2, 1)).max() < 1e-12


def test_integrators_agree_in_the_small_step_limit(oracle, games):
    import copy
    g, P, _ = games['kb_chicane_N15']
    q = np.array([0.5, 0.1, 2.5, 0.05, 0.5, 0.1])
    u = np.array([0.3, 0.1])
    ref = None
    for integ, sub in ((1, 50), (2, 50), (3, 200), (0, 1)):
        Pc = copy.copy(P)
        Pc.integrator, Pc.substeps = integ, sub
        qn = oracle.dynamics(Pc, 0, q, u, derivs=False)[1]
        if ref is None:
            ref = qn
        tol = 3e-2 if integ == 0 else (1e-5 if integ == 3 else 1e-8)   # euler is first order in dt = 0.1
        np.testing.assert_allclose(qn, ref, atol=tol)


# ---------------------------------------------------------------------------------------------
# _evaluate: G, q, Q against finite differences of g, J^a and grad_{u^a} L^a   
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('name', ['kb_chicane_N15', 'dyn_curve_N15', 'merge_N8'])
def test_evaluate_against_finite_differences(oracle, games, name):
    from dgsqp_amd.montecarlo import sample_scenarios
    g, P, par = games[name]
    d = oracle.dims(P)
    n, nc = d['n'], d['nc']
    x0, u_tm = sample_scenarios(g, 1, seed=5)
    rng = np.random.default_rng(1)
    u = agent_major(u_tm)[0] + (0.3 if name.startswith('merge') else 0.01) * rng.standard_normal(n)   # merge starts from zero inputs
    l = np.maximum(0, rng.standard_normal(nc))
    ev = oracle.evaluate(P, x0[0], u, l, hessian=1)
    lit = oracle.evaluate(P, x0[0], u, l, hessian=2)            # literal per-row DP (DGSQP.py:829-877)
    assert np.abs(ev['Q'] - lit['Q']).max() < 1e-9 * max(1.0, np.abs(ev['Q']).max())
    eps = 1e-6
    Gfd, qfd, Qfd = np.zeros((nc, n)), np.zeros(n), np.zeros((n, n))

    def grad_lagrangian(uu):
        e = oracle.evaluate(P, x0[0], uu, None, 0)
        return e['q'] + e['G'].T @ l
    for i in range(n):
        up, um = u.copy(), u.copy()
        up[i] += eps
        um[i] -= eps
        ep, em = oracle.evaluate(P, x0[0], up, None, 0), oracle.evaluate(P, x0[0], um, None, 0)
        Gfd[:, i] = (ep['g'] - em['g']) / (2 * eps)
        a = i // (n // d['M'])
        qfd[i] = (ep['J'][a] - em['J'][a]) / (2 * eps)
        Qfd[:, i] = ((ep['q'] + ep['G'].T @ l) - (em['q'] + em['G'].T @ l)) / (2 * eps)
    assert np.abs(Gfd - ev['G']).max() < 1e-6 * max(1, np.abs(ev['G']).max())
    assert np.abs(qfd - ev['q']).max() < 1e-6 * max(1, np.abs(ev['q']).max())
    assert np.abs(Qfd - ev['Q']).max() < 2e-6 * max(1, np.abs(ev['Q']).max())
    if not name.startswith('merge'):                       # (the merge game's costs are decoupled: a potential game)
        assert np.abs(ev['Q'] - ev['Q'].T).max() > 1e-3    # the game Hessian is NOT symmetric


def test_unicycle_and_merge_rows(oracle, games):
    """Kinematic unicycle (dynamics_models.py:331-339) and the merge game's row layout (merge.py:316-356): 18 / 27 / 15 rows
    per stage for three cars (36 / 63 / 39 for six), lane rows at k = 0 with a zero gradient."""
    from dgsqp_amd.montecarlo import sample_scenarios
    g, P, par = games['merge_N8']
    mdl = g.joint_model.dynamics_models[0]
    rng = np.random.default_rng(0)
    q, u = np.array([0.3, 0.1, 0.4, 0.2]), np.array([0.5, -0.7])
    dq, qn, J, H = oracle.dynamics(P, 0, q, u)
    np.testing.assert_allclose(dq, mdl.fc(q, u), rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(qn, mdl.fd(q, u), rtol=1e-13, atol=1e-15)          # rk3, one sub-step (:200-211)
    eps = 1e-6
    for i in range(6):
        z = np.concatenate([q, u]); zp, zm = z.copy(), z.copy(); zp[i] += eps; zm[i] -= eps
        fp, fm = oracle.dynamics(P, 0, zp[:4], zp[4:])[1], oracle.dynamics(P, 0, zm[:4], zm[4:])[1]
        np.testing.assert_allclose(J[:, i], (fp - fm) / (2 * eps), rtol=1e-6, atol=1e-8)
    rows = oracle.rows(P)
    per_stage = np.bincount(rows[:, 1], minlength=9)
    assert per_stage[0] == 18 and (per_stage[1:8] == 27).all() and per_stage[8] == 15 and len(rows) == 18 + 7 * 27 + 15
    x0, u_tm = sample_scenarios(g, 4, seed=1)
    assert np.all(u_tm == 0)
    ev = oracle.evaluate(P, x0[0], agent_major(u_tm)[0], None, 0)
    k0 = np.where(rows[:, 1] == 0)[0]
    lanes0 = [r for r in k0 if rows[r, 0] == 7]
    assert len(lanes0) == 6 and np.all(ev['G'][lanes0] == 0) and np.all(ev['g'][lanes0] < 0)
    l0 = oracle.dual_init(P, par, x0[0], agent_major(u_tm)[0])                  # G G^T is singular: min-norm LSQR solution
    assert np.all(np.isfinite(l0)) and np.all(l0[lanes0] == 0)
    res = oracle.solve_batch(P, par, x0, agent_major(u_tm))
    assert np.all(res['status'] <= 1) and np.all(res['num_iters'] < 30)


def test_constraint_row_order_and_counts(oracle, games):
    """Row layout of DGSQP.py:732-821: 16 / 21 / 5 rows per stage for the 2-agent chicane game."""
    _, P, _ = games['kb_chicane_N25']
    rows = oracle.rows(P)
    assert len(rows) == 525
    per_stage = np.bincount(rows[:, 1], minlength=26)
    assert per_stage[0] == 16 and (per_stage[1:25] == 21).all() and per_stage[25] == 5
    k1 = rows[rows[:, 1] == 1]
    # [obstacle ; agent0: rate(ub,lb,ub,lb), in ub x2, in lb x2, state ub, state lb ; agent1 ...]
    assert list(k1[:11, 0]) == [0, 1, 2, 1, 2, 3, 3, 4, 4, 5, 6]
    assert list(k1[:11, 2]) == [0] * 11 and list(k1[11:, 2]) == [1] * 10


# ---------------------------------------------------------------------------------------------
# third-party pieces pinned against the installed third party itself
# ---------------------------------------------------------------------------------------------
def test_lsqr_matches_scipy(oracle):
    """scipy.sparse.linalg.lsqr is the routine DGSQP.py:324 calls; same algorithm, same defaults."""
    rng = np.random.default_rng(3)
    for m, n in ((30, 30), (60, 40), (25, 50)):
        A = rng.standard_normal((m, n))
        b = rng.standard_normal(m)
        x, istop, itn = oracle.lsqr(A, b)
        ref = spl.lsqr(sp.csr_matrix(A), b)
        assert istop == ref[1] and itn == ref[2]
        # At the default 1e-6 tolerance LSQR's iterates are sensitive to summation order: scipy's own
        # dense and sparse operator paths differ by ~5e-5 on the 30x30 case.  Use that spread as the yardstick.
        spread = np.abs(spl.lsqr(A, b)[0] - ref[0]).max()
        assert np.abs(x - ref[0]).max() <= max(10 * spread, 1e-8)
        xt, _, itt = oracle.lsqr(A, b, atol=1e-12, btol=1e-12)
        reft = spl.lsqr(sp.csr_matrix(A), b, atol=1e-12, btol=1e-12)
        assert itt == reft[2]
        np.testing.assert_allclose(xt, reft[0], rtol=0, atol=1e-9)
    # singular symmetric G G^T system as in the dual start: min-norm solution within LSQR's own tolerance
    Gm = rng.standard_normal((40, 12))
    A, b = Gm @ Gm.T, Gm @ rng.standard_normal(12)
    x, _, _ = oracle.lsqr(A, b)
    ref = spl.lsqr(sp.csr_matrix(A), b)[0]
    assert np.linalg.norm(x - ref) < 1e-4 * np.linalg.norm(ref)


def test_dual_init_matches_reference_formula(oracle, games):
    from dgsqp_amd.montecarlo import sample_scenarios
    g, P, par = games['kb_chicane_N15']
    x0, u_tm = sample_scenarios(g, 2, seed=2)
    u = agent_major(u_tm)
    for b in range(2):
        ev = oracle.evaluate(P, x0[b], u[b], None, 0)
        Gs = sp.csc_matrix(ev['G'])
        ref = np.maximum(0, -spl.lsqr(Gs @ Gs.T, Gs @ ev['q'])[0])      # DGSQP.py:323-324 verbatim
        l0 = oracle.dual_init(P, par, x0[b], u[b])
        assert np.linalg.norm(l0 - ref) < 2e-3 * np.linalg.norm(ref)


def test_nearest_pd_matches_numpy_eigh(oracle):
    rng = np.random.default_rng(4)
    for n in (5, 40, 100):
        A = rng.standard_normal((n, n))
        Bm = (A + A.T) / 2                                            # DGSQP.py:1290-1296 verbatim
        s, U = np.linalg.eigh(Bm)
        s[np.where(s < 0)[0]] = 1e-10
        C = U @ np.diag(s) @ U.T
        ref = (C + C.T) / 2 + 1e-3 * np.eye(n)
        out = oracle.nearest_pd(A, 1e-3)
        assert np.abs(out - ref).max() < 1e-11 * max(1, np.abs(ref).max())
        assert np.linalg.eigvalsh(out).min() > 1e-3 - 1e-9
        se, Ue = oracle.eigh(Bm)
        np.testing.assert_allclose(np.sort(se), np.linalg.eigvalsh(Bm), atol=1e-11 * np.abs(Bm).max() * n)
    Ppd = A @ A.T + np.eye(n)
    assert np.abs(oracle.nearest_pd(Ppd, 0.0) - Ppd).max() < 1e-10 * np.abs(Ppd).max()   # identity on PD input


def _kkt(H, c, G, g, x, lam):
    stat = np.abs(H @ x + c + G.T @ lam).max()
    return stat, (G @ x + g).max(), lam.min(), np.abs(lam * (G @ x + g)).max()


def test_qp_kkt_and_scipy_crosscheck(oracle):
    rng = np.random.default_rng(5)
    for n, m in ((6, 10), (20, 60), (40, 150)):
        A = rng.standard_normal((n, n))
        H = A @ A.T + 0.1 * np.eye(n)
        c = rng.standard_normal(n) * 3
        G = rng.standard_normal((m, n))
        g = -rng.random(m) * 0.5                           # x = 0 strictly feasible
        x, lam, flag = oracle.qp(H, c, G, g)
        assert flag == 0
        stat, pf, lmin, comp = _kkt(H, c, G, g, x, lam)
        assert stat < 1e-9 and pf < 1e-9 and lmin >= 0 and comp < 1e-9
        if n <= 20:
            res = scipy.optimize.minimize(lambda v: 0.5 * v @ H @ v + c @ v, np.zeros(n), jac=lambda v: H @ v + c,
                                          constraints=[dict(type='ineq', fun=lambda v: -(G @ v + g), jac=lambda v: -G)],
                                          method='SLSQP', options=dict(ftol=1e-14, maxiter=500))
            np.testing.assert_allclose(x, res.x, atol=2e-5)
    # infeasible: x <= -1 and x >= 1
    x, lam, flag = oracle.qp(np.eye(2), np.zeros(2), np.array([[1.0, 0], [-1.0, 0]]), np.array([1.0, 1.0]))
    assert flag == 1
    # linearly dependent active rows (duplicated constraint) are handled
    x, lam, flag = oracle.qp(np.eye(2), np.array([-2.0, 0]), np.array([[1.0, 0], [1.0, 0]]), np.array([-1.0, -1.0]))
    assert flag == 0 and x[0] == pytest.approx(1.0) and lam.sum() == pytest.approx(1.0)


def test_game_qp_satisfies_kkt(oracle, games):
    from dgsqp_amd.montecarlo import sample_scenarios
    g, P, par = games['kb_chicane_N25']
    x0, u_tm = sample_scenarios(g, 2, seed=7)
    u = agent_major(u_tm)
    for b in range(2):
        l0 = oracle.dual_init(P, par, x0[b], u[b])
        ev = oracle.evaluate(P, x0[b], u[b], l0, 1)
        Qpd = oracle.nearest_pd(ev['Q'], par.reg)
        du, lam, flag = oracle.qp(Qpd, ev['q'], ev['G'], ev['g'])
        assert flag == 0
        stat, pf, lmin, comp = _kkt(Qpd, ev['q'], ev['G'], ev['g'], du, lam)
        assert stat < 1e-8 and pf < 1e-9 and lmin >= 0 and comp < 1e-8


# ---------------------------------------------------------------------------------------------
# merit function pieces (DGSQP.py:949-979, :559-585)
# ---------------------------------------------------------------------------------------------
def test_merit_and_mu_formulas(oracle, games):
    from dgsqp_amd.montecarlo import sample_scenarios
    g, P, par = games['kb_chicane_N15']
    d = oracle.dims(P)
    x0, u_tm = sample_scenarios(g, 1, seed=9)
    u = agent_major(u_tm)[0]
    rng = np.random.default_rng(2)
    l = np.maximum(0, rng.standard_normal(d['nc']))
    ev = oracle.evaluate(P, x0[0], u, l, 1)
    Q, q, G, gg = ev['Q'], ev['q'], ev['G'], ev['g']
    du, dl = rng.standard_normal(d['n']) * 0.1, rng.standard_normal(d['nc']) * 0.1
    s = np.minimum(0, gg)
    mu_in = 0.7
    phi, dphi, mu = oracle.merit(P, par, Q, q, G, gg, l, s, du, dl, mu_in)
    stat = np.concatenate([q + G.T @ l, [l @ gg]])
    assert phi == pytest.approx(0.5 * stat @ stat + mu_in * np.sum(gg - s), rel=1e-12)
    dstat = (q + G.T @ l) @ np.hstack([Q, G.T]) @ np.concatenate([du, dl]) + (l @ gg) * (l @ G @ du + dl @ gg)
    assert dphi == pytest.approx(dstat - mu_in * np.sum(gg - s), rel=1e-10)
    vio = np.sum(gg - s)
    assert mu == pytest.approx(abs(dstat) / (0.5 * vio) if vio > 0 else 0.0, rel=1e-10)


# ---------------------------------------------------------------------------------------------
# full solves: invariants + committed fixtures
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('name', ['kb_chicane_N15', 'kb_curve_N10', 'dyn_curve_N15', 'kb_barc2_N15', 'merge_N8'])
def test_solve_reproduces_golden_and_invariants(oracle, games, name):
    g, P, par = games[name]
    gold = np.load(GOLD / f'{name}.npz')
    B = 8
    x0, u = gold['x0'][:B], agent_major(gold['u_ws'][:B])
    out = oracle.solve_batch(P, tight_lsqr(par), x0, u, nthreads=8)
    assert (out['status'] == gold['status'][:B]).all()
    assert (out['num_iters'] == gold['num_iters'][:B]).all()
    assert (out['qp_solves'] == gold['qp_solves'][:B]).all()
    np.testing.assert_allclose(out['u'], gold['u'][:B], rtol=0, atol=1e-9)
    for b in range(B):
        if out['status'][b] == 0:                      # conv_abs_tol => recomputed optimality measures below tolerance
            ev = oracle.evaluate(P, x0[b], out['u'][b], out['l'][b], 0)
            assert max(0, ev['g'].max()) < par.p_tol
            assert np.abs(ev['g'] * out['l'][b]).max() < par.d_tol
            assert np.abs(ev['q'] + ev['G'].T @ out['l'][b]).max() < par.d_tol
            assert (out['l'][b] >= 0).all()


def test_literal_per_row_hessian_gives_same_solve(oracle, games):
    g, P, par = games['kb_curve_N10']
    gold = np.load(GOLD / 'kb_curve_N10.npz')
    x0, u = gold['x0'][:2], agent_major(gold['u_ws'][:2])
    a = oracle.solve_batch(P, tight_lsqr(par), x0, u, literal=0)
    b = oracle.solve_batch(P, tight_lsqr(par), x0, u, literal=1)
    assert (a['status'] == b['status']).all() and (a['num_iters'] == b['num_iters']).all()
    np.testing.assert_allclose(a['u'], b['u'], atol=1e-8)


def test_restated_osqp_solves_small_qps_and_flags_infeasibility(oracle):
    """oracle/osqp_restate.py (restatement of OSQP's published algorithm, the QP the reference calls): on strictly convex
    random QPs the polished point is the exact minimiser (KKT to 1e-8, equal to the active-set solution); an infeasible
    system yields the primal-infeasibility certificate and NaNs, as OSQP stores them."""
    from oracle import osqp_restate
    rng = np.random.default_rng(4)
    polished = 0
    for trial in range(12):
        n, m = 12, 30
        A = rng.standard_normal((n, n))
        H = A @ A.T + 0.1 * np.eye(n)
        q = rng.standard_normal(n)
        G = rng.standard_normal((m, n))
        g = -rng.random(m) - 0.05                       # x = 0 strictly feasible
        x, lam, info = osqp_restate.conic(H, q, G, -g)
        assert info['status'] == osqp_restate.SOLVED and info['iters'] % 25 == 0
        xe, le, flag = oracle.qp(H, q, G, g)
        assert flag == 0
        if info['polished'] == 1:
            polished += 1
            assert np.abs(H @ x + q + G.T @ lam).max() < 1e-8 and (G @ x + g).max() < 1e-8
            if lam.min() > -1e-9:                       # polish does not check multiplier signs; when they are fine it IS the minimiser
                assert np.abs(x - xe).max() < 1e-6 and np.abs(lam - le).max() < 1e-5
        else:                                           # ADMM point at eps = 1e-3
            assert np.abs(x - xe).max() < 5e-2
    assert polished >= 8
    G = np.array([[1.0, 0.0], [-1.0, 0.0]])
    x, lam, info = osqp_restate.conic(np.eye(2), np.zeros(2), G, -np.array([1.0, 1.0]))      # x0 <= -1 and x0 >= 1
    assert info['status'] == osqp_restate.PRIMAL_INFEASIBLE and np.isnan(x).all()


def test_cpp_oracle_follows_the_numpy_loop(oracle, games):
    """oracle/pyref.py restates DGSQP.solve() line by line in numpy with the library routines the reference itself calls
    (numpy.linalg.eigh, scipy.sparse.linalg.lsqr at its defaults).  With the same exact QP plugged in, the C++ oracle must take
    the same path: this separates the SQP logic / _nearestPD / LSQR of the C++ oracle from its QP.  With the restated OSQP
    as the QP (what the reference runs) the flags and counts still agree on these well-conditioned scenarios."""
    from dgsqp_amd.montecarlo import sample_scenarios
    from oracle import pyref
    g, P, par = games['kb_chicane_N15']
    x0, u_tm = sample_scenarios(g, 6, seed=3)
    u = agent_major(u_tm)
    ref = oracle.solve_batch(P, par, x0, u, nthreads=6)
    code = {'conv_abs_tol': 0, 'conv_rel_tol': 1, 'max_it': 2, 'diverged': 3, 'exception': 4}
    for kind, tol in (('gi', 1e-7), ('osqp', 5e-3)):
        for b in range(6 if kind == 'gi' else 3):
            s = pyref.PyRef(P, par, qp=kind).solve(x0[b], u[b])
            assert (code[s['msg']], s['num_iters'], s['qp_solves']) == (ref['status'][b], ref['num_iters'][b], ref['qp_solves'][b]), (kind, b)
            assert np.abs(s['l_init'] - ref['l_init'][b]).max() < 1e-3          # two LSQR implementations at tolerance 1e-6
            assert np.abs(s['u'] - ref['u'][b]).max() < tol * max(1.0, np.abs(ref['u'][b]).max()), (kind, b)


def test_cpp_osqp_follows_the_numpy_restatement(oracle, games):
    """oracle/osqp.hpp (the OSQP the C++ oracle solves its QPs with when qp_method = 1, and the checker of the device's ADMM kernel)
    is the statement-by-statement C++ twin of oracle/osqp_restate.py: same status, ADMM iteration count, rho, polish verdict, and
    x / lambda to rounding level -- on random strictly convex QPs, on an infeasible one (NaN answer), and on QPs of two games
    (reg = 1e-3 and the literal reg = 0 projection) at the start point of four scenarios each."""
    from dgsqp_amd.montecarlo import sample_scenarios
    from oracle import osqp_restate
    rng = np.random.default_rng(11)

    def both(H, q, G, g, tol=1e-8):
        xn, ln, inn = osqp_restate.conic(H, q, G, -g)
        xc, lc, ic = oracle.osqp(H, q, G, g)
        assert (ic['status'], ic['iters'], ic['polished']) == (inn['status'], inn['iters'], inn['polished'])
        if inn['status'] in (osqp_restate.PRIMAL_INFEASIBLE, osqp_restate.DUAL_INFEASIBLE):
            assert np.isnan(xc).all() and np.isnan(lc).all()
            return inn
        assert abs(ic['rho'] - inn['rho']) <= 1e-9 * inn['rho']
        assert np.abs(xc - xn).max() <= tol * max(1.0, np.abs(xn).max()) and np.abs(lc - ln).max() <= tol * max(1.0, np.abs(ln).max())
        return inn
    for trial in range(6):
        n, m = 10, 24
        A = rng.standard_normal((n, n))
        both(A @ A.T + 0.1 * np.eye(n), rng.standard_normal(n), rng.standard_normal((m, n)), -rng.random(m) - 0.05)
    inn = both(np.eye(2), np.zeros(2), np.array([[1.0, 0.0], [-1.0, 0.0]]), np.array([1.0, 1.0]))
    assert inn['status'] == osqp_restate.PRIMAL_INFEASIBLE
    x, lam, info = oracle.osqp(np.array([[np.nan, 0.0], [0.0, 1.0]]), np.zeros(2), np.eye(2), -np.ones(2))      # non-finite data: NaN answer
    assert info['status'] == -10 and np.isnan(x).all()
    seen_rho_update = 0
    for name in ('kb_chicane_N15', 'kb_barc2_N15'):
        g, P, par = games[name]
        x0, u_tm = sample_scenarios(g, 4, seed=5)
        u = agent_major(u_tm)
        for b in range(4):
            l0 = oracle.dual_init(P, par, x0[b], u[b])
            ev = oracle.evaluate(P, x0[b], u[b], l0, 1)
            inn = both(oracle.nearest_pd(ev['Q'], par.reg, par.eig_floor), ev['q'], ev['G'], ev['g'], tol=1e-7)
            seen_rho_update += inn['rho'] != 0.1
    assert seen_rho_update >= 1        # (the adaptive-rho branch was exercised)


def test_cpp_oracle_with_osqp_follows_the_numpy_loop(oracle, games):
    """The C++ oracle with qp_method = 1 (its OSQP restatement inside solve(), incl. "an infeasible QP ends the solve") against the
    numpy loop with the numpy restatement, the stand-in for the reference's own iterates: same status, iterations and QP solves on
    well-conditioned scenarios, iterates to 1e-6 (both polish; the differences are the two LSQR implementations at tolerance 1e-6)."""
    import copy
    from dgsqp_amd.montecarlo import sample_scenarios
    from oracle import pyref
    g, P, par0 = games['kb_chicane_N15']
    par = copy.copy(par0)
    par.qp_method = 1
    x0, u_tm = sample_scenarios(g, 4, seed=3)
    u = agent_major(u_tm)
    ref = oracle.solve_batch(P, par, x0, u, nthreads=4)
    code = {'conv_abs_tol': 0, 'conv_rel_tol': 1, 'max_it': 2, 'diverged': 3, 'exception': 4}
    for b in range(4):
        s = pyref.PyRef(P, par, qp='osqp').solve(x0[b], u[b])
        assert (code[s['msg']], s['num_iters'], s['qp_solves']) == (ref['status'][b], ref['num_iters'][b], ref['qp_solves'][b]), b
        assert np.abs(s['u'] - ref['u'][b]).max() < 1e-6 * max(1.0, np.abs(ref['u'][b]).max()), b


def test_v2_restatement_invariants(oracle):
    """Oracle restatement of DG-SQP v2 (DGSQP_v2.py:322-720).  With the parameters of the reference's study
    (comparison_study_barc/globals.py:27-55) the regularisation has to decay from 100 before the steps grow: ~350 iterations, all
    m-steps accepted, 'conv_abs_tol' with the measures below 1e-4, one QP per iteration.  With the DGSQPV2Params defaults
    (frequency 5, memory 3) the kinematic game exercises the other branches: the m-step's merit test fails, the watchdog returns to
    the checkpoint, the Armijo search along a heavily regularised step makes no progress and after rel_tol_req = 10 such m-steps
    the solve ends with 'conv_rel_tol' (DGSQP_v2.py:549-556) -- 60 iterations = 10 x (5 d-steps + 1 m-step)."""
    from dgsqp_amd import montecarlo as mc
    from dgsqp_amd.solver import build_problem, build_params
    from dgsqp_amd.solver_types import DGSQPV2Params
    g = mc.kinematic_racing_game('curve', N=8)
    g.params = DGSQPV2Params(dt=0.1, N=8)
    P, par = build_problem(*g.solver_args()), build_params(g.params)
    assert (par.variant, par.nms, par.nms_frequency, par.nms_memory_size, par.merit_decrease_condition) == (1, 1, 5, 3, 0)
    assert (par.reg, par.reg_decay, par.delta_decay, par.merit_decrease, par.merit_parameter) == (100.0, 0.95, 0.95, 0.01, -1.0)
    x0, u_tm = mc.sample_scenarios(g, 6, seed=3)
    u = agent_major(u_tm)
    res = oracle.solve_batch(P, par, x0, u, nthreads=6)
    stalled = res['status'] == 1
    assert stalled.sum() >= 3 and (res['num_iters'][stalled] == 60).all() and (res['cond'][stalled, 2] > 1e-2).all()
    g.params = DGSQPV2Params(dt=0.1, N=8, nms=True, nms_frequency=10, nms_memory_size=10, line_search_iters=20, sqp_iters=500, reg=1e2,
                             reg_decay=0.95, delta_decay=0.99, merit_decrease=0.01, beta=0.01, tau=0.5)
    res = oracle.solve_batch(P, build_params(g.params), x0, u, nthreads=6)
    ok = res['status'] == 0
    assert ok.sum() >= 2 and (res['num_iters'][ok] > 200).all() and (res['cond'][ok] < 1e-4).all() and (res['status'] <= 1).all()
    assert (res['qp_solves'] == res['num_iters']).all()            # one QP per iteration; the final iteration only tests convergence
    # merit 'sum_obj_l1' (DGSQP_v2.py:1151-1152, 1161-1164): sum of the agents' costs + mu * violation.  Its gradient (every agent's cost
    # w.r.t. every input, through the dense Du_x) against central differences of the costs; a solve with it
    obj, grad = oracle.sum_obj(P, x0[0], u[0])
    assert obj == pytest.approx(oracle.evaluate(P, x0[0], u[0], hessian=0)['J'].sum(), rel=1e-14)
    h, fd = 1e-6, np.zeros_like(grad)
    for i in range(len(grad)):
        e = np.zeros_like(grad); e[i] = h
        fd[i] = (oracle.evaluate(P, x0[0], u[0] + e, hessian=0)['J'].sum() - oracle.evaluate(P, x0[0], u[0] - e, hessian=0)['J'].sum()) / (2 * h)
    assert np.abs(grad - fd).max() < 1e-6 * max(1.0, np.abs(grad).max())
    g.params.merit_function = 'sum_obj_l1'
    par_obj = build_params(g.params)
    assert par_obj.merit_function == 2 and par_obj.variant == 1
    res_obj = oracle.solve_batch(P, par_obj, x0, u, nthreads=6)
    assert (res_obj['status'] <= 2).all() and (res_obj['num_iters'] > 20).all()
    assert not np.array_equal(res_obj['num_iters'], res['num_iters'])            # (a different merit: different paths)
    with pytest.raises(ValueError):
        g.params.merit_function = 'nope'
        build_params(g.params)


def test_spline_track_game_derivatives(oracle):
    """BASELINE configs[3]'s track: cubic-spline centre line (CasadiBSplineTrack, casadi_bspline_track.py:56-71, :122-149).  The
    host class, the oracle's track function and its derivatives through the game: tangent / curvature agree between host and
    oracle, d(tangent)/ds from the jets == finite differences, G and the game Hessian Q == finite differences of g and of the
    Lagrangian gradients (the curvature now has non-zero first and second derivatives, unlike the arc tracks)."""
    from dgsqp_amd import montecarlo as mc
    from dgsqp_amd.solver import build_problem, build_params
    for model in ('kinematic', 'dynamic'):
        g = mc.f1_racing_game(N=6, model=model, rk4_substeps=2)
        P = build_problem(*g.solver_args())
        assert P.track_kind == 1 and P.n_knots == 1103 and abs(P.track_L - 550.8628177372308) < 1e-9
        for s in (0.0, 10.3, 275.0, 550.5, 551.2, -3.0):
            c, t, dt = oracle.track(P, s)
            ch, th = g.track.lookup(np.array([s]))
            assert abs(c - ch[0]) < 1e-12 and abs(t - th[0]) < 1e-12
            h = 1e-6
            if s not in (0.0,):          # (the interpolant is not periodic: its tangent jumps by 9e-7 at the seam s = 0 = L)
                assert abs((oracle.track(P, s + h)[1] - oracle.track(P, s - h)[1]) / (2 * h) - dt) < 1e-7
        x0, u_tm = mc.sample_scenarios(g, 2, seed=0)
        u = agent_major(u_tm)[0] + 0.05 * np.random.default_rng(1).standard_normal(u_tm.shape[1] * u_tm.shape[2])
        nc, n = oracle.dims(P)['nc'], u.size
        l = np.abs(np.random.default_rng(0).standard_normal(nc))
        ev = oracle.evaluate(P, x0[0], u, l, 1)
        Gfd, Qfd, h = np.zeros((nc, n)), np.zeros((n, n)), 1e-6
        for i in range(n):
            up, um = u.copy(), u.copy()
            up[i] += h; um[i] -= h
            ep, em = oracle.evaluate(P, x0[0], up, l, 0), oracle.evaluate(P, x0[0], um, l, 0)
            Gfd[:, i] = (ep['g'] - em['g']) / (2 * h)
            Qfd[:, i] = ((ep['q'] + ep['G'].T @ l) - (em['q'] + em['G'].T @ l)) / (2 * h)
        assert np.abs(Gfd - ev['G']).max() < 1e-6 * max(1.0, np.abs(ev['G']).max())
        assert np.abs(Qfd - ev['Q']).max() < 1e-6 * max(1.0, np.abs(ev['Q']).max())


SYMPY_FIELDS = {'wheel_dist_front': 'L_f', 'wheel_dist_rear': 'L_r', 'mass': 'mass', 'yaw_inertia': 'I_z', 'gravity': 'gravity',
                'drag_coefficient': 'c_dr', 'damping_coefficient': 'c_da', 'slip_coefficient': 'c_s', 'rolling_resistance': 'c_r',
                'rolling_resistance_exponent': 'p_r', 'pacejka_b_front': 'pac_Bf', 'pacejka_b_rear': 'pac_Br', 'pacejka_c_front': 'pac_Cf',
                'pacejka_c_rear': 'pac_Cr', 'pacejka_d_front': 'pac_Df', 'pacejka_d_rear': 'pac_Dr'}


def sympy_kat_problem(games, kind):
    """The problem record whose agent 0 is the vehicle of tests/golden/sympy_fd_<kind>.npz (the REFERENCE's config defaults,
    model_types.py) on the curve track's arc; returns (P, kat)."""
    import copy
    kat = np.load(GOLD / f'sympy_fd_{kind}.npz')
    g, P, _ = games[{'kin': 'kb_curve_N10', 'dyn': 'dyn_curve_N15', 'uni': 'merge_N8'}[kind]]
    P = copy.deepcopy(P)
    A = P.agents[0]
    for name, fld in SYMPY_FIELDS.items():
        if 'param_' + name in kat.files:
            setattr(A, fld, float(kat['param_' + name]))
    if kind == 'dyn':
        assert str(kat['param_tire_model']) == 'pacejka'
        A.tire_model, A.simple_slip = 0, int(bool(kat['param_simple_slip']))
        A.drive_wheels = 0 if str(kat['param_drive_wheels']) == 'all' else 1
    return P, kat


@pytest.mark.parametrize('kind', ['uni', 'kin', 'dyn'])
def test_dynamics_tensors_against_sympy(oracle, games, kind):
    """SURVEY.md section 8c, KAT (1): f_d, [fAd fBd] and [fEd fGd; fGd^T fFd] (dynamics_models.py:128-144) of the oracle's jet
    arithmetic against EXACT symbolic derivatives -- f_c transcribed into sympy from dynamics_models.py:331-339 / :1046-1070 /
    :2013-2062 by tools/make_sympy_kats.py, independently of oracle/, with the reference's own config defaults; euler and rk4.
    1e-12 relative to the largest entry of each tensor (finite differences pin the same tensors to 1e-6 only)."""
    P, kat = sympy_kat_problem(games, kind)
    if kind != 'uni':            # the KAT's track is the arc of the curve track (c, s0, psi0 stored): the oracle's tables must say the same
        c, s0, psi0 = kat['track']
        for s in (3.0, 5.5):
            cc, tt, _ = oracle.track(P, s)
            assert cc == pytest.approx(c, rel=1e-15) and tt == pytest.approx(psi0 + c * (s - s0), rel=1e-14)
    nq = kat['points'].shape[1] - 2
    for tag, integ in (('euler', 0), ('rk4', 1)):
        if f'{tag}_M' not in kat.files:
            assert kind == 'dyn' and tag == 'rk4'       # (sympy needs about an hour for the Pacejka model's rk4 step: the file may hold euler only)
            continue
        P.integrator, P.substeps = integ, int(kat[f'{tag}_M'])
        assert P.dt == float(kat['dt'])
        for k, z in enumerate(kat['points']):
            dq, qn, J, H = oracle.dynamics(P, 0, z[:nq], z[nq:])
            for got, want, what in ((qn, kat[f'{tag}_fd'][k], 'fd'), (J, kat[f'{tag}_jac'][k], 'jac'), (H, kat[f'{tag}_hes'][k], 'hes')):
                err = np.abs(got - want).max() / max(1e-300, np.abs(want).max())
                assert err < 1e-12, (kind, tag, k, what, err)


@pytest.mark.parametrize('kind,method', [('kin', 'euler'), ('kin', 'rk4'), ('dyn', 'euler'), ('dyn', 'rk4')])
def test_one_stage_game_hessian_from_sympy_tensors(oracle, kind, method):
    """f_Q (DGSQP.py:678-727, :828-877) on a one-stage race: rows a of Duu J^a + B^T (D2 phi^a) B + sum_i (D phi^a)_i F_i with
    sympy's exact B = fBd, F = fFd (conftest.sympy_one_stage_Q) against the oracle's Q, multiplier 0.7 on the obstacle row."""
    from conftest import sympy_one_stage_game, sympy_one_stage_Q
    from dgsqp_amd.solver import build_problem
    if method == 'rk4' and 'rk4_M' not in np.load(GOLD / f'sympy_fd_{kind}.npz').files:
        pytest.skip('tests/golden/sympy_fd_dyn.npz holds the euler step only (tools/make_sympy_kats.py dyn: rk4 takes about an hour of sympy)')
    g, kat = sympy_one_stage_game(kind, method)
    P = build_problem(*g.solver_args())
    nqa = g.joint_model.dynamics_models[0].n_q
    s_idx = g.joint_model.dynamics_models[0].s_idx
    rows = oracle.rows(P)
    obs = int(np.nonzero(rows[:, 0] == 0)[0][0])
    pts = kat['points']
    for k1, k2 in ((0, 1), (2, 3), (3, 0)):
        x0 = np.concatenate([pts[k1][:nqa], pts[k2][:nqa]])
        u = np.concatenate([pts[k1][nqa:], pts[k2][nqa:]])
        l = np.zeros(len(rows))
        l[obs] = 0.7
        ev = oracle.evaluate(P, x0, u, l, 1)
        want = sympy_one_stage_Q(kat, method, k1, k2, 0.7, nqa, s_idx)
        assert np.abs(ev['Q'] - want).max() < 1e-12 * np.abs(want).max(), (kind, method, k1, k2)
        np.testing.assert_allclose(ev['x'].reshape(2, -1)[1], np.concatenate([kat[f'{method}_fd'][k1], kat[f'{method}_fd'][k2]]), rtol=1e-13, atol=1e-14)


@pytest.mark.parametrize('name', ['kb_barc3_N25', 'kb_f1_N50'])
def test_infeasible_verdicts_are_backed_by_an_lp(oracle, name):
    """The active-set QP (shared by oracle and device) ends 88 % / 30 % of the configs[2] / configs[3] solves with 'qp_fail'.  Its
    verdict is checked here against an algorithm that shares nothing with it: the LP  min t s.t. G du - t <= -g  (HiGHS) on QPs
    harvested from those games by tools/qp_infeasibility_lp.py (the numpy loop's iterates; up to 12 QPs the active-set method
    called infeasible + first QPs it solved).  t* > 0 <=> no du meets the linearised constraints, whatever the Hessian; the
    reference's OSQP would return 'primal infeasible' and a NaN step there (DGSQP.py:186 error_on_fail=False)."""
    import scipy.optimize
    path = GOLD / f'qp_infeasible_{name}.npz'
    if not path.exists():
        pytest.skip('fixture not generated')
    d = np.load(path)
    assert d['infeasible'].sum() >= 1
    for G, g, t_gold, bad in zip(d['G'], d['g'], d['t'], d['infeasible']):
        nc, n = G.shape
        A = np.hstack([G, -np.ones((nc, 1))])
        c = np.zeros(n + 1); c[-1] = 1.0
        r = scipy.optimize.linprog(c, A_ub=A, b_ub=-g, bounds=[(None, None)] * n + [(-1.0, None)], method='highs')
        assert r.status == 0 and r.x[-1] == pytest.approx(t_gold, abs=1e-7)
        assert (r.x[-1] > 1e-6) == bool(bad), (r.x[-1], bad)          # the verdict of the active-set method == the LP's
        # ... and the oracle's QP routine itself on this G, g (feasibility does not depend on the objective: identity Hessian)
        du, lam, flag = oracle.qp(np.eye(n), np.zeros(n), G, g)
        assert (flag != 0) == bool(bad)
        if not bad:
            assert (G @ du + g).max() < 1e-8
