"""Restatement of OSQP (the QP solver the reference calls through ``ca.conic('qp', 'osqp', ...)``,
DGSQP/solvers/DGSQP.py:186,200,246) in numpy.

TEST INFRASTRUCTURE ONLY -- never imported by dgsqp_amd/.  PARITY UNPINNED: OSQP is a third-party
dependency of the reference (``setup.py:15``, no version pin; CasADi 3.5/3.6 bundle OSQP 0.6.x) that
is absent from /root/reference and from this image.  This file restates its *published* algorithm:

  B. Stellato, G. Banjac, P. Goulart, A. Bemporad, S. Boyd, "OSQP: an operator splitting solver for
  quadratic programs", Math. Prog. Comp. 12 (2020) -- Algorithm 1 (ADMM), section 3.4 (termination),
  3.4/Prop. (infeasibility certificates), 4 (polish), 5.1 (Ruiz equilibration), 5.2 (rho selection /
  adaptation), with the documented default settings of OSQP 0.6: rho 0.1, sigma 1e-6, alpha 1.6,
  eps_abs = eps_rel 1e-3, eps_prim_inf = eps_dual_inf 1e-4, max_iter 4000, scaling 10,
  adaptive_rho on (tolerance 5), check_termination 25, polish delta 1e-6 with 3 refinement steps
  (the reference switches ``polish=True`` on, DGSQP.py:186).

How CasADi's conic plugin poses the problem (restated from its interface; source not in tree):
the decision-variable bounds ``lbx <= x <= ubx`` are appended as identity rows ABOVE the user's
``a`` rows, ``l = [lbx; lba]``, ``u = [ubx; uba]``.  The reference passes neither ``lbx/ubx`` nor
``lba`` (DGSQP.py:246) so they are -inf/+inf; OSQP treats rows whose two bounds are beyond
+-1e26 as "loose" (rho_i = 1e-6) and rows with l = u as equalities (rho_i = 1e3 rho).  Every call
starts from x = 0, y = 0 (``x0=0``, DGSQP.py:240-241).

Deliberate, stated deviations of this restatement (things the reference makes irreproducible):
  * adaptive-rho interval: OSQP picks it from wall-clock time of the first solve (a multiple of 25
    iterations); fixed to ``adaptive_rho_interval`` = 25 here;
  * the rho reached by adaptation persists inside CasADi's plugin from one solve to the next
    (SURVEY hazard 7); every call here starts from rho = 0.1;
  * the KKT systems are solved by dense LU instead of QDLDL (rounding-level differences).
"""
from __future__ import annotations

import numpy as np
import scipy.linalg as sla

OSQP_INFTY = 1e30
MIN_SCALING, MAX_SCALING = 1e-4, 1e4
RHO_MIN, RHO_MAX, RHO_TOL, RHO_EQ_OVER_RHO_INEQ = 1e-6, 1e6, 1e-4, 1e3

SOLVED, SOLVED_INACCURATE, MAX_ITER, PRIMAL_INFEASIBLE, DUAL_INFEASIBLE = 1, 2, -2, -3, -4
PRIMAL_INFEASIBLE_INACCURATE, DUAL_INFEASIBLE_INACCURATE = 3, 4


def _limit(v):
    v = np.where(v < MIN_SCALING, 1.0, v)
    return np.minimum(v, MAX_SCALING)


def _ruiz(P, q, A, iters):
    """Section 5.1 / OSQP scale_data(): D, E, c and the scaled data."""
    n, m = P.shape[0], A.shape[0]
    D, E, c = np.ones(n), np.ones(m), 1.0
    P, q, A = P.copy(), q.copy(), A.copy()
    for _ in range(iters):
        dn = np.maximum(np.abs(P).max(axis=0), np.abs(A).max(axis=0) if m else 0.0)      # column norms of the KKT matrix
        en = np.abs(A).max(axis=1) if m else np.zeros(0)
        dt, et = 1.0 / np.sqrt(_limit(dn)), 1.0 / np.sqrt(_limit(en))
        P = dt[:, None] * P * dt[None, :]
        A = et[:, None] * A * dt[None, :]
        q = dt * q
        D *= dt
        E *= et
        ct = _limit(np.array([np.abs(P).max(axis=0).mean()]))[0]
        qn = np.abs(q).max()
        qn = 1.0 if qn < MIN_SCALING else min(qn, MAX_SCALING)
        ct = 1.0 / max(ct, qn)
        P *= ct
        q *= ct
        c *= ct
    return P, q, A, D, E, c


class _KKT:
    def __init__(self, P, A, sigma, rho_vec):
        n, m = P.shape[0], A.shape[0]
        K = np.zeros((n + m, n + m))
        K[:n, :n] = P + sigma * np.eye(n)
        K[:n, n:] = A.T
        K[n:, :n] = A
        K[n:, n:] = -np.diag(1.0 / rho_vec)
        self.lu = sla.lu_factor(K)

    def solve(self, rhs):
        return sla.lu_solve(self.lu, rhs)


def solve(P, q, A, l, u, *, rho=0.1, sigma=1e-6, alpha=1.6, eps_abs=1e-3, eps_rel=1e-3, eps_prim_inf=1e-4,
          eps_dual_inf=1e-4, max_iter=4000, scaling=10, adaptive_rho=True, adaptive_rho_interval=25,
          adaptive_rho_tolerance=5.0, check_termination=25, polish=True, delta=1e-6, polish_refine_iter=3):
    """min 1/2 x'Px + q'x  s.t. l <= Ax <= u.  Returns dict(x, y, status, iters, polished, rho)."""
    P = 0.5 * (np.asarray(P, float) + np.asarray(P, float).T)    # OSQP holds the upper triangle only
    q, A = np.asarray(q, float), np.asarray(A, float)
    l = np.maximum(np.asarray(l, float), -OSQP_INFTY)
    u = np.minimum(np.asarray(u, float), OSQP_INFTY)
    n, m = P.shape[0], A.shape[0]
    Ps, qs, As, D, E, c = _ruiz(P, q, A, scaling)
    ls, us = E * l, E * u
    Dinv, Einv, cinv = 1.0 / D, 1.0 / E, 1.0 / c

    def make_rho_vec(r):
        loose = (ls < -OSQP_INFTY * MIN_SCALING) & (us > OSQP_INFTY * MIN_SCALING)
        eq = (us - ls) < RHO_TOL
        return np.where(loose, RHO_MIN, np.where(eq, RHO_EQ_OVER_RHO_INEQ * r, r))

    rho_vec = make_rho_vec(rho)
    kkt = _KKT(Ps, As, sigma, rho_vec)
    x, z, y = np.zeros(n), np.zeros(m), np.zeros(m)
    status, it = MAX_ITER, 0
    pri_res = dua_res = np.inf
    delta_x = np.zeros(n)
    delta_y = np.zeros(m)

    def residuals():
        Ax, Px, Aty = As @ x, Ps @ x, As.T @ y
        pr = np.abs(Einv * (Ax - z)).max() if m else 0.0
        dr = cinv * np.abs(Dinv * (Px + qs + Aty)).max()
        eps_p = eps_abs + eps_rel * max(np.abs(Einv * z).max(), np.abs(Einv * Ax).max()) if m else eps_abs
        eps_d = eps_abs + eps_rel * cinv * max(np.abs(Dinv * qs).max(), np.abs(Dinv * Aty).max(), np.abs(Dinv * Px).max())
        return pr, dr, eps_p, eps_d, Ax, Px, Aty

    def primal_infeasible(eps):
        dy = delta_y.copy()
        inf_u, inf_l = us > OSQP_INFTY * MIN_SCALING, ls < -OSQP_INFTY * MIN_SCALING
        dy = np.where(inf_u & inf_l, 0.0, np.where(inf_u, np.minimum(dy, 0.0), np.where(inf_l, np.maximum(dy, 0.0), dy)))
        nrm = np.abs(E * dy).max() if m else 0.0
        if nrm <= 1.0 / OSQP_INFTY:
            return False
        fin_u, fin_l = ~inf_u, ~inf_l
        lhs = (us[fin_u] * np.maximum(dy[fin_u], 0.0)).sum() + (ls[fin_l] * np.minimum(dy[fin_l], 0.0)).sum()
        if lhs < -eps * nrm:
            return np.abs(Dinv * (As.T @ dy)).max() < eps * nrm
        return False

    def dual_infeasible(eps):
        nrm = np.abs(D * delta_x).max()
        if nrm <= 1.0 / OSQP_INFTY:
            return False
        cost_scaling = c
        if qs @ delta_x < -cost_scaling * eps * nrm:
            if np.abs(Dinv * (Ps @ delta_x)).max() < cost_scaling * eps * nrm:
                Adx = Einv * (As @ delta_x)
                ok_u = (us > OSQP_INFTY * MIN_SCALING) | (Adx < eps * nrm)
                ok_l = (ls < -OSQP_INFTY * MIN_SCALING) | (Adx > -eps * nrm)
                return bool(np.all(ok_u & ok_l))
        return False

    for it in range(1, max_iter + 1):
        x_prev, z_prev = x, z
        sol = kkt.solve(np.concatenate([sigma * x_prev - qs, z_prev - y / rho_vec]))
        xt = sol[:n]
        zt = z_prev + (sol[n:] - y) / rho_vec
        x = alpha * xt + (1 - alpha) * x_prev
        delta_x = x - x_prev
        zr = alpha * zt + (1 - alpha) * z_prev
        z = np.minimum(np.maximum(zr + y / rho_vec, ls), us)
        delta_y = rho_vec * (zr - z)
        y = y + delta_y
        check = check_termination and it % check_termination == 0
        adapt = adaptive_rho and adaptive_rho_interval and it % adaptive_rho_interval == 0
        if check:
            pri_res, dua_res, eps_p, eps_d, *_ = residuals()
            if pri_res <= eps_p and dua_res <= eps_d:
                status = SOLVED
                break
            if primal_infeasible(eps_prim_inf):
                status = PRIMAL_INFEASIBLE
                break
            if dual_infeasible(eps_dual_inf):
                status = DUAL_INFEASIBLE
                break
        if adapt:
            Ax, Px, Aty = As @ x, Ps @ x, As.T @ y
            pr = np.abs(Ax - z).max() / (max(np.abs(z).max(), np.abs(Ax).max()) + 1e-10) if m else 0.0
            dr = np.abs(Px + qs + Aty).max() / (max(np.abs(qs).max(), np.abs(Aty).max(), np.abs(Px).max()) + 1e-10)
            rho_new = min(max(rho * np.sqrt(pr / (dr + 1e-10)), RHO_MIN), RHO_MAX)
            if rho_new > rho * adaptive_rho_tolerance or rho_new < rho / adaptive_rho_tolerance:
                rho = rho_new
                rho_vec = make_rho_vec(rho)
                kkt = _KKT(Ps, As, sigma, rho_vec)
    else:
        pri_res, dua_res, eps_p, eps_d, *_ = residuals()
        # OSQP re-checks with 10x tolerances at the iteration limit ("inaccurate" statuses)
        # (check_termination(work, approximate = 1): residual tolerances AND both infeasibility tolerances times ten, same order)
        if pri_res <= 10 * eps_p and dua_res <= 10 * eps_d:
            status = SOLVED_INACCURATE
        elif primal_infeasible(10 * eps_prim_inf):
            status = PRIMAL_INFEASIBLE_INACCURATE
        elif dual_infeasible(10 * eps_dual_inf):
            status = DUAL_INFEASIBLE_INACCURATE
        else:
            status = MAX_ITER

    polished = 0
    if status == SOLVED and polish:
        pri_res, dua_res, *_ = residuals()
        low = (z - ls) < -y
        upp = (us - z) < y
        act = low | upp
        Ared = As[act]
        rhs_b = np.where(low, ls, us)[act]
        na = int(act.sum())
        K = np.zeros((n + na, n + na))
        K[:n, :n] = Ps
        K[:n, n:] = Ared.T
        K[n:, :n] = Ared
        Kreg = K.copy()
        Kreg[:n, :n] += delta * np.eye(n)
        Kreg[n:, n:] -= delta * np.eye(na)
        rhs = np.concatenate([-qs, rhs_b])
        try:
            lu = sla.lu_factor(Kreg)
            sol = sla.lu_solve(lu, rhs)
            for _ in range(polish_refine_iter):
                sol = sol + sla.lu_solve(lu, rhs - K @ sol)
            xp = sol[:n]
            yp = np.zeros(m)
            yp[act] = sol[n:]
            zp = As @ xp
            pr_p = np.abs(Einv * (zp - np.minimum(np.maximum(zp, ls), us))).max() if m else 0.0
            dr_p = cinv * np.abs(Dinv * (Ps @ xp + qs + As.T @ yp)).max()
            ok = (pr_p < pri_res and dr_p < dua_res) or (pr_p < pri_res and dua_res < 1e-10) or (dr_p < dua_res and pri_res < 1e-10)
            if ok and np.all(np.isfinite(sol)):
                x, y, z = xp, yp, zp
                polished = 1
            else:
                polished = -1
        except (sla.LinAlgError, ValueError):
            polished = -1

    if status in (PRIMAL_INFEASIBLE, DUAL_INFEASIBLE, PRIMAL_INFEASIBLE_INACCURATE, DUAL_INFEASIBLE_INACCURATE):
        xo, yo = np.full(n, np.nan), np.full(m, np.nan)       # OSQP stores NaN when there is no solution
    else:
        xo, yo = D * x, cinv * E * y
    return dict(x=xo, y=yo, status=status, iters=it, polished=polished, rho=rho)


def conic(H, g, A, uba, **kw):
    """The call of DGSQP.py:246 -- ``solver(h=Q, g=q, a=G, uba=-g, x0=0)`` -- as CasADi's OSQP plugin poses it:
    identity rows for the (absent) variable bounds above the ``a`` rows.  Returns (x, lam_a, info)."""
    n = H.shape[0]
    if not (np.all(np.isfinite(H)) and np.all(np.isfinite(g)) and np.all(np.isfinite(A)) and not np.any(np.isnan(uba))):
        return np.full(n, np.nan), np.full(A.shape[0], np.nan), dict(status=-10, iters=0, polished=0, rho=np.nan)   # NaN data: no defined result
    Afull = np.vstack([np.eye(n), A])
    l = np.full(n + A.shape[0], -np.inf)
    u = np.concatenate([np.full(n, np.inf), uba])
    r = solve(H, g, Afull, l, u, **kw)
    return r['x'], r['y'][n:], r
