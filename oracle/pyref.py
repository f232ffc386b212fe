"""Second, independent restatement of the reference's SQP loop -- plain numpy/scipy, line by line.

TEST INFRASTRUCTURE ONLY (never imported by dgsqp_amd/).  PARITY UNPINNED: the reference has no golden
vectors and its CasADi/OSQP arithmetic cannot run here; this file exists to reduce the shared fate of
oracle/dgsqp_oracle.cpp and the device code: it uses the very library routines the reference calls --
``numpy.linalg.eigh`` for ``_nearestPD`` (DGSQP.py:1290-1296), ``scipy.sparse.linalg.lsqr`` for the dual
start (DGSQP.py:320-327) -- and, for the QP, a restatement of OSQP itself (oracle/osqp_restate.py)
instead of the active-set method the C++ oracle and the device share.  Only the derivatives
(Q, q, G, g of ``_evaluate``, DGSQP.py:509-533) come from the C++ oracle; those are pinned against finite
differences in tests/test_oracle.py.

Every function cites the reference lines it follows.  NaN handling is the reference's: comparisons with
NaN are False, ``_get_mu`` raises UnboundLocalError on a NaN directional derivative (DGSQP.py:566-585) --
reported here as msg 'exception' (the Monte-Carlo scripts have no try/except around solve()).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse
import scipy.sparse.linalg

from . import oracle, osqp_restate


class PyRef:
    def __init__(self, P, par, qp='osqp', qp_opts=None):
        """P, par: the dgsqp_problem_t / dgsqp_params_t PODs (only N, tolerances etc. are read from par here)."""
        self.P, self.par = P, par
        self.d = oracle.dims(P)
        self.qp_kind = qp
        self.qp_opts = qp_opts or {}
        self.qp_log = []           # (status, iters, polished) of every OSQP call
        self.rel_tol_req = 3       # DGSQP.py:56

    # ---- _evaluate (DGSQP.py:509-533): derivatives from the C++ oracle --------------------------------
    def evaluate(self, u, l, hessian=True):
        ev = oracle.evaluate(self.P, self.x0, u, l if l is not None else np.zeros(self.d['nc']), 1 if hessian else 0)
        if hessian:
            return ev['Q'], ev['q'], ev['G'], ev['g']
        return ev['q'], ev['G'], ev['g']

    # ---- _nearestPD (DGSQP.py:1290-1296) ---------------------------------------------------------------
    @staticmethod
    def nearest_pd(A):
        B = (A + A.T) / 2
        s, U = np.linalg.eigh(B)
        s[np.where(s < 0)[0]] = 1e-10
        C = U @ np.diag(s) @ U.T
        return (C + C.T) / 2

    # ---- _solve_qp (DGSQP.py:232-266) ------------------------------------------------------------------
    def solve_qp(self, Q, q, G, g):
        Q = self.nearest_pd(Q)
        if self.par.reg > 0:
            Q = Q + self.par.reg * np.eye(Q.shape[0])
        if self.qp_kind == 'osqp':
            opts = dict(self.qp_opts)
            if getattr(self.par, 'osqp_rho_carry', 0):        # the previous call's adapted rho (CasADi's conic plugin keeps its OSQP workspace)
                opts['rho'] = self.osqp_rho
            du, lhat, info = osqp_restate.conic(Q, q, G, -g, **opts)
            if getattr(self.par, 'osqp_rho_carry', 0) and np.isfinite(info['rho']):
                self.osqp_rho = float(info['rho'])
            self.qp_log.append((info['status'], info['iters'], info['polished']))
            return du, lhat
        if self.qp_kind == 'gi':           # the C++ oracle's Goldfarb-Idnani (exact minimiser); infeasible -> NaN like OSQP
            du, lhat, flag = oracle.qp(Q, q, G, g)
            self.qp_log.append((1 if flag == 0 else -3, 0, 0))
            if flag != 0:
                return np.full_like(du, np.nan), np.full_like(lhat, np.nan)
            return du, lhat
        raise ValueError(self.qp_kind)

    # ---- merit function (DGSQP.py:949-979) -------------------------------------------------------------
    def f_phi(self, l, s, q, G, g, mu):
        stat = np.concatenate([q + G.T @ l, [l @ g]])
        phi = 0.5 * float(stat @ stat)
        if self.par.merit_function == 0:                       # 'stat_l1'
            phi += mu * float(np.sum(g - s))
        return phi

    @staticmethod
    def f_dstat_norm(du, l, dl, Q, q, G, g):
        d = q + G.T @ l
        return float(d @ (np.hstack([Q, G.T]) @ np.concatenate([du, dl])) + (l @ g) * (l @ (G @ du) + dl @ g))

    def f_dphi(self, du, l, dl, s, Q, q, G, g, mu):
        d = self.f_dstat_norm(du, l, dl, Q, q, G, g)
        if self.par.merit_function == 0:
            d += -mu * float(np.sum(g - s))
        return d

    # ---- _get_mu (DGSQP.py:559-585) --------------------------------------------------------------------
    def get_mu(self, du, l, dl, s, Q, q, G, g):
        thresh = 0
        if self.par.merit_function == 0:
            constr_vio = g - s
            d_stat_norm = self.f_dstat_norm(du, l, dl, Q, q, G, g)
            rho = 0.5
            if d_stat_norm < 0 and np.sum(constr_vio) > thresh:
                mu = -d_stat_norm / ((1 - rho) * np.sum(constr_vio))
            elif d_stat_norm < 0 and np.sum(constr_vio) <= thresh:
                mu = 0
            elif d_stat_norm >= 0 and np.sum(constr_vio) > thresh:
                mu = d_stat_norm / ((1 - rho) * np.sum(constr_vio))
            elif d_stat_norm >= 0 and np.sum(constr_vio) <= thresh:
                mu = 0
        else:
            mu = 0
        return mu          # UnboundLocalError when d_stat_norm is NaN, as in the reference

    # ---- _line_search_3 (DGSQP.py:1057-1081) -----------------------------------------------------------
    def line_search_3(self, u, du, l, dl, s, ds, Q, q, G, g, mu):
        phi = self.f_phi(l, s, q, G, g, mu)
        dphi = self.f_dphi(du, l, dl, s, Q, q, G, g, mu)
        alpha = 1.0
        for _ in range(self.par.line_search_iters):
            u_trial, l_trial, s_trial = u + alpha * du, l + alpha * dl, s + alpha * ds
            q_t, G_t, g_t = self.evaluate(u_trial, l_trial, False)
            phi_trial = self.f_phi(l_trial, s_trial, q_t, G_t, g_t, mu)
            self.tr(30, alpha)
            self.tr(31, phi_trial)
            if phi_trial <= phi + self.par.beta * alpha * dphi:
                break
            alpha *= self.par.tau
        return u_trial, l_trial, phi_trial

    # ---- _watchdog_line_search_4 (DGSQP.py:1174-1288) --------------------------------------------------
    def watchdog_4(self, u_k, du_k, l_k, dl_k, s_k, ds_k, Q_k, q_k, G_k, g_k, mu, merit_max=1e6):
        beta = self.par.beta
        qp_solves, t_hat = 0, 5
        phi_k = self.f_phi(l_k, s_k, q_k, G_k, g_k, mu)
        dphi_k = self.f_dphi(du_k, l_k, dl_k, s_k, Q_k, q_k, G_k, g_k, mu)
        u_kp1, l_kp1, s_kp1 = u_k + du_k, l_k + dl_k, s_k + ds_k
        q1, G1, g1 = self.evaluate(u_kp1, l_kp1, False)
        phi_kp1 = self.f_phi(l_kp1, s_kp1, q1, G1, g1, mu)
        self.tr(20, phi_kp1)
        if phi_kp1 <= phi_k + beta * dphi_k:
            return u_kp1, l_kp1, qp_solves
        fail = False
        u_t, l_t = u_kp1, l_kp1
        for _t in range(t_hat):
            Q_t, q_t, G_t, g_t = self.evaluate(u_t, l_t, True)
            du_t, l_hat = self.solve_qp(Q_t, q_t, G_t, g_t)
            qp_solves += 1
            dl_t = l_hat - l_t
            s_t = np.minimum(0, g_t)
            ds_t = g_t + G_t @ du_t - s_t
            u_tp1, l_tp1, s_tp1 = u_t + du_t, l_hat, s_t + ds_t
            q2, G2, g2 = self.evaluate(u_tp1, l_tp1, False)
            phi_tp1 = self.f_phi(l_tp1, s_tp1, q2, G2, g2, mu)
            self.tr(21, phi_tp1)
            if phi_tp1 > merit_max:
                break
            if phi_tp1 <= phi_k + beta * dphi_k:
                return u_tp1, l_tp1, qp_solves
            u_t, l_t = u_tp1, l_tp1
        Q_t, q_t, G_t, g_t = self.evaluate(u_t, l_t, True)
        du_t, l_hat = self.solve_qp(Q_t, q_t, G_t, g_t)
        qp_solves += 1
        dl_t = l_hat - l_t
        s_t = np.minimum(0, g_t)
        ds_t = g_t + G_t @ du_t - s_t
        u_tp1, l_tp1, phi_tp1 = self.line_search_3(u_t, du_t, l_t, dl_t, s_t, ds_t, Q_t, q_t, G_t, g_t, mu)
        self.tr(22, phi_tp1)
        if not fail:
            if phi_tp1 <= phi_k + beta * dphi_k:
                return u_tp1, l_tp1, qp_solves
            elif phi_tp1 > phi_k:
                fail = True
            else:
                Q2, q2, G2, g2 = self.evaluate(u_tp1, l_tp1, True)
                du2, l_hat = self.solve_qp(Q2, q2, G2, g2)
                qp_solves += 1
                dl2 = l_hat - l_tp1
                s2 = np.minimum(0, g2)
                ds2 = g2 + G2 @ du2 - s2
                u_tp2, l_tp2, _ = self.line_search_3(u_tp1, du2, l_tp1, dl2, s2, ds2, Q2, q2, G2, g2, mu)
                return u_tp2, l_tp2, qp_solves
        u_kp1, l_kp1, _ = self.line_search_3(u_k, du_k, l_k, dl_k, s_k, ds_k, Q_k, q_k, G_k, g_k, mu)
        return u_kp1, l_kp1, qp_solves

    def tr(self, code, v):
        if self.trace is not None:
            self.trace.append((code, float(v)))

    # ---- solve (DGSQP.py:302-507) ----------------------------------------------------------------------
    def solve(self, x0, u_ws, trace=False, lsqr_kw=None):
        """u_ws agent-major [n].  Returns dict(u, l, status(bool), msg, num_iters, qp_solves, cond, l_init)."""
        self.x0 = np.ascontiguousarray(x0, float)
        self.osqp_rho = 0.1
        self.trace = [] if trace else None
        par = self.par
        u = np.array(u_ws, float)
        q, G, _ = self.evaluate(u, None, False)
        Gs = scipy.sparse.csc_matrix(G)
        l = np.maximum(0, -scipy.sparse.linalg.lsqr(Gs @ Gs.T, Gs @ q, **(lsqr_kw or {}))[0])
        l_init = l.copy()
        rel_tol_its, sqp_it, total_qp = 0, 0, 0
        msg, converged = 'max_it', False
        cond = {}
        try:
            while True:
                qp_solves = 0
                Q_i, q_i, G_i, g_i = self.evaluate(u, l, True)
                d_i = q_i + G_i.T @ l
                u_im1, l_im1 = u.copy(), l.copy()
                xtol, ltol = par.p_tol, par.d_tol
                p_feas = max(0, np.amax(g_i))
                comp = np.linalg.norm(g_i * l, ord=np.inf)
                stat = np.linalg.norm(d_i, ord=np.inf)
                cond = {'p_feas': p_feas, 'comp': comp, 'stat': stat}
                self.tr(1, stat); self.tr(2, p_feas); self.tr(3, comp)
                if stat > 1e5:
                    msg = 'diverged'
                    break
                if p_feas < xtol and comp < ltol and stat < ltol:
                    converged, msg = True, 'conv_abs_tol'
                    break
                du, l_hat = self.solve_qp(Q_i, q_i, G_i, g_i)
                qp_solves += 1
                dl = l_hat - l
                s = np.minimum(0, g_i)
                ds = g_i + G_i @ du - s
                mu = self.get_mu(du, l, dl, s, Q_i, q_i, G_i, g_i)
                self.tr(10, du @ du); self.tr(11, mu)
                self.tr(12, self.f_phi(l, s, q_i, G_i, g_i, mu)); self.tr(13, self.f_dphi(du, l, dl, s, Q_i, q_i, G_i, g_i, mu))
                if par.nonmono_ls:
                    u, l, n_qp = self.watchdog_4(u, du, l, dl, s, ds, Q_i, q_i, G_i, g_i, mu)
                    qp_solves += n_qp
                else:
                    u, l, _ = self.line_search_3(u, du, l, dl, s, ds, Q_i, q_i, G_i, g_i, mu)
                total_qp += qp_solves
                self.tr(40, qp_solves)
                if np.linalg.norm(u - u_im1) < xtol / 2 and np.linalg.norm(l - l_im1) < ltol / 2:
                    rel_tol_its += 1
                    if rel_tol_its >= self.rel_tol_req and p_feas < xtol:
                        converged, msg = True, 'conv_rel_tol'
                        break
                else:
                    rel_tol_its = 0
                sqp_it += 1
                if sqp_it >= par.sqp_iters:
                    msg = 'max_it'
                    break
        except UnboundLocalError:
            total_qp += qp_solves
            msg = 'exception'        # NaN step from a failed QP reached _get_mu (DGSQP.py:566-585): solve() raises in the reference
        return dict(u=u, l=l, status=converged, msg=msg, num_iters=sqp_it, qp_solves=total_qp, cond=cond, l_init=l_init,
                    trace=np.array(self.trace) if trace else None)
