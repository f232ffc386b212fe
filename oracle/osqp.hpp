// =============================================================================
// ORACLE -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.
//
// Dense C++ restatement of OSQP as the reference calls it through CasADi:
//   ca.conic('qp', 'osqp', ..., {polish: True})   DGSQP/solvers/DGSQP.py:183-201
//   solver(h=Q, g=q, a=G, uba=-g, x0=0)            DGSQP/solvers/DGSQP.py:246-249
// OSQP itself is a third-party dependency of the reference (setup.py:15, no version
// pin; CasADi 3.5 / 3.6 bundle OSQP 0.6.x) that is absent from /root/reference and from
// this image.  This file follows oracle/osqp_restate.py statement by statement (that
// file restates the published algorithm: Stellato et al., Math. Prog. Comp. 12 (2020),
// Algorithm 1, sections 3.4, 4, 5.1, 5.2, with the OSQP 0.6 default settings); the
// two are held together by tests/test_oracle.py::test_cpp_osqp_follows_the_numpy_restatement.
// Same stated deviations: adaptive-rho interval fixed at 25 iterations (OSQP derives
// it from wall-clock time), every call starts from rho = 0.1, dense LU with partial
// pivoting instead of QDLDL.
// =============================================================================
#pragma once
#include <algorithm>
#include <cmath>
#include <limits>
#include <vector>

namespace osqp_restate {

typedef std::vector<double> vec;

static const double OSQP_INFTY = 1e30, MIN_SCALING = 1e-4, MAX_SCALING = 1e4;
static const double RHO_MIN = 1e-6, RHO_MAX = 1e6, RHO_TOL = 1e-4, RHO_EQ_OVER_RHO_INEQ = 1e3;
enum { SOLVED = 1, SOLVED_INACCURATE = 2, PRIMAL_INFEASIBLE_INACCURATE = 3, DUAL_INFEASIBLE_INACCURATE = 4, MAX_ITER = -2, PRIMAL_INFEASIBLE = -3, DUAL_INFEASIBLE = -4, NAN_DATA = -10 };

struct Settings {
  double rho = 0.1, sigma = 1e-6, alpha = 1.6, eps_abs = 1e-3, eps_rel = 1e-3, eps_prim_inf = 1e-4, eps_dual_inf = 1e-4;
  int max_iter = 4000, scaling = 10, adaptive_rho = 1, adaptive_rho_interval = 25, check_termination = 25, polish = 1, polish_refine_iter = 3;
  double adaptive_rho_tolerance = 5.0, delta = 1e-6;
};
struct Info {
  int status = MAX_ITER, iters = 0, polished = 0, rho_updates = 0, n_active = 0;
  double rho = 0.1, pri_res = 0, dua_res = 0;
};

// dense LU with partial pivoting (what scipy.linalg.lu_factor / lu_solve do through LAPACK getrf / getrs)
struct LU {
  int n = 0;
  vec a;
  std::vector<int> piv;
  bool singular = false;
  void factor(int n_, const vec& A) {
    n = n_; a = A; piv.assign(n, 0); singular = false;
    for (int k = 0; k < n; k++) {
      int p = k; double best = std::fabs(a[(size_t)k * n + k]);
      for (int i = k + 1; i < n; i++) { const double v = std::fabs(a[(size_t)i * n + k]); if (v > best) { best = v; p = i; } }
      piv[k] = p;
      if (!(best > 0.0)) { singular = true; continue; }
      if (p != k) for (int j = 0; j < n; j++) std::swap(a[(size_t)k * n + j], a[(size_t)p * n + j]);
      const double d = 1.0 / a[(size_t)k * n + k];
      for (int i = k + 1; i < n; i++) {
        const double m = a[(size_t)i * n + k] * d;
        if (m == 0.0) { a[(size_t)i * n + k] = 0.0; continue; }
        a[(size_t)i * n + k] = m;
        double* ri = &a[(size_t)i * n];
        const double* rk = &a[(size_t)k * n];
        for (int j = k + 1; j < n; j++) ri[j] -= m * rk[j];
      }
    }
  }
  vec solve(vec b) const {
    for (int k = 0; k < n; k++) if (piv[k] != k) std::swap(b[k], b[piv[k]]);
    for (int i = 0; i < n; i++) { double s = b[i]; const double* ri = &a[(size_t)i * n]; for (int j = 0; j < i; j++) s -= ri[j] * b[j]; b[i] = s; }
    for (int i = n - 1; i >= 0; i--) { double s = b[i]; const double* ri = &a[(size_t)i * n]; for (int j = i + 1; j < n; j++) s -= ri[j] * b[j]; b[i] = s / ri[i]; }
    return b;
  }
};

static inline double limit_scaling(double v) { v = v < MIN_SCALING ? 1.0 : v; return std::min(v, MAX_SCALING); }
static inline double inf_norm(const vec& v) { double m = 0; for (double e : v) m = std::max(m, std::fabs(e)); return m; }

// min 1/2 x'Px + q'x  s.t.  l <= Ax <= u  (P n x n, A m x n, row-major).  x[n], y[m] out (NaN when infeasible).
static Info solve(int n, int m, const double* Pin, const double* qin, const double* Ain, const double* lin, const double* uin,
                  double* xout, double* yout, const Settings& S = Settings()) {
  Info info;
  vec P((size_t)n * n), q(qin, qin + n), A(Ain, Ain + (size_t)m * n), l(m), u(m);
  for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) P[(size_t)i * n + j] = 0.5 * (Pin[(size_t)i * n + j] + Pin[(size_t)j * n + i]);   // OSQP holds the upper triangle only
  for (int i = 0; i < m; i++) { l[i] = std::max(lin[i], -OSQP_INFTY); u[i] = std::min(uin[i], OSQP_INFTY); }
  // ---- Ruiz equilibration (section 5.1, scale_data())
  vec D(n, 1.0), E(m, 1.0);
  double c = 1.0;
  for (int it = 0; it < S.scaling; it++) {
    vec dn(n, 0.0), en(m, 0.0);
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) dn[j] = std::max(dn[j], std::fabs(P[(size_t)i * n + j]));
    for (int i = 0; i < m; i++) for (int j = 0; j < n; j++) { const double v = std::fabs(A[(size_t)i * n + j]); dn[j] = std::max(dn[j], v); en[i] = std::max(en[i], v); }
    vec dt(n), et(m);
    for (int j = 0; j < n; j++) dt[j] = 1.0 / std::sqrt(limit_scaling(dn[j]));
    for (int i = 0; i < m; i++) et[i] = 1.0 / std::sqrt(limit_scaling(en[i]));
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) P[(size_t)i * n + j] = dt[i] * P[(size_t)i * n + j] * dt[j];
    for (int i = 0; i < m; i++) for (int j = 0; j < n; j++) A[(size_t)i * n + j] = et[i] * A[(size_t)i * n + j] * dt[j];
    for (int j = 0; j < n; j++) { q[j] = dt[j] * q[j]; D[j] *= dt[j]; }
    for (int i = 0; i < m; i++) E[i] *= et[i];
    double cm = 0;
    for (int j = 0; j < n; j++) { double mx = 0; for (int i = 0; i < n; i++) mx = std::max(mx, std::fabs(P[(size_t)i * n + j])); cm += mx; }
    double ct = limit_scaling(cm / n);
    double qn = inf_norm(q);
    qn = qn < MIN_SCALING ? 1.0 : std::min(qn, MAX_SCALING);
    ct = 1.0 / std::max(ct, qn);
    for (double& e : P) e *= ct;
    for (double& e : q) e *= ct;
    c *= ct;
  }
  vec ls(m), us(m), Dinv(n), Einv(m);
  for (int i = 0; i < m; i++) { ls[i] = E[i] * l[i]; us[i] = E[i] * u[i]; Einv[i] = 1.0 / E[i]; }
  for (int j = 0; j < n; j++) Dinv[j] = 1.0 / D[j];
  const double cinv = 1.0 / c;
  double rho = S.rho;
  vec rho_vec(m);
  auto make_rho_vec = [&](double r) {
    for (int i = 0; i < m; i++) {
      const bool loose = ls[i] < -OSQP_INFTY * MIN_SCALING && us[i] > OSQP_INFTY * MIN_SCALING;
      const bool eq = (us[i] - ls[i]) < RHO_TOL;
      rho_vec[i] = loose ? RHO_MIN : (eq ? RHO_EQ_OVER_RHO_INEQ * r : r);
    }
  };
  LU kkt;
  auto factor_kkt = [&]() {
    const int N = n + m;
    vec K((size_t)N * N, 0.0);
    for (int i = 0; i < n; i++) { for (int j = 0; j < n; j++) K[(size_t)i * N + j] = P[(size_t)i * n + j]; K[(size_t)i * N + i] += S.sigma; }
    for (int r = 0; r < m; r++) {
      for (int j = 0; j < n; j++) { K[(size_t)(n + r) * N + j] = A[(size_t)r * n + j]; K[(size_t)j * N + n + r] = A[(size_t)r * n + j]; }
      K[(size_t)(n + r) * N + n + r] = -1.0 / rho_vec[r];
    }
    kkt.factor(N, K);
  };
  make_rho_vec(rho);
  factor_kkt();
  vec x(n, 0.0), z(m, 0.0), y(m, 0.0), delta_x(n, 0.0), delta_y(m, 0.0);
  vec Ax(m), Px(n), Aty(n);
  auto products = [&]() {
    for (int r = 0; r < m; r++) { double s = 0; const double* a = &A[(size_t)r * n]; for (int j = 0; j < n; j++) s += a[j] * x[j]; Ax[r] = s; }
    for (int i = 0; i < n; i++) { double s = 0; const double* p = &P[(size_t)i * n]; for (int j = 0; j < n; j++) s += p[j] * x[j]; Px[i] = s; }
    std::fill(Aty.begin(), Aty.end(), 0.0);
    for (int r = 0; r < m; r++) { const double yr = y[r]; if (yr == 0.0) continue; const double* a = &A[(size_t)r * n]; for (int j = 0; j < n; j++) Aty[j] += a[j] * yr; }
  };
  double pri_res = INFINITY, dua_res = INFINITY, eps_p = 0, eps_d = 0;
  auto residuals = [&]() {
    products();
    double pr = 0, nz = 0, nax = 0;
    for (int r = 0; r < m; r++) { pr = std::max(pr, std::fabs(Einv[r] * (Ax[r] - z[r]))); nz = std::max(nz, std::fabs(Einv[r] * z[r])); nax = std::max(nax, std::fabs(Einv[r] * Ax[r])); }
    double dr = 0, nq = 0, naty = 0, npx = 0;
    for (int j = 0; j < n; j++) {
      dr = std::max(dr, std::fabs(Dinv[j] * (Px[j] + q[j] + Aty[j])));
      nq = std::max(nq, std::fabs(Dinv[j] * q[j])); naty = std::max(naty, std::fabs(Dinv[j] * Aty[j])); npx = std::max(npx, std::fabs(Dinv[j] * Px[j]));
    }
    pri_res = m ? pr : 0.0;
    dua_res = cinv * dr;
    eps_p = m ? S.eps_abs + S.eps_rel * std::max(nz, nax) : S.eps_abs;
    eps_d = S.eps_abs + S.eps_rel * cinv * std::max(nq, std::max(naty, npx));
  };
  auto primal_infeasible = [&](double eps) {
    vec dy(m);
    double nrm = 0, lhs = 0;
    for (int r = 0; r < m; r++) {
      const bool inf_u = us[r] > OSQP_INFTY * MIN_SCALING, inf_l = ls[r] < -OSQP_INFTY * MIN_SCALING;
      double v = delta_y[r];
      v = (inf_u && inf_l) ? 0.0 : (inf_u ? std::min(v, 0.0) : (inf_l ? std::max(v, 0.0) : v));
      dy[r] = v;
      nrm = std::max(nrm, std::fabs(E[r] * v));
    }
    if (nrm <= 1.0 / OSQP_INFTY) return false;
    for (int r = 0; r < m; r++) {
      const bool inf_u = us[r] > OSQP_INFTY * MIN_SCALING, inf_l = ls[r] < -OSQP_INFTY * MIN_SCALING;
      if (!inf_u) lhs += us[r] * std::max(dy[r], 0.0);
      if (!inf_l) lhs += ls[r] * std::min(dy[r], 0.0);
    }
    if (lhs < -eps * nrm) {
      double mx = 0;
      for (int j = 0; j < n; j++) { double s = 0; for (int r = 0; r < m; r++) s += A[(size_t)r * n + j] * dy[r]; mx = std::max(mx, std::fabs(Dinv[j] * s)); }
      return mx < eps * nrm;
    }
    return false;
  };
  auto dual_infeasible = [&](double eps) {
    double nrm = 0;
    for (int j = 0; j < n; j++) nrm = std::max(nrm, std::fabs(D[j] * delta_x[j]));
    if (nrm <= 1.0 / OSQP_INFTY) return false;
    double qdx = 0;
    for (int j = 0; j < n; j++) qdx += q[j] * delta_x[j];
    if (qdx < -c * eps * nrm) {
      double mx = 0;
      for (int i = 0; i < n; i++) { double s = 0; for (int j = 0; j < n; j++) s += P[(size_t)i * n + j] * delta_x[j]; mx = std::max(mx, std::fabs(Dinv[i] * s)); }
      if (mx < c * eps * nrm) {
        for (int r = 0; r < m; r++) {
          double s = 0; for (int j = 0; j < n; j++) s += A[(size_t)r * n + j] * delta_x[j];
          const double adx = Einv[r] * s;
          const bool ok_u = us[r] > OSQP_INFTY * MIN_SCALING || adx < eps * nrm;
          const bool ok_l = ls[r] < -OSQP_INFTY * MIN_SCALING || adx > -eps * nrm;
          if (!(ok_u && ok_l)) return false;
        }
        return true;
      }
    }
    return false;
  };

  int status = MAX_ITER, it = 0;
  bool stopped = false;
  for (it = 1; it <= S.max_iter; it++) {
    const vec x_prev = x, z_prev = z;
    vec rhs(n + m);
    for (int j = 0; j < n; j++) rhs[j] = S.sigma * x_prev[j] - q[j];
    for (int r = 0; r < m; r++) rhs[n + r] = z_prev[r] - y[r] / rho_vec[r];
    const vec sol = kkt.solve(rhs);
    for (int j = 0; j < n; j++) { x[j] = S.alpha * sol[j] + (1 - S.alpha) * x_prev[j]; delta_x[j] = x[j] - x_prev[j]; }
    for (int r = 0; r < m; r++) {
      const double zt = z_prev[r] + (sol[n + r] - y[r]) / rho_vec[r];
      const double zr = S.alpha * zt + (1 - S.alpha) * z_prev[r];
      z[r] = std::min(std::max(zr + y[r] / rho_vec[r], ls[r]), us[r]);
      delta_y[r] = rho_vec[r] * (zr - z[r]);
      y[r] = y[r] + delta_y[r];
    }
    const bool check = S.check_termination && it % S.check_termination == 0;
    const bool adapt = S.adaptive_rho && S.adaptive_rho_interval && it % S.adaptive_rho_interval == 0;
    if (check) {
      residuals();
      if (pri_res <= eps_p && dua_res <= eps_d) { status = SOLVED; stopped = true; break; }
      if (primal_infeasible(S.eps_prim_inf)) { status = PRIMAL_INFEASIBLE; stopped = true; break; }
      if (dual_infeasible(S.eps_dual_inf)) { status = DUAL_INFEASIBLE; stopped = true; break; }
    }
    if (adapt) {
      products();
      double pr = 0, nz = 0, nax = 0, dr = 0, nq = 0, naty = 0, npx = 0;
      for (int r = 0; r < m; r++) { pr = std::max(pr, std::fabs(Ax[r] - z[r])); nz = std::max(nz, std::fabs(z[r])); nax = std::max(nax, std::fabs(Ax[r])); }
      for (int j = 0; j < n; j++) { dr = std::max(dr, std::fabs(Px[j] + q[j] + Aty[j])); nq = std::max(nq, std::fabs(q[j])); naty = std::max(naty, std::fabs(Aty[j])); npx = std::max(npx, std::fabs(Px[j])); }
      pr = m ? pr / (std::max(nz, nax) + 1e-10) : 0.0;
      dr = dr / (std::max(nq, std::max(naty, npx)) + 1e-10);
      const double rho_new = std::min(std::max(rho * std::sqrt(pr / (dr + 1e-10)), RHO_MIN), RHO_MAX);
      if (rho_new > rho * S.adaptive_rho_tolerance || rho_new < rho / S.adaptive_rho_tolerance) {
        rho = rho_new;
        make_rho_vec(rho);
        factor_kkt();
        info.rho_updates++;
      }
    }
  }
  if (!stopped) {
    it = S.max_iter;
    residuals();
    // OSQP re-checks with 10x tolerances at the iteration limit ("inaccurate" statuses)
    // (check_termination(work, approximate = 1): residual tolerances AND both infeasibility tolerances times ten, same order)
    if (pri_res <= 10 * eps_p && dua_res <= 10 * eps_d) status = SOLVED_INACCURATE;
    else if (primal_infeasible(10 * S.eps_prim_inf)) status = PRIMAL_INFEASIBLE_INACCURATE;
    else if (dual_infeasible(10 * S.eps_dual_inf)) status = DUAL_INFEASIBLE_INACCURATE;
    else status = MAX_ITER;
  }

  int polished = 0;
  if (status == SOLVED && S.polish) {
    residuals();
    std::vector<int> act;
    vec rhs_b;
    for (int r = 0; r < m; r++) {
      const bool low = (z[r] - ls[r]) < -y[r], upp = (us[r] - z[r]) < y[r];
      if (low || upp) { act.push_back(r); rhs_b.push_back(low ? ls[r] : us[r]); }
    }
    const int na = (int)act.size(), N = n + na;
    info.n_active = na;
    vec K((size_t)N * N, 0.0);
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) K[(size_t)i * N + j] = P[(size_t)i * n + j];
    for (int k = 0; k < na; k++) for (int j = 0; j < n; j++) { K[(size_t)(n + k) * N + j] = A[(size_t)act[k] * n + j]; K[(size_t)j * N + n + k] = A[(size_t)act[k] * n + j]; }
    vec Kreg = K;
    for (int i = 0; i < n; i++) Kreg[(size_t)i * N + i] += S.delta;
    for (int k = 0; k < na; k++) Kreg[(size_t)(n + k) * N + n + k] -= S.delta;
    vec rhs(N);
    for (int j = 0; j < n; j++) rhs[j] = -q[j];
    for (int k = 0; k < na; k++) rhs[n + k] = rhs_b[k];
    LU lu;
    lu.factor(N, Kreg);
    if (lu.singular) polished = -1;
    else {
      vec sol = lu.solve(rhs);
      for (int rf = 0; rf < S.polish_refine_iter; rf++) {
        vec res(N);
        for (int i = 0; i < N; i++) { double s = rhs[i]; const double* ki = &K[(size_t)i * N]; for (int j = 0; j < N; j++) s -= ki[j] * sol[j]; res[i] = s; }
        const vec dsol = lu.solve(res);
        for (int i = 0; i < N; i++) sol[i] += dsol[i];
      }
      bool finite = true;
      for (double e : sol) finite = finite && std::isfinite(e);
      vec xp(sol.begin(), sol.begin() + n), yp(m, 0.0), zp(m);
      for (int k = 0; k < na; k++) yp[act[k]] = sol[n + k];
      double pr_p = 0, dr_p = 0;
      for (int r = 0; r < m; r++) {
        double s = 0; const double* a = &A[(size_t)r * n]; for (int j = 0; j < n; j++) s += a[j] * xp[j];
        zp[r] = s;
        pr_p = std::max(pr_p, std::fabs(Einv[r] * (s - std::min(std::max(s, ls[r]), us[r]))));
      }
      if (!m) pr_p = 0.0;
      for (int j = 0; j < n; j++) {
        double s = q[j];
        for (int i = 0; i < n; i++) s += P[(size_t)j * n + i] * xp[i];
        for (int k = 0; k < na; k++) s += A[(size_t)act[k] * n + j] * sol[n + k];
        dr_p = std::max(dr_p, std::fabs(Dinv[j] * s));
      }
      dr_p *= cinv;
      const bool ok = (pr_p < pri_res && dr_p < dua_res) || (pr_p < pri_res && dua_res < 1e-10) || (dr_p < dua_res && pri_res < 1e-10);
      if (ok && finite) { x = xp; y = yp; z = zp; polished = 1; }
      else polished = -1;
    }
  }
  const double qnan = std::numeric_limits<double>::quiet_NaN();
  if (status == PRIMAL_INFEASIBLE || status == DUAL_INFEASIBLE || status == PRIMAL_INFEASIBLE_INACCURATE || status == DUAL_INFEASIBLE_INACCURATE) {
    for (int j = 0; j < n; j++) xout[j] = qnan;         // OSQP stores NaN when there is no solution
    for (int r = 0; r < m; r++) yout[r] = qnan;
  } else {
    for (int j = 0; j < n; j++) xout[j] = D[j] * x[j];
    for (int r = 0; r < m; r++) yout[r] = cinv * E[r] * y[r];
  }
  info.status = status; info.iters = it; info.polished = polished; info.rho = rho; info.pri_res = pri_res; info.dua_res = dua_res;
  return info;
}

// The call of DGSQP.py:246 -- solver(h=Q, g=q, a=G, uba=-g, x0=0) -- as CasADi's OSQP plugin poses it: identity rows for the
// (absent) variable bounds above the `a` rows.  x[n], lam[nc] out.
static Info conic(int n, int nc, const double* H, const double* g, const double* A, const double* uba, double* x, double* lam,
                  const Settings& S = Settings()) {
  bool finite = true;
  for (size_t i = 0; i < (size_t)n * n && finite; i++) finite = std::isfinite(H[i]);
  for (int i = 0; i < n && finite; i++) finite = std::isfinite(g[i]);
  for (size_t i = 0; i < (size_t)nc * n && finite; i++) finite = std::isfinite(A[i]);
  for (int i = 0; i < nc && finite; i++) finite = !std::isnan(uba[i]);
  const double qnan = std::numeric_limits<double>::quiet_NaN();
  if (!finite) {      // NaN data: no defined result
    for (int j = 0; j < n; j++) x[j] = qnan;
    for (int r = 0; r < nc; r++) lam[r] = qnan;
    Info info; info.status = NAN_DATA; info.rho = qnan;
    return info;
  }
  const int m = n + nc;
  vec Af((size_t)m * n, 0.0), l(m, -INFINITY), u(m, INFINITY), y(m);
  for (int j = 0; j < n; j++) Af[(size_t)j * n + j] = 1.0;
  std::copy(A, A + (size_t)nc * n, Af.begin() + (size_t)n * n);
  for (int r = 0; r < nc; r++) u[n + r] = uba[r];
  const Info info = solve(n, m, H, g, Af.data(), l.data(), u.data(), x, y.data(), S);
  for (int r = 0; r < nc; r++) lam[r] = y[n + r];
  return info;
}

}  // namespace osqp_restate
