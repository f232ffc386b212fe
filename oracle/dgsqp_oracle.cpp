// =============================================================================
// ORACLE -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.
//
// CPU (fp64) restatement of the reference hot path DGSQP.solve()
// (/root/reference/DGSQP/solvers/DGSQP.py:302-507) and everything it calls.
// Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may
// load this library; the product (dgsqp_amd/) never does.
//
// "Parity unpinned": the reference has no tests / golden vectors, and its
// arithmetic lives in CasADi + OSQP which are absent from this image and from
// /root/reference (setup.py:10-16, un-pinned), so the reference cannot run
// here.  What IS pinned (tests/test_oracle_*.py): LSQR against the installed
// scipy 1.15.3 `scipy.sparse.linalg.lsqr` (the exact routine DGSQP.py:324
// calls), `_nearestPD` against numpy.linalg.eigh (DGSQP.py:1290-1296), all
// derivatives against finite differences, the QP against KKT conditions and
// scipy.optimize, track tables against SURVEY.md Appendix B numbers.
//
// Third-party algorithms restated here:
//  * CasADi symbolic AD  -> second-order forward jets (jet.hpp)
//  * scipy.sparse.linalg.lsqr (Paige & Saunders 1982; scipy 1.15.3)
//  * numpy.linalg.eigh   -> cyclic Jacobi eigenvalue iteration
//  * OSQP(polish=True) through ca.conic (DGSQP.py:186,200,246): the polished
//    OSQP answer is the exact KKT point of the strictly convex QP; it is
//    computed here by the Goldfarb-Idnani dual active-set method (1983).
//    Where OSQP's polish fails the reference output is not reproducible
//    (time-based adaptive rho), see DESIGN.md.
//
// Everything is dense and literal on purpose: Du_x, G, Q are formed exactly
// the way _build_solver (DGSQP.py:587-979) writes them down.
// =============================================================================
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <chrono>
#include <cstring>
#include <limits>
#include <thread>
#include <atomic>
#include <vector>

#include "../include/dgsqp.h"
#include "jet.hpp"
#include "osqp.hpp"

thread_local int Jet::nv = 0;

using std::vector;
typedef vector<double> vec;
static const double INF = std::numeric_limits<double>::infinity();

// -----------------------------------------------------------------------------
// layout (DGSQP.py:150-170, :729-821)
// -----------------------------------------------------------------------------
enum RowType { R_OBS = 0, R_RATE_UB, R_RATE_LB, R_IN_UB, R_IN_LB, R_ST_UB, R_ST_LB, R_LANE };
struct Row {
  int type, k, a, b, idx;
};
struct Layout {
  int M, N, nq, nu, n, nc;
  int nqa[DGSQP_MAX_AGENTS], qoff[DGSQP_MAX_AGENTS], s_idx[DGSQP_MAX_AGENTS], ey_idx[DGSQP_MAX_AGENTS];
  vector<Row> rows;
  vector<int> stage_row0;  // N+2 entries
  // agent-major index of time-major input (k, joint input j)   (DGSQP.py:170 ua_idxs)
  int am(int k, int ju) const { int a = ju / DGSQP_NUA, j = ju % DGSQP_NUA; return a * N * DGSQP_NUA + k * DGSQP_NUA + j; }
  int col(int a, int k, int j) const { return a * N * DGSQP_NUA + k * DGSQP_NUA + j; }
};

static int model_nq(int model) { return model == DGSQP_MODEL_DYN_BICYCLE ? 8 : (model == DGSQP_MODEL_UNICYCLE ? 4 : 6); }

static Layout make_layout(const dgsqp_problem_t& P) {
  Layout L;
  L.M = P.M; L.N = P.N; L.nq = 0; L.nu = P.M * DGSQP_NUA;
  for (int a = 0; a < P.M; a++) {
    L.nqa[a] = model_nq(P.agents[a].model);
    L.qoff[a] = L.nq;
    L.nq += L.nqa[a];
    // Frenet states s, e_y: the last two of both bicycles (the unicycle has none; its slots only meet zero weights)
    L.s_idx[a] = L.nqa[a] - 2;
    L.ey_idx[a] = L.nqa[a] - 1;
  }
  L.n = L.N * L.nu;
  // row order per stage: [shared ; agent0: fn rows, input ub, input lb, (k>0) state ub, state lb ; agent1 ...]
  // terminal: [shared ; per agent: state ub, state lb]          (DGSQP.py:732-821)
  for (int k = 0; k <= P.N; k++) {
    L.stage_row0.push_back((int)L.rows.size());
    if (P.obstacle_rows && k >= 1)  // shared constraint is None at k=0 (chicane.py:325-330)
      for (int i = 0; i < P.M; i++)
        for (int j = i + 1; j < P.M; j++) L.rows.push_back({R_OBS, k, i, j, 0});
    for (int a = 0; a < P.M; a++) {
      const dgsqp_agent_t& ag = P.agents[a];
      if (k < P.N) {
        if (ag.has_rate)
          for (int j = 0; j < DGSQP_NUA; j++) {  // chicane.py:282-285 order
            L.rows.push_back({R_RATE_UB, k, a, -1, j});
            L.rows.push_back({R_RATE_LB, k, a, -1, j});
          }
      }
      // agent constraint function rows of the merge game: lane half-planes at every stage incl. k = 0 and k = N
      // (merge.py:316-342; rows of stage 0 depend on x_0 only and keep a zero gradient, SURVEY.md hazard 8)
      for (int j = 0; j < ag.n_lane; j++) L.rows.push_back({R_LANE, k, a, -1, j});
      if (k < P.N) {
        for (int j = 0; j < DGSQP_NUA; j++)
          if (ag.in_ub[j] < INF) L.rows.push_back({R_IN_UB, k, a, -1, j});
        for (int j = 0; j < DGSQP_NUA; j++)
          if (ag.in_lb[j] > -INF) L.rows.push_back({R_IN_LB, k, a, -1, j});
      }
      if (k > 0) {
        for (int i = 0; i < L.nqa[a]; i++)
          if (ag.st_ub[i] < INF) L.rows.push_back({R_ST_UB, k, a, -1, i});
        for (int i = 0; i < L.nqa[a]; i++)
          if (ag.st_lb[i] > -INF) L.rows.push_back({R_ST_LB, k, a, -1, i});
      }
    }
  }
  L.stage_row0.push_back((int)L.rows.size());
  L.nc = (int)L.rows.size();
  return L;
}

// -----------------------------------------------------------------------------
// track functions (radius_arclength_track.py:199-225; CasADi pw_const/pw_lin)
// -----------------------------------------------------------------------------
// CasadiBSplineTrack (casadi_bspline_track.py:122-149): curvature and tangent of the cubic-spline centre line
template <class T>
static void track_eval_spline(const dgsqp_problem_t& P, const T& s, T& curv, T& psi_t) {
  using std::sqrt; using std::atan2;
  const double L = P.track_L, sv = val(s);
  const double sbar = std::fmod(std::fmod(sv, L) + L, L);
  const int nk = P.n_knots;
  const double* kn = P.spline;
  int i = (int)(std::upper_bound(kn, kn + nk, sbar) - kn) - 1;
  i = std::max(0, std::min(nk - 2, i));
  const double* cx = P.spline + nk + 4 * (size_t)i;
  const double* cy = P.spline + nk + 4 * (size_t)(nk - 1) + 4 * (size_t)i;
  T t = s + (sbar - sv - kn[i]);                       // d sbar / d s = 1 (fmod)
  T dx = (3.0 * cx[3] * t + 2.0 * cx[2]) * t + cx[1], dy = (3.0 * cy[3] * t + 2.0 * cy[2]) * t + cy[1];
  T ddx = 6.0 * cx[3] * t + 2.0 * cx[2], ddy = 6.0 * cy[3] * t + 2.0 * cy[2];
  T n2 = dx * dx + dy * dy;
  curv = (dx * ddy - dy * ddx) / (n2 * sqrt(n2));
  psi_t = atan2(dy, dx);
}
template <class T>
static void track_eval(const dgsqp_problem_t& P, const T& s, T& curv, T& psi_t) {
  if (P.track_kind == DGSQP_TRACK_SPLINE) { track_eval_spline(P, s, curv, psi_t); return; }
  const double L = P.track_L;
  const double sv = val(s);
  const double sbar = std::fmod(std::fmod(sv, L) + L, L);
  const int ns = P.n_segs;
  // pw_const(sbar, key_pts[1:-1,3], key_pts[1:,5]) = v0 + sum (v_{i+1}-v_i)*(t>=t_i)
  double c = P.seg_curv[0];
  for (int i = 0; i + 1 < ns; i++) c += (P.seg_curv[i + 1] - P.seg_curv[i]) * (sbar >= P.seg_s[i + 1] ? 1.0 : 0.0);
  curv = T(c);
  // pw_lin(sbar, key_pts[:,3], abs_angs); d sbar / d s = 1 (fmod)
  T sb = s + (sbar - sv);
  auto lseg = [&](int i) -> T {
    double gi = (P.seg_ang[i + 1] - P.seg_ang[i]) / (P.seg_s[i + 1] - P.seg_s[i]);
    return P.seg_ang[i] + gi * (sb - P.seg_s[i]);
  };
  T ret = lseg(0);
  for (int i = 0; i + 1 < ns; i++)
    if (sbar >= P.seg_s[i + 1]) ret = ret + (lseg(i + 1) - lseg(i));
  psi_t = ret;
}

template <class T> static T ca_abs(const T& x) { return val(x) > 0 ? x : -x; }          // dynamics_models.py:228-234
template <class T> static T ca_sign(const T& x) { using std::sqrt; return x / sqrt(x * x + 1e-6); }  // :236-238 (eps=1e-3)

// continuous-time kinematic bicycle, Frenet-combined (dynamics_models.py:1046-1070)
template <class T>
static void fc_kin(const dgsqp_problem_t& P, const dgsqp_agent_t& ag, const T* q, const T* u, T* dq) {
  using std::sin; using std::cos; using std::tan; using std::atan2;
  const T &v = q[2], &epsi = q[3], &s = q[4], &ey = q[5];
  const T &ua = u[0], &us = u[1];
  T beta = atan2(tan(us) * ag.L_r, T(ag.L_f + ag.L_r));
  T psidot = v / ag.L_r * sin(beta);
  T F_ext = -ag.c_da * v - ag.c_dr * v * ca_abs(v) - ag.c_s * (psidot * psidot);
  if (ag.c_r != 0.0) F_ext = F_ext - ag.c_r * powc(ca_abs(v), ag.p_r) * ca_sign(v);
  T c; T psi_t;
  track_eval(P, s, c, psi_t);
  T den = 1.0 - ey * c;
  dq[0] = v * cos(beta + psi_t + epsi);
  dq[1] = v * sin(beta + psi_t + epsi);
  dq[2] = ua + F_ext / ag.mass;
  dq[3] = psidot - c * v * cos(beta + epsi) / den;
  dq[4] = v * cos(beta + epsi) / den;
  dq[5] = v * sin(beta + epsi);
}

// continuous-time dynamic bicycle, Frenet-combined, Pacejka/linear tyres (dynamics_models.py:2008-2062)
template <class T>
static void fc_dyn(const dgsqp_problem_t& P, const dgsqp_agent_t& ag, const T* q, const T* u, T* dq) {
  using std::sin; using std::cos; using std::atan; using std::atan2;
  const T &vx = q[2], &vy = q[3], &w = q[4], &epsi = q[5], &s = q[6], &ey = q[7];
  const T &ua = u[0], &us = u[1];
  T c; T psi_t;
  track_eval(P, s, c, psi_t);
  T alpha_f, alpha_r;
  if (ag.simple_slip)
    alpha_f = -atan2(vy + ag.L_f * w, vx) + us;
  else
    alpha_f = -atan2((vy + ag.L_f * w) * cos(us) - vx * sin(us), vx * cos(us) + (vy + ag.L_f * w) * sin(us));
  alpha_r = -atan2(vy - ag.L_r * w, vx);
  T fyf, fyr;
  if (ag.tire_model == 0) {
    fyf = ag.pac_Df * sin(ag.pac_Cf * atan(ag.pac_Bf * alpha_f));
    fyr = ag.pac_Dr * sin(ag.pac_Cr * atan(ag.pac_Br * alpha_r));
  } else {
    fyf = (ag.lin_Bf * ag.mass * ag.gravity * ag.L_r / (ag.L_f + ag.L_r)) * alpha_f;
    fyr = (ag.lin_Br * ag.mass * ag.gravity * ag.L_f / (ag.L_f + ag.L_r)) * alpha_r;
  }
  T F_ext = -ag.c_da * vx - ag.c_dr * vx * ca_abs(vx);
  if (ag.c_r != 0.0) F_ext = F_ext - ag.c_r * powc(ca_abs(vx), ag.p_r) * ca_sign(vx);
  T ar, af;
  if (ag.drive_wheels == 0) { ar = ua / 2.0; af = ua / 2.0; } else { ar = ua; af = T(0.0); }
  T ax = ar + af * cos(us) + (F_ext - fyf * sin(us)) / ag.mass;
  T ay = af * sin(us) + (fyf * cos(us) + fyr) / ag.mass;
  T alphaz = (ag.L_f * fyf * cos(us) - ag.L_r * fyr) / ag.I_z;
  T den = 1.0 - ey * c;
  T vlon = vx * cos(epsi) - vy * sin(epsi);
  dq[0] = vx * cos(epsi + psi_t) - vy * sin(epsi + psi_t);
  dq[1] = vy * cos(epsi + psi_t) + vx * sin(epsi + psi_t);
  dq[2] = ax + w * vy;
  dq[3] = ay - w * vx;
  dq[4] = alphaz;
  dq[5] = w - c * vlon / den;
  dq[6] = vlon / den;
  dq[7] = vx * sin(epsi) + vy * cos(epsi);
}

// kinematic unicycle in the global frame (dynamics_models.py:331-339): state [x, y, v, psi], input [F, omega]
template <class T>
static void fc_uni(const dgsqp_agent_t& ag, const T* q, const T* u, T* dq) {
  using std::sin; using std::cos;
  dq[0] = q[2] * cos(q[3]);
  dq[1] = q[2] * sin(q[3]);
  dq[2] = u[0] / ag.mass;
  dq[3] = u[1];
}

template <class T>
static void fc(const dgsqp_problem_t& P, const dgsqp_agent_t& ag, const T* q, const T* u, T* dq) {
  if (ag.model == DGSQP_MODEL_DYN_BICYCLE) fc_dyn(P, ag, q, u, dq);
  else if (ag.model == DGSQP_MODEL_UNICYCLE) fc_uni(ag, q, u, dq);
  else fc_kin(P, ag, q, u, dq);
}

// discretisation of the JOINT model's config applied per (decoupled) agent
// (dynamics_models.py:88-99 euler, :188-219 rk4/rk3/rk2, :2521-2528 joint = concat of agents)
template <class T>
static void fd(const dgsqp_problem_t& P, const dgsqp_agent_t& ag, int nqa, const T* q, const T* u, T* qn) {
  T x[DGSQP_MAX_NQA], a1[DGSQP_MAX_NQA], a2[DGSQP_MAX_NQA], a3[DGSQP_MAX_NQA], a4[DGSQP_MAX_NQA], t[DGSQP_MAX_NQA];
  for (int i = 0; i < nqa; i++) x[i] = q[i];
  if (P.integrator == DGSQP_INT_EULER) {
    fc(P, ag, x, u, a1);
    for (int i = 0; i < nqa; i++) qn[i] = x[i] + P.dt * a1[i];
    return;
  }
  const int Ms = P.substeps;
  const double h = P.dt / Ms;
  for (int m = 0; m < Ms; m++) {
    if (P.integrator == DGSQP_INT_RK4) {
      fc(P, ag, x, u, a1);
      for (int i = 0; i < nqa; i++) t[i] = x[i] + (h / 2) * a1[i];
      fc(P, ag, t, u, a2);
      for (int i = 0; i < nqa; i++) t[i] = x[i] + (h / 2) * a2[i];
      fc(P, ag, t, u, a3);
      for (int i = 0; i < nqa; i++) t[i] = x[i] + h * a3[i];
      fc(P, ag, t, u, a4);
      for (int i = 0; i < nqa; i++) x[i] = x[i] + h * (a1[i] + 2.0 * a2[i] + 2.0 * a3[i] + a4[i]) / 6.0;
    } else if (P.integrator == DGSQP_INT_RK3) {
      fc(P, ag, x, u, a1);
      for (int i = 0; i < nqa; i++) { a1[i] = h * a1[i]; t[i] = x[i] + a1[i] / 2.0; }
      fc(P, ag, t, u, a2);
      for (int i = 0; i < nqa; i++) { a2[i] = h * a2[i]; t[i] = x[i] - a1[i] + 2.0 * a2[i]; }
      fc(P, ag, t, u, a3);
      for (int i = 0; i < nqa; i++) { a3[i] = h * a3[i]; x[i] = x[i] + (a1[i] + 4.0 * a2[i] + a3[i]) / 6.0; }
    } else {  // rk2 (Heun)
      fc(P, ag, x, u, a1);
      for (int i = 0; i < nqa; i++) t[i] = x[i] + h * a1[i];
      fc(P, ag, t, u, a2);
      for (int i = 0; i < nqa; i++) x[i] = x[i] + h * (a1[i] + a2[i]) / 2.0;
    }
  }
  for (int i = 0; i < nqa; i++) qn[i] = x[i];
}

// -----------------------------------------------------------------------------
// state-dependent part of agent a's cost at one stage (chicane.py:239-256,
// ablation.py:229-262, agents.py:179-184, exact_dynamic_game_dynamic.py:146-147)
// -----------------------------------------------------------------------------
template <class T>
static T state_cost(const dgsqp_problem_t& P, const Layout& L, int a, const T* x, bool terminal) {
  using std::sqrt; using std::atan;
  const dgsqp_agent_t& ag = P.agents[a];
  T J(0.0);
  // goal tracking 1/2 (q - goal)^T diag(w) (q - goal), terminal = mult x stage (merge.py:253-261)
  for (int i = 0; i < L.nqa[a]; i++)
    if (ag.w_goal[i] != 0.0) {
      T d = x[L.qoff[a] + i] - ag.goal[i];
      J = J + (terminal ? ag.goal_term_mult : 1.0) * 0.5 * ag.w_goal[i] * (d * d);
    }
  for (int b = 0; b < P.M; b++) {
    if (b == a) continue;
    if (ag.w_block != 0.0) {
      T d = x[L.qoff[a] + L.ey_idx[a]] - x[L.qoff[b] + L.ey_idx[b]];
      J = J + 0.5 * ag.w_block * (d * d);
    }
    if (ag.w_obs != 0.0) {
      T dx = x[L.qoff[a]] - x[L.qoff[b]], dy = x[L.qoff[a] + 1] - x[L.qoff[b] + 1];
      T z = (ag.obs_cost_r + P.agents[b].obs_cost_r) - sqrt(dx * dx + dy * dy);
      if (val(z) > 0) J = J + 0.5 * ag.w_obs * (z * z);  // fmax(0,z)^2
    }
  }
  if (terminal) {
    const T& sa = x[L.qoff[a] + L.s_idx[a]];
    J = J - ag.w_prog * sa;
    for (int b = 0; b < P.M; b++) {
      if (b == a) continue;
      T d = x[L.qoff[b] + L.s_idx[b]] - sa;
      if (ag.comp_type == DGSQP_COMP_ATAN) J = J + ag.w_comp * atan(d); else J = J + ag.w_comp * d;
    }
  }
  return J;
}

// -----------------------------------------------------------------------------
// _evaluate (DGSQP.py:509-533)
// -----------------------------------------------------------------------------
struct Eval {
  vec x;            // (N+1)*nq
  vec A, B;         // N*nq*nq, N*nq*nu (joint, block-diagonal)
  vec E, F, Gd;     // N*nq*(nq*nq), N*nq*(nu*nu), N*nq*(nu*nq)
  vec Dux;          // (N+1)nq x n, agent-major columns (f_Du_x, DGSQP.py:642-650)
  vec g, G, q, Q;   // n_c, n_c x n, n, n x n
};

static void rollout(const dgsqp_problem_t& P, const Layout& L, const double* u, const double* x0, vec& x) {
  // evaluate_dynamics (DGSQP.py:597-601)
  x.assign((size_t)(L.N + 1) * L.nq, 0.0);
  for (int i = 0; i < L.nq; i++) x[i] = x0[i];
  for (int k = 0; k < L.N; k++)
    for (int a = 0; a < L.M; a++) {
      double ua[DGSQP_NUA] = {u[L.col(a, k, 0)], u[L.col(a, k, 1)]};
      fd<double>(P, P.agents[a], L.nqa[a], &x[(size_t)k * L.nq + L.qoff[a]], ua, &x[(size_t)(k + 1) * L.nq + L.qoff[a]]);
    }
}

static void dyn_derivs(const dgsqp_problem_t& P, const Layout& L, const double* u, Eval& ev, bool hessian) {
  // evaluate_jacobian_A/B (DGSQP.py:606-612), evaluate_hessian_E/F/G (:620-628)
  const int nq = L.nq, nu = L.nu, N = L.N;
  ev.A.assign((size_t)N * nq * nq, 0.0);
  ev.B.assign((size_t)N * nq * nu, 0.0);
  if (hessian) {
    ev.E.assign((size_t)N * nq * nq * nq, 0.0);
    ev.F.assign((size_t)N * nq * nu * nu, 0.0);
    ev.Gd.assign((size_t)N * nq * nu * nq, 0.0);
  }
  for (int k = 0; k < N; k++)
    for (int a = 0; a < L.M; a++) {
      const int nqa = L.nqa[a], qo = L.qoff[a], uo = a * DGSQP_NUA;
      Jet::nv = nqa + DGSQP_NUA;
      Jet qj[DGSQP_MAX_NQA], uj[DGSQP_NUA], out[DGSQP_MAX_NQA];
      for (int i = 0; i < nqa; i++) qj[i] = Jet::var(ev.x[(size_t)k * nq + qo + i], i);
      for (int j = 0; j < DGSQP_NUA; j++) uj[j] = Jet::var(u[L.col(a, k, j)], nqa + j);
      fd<Jet>(P, P.agents[a], nqa, qj, uj, out);
      for (int i = 0; i < nqa; i++) {
        for (int j = 0; j < nqa; j++) ev.A[((size_t)k * nq + qo + i) * nq + qo + j] = out[i].g[j];
        for (int j = 0; j < DGSQP_NUA; j++) ev.B[((size_t)k * nq + qo + i) * nu + uo + j] = out[i].g[nqa + j];
        if (hessian) {
          double* Ei = &ev.E[((size_t)k * nq + qo + i) * nq * nq];
          double* Fi = &ev.F[((size_t)k * nq + qo + i) * nu * nu];
          double* Gi = &ev.Gd[((size_t)k * nq + qo + i) * nu * nq];
          for (int r = 0; r < nqa; r++)
            for (int c = 0; c < nqa; c++) Ei[(qo + r) * nq + qo + c] = out[i].H(r, c);
          for (int r = 0; r < DGSQP_NUA; r++)
            for (int c = 0; c < DGSQP_NUA; c++) Fi[(uo + r) * nu + uo + c] = out[i].H(nqa + r, nqa + c);
          for (int r = 0; r < DGSQP_NUA; r++)
            for (int c = 0; c < nqa; c++) Gi[(uo + r) * nq + qo + c] = out[i].H(nqa + r, c);
        }
      }
    }
}

static void build_Dux(const Layout& L, Eval& ev) {
  // f_Du_x (DGSQP.py:642-650): column block k = [0 (k+1 blocks); B_k; A_{k+1}B_k; ...], then agent-major permutation
  const int nq = L.nq, nu = L.nu, N = L.N, n = L.n;
  ev.Dux.assign((size_t)(N + 1) * nq * n, 0.0);
  vec cur(nq * nu), nxt(nq * nu);
  for (int k = 0; k < N; k++) {
    for (int i = 0; i < nq * nu; i++) cur[i] = ev.B[(size_t)k * nq * nu + i];
    for (int t = k + 1; t <= N; t++) {
      for (int i = 0; i < nq; i++)
        for (int j = 0; j < nu; j++) ev.Dux[((size_t)t * nq + i) * n + L.am(k, j)] = cur[i * nu + j];
      if (t < N) {
        const double* At = &ev.A[(size_t)t * nq * nq];
        for (int i = 0; i < nq; i++)
          for (int j = 0; j < nu; j++) {
            double s = 0;
            for (int l = 0; l < nq; l++) s += At[i * nq + l] * cur[l * nu + j];
            nxt[i * nu + j] = s;
          }
        cur.swap(nxt);
      }
    }
  }
}

// constraint values and first derivatives (f_Cxu DGSQP.py:729-821,911 ; f_Du_C :823-826,918)
static void constraints(const dgsqp_problem_t& P, const Layout& L, const double* u, Eval& ev, bool jac) {
  const int nq = L.nq, n = L.n;
  ev.g.assign(L.nc, 0.0);
  if (jac) ev.G.assign((size_t)L.nc * n, 0.0);
  for (int r = 0; r < L.nc; r++) {
    const Row& R = L.rows[r];
    const dgsqp_agent_t& ag = P.agents[R.a];
    double* Gr = jac ? &ev.G[(size_t)r * n] : nullptr;
    const double* xk = &ev.x[(size_t)R.k * nq];
    switch (R.type) {
      case R_OBS: {
        const int ia = L.qoff[R.a], ib = L.qoff[R.b];
        const double dx = xk[ia] - xk[ib], dy = xk[ia + 1] - xk[ib + 1];
        const double d = ag.radius + P.agents[R.b].radius;
        ev.g[r] = d * d - (dx * dx + dy * dy);  // chicane.py:292-293
        if (jac) {
          const double* Da = &ev.Dux[((size_t)R.k * nq + ia) * n];
          const double* Db = &ev.Dux[((size_t)R.k * nq + ib) * n];
          for (int c = 0; c < n; c++) Gr[c] = -2 * dx * (Da[c] - Db[c]) - 2 * dy * (Da[n + c] - Db[n + c]);
        }
      } break;
      case R_RATE_UB:
      case R_RATE_LB: {
        const double uk = u[L.col(R.a, R.k, R.idx)];
        const double um = R.k > 0 ? u[L.col(R.a, R.k - 1, R.idx)] : 0.0;  // up = 0 (DGSQP.py:305,308)
        if (R.type == R_RATE_UB) {
          ev.g[r] = (uk - um) - P.dt * ag.rate_ub[R.idx];
          if (jac) { Gr[L.col(R.a, R.k, R.idx)] += 1.0; if (R.k > 0) Gr[L.col(R.a, R.k - 1, R.idx)] -= 1.0; }
        } else {
          ev.g[r] = P.dt * ag.rate_lb[R.idx] - (uk - um);
          if (jac) { Gr[L.col(R.a, R.k, R.idx)] -= 1.0; if (R.k > 0) Gr[L.col(R.a, R.k - 1, R.idx)] += 1.0; }
        }
      } break;
      case R_IN_UB:
        ev.g[r] = u[L.col(R.a, R.k, R.idx)] - ag.in_ub[R.idx];
        if (jac) Gr[L.col(R.a, R.k, R.idx)] = 1.0;
        break;
      case R_IN_LB:
        ev.g[r] = ag.in_lb[R.idx] - u[L.col(R.a, R.k, R.idx)];
        if (jac) Gr[L.col(R.a, R.k, R.idx)] = -1.0;
        break;
      case R_LANE: {
        // n(p_x)^T (p - (anchor - r n(p_x))), n = pw_const(p_x, brk, [n_lo, n_hi]) (merge.py:66-74); d pw_const / d p_x = 0
        const auto& ln = ag.lane[R.idx];
        const int ia = L.qoff[R.a];
        const double hi = xk[ia] >= ln.brk ? 1.0 : 0.0;
        const double nx = ln.n_lo[0] + (ln.n_hi[0] - ln.n_lo[0]) * hi, ny = ln.n_lo[1] + (ln.n_hi[1] - ln.n_lo[1]) * hi;
        ev.g[r] = nx * (xk[ia] - (ln.anchor[0] - ln.r * nx)) + ny * (xk[ia + 1] - (ln.anchor[1] - ln.r * ny));
        if (jac) {
          const double* Dp = &ev.Dux[((size_t)R.k * nq + ia) * n];
          for (int c = 0; c < n; c++) Gr[c] = nx * Dp[c] + ny * Dp[n + c];
        }
      } break;
      case R_ST_UB:
      case R_ST_LB: {
        const int xi = L.qoff[R.a] + R.idx;
        const double sgn = R.type == R_ST_UB ? 1.0 : -1.0;
        ev.g[r] = R.type == R_ST_UB ? xk[xi] - ag.st_ub[R.idx] : ag.st_lb[R.idx] - xk[xi];
        if (jac) {
          const double* D = &ev.Dux[((size_t)R.k * nq + xi) * n];
          for (int c = 0; c < n; c++) Gr[c] = sgn * D[c];
        }
      } break;
    }
  }
}

// f_J (DGSQP.py:889-893): per-agent cost of an input sequence
static void costs(const dgsqp_problem_t& P, const Layout& L, const double* u, const vec& x, double* J) {
  for (int a = 0; a < L.M; a++) {
    const dgsqp_agent_t& ag = P.agents[a];
    double s = 0;
    for (int k = 0; k < L.N; k++) {
      for (int j = 0; j < DGSQP_NUA; j++) {
        const double uk = u[L.col(a, k, j)], um = k > 0 ? u[L.col(a, k - 1, j)] : 0.0;
        s += 0.5 * ag.w_in[j] * uk * uk + 0.5 * ag.w_rate[j] * (uk - um) * (uk - um);
      }
      s += state_cost<double>(P, L, a, &x[(size_t)k * L.nq], false);
    }
    s += state_cost<double>(P, L, a, &x[(size_t)L.N * L.nq], true);
    J[a] = s;
  }
}

// f_q (DGSQP.py:672-676, 898-899)
static void cost_gradient(const dgsqp_problem_t& P, const Layout& L, const double* u, Eval& ev) {
  const int nq = L.nq, n = L.n, N = L.N;
  ev.q.assign(n, 0.0);
  vec DxJ((size_t)(N + 1) * nq), DuJ(n);
  for (int a = 0; a < L.M; a++) {
    const dgsqp_agent_t& ag = P.agents[a];
    std::fill(DxJ.begin(), DxJ.end(), 0.0);
    std::fill(DuJ.begin(), DuJ.end(), 0.0);
    Jet::nv = nq;
    vector<Jet> xj(nq);
    for (int k = 0; k <= N; k++) {
      for (int i = 0; i < nq; i++) xj[i] = Jet::var(ev.x[(size_t)k * nq + i], i);
      Jet Jk = state_cost<Jet>(P, L, a, xj.data(), k == N);
      for (int i = 0; i < nq; i++) DxJ[(size_t)k * nq + i] = Jk.g[i];
    }
    for (int k = 0; k < N; k++)
      for (int j = 0; j < DGSQP_NUA; j++) {
        const double uk = u[L.col(a, k, j)], um = k > 0 ? u[L.col(a, k - 1, j)] : 0.0;
        double d = ag.w_in[j] * uk + ag.w_rate[j] * (uk - um);
        if (k + 1 < N) d -= ag.w_rate[j] * (u[L.col(a, k + 1, j)] - uk);
        DuJ[L.col(a, k, j)] = d;
      }
    // Du_J = (Du_Jxu + Dx_Jxu @ Du_x)^T, keep agent a's own rows
    for (int k = 0; k < N; k++)
      for (int j = 0; j < DGSQP_NUA; j++) {
        const int c = L.col(a, k, j);
        double s = DuJ[c];
        for (size_t r = 0; r < (size_t)(N + 1) * nq; r++) s += DxJ[r] * ev.Dux[r * n + c];
        ev.q[c] = s;
      }
  }
}

// DG-SQP v2's merit 'sum_obj_l1' (DGSQP_v2.py:1151-1152,1161-1164): obj = sum_a J^a(u) and dobj = Du(obj) . du with the gradient of
// EVERY agent's cost w.r.t. EVERY input (f_q above keeps only the agent's own rows): Du_J^a = Du_Jxu + Dx_Jxu Du_x, summed over a.
static void sum_obj_terms(const dgsqp_problem_t& P, const Layout& L, const double* u, const Eval& ev, double& obj, vec* grad) {
  const int nq = L.nq, n = L.n, N = L.N;
  double J[DGSQP_MAX_AGENTS];
  costs(P, L, u, ev.x, J);
  obj = 0;
  for (int a = 0; a < L.M; a++) obj += J[a];
  if (!grad) return;
  grad->assign(n, 0.0);
  vec DxJ((size_t)(N + 1) * nq);
  for (int a = 0; a < L.M; a++) {
    const dgsqp_agent_t& ag = P.agents[a];
    Jet::nv = nq;
    vector<Jet> xj(nq);
    for (int k = 0; k <= N; k++) {
      for (int i = 0; i < nq; i++) xj[i] = Jet::var(ev.x[(size_t)k * nq + i], i);
      Jet Jk = state_cost<Jet>(P, L, a, xj.data(), k == N);
      for (int i = 0; i < nq; i++) DxJ[(size_t)k * nq + i] = Jk.g[i];
    }
    for (int k = 0; k < N; k++)
      for (int j = 0; j < DGSQP_NUA; j++) {
        const double uk = u[L.col(a, k, j)], um = k > 0 ? u[L.col(a, k - 1, j)] : 0.0;
        double d = ag.w_in[j] * uk + ag.w_rate[j] * (uk - um);
        if (k + 1 < N) d -= ag.w_rate[j] * (u[L.col(a, k + 1, j)] - uk);
        (*grad)[L.col(a, k, j)] += d;
      }
    for (int c = 0; c < n; c++) {
      double s = 0;
      for (size_t r = 0; r < (size_t)(N + 1) * nq; r++) s += DxJ[r] * ev.Dux[r * n + c];
      (*grad)[c] += s;
    }
  }
}

// One backward dynamic-programming sweep for the Hessian w.r.t. the input
// sequence of a scalar function of the trajectory (DGSQP.py:679-727 for costs,
// :828-877 for one constraint row).  `kstart` is the stage whose (Dx, Dxx)
// initialise the recursion; stage injections are supplied for the cost DP.
// Accumulates weight * Hessian into time-major Htm (n x n).
struct StageInj {
  vec Dx, Dxx;     // nq, nq*nq          (Dx_Jk, Dxx_Jk)
  vec Duu, Duu2;   // nu*nu, nu*nu       (Duu_Jk, Duu_Jk2 = d2/du_{k+1} du_k)
};
static void hess_dp(const Layout& L, const Eval& ev, int kstart, const vec& Dx0, const vec& Dxx0,
                    const vector<StageInj>* inj, double weight, vec& Htm) {
  const int nq = L.nq, nu = L.nu, n = L.n;
  vec Dx = Dx0, Dxx = Dxx0;
  vec Dxu((size_t)n * nq, 0.0);  // row block t = d2/du_t dx_k (time-major rows)
  vec tmpQB(nq * nu), A1(nu * nu), A2(nu * nq), B1(nu * nu), nDxx(nq * nq), nDx(nq), tmpQA(nq * nq), rowA(nq);
  for (int k = kstart - 1; k >= 0; k--) {
    const double* Ak = &ev.A[(size_t)k * nq * nq];
    const double* Bk = &ev.B[(size_t)k * nq * nu];
    // Dxx_Q[-1] @ B_k and @ A_k
    for (int i = 0; i < nq; i++) {
      for (int j = 0; j < nu; j++) { double s = 0; for (int l = 0; l < nq; l++) s += Dxx[i * nq + l] * Bk[l * nu + j]; tmpQB[i * nu + j] = s; }
      for (int j = 0; j < nq; j++) { double s = 0; for (int l = 0; l < nq; l++) s += Dxx[i * nq + l] * Ak[l * nq + j]; tmpQA[i * nq + j] = s; }
    }
    // A1 = Duu_Jk + B^T Dxx B + sum_i Dx[i] F_k[i]                (:698-700 / :847-849)
    for (int i = 0; i < nu; i++)
      for (int j = 0; j < nu; j++) {
        double s = inj ? (*inj)[k].Duu[i * nu + j] : 0.0;
        for (int l = 0; l < nq; l++) s += Bk[l * nu + i] * tmpQB[l * nu + j];
        for (int l = 0; l < nq; l++) s += Dx[l] * ev.F[((size_t)k * nq + l) * nu * nu + i * nu + j];
        A1[i * nu + j] = s;
      }
    for (int i = 0; i < nu; i++)
      for (int j = 0; j < nu; j++) Htm[(size_t)(k * nu + i) * n + k * nu + j] += weight * A1[i * nu + j];
    // B1 = Dxu_Q[-1] @ B_k (+ Duu_Jk2 on its first block)         (:704-706 / :853-854)
    for (int t = k + 1; t < kstart && t < L.N; t++) {
      for (int i = 0; i < nu; i++)
        for (int j = 0; j < nu; j++) {
          double s = 0;
          for (int l = 0; l < nq; l++) s += Dxu[(size_t)(t * nu + i) * nq + l] * Bk[l * nu + j];
          if (inj && t == k + 1) s += (*inj)[k].Duu2[i * nu + j];
          B1[i * nu + j] = s;
        }
      for (int i = 0; i < nu; i++)
        for (int j = 0; j < nu; j++) {
          Htm[(size_t)(t * nu + i) * n + k * nu + j] += weight * B1[i * nu + j];
          Htm[(size_t)(k * nu + j) * n + t * nu + i] += weight * B1[i * nu + j];
        }
    }
    // A2 = Dxu_Jk + B^T Dxx A + sum_i Dx[i] G_k[i]                 (:708-710 / :856-858)
    for (int i = 0; i < nu; i++)
      for (int j = 0; j < nq; j++) {
        double s = 0;
        for (int l = 0; l < nq; l++) s += Bk[l * nu + i] * tmpQA[l * nq + j];
        for (int l = 0; l < nq; l++) s += Dx[l] * ev.Gd[((size_t)k * nq + l) * nu * nq + i * nq + j];
        A2[i * nq + j] = s;
      }
    // Dxu_Qk = [A2 ; Dxu_Q[-1] @ A_k]                              (:714-715 / :862-863)
    for (int t = k + 1; t < kstart && t < L.N; t++)
      for (int i = 0; i < nu; i++) {
        double* row = &Dxu[(size_t)(t * nu + i) * nq];
        for (int j = 0; j < nq; j++) { double s = 0; for (int l = 0; l < nq; l++) s += row[l] * Ak[l * nq + j]; rowA[j] = s; }
        for (int j = 0; j < nq; j++) row[j] = rowA[j];
      }
    for (int i = 0; i < nu; i++)
      for (int j = 0; j < nq; j++) Dxu[(size_t)(k * nu + i) * nq + j] = A2[i * nq + j];
    // Dxx_Qk = Dxx_Jk + A^T Dxx A + sum_i Dx[i] E_k[i]             (:717-719 / :865-867)
    for (int i = 0; i < nq; i++)
      for (int j = 0; j < nq; j++) {
        double s = inj ? (*inj)[k].Dxx[i * nq + j] : 0.0;
        for (int l = 0; l < nq; l++) s += Ak[l * nq + i] * tmpQA[l * nq + j];
        for (int l = 0; l < nq; l++) s += Dx[l] * ev.E[((size_t)k * nq + l) * nq * nq + i * nq + j];
        nDxx[i * nq + j] = s;
      }
    // Dx_Qk = Dx_Jk + Dx_Q[-1] @ A_k                               (:696 / :845)
    for (int j = 0; j < nq; j++) {
      double s = inj ? (*inj)[k].Dx[j] : 0.0;
      for (int l = 0; l < nq; l++) s += Dx[l] * Ak[l * nq + j];
      nDx[j] = s;
    }
    Dx = nDx;
    Dxx = nDxx;
  }
}

// f_Q (DGSQP.py:920-934).  literal=1: one DP per constraint row (exactly the
// reference's loop :830-877); literal=0: rows of one stage are summed with
// their multipliers first (the DP is linear in its terminal data).
static void lagrangian_hessian(const dgsqp_problem_t& P, const Layout& L, const double* u, const double* l,
                               Eval& ev, int literal) {
  const int nq = L.nq, nu = L.nu, n = L.n, N = L.N;
  ev.Q.assign((size_t)n * n, 0.0);
  vec lDuuC((size_t)n * n, 0.0);
  // ---- constraints: only rows with x-dependence have non-zero Hessians (rate / input-box rows are affine in u)
  for (int k = 1; k <= N; k++) {
    vec Dx(nq, 0.0), Dxx((size_t)nq * nq, 0.0);
    bool any = false;
    for (int r = L.stage_row0[k]; r < L.stage_row0[k + 1]; r++) {
      const Row& R = L.rows[r];
      if (R.type != R_OBS && R.type != R_ST_UB && R.type != R_ST_LB && R.type != R_LANE) continue;
      vec dx(nq, 0.0), dxx((size_t)nq * nq, 0.0);
      const double* xk = &ev.x[(size_t)k * nq];
      if (R.type == R_OBS) {
        const int ia = L.qoff[R.a], ib = L.qoff[R.b];
        const double ddx = xk[ia] - xk[ib], ddy = xk[ia + 1] - xk[ib + 1];
        dx[ia] = -2 * ddx; dx[ia + 1] = -2 * ddy; dx[ib] = 2 * ddx; dx[ib + 1] = 2 * ddy;
        for (int c = 0; c < 2; c++) {
          dxx[(ia + c) * nq + ia + c] = -2; dxx[(ib + c) * nq + ib + c] = -2;
          dxx[(ia + c) * nq + ib + c] = 2;  dxx[(ib + c) * nq + ia + c] = 2;
        }
      } else if (R.type == R_LANE) {
        const auto& ln = P.agents[R.a].lane[R.idx];
        const double hi = xk[L.qoff[R.a]] >= ln.brk ? 1.0 : 0.0;
        dx[L.qoff[R.a]] = ln.n_lo[0] + (ln.n_hi[0] - ln.n_lo[0]) * hi;
        dx[L.qoff[R.a] + 1] = ln.n_lo[1] + (ln.n_hi[1] - ln.n_lo[1]) * hi;
      } else {
        dx[L.qoff[R.a] + R.idx] = R.type == R_ST_UB ? 1.0 : -1.0;
      }
      if (literal) {
        hess_dp(L, ev, k, dx, dxx, nullptr, l[r], lDuuC);
      } else {
        for (int i = 0; i < nq; i++) Dx[i] += l[r] * dx[i];
        for (int i = 0; i < nq * nq; i++) Dxx[i] += l[r] * dxx[i];
        any = true;
      }
    }
    if (!literal && any) hess_dp(L, ev, k, Dx, Dxx, nullptr, 1.0, lDuuC);
  }
  // ---- per-agent cost Hessians (:679-727)
  for (int a = 0; a < L.M; a++) {
    const dgsqp_agent_t& ag = P.agents[a];
    vector<StageInj> inj(N);
    Jet::nv = nq;
    vector<Jet> xj(nq);
    auto state_derivs = [&](int k, bool term, vec& Dx, vec& Dxx) {
      for (int i = 0; i < nq; i++) xj[i] = Jet::var(ev.x[(size_t)k * nq + i], i);
      Jet Jk = state_cost<Jet>(P, L, a, xj.data(), term);
      Dx.assign(nq, 0.0); Dxx.assign((size_t)nq * nq, 0.0);
      for (int i = 0; i < nq; i++) { Dx[i] = Jk.g[i]; for (int j = 0; j < nq; j++) Dxx[i * nq + j] = Jk.H(i, j); }
    };
    for (int k = 0; k < N; k++) {
      state_derivs(k, false, inj[k].Dx, inj[k].Dxx);
      inj[k].Duu.assign((size_t)nu * nu, 0.0);
      inj[k].Duu2.assign((size_t)nu * nu, 0.0);
      for (int j = 0; j < DGSQP_NUA; j++) {
        const int c = a * DGSQP_NUA + j;
        // Jk = J[k] + J[k+1] picks up the rate coupling of the next stage (:686-689)
        inj[k].Duu[c * nu + c] = ag.w_in[j] + ag.w_rate[j] + (k + 1 < N ? ag.w_rate[j] : 0.0);
        if (k + 1 < N) inj[k].Duu2[c * nu + c] = -ag.w_rate[j];  // d2 J_{k+1} / du_{k+1} du_k (:694,705)
      }
    }
    vec DxN, DxxN;
    state_derivs(N, true, DxN, DxxN);
    vec Htm((size_t)n * n, 0.0);
    hess_dp(L, ev, N, DxN, DxxN, &inj, 1.0, Htm);
    // agent-major permutation (:725-726), keep rows of agent a (:933)
    for (int k = 0; k < N; k++)
      for (int j = 0; j < DGSQP_NUA; j++) {
        const int rt = k * nu + a * DGSQP_NUA + j, ra = L.col(a, k, j);
        for (int k2 = 0; k2 < N; k2++)
          for (int j2 = 0; j2 < nu; j2++) ev.Q[(size_t)ra * n + L.am(k2, j2)] = Htm[(size_t)rt * n + k2 * nu + j2];
      }
  }
  for (int k = 0; k < N; k++)
    for (int j = 0; j < nu; j++)
      for (int k2 = 0; k2 < N; k2++)
        for (int j2 = 0; j2 < nu; j2++)
          ev.Q[(size_t)L.am(k, j) * n + L.am(k2, j2)] += lDuuC[(size_t)(k * nu + j) * n + k2 * nu + j2];
}

static void evaluate(const dgsqp_problem_t& P, const Layout& L, const double* u, const double* l, const double* x0,
                     bool hessian, int literal, Eval& ev) {
  rollout(P, L, u, x0, ev.x);
  dyn_derivs(P, L, u, ev, hessian);
  build_Dux(L, ev);
  constraints(P, L, u, ev, true);
  cost_gradient(P, L, u, ev);
  if (hessian) lagrangian_hessian(P, L, u, l, ev, literal);
}

// -----------------------------------------------------------------------------
// numpy.linalg.eigh stand-in: cyclic Jacobi; returns eigenvalues s, eigenvectors U (columns)
// -----------------------------------------------------------------------------
static void jacobi_eigh(int n, vec A, vec& s, vec& U) {
  U.assign((size_t)n * n, 0.0);
  for (int i = 0; i < n; i++) U[(size_t)i * n + i] = 1.0;
  for (int sweep = 0; sweep < 100; sweep++) {
    double off = 0, diag = 0;
    for (int i = 0; i < n; i++) { diag += A[(size_t)i * n + i] * A[(size_t)i * n + i]; for (int j = i + 1; j < n; j++) off += A[(size_t)i * n + j] * A[(size_t)i * n + j]; }
    if (off <= 1e-60 || off <= 1e-32 * diag) break;
    for (int p = 0; p < n - 1; p++)
      for (int q = p + 1; q < n; q++) {
        const double apq = A[(size_t)p * n + q];
        if (apq == 0.0) continue;
        const double app = A[(size_t)p * n + p], aqq = A[(size_t)q * n + q];
        const double theta = (aqq - app) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
        for (int k = 0; k < n; k++) {
          const double akp = A[(size_t)k * n + p], akq = A[(size_t)k * n + q];
          A[(size_t)k * n + p] = c * akp - sn * akq;
          A[(size_t)k * n + q] = sn * akp + c * akq;
        }
        for (int k = 0; k < n; k++) {
          const double apk = A[(size_t)p * n + k], aqk = A[(size_t)q * n + k];
          A[(size_t)p * n + k] = c * apk - sn * aqk;
          A[(size_t)q * n + k] = sn * apk + c * aqk;
        }
        for (int k = 0; k < n; k++) {
          const double ukp = U[(size_t)k * n + p], ukq = U[(size_t)k * n + q];
          U[(size_t)k * n + p] = c * ukp - sn * ukq;
          U[(size_t)k * n + q] = sn * ukp + c * ukq;
        }
      }
  }
  s.resize(n);
  for (int i = 0; i < n; i++) s[i] = A[(size_t)i * n + i];
}

// _nearestPD (DGSQP.py:1290-1296) followed by Q += reg*I (:238-239)
static void nearest_pd(int n, const double* Qin, double reg, vec& out, double floor_ = 1e-10) {
  if (!(floor_ > 0)) floor_ = 1e-10;
  vec Bm((size_t)n * n), s, U;
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) Bm[(size_t)i * n + j] = (Qin[(size_t)i * n + j] + Qin[(size_t)j * n + i]) / 2;
  jacobi_eigh(n, Bm, s, U);
  for (int i = 0; i < n; i++) if (s[i] < 0) s[i] = floor_;
  vec C((size_t)n * n, 0.0);
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) {
      double a = 0;
      for (int k = 0; k < n; k++) a += U[(size_t)i * n + k] * s[k] * U[(size_t)j * n + k];
      C[(size_t)i * n + j] = a;
    }
  out.assign((size_t)n * n, 0.0);
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) out[(size_t)i * n + j] = (C[(size_t)i * n + j] + C[(size_t)j * n + i]) / 2;
  if (reg > 0) for (int i = 0; i < n; i++) out[(size_t)i * n + i] += reg;
}

// -----------------------------------------------------------------------------
// QP: min 1/2 x'Hx + c'x  s.t. Gx <= -g   (what ca.conic(h=Q,g=q,a=G,uba=-g) poses, DGSQP.py:246)
// Goldfarb-Idnani dual active set.  Returns 0 ok, 1 infeasible, 2 numerical failure.
// lam = multipliers of G x <= -g (>= 0), i.e. CasADi's lam_a.
// -----------------------------------------------------------------------------
// KKT polish on the final active set A (iq rows): the dual method above reaches the optimal active set, but on the literal reg = 0
// projection (eigenvalues floored at 1e-10, condition 1e12..1e13) its iterates start from x = -H^-1 c, of size 1e10 |c| along the
// floored directions, and cancel back to O(1): the point it ends on carries relative errors up to 0.5 (tools/reg0_qp_study.py,
// profiles/r03_reg0_qp_study.txt), although the minimiser itself is perfectly well determined -- the active rows pin those
// directions.  What the reference returns is OSQP's POLISHED point: the solution of the KKT system of the active set
// (osqp/src/polish.c: [H A'; A 0] regularised by delta = 1e-6 and iteratively refined).  Here: the same system, dense LU with
// partial pivoting, two steps of iterative refinement.  The polished point replaces the iterate when it is primal and dual
// feasible to the method's own tolerance; otherwise the iterate is kept (OSQP's "polish unsuccessful").  Returns 1 when taken.
static bool qp_polish_enabled = true;
static std::atomic<long> qp_polish_stats[4];   // QPs polished, rejected on a negative multiplier, rejected on a violated row, singular
static int qp_kkt_polish(int n, int m, const double* H, const double* c, const double* G, const double* g, const int* A, int iq,
                         double* x, double* lamA) {
  const int s = n + iq;
  vec K((size_t)s * s, 0.0), K0, rhs(s), sol(s), res(s);
  for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) K[(size_t)i * s + j] = H[(size_t)i * n + j];
  for (int k = 0; k < iq; k++)
    for (int j = 0; j < n; j++) { const double a = G[(size_t)A[k] * n + j]; K[(size_t)(n + k) * s + j] = a; K[(size_t)j * s + n + k] = a; }
  for (int i = 0; i < n; i++) rhs[i] = -c[i];
  for (int k = 0; k < iq; k++) rhs[n + k] = -g[A[k]];
  K0 = K;
  vector<int> piv(s);
  for (int k = 0; k < s; k++) {       // LU with partial pivoting (lowest index on ties), L below the diagonal
    int p = k; double best = std::fabs(K[(size_t)k * s + k]);
    for (int i = k + 1; i < s; i++) { const double v = std::fabs(K[(size_t)i * s + k]); if (v > best) { best = v; p = i; } }
    piv[k] = p;
    if (!(best > 0.0)) { qp_polish_stats[3]++; return 0; }
    if (p != k) for (int j = 0; j < s; j++) std::swap(K[(size_t)k * s + j], K[(size_t)p * s + j]);
    const double inv = 1.0 / K[(size_t)k * s + k];
    for (int i = k + 1; i < s; i++) {
      const double l = K[(size_t)i * s + k] * inv;
      K[(size_t)i * s + k] = l;
      if (l != 0.0) for (int j = k + 1; j < s; j++) K[(size_t)i * s + j] -= l * K[(size_t)k * s + j];
    }
  }
  auto lu_solve = [&](vec& b) {        // in place
    for (int k = 0; k < s; k++) if (piv[k] != k) std::swap(b[k], b[piv[k]]);      // P b (whole rows were swapped: P A = L U)
    for (int k = 0; k < s; k++) for (int i = k + 1; i < s; i++) b[i] -= K[(size_t)i * s + k] * b[k];
    for (int k = s - 1; k >= 0; k--) { double t = b[k]; for (int j = k + 1; j < s; j++) t -= K[(size_t)k * s + j] * b[j]; b[k] = t / K[(size_t)k * s + k]; }
  };
  sol = rhs;
  lu_solve(sol);
  for (int it = 0; it < 2; it++) {     // iterative refinement with fp64 residuals
    for (int i = 0; i < s; i++) { double t = rhs[i]; for (int j = 0; j < s; j++) t -= K0[(size_t)i * s + j] * sol[j]; res[i] = t; }
    lu_solve(res);
    for (int i = 0; i < s; i++) sol[i] += res[i];
  }
  for (int i = 0; i < s; i++) if (!std::isfinite(sol[i])) return 0;
  // accept only a primal / dual feasible point (tolerances of the active-set loop, relative for the multipliers)
  double lmax = 0.0, lmin = 0.0;
  for (int k = 0; k < iq; k++) { lmax = std::max(lmax, std::fabs(sol[n + k])); lmin = std::min(lmin, sol[n + k]); }
  if (lmin < -1e-9 * (1.0 + lmax)) { qp_polish_stats[1]++; return 0; }
  vector<char> act(m, 0);
  for (int k = 0; k < iq; k++) act[A[k]] = 1;
  for (int i = 0; i < m; i++) {
    if (act[i]) continue;
    double v = g[i];
    for (int j = 0; j < n; j++) v += G[(size_t)i * n + j] * sol[j];
    if (v > 1e-9) { qp_polish_stats[2]++; return 0; }
  }
  for (int i = 0; i < n; i++) x[i] = sol[i];
  for (int k = 0; k < iq; k++) lamA[k] = std::max(sol[n + k], 0.0);
  qp_polish_stats[0]++;
  return 1;
}

static int qp_gi(int n, int m, const double* H, const double* c, const double* G, const double* g, double* x, double* lam) {
  vec Lc((size_t)n * n, 0.0);
  for (int j = 0; j < n; j++) {  // Cholesky H = L L'
    double d = H[(size_t)j * n + j];
    for (int k = 0; k < j; k++) d -= Lc[(size_t)j * n + k] * Lc[(size_t)j * n + k];
    if (!(d > 0)) return 2;
    d = std::sqrt(d);
    Lc[(size_t)j * n + j] = d;
    for (int i = j + 1; i < n; i++) {
      double s = H[(size_t)i * n + j];
      for (int k = 0; k < j; k++) s -= Lc[(size_t)i * n + k] * Lc[(size_t)j * n + k];
      Lc[(size_t)i * n + j] = s / d;
    }
  }
  // J = L^-T
  vec J((size_t)n * n, 0.0);
  for (int col = 0; col < n; col++) {  // solve L' y = e_col  -> column col of L^-T
    for (int i = n - 1; i >= 0; i--) {
      double s = (i == col) ? 1.0 : 0.0;
      for (int k = i + 1; k < n; k++) s -= Lc[(size_t)k * n + i] * J[(size_t)k * n + col];
      J[(size_t)i * n + col] = s / Lc[(size_t)i * n + i];
    }
  }
  // x = -H^-1 c
  {
    vec y(n);
    for (int i = 0; i < n; i++) { double s = -c[i]; for (int k = 0; k < i; k++) s -= Lc[(size_t)i * n + k] * y[k]; y[i] = s / Lc[(size_t)i * n + i]; }
    for (int i = n - 1; i >= 0; i--) { double s = y[i]; for (int k = i + 1; k < n; k++) s -= Lc[(size_t)k * n + i] * x[k]; x[i] = s / Lc[(size_t)i * n + i]; }
  }
  vec R((size_t)n * n, 0.0), d(n), z(n), r(n), uu(n + 1, 0.0), np(n);
  vector<int> A(n + 1, -1);
  vector<char> active(m, 0);
  int iq = 0;
  const double TOL = 1e-10;
  for (int iter = 0; iter < 20 * (n + m); iter++) {
    // step 1: most violated constraint  s_i = -(G x + g)_i >= 0 required
    int ip = -1; double ss = -TOL;
    for (int i = 0; i < m; i++) {
      if (active[i]) continue;
      double s = -g[i];
      const double* Gi = &G[(size_t)i * n];
      for (int k = 0; k < n; k++) s -= Gi[k] * x[k];
      if (s < ss) { ss = s; ip = i; }
    }
    if (ip < 0) {
      if (qp_polish_enabled && iq > 0) (void)qp_kkt_polish(n, m, H, c, G, g, A.data(), iq, x, uu.data());
      for (int i = 0; i < m; i++) lam[i] = 0.0;
      for (int k = 0; k < iq; k++) lam[A[k]] = uu[k];
      return 0;
    }
    for (int k = 0; k < n; k++) np[k] = -G[(size_t)ip * n + k];
    uu[iq] = 0.0;
    A[iq] = ip;
    double sp = ss;
    for (int inner = 0; inner < 10 * (n + m); inner++) {
      // step 2a: d = J' np ; z = J2 d2 ; r = R^-1 d1
      for (int i = 0; i < n; i++) { double s = 0; for (int k = 0; k < n; k++) s += J[(size_t)k * n + i] * np[k]; d[i] = s; }
      for (int i = 0; i < n; i++) { double s = 0; for (int k = iq; k < n; k++) s += J[(size_t)i * n + k] * d[k]; z[i] = s; }
      for (int i = iq - 1; i >= 0; i--) { double s = d[i]; for (int k = i + 1; k < iq; k++) s -= R[(size_t)i * n + k] * r[k]; r[i] = s / R[(size_t)i * n + i]; }
      // step 2b: step lengths
      int lidx = -1; double t1 = INF;
      for (int k = 0; k < iq; k++) if (r[k] > 0 && uu[k] / r[k] < t1) { t1 = uu[k] / r[k]; lidx = k; }
      // z'np = |d2|^2 exactly (J'HJ = I); a (numerically) dependent normal has d2 = 0
      double znp = 0, npnp = 0;
      for (int k = iq; k < n; k++) znp += d[k] * d[k];
      for (int k = 0; k < n; k++) npnp += np[k] * np[k];
      double t2 = (znp > 1e-18 * npnp) ? -sp / znp : INF;
      const double t = std::min(t1, t2);
      if (t >= INF) return 1;  // infeasible
      auto drop = [&](int l) {  // remove active constraint at position l, restore R triangular
        active[A[l]] = 0;
        for (int k = l; k < iq - 1; k++) {
          A[k] = A[k + 1]; uu[k] = uu[k + 1];
          for (int i = 0; i < n; i++) R[(size_t)i * n + k] = R[(size_t)i * n + k + 1];
        }
        A[iq - 1] = A[iq]; uu[iq - 1] = uu[iq]; uu[iq] = 0; A[iq] = -1;
        for (int i = 0; i < n; i++) R[(size_t)i * n + iq - 1] = 0.0;
        iq--;
        for (int k = l; k < iq; k++) {
          const double a = R[(size_t)k * n + k], b = R[(size_t)(k + 1) * n + k];
          const double h = std::hypot(a, b);
          if (h == 0) continue;
          const double cc = a / h, s2 = b / h;
          for (int j2 = k; j2 < iq; j2++) {
            const double ra = R[(size_t)k * n + j2], rb = R[(size_t)(k + 1) * n + j2];
            R[(size_t)k * n + j2] = cc * ra + s2 * rb;
            R[(size_t)(k + 1) * n + j2] = -s2 * ra + cc * rb;
          }
          for (int i = 0; i < n; i++) {
            const double ja = J[(size_t)i * n + k], jb = J[(size_t)i * n + k + 1];
            J[(size_t)i * n + k] = cc * ja + s2 * jb;
            J[(size_t)i * n + k + 1] = -s2 * ja + cc * jb;
          }
        }
      };
      if (t2 >= INF) {  // dual step only
        for (int k = 0; k < iq; k++) uu[k] -= t * r[k];
        uu[iq] += t;
        drop(lidx);
        continue;
      }
      for (int k = 0; k < n; k++) x[k] += t * z[k];
      for (int k = 0; k < iq; k++) uu[k] -= t * r[k];
      uu[iq] += t;
      if (t == t2) {  // full step: add constraint ip
        // Givens rotations zeroing d[iq+1..n-1] into d[iq], applied to J's columns
        for (int j2 = n - 1; j2 > iq; j2--) {
          const double a = d[j2 - 1], b = d[j2];
          const double h = std::hypot(a, b);
          if (h == 0) continue;
          const double cc = a / h, s2 = b / h;
          d[j2 - 1] = h; d[j2] = 0;
          for (int i = 0; i < n; i++) {
            const double ja = J[(size_t)i * n + j2 - 1], jb = J[(size_t)i * n + j2];
            J[(size_t)i * n + j2 - 1] = cc * ja + s2 * jb;
            J[(size_t)i * n + j2] = -s2 * ja + cc * jb;
          }
        }
        for (int i = 0; i <= iq; i++) R[(size_t)i * n + iq] = d[i];
        active[ip] = 1;
        iq++;
        break;  // back to step 1
      }
      // partial step: drop blocking constraint, recompute slack of ip
      drop(lidx);
      sp = -g[ip];
      for (int k = 0; k < n; k++) sp -= G[(size_t)ip * n + k] * x[k];
    }
  }
  return 2;
}

// -----------------------------------------------------------------------------
// scipy.sparse.linalg.lsqr (scipy 1.15.3 _isolve/lsqr.py; damp=0, x0=None,
// conlim=1e8) on a dense symmetric operator A (m x m)   (DGSQP.py:324)
// -----------------------------------------------------------------------------
static void sym_ortho(double a, double b, double& c, double& s, double& r) {
  auto sgn = [](double v) { return (v > 0) - (v < 0); };
  if (b == 0) { c = sgn(a); s = 0; r = std::fabs(a); }
  else if (a == 0) { c = 0; s = sgn(b); r = std::fabs(b); }
  else if (std::fabs(b) > std::fabs(a)) { double tau = a / b; s = sgn(b) / std::sqrt(1 + tau * tau); c = s * tau; r = b / s; }
  else { double tau = b / a; c = sgn(a) / std::sqrt(1 + tau * tau); s = c * tau; r = a / c; }
}
static double nrm2(const vec& v) { double s = 0; for (double e : v) s += e * e; return std::sqrt(s); }
static int lsqr_dense(int m, int ncol, const double* A, const double* b, double atol, double btol, int iter_lim, double* xout, int* itn_out) {
  const double eps = std::numeric_limits<double>::epsilon();
  const double conlim = 1e8, ctol = 1 / conlim;
  if (iter_lim <= 0) iter_lim = 2 * ncol;
  auto matvec = [&](const vec& v, vec& out) { out.assign(m, 0.0); for (int i = 0; i < m; i++) { double s = 0; for (int j = 0; j < ncol; j++) s += A[(size_t)i * ncol + j] * v[j]; out[i] = s; } };
  auto rmatvec = [&](const vec& v, vec& out) { out.assign(ncol, 0.0); for (int i = 0; i < m; i++) { const double vi = v[i]; if (vi == 0) continue; for (int j = 0; j < ncol; j++) out[j] += A[(size_t)i * ncol + j] * vi; } };
  int itn = 0, istop = 0;
  double anorm = 0, acond = 0, ddnorm = 0, res2 = 0, xnorm = 0, xxnorm = 0, z = 0, cs2 = -1, sn2 = 0;
  vec u(b, b + m), x(ncol, 0.0), v, w, tmp;
  const double bnorm = nrm2(u);
  double beta = bnorm, alfa = 0;
  if (beta > 0) { for (double& e : u) e *= 1 / beta; rmatvec(u, v); alfa = nrm2(v); } else { v = x; alfa = 0; }
  if (alfa > 0) for (double& e : v) e *= 1 / alfa;
  w = v;
  double rhobar = alfa, phibar = beta, rnorm = beta, arnorm = alfa * beta;
  if (arnorm == 0) { for (int i = 0; i < ncol; i++) xout[i] = x[i]; if (itn_out) *itn_out = 0; return 0; }
  while (itn < iter_lim) {
    itn++;
    matvec(v, tmp);
    for (int i = 0; i < m; i++) u[i] = tmp[i] - alfa * u[i];
    beta = nrm2(u);
    if (beta > 0) {
      for (double& e : u) e *= 1 / beta;
      anorm = std::sqrt(anorm * anorm + alfa * alfa + beta * beta);
      rmatvec(u, tmp);
      for (int i = 0; i < ncol; i++) v[i] = tmp[i] - beta * v[i];
      alfa = nrm2(v);
      if (alfa > 0) for (double& e : v) e *= 1 / alfa;
    }
    const double rhobar1 = rhobar, psi = 0.0;
    double cs, sn, rho;
    sym_ortho(rhobar1, beta, cs, sn, rho);
    const double theta = sn * alfa;
    rhobar = -cs * alfa;
    const double phi = cs * phibar;
    phibar = sn * phibar;
    const double tau = sn * phi;
    const double t1 = phi / rho, t2 = -theta / rho;
    double dk2 = 0;
    for (int i = 0; i < ncol; i++) { const double dk = (1 / rho) * w[i]; dk2 += dk * dk; }
    for (int i = 0; i < ncol; i++) { x[i] = x[i] + t1 * w[i]; w[i] = v[i] + t2 * w[i]; }
    ddnorm = ddnorm + dk2;
    const double delta = sn2 * rho, gambar = -cs2 * rho, rhs = phi - delta * z, zbar = rhs / gambar;
    xnorm = std::sqrt(xxnorm + zbar * zbar);
    const double gamma = std::sqrt(gambar * gambar + theta * theta);
    cs2 = gambar / gamma; sn2 = theta / gamma; z = rhs / gamma;
    xxnorm = xxnorm + z * z;
    acond = anorm * std::sqrt(ddnorm);
    const double res1 = phibar * phibar;
    res2 = res2 + psi * psi;
    rnorm = std::sqrt(res1 + res2);
    arnorm = alfa * std::fabs(tau);
    const double test1 = rnorm / bnorm, test2 = arnorm / (anorm * rnorm + eps), test3 = 1 / (acond + eps);
    const double tt1 = test1 / (1 + anorm * xnorm / bnorm), rtol = btol + atol * anorm * xnorm / bnorm;
    if (itn >= iter_lim) istop = 7;
    if (1 + test3 <= 1) istop = 6;
    if (1 + test2 <= 1) istop = 5;
    if (1 + tt1 <= 1) istop = 4;
    if (test3 <= ctol) istop = 3;
    if (test2 <= atol) istop = 2;
    if (test1 <= rtol) istop = 1;
    if (istop != 0) break;
  }
  for (int i = 0; i < ncol; i++) xout[i] = x[i];
  if (itn_out) *itn_out = itn;
  return istop;
}

// dual initialisation  l = max(0, -lsqr(G G^T, G q))  (DGSQP.py:320-327)
static void dual_init(const dgsqp_params_t& par, const Layout& L, const Eval& ev, vec& l) {
  const int nc = L.nc, n = L.n;
  vec GGt((size_t)nc * nc), Gq(nc);
  for (int i = 0; i < nc; i++) {
    const double* Gi = &ev.G[(size_t)i * n];
    double s = 0;
    for (int k = 0; k < n; k++) s += Gi[k] * ev.q[k];
    Gq[i] = s;
    for (int j = i; j < nc; j++) {
      const double* Gj = &ev.G[(size_t)j * n];
      double t = 0;
      for (int k = 0; k < n; k++) t += Gi[k] * Gj[k];
      GGt[(size_t)i * nc + j] = t;
      GGt[(size_t)j * nc + i] = t;
    }
  }
  vec sol(nc);
  lsqr_dense(nc, nc, GGt.data(), Gq.data(), par.lsqr_atol, par.lsqr_btol, par.lsqr_iter_lim, sol.data(), nullptr);
  l.resize(nc);
  for (int i = 0; i < nc; i++) l[i] = std::max(0.0, -sol[i]);
}

// -----------------------------------------------------------------------------
// merit function pieces (DGSQP.py:949-979)
// -----------------------------------------------------------------------------
struct Lin {  // one SQP linearisation
  vec Q, q, G, g;
  double obj = 0;   // DG-SQP v2, merit 'sum_obj_l1': sum of the agents' costs at this point ...
  vec gobj;         // ... and its gradient w.r.t. ALL inputs (filled with the Hessian pass only)
};
static double f_phi(const Layout& L, const dgsqp_params_t& par, const vec& l, const vec& s, const vec& q, const vec& G, const vec& g, double mu) {
  const int n = L.n, nc = L.nc;
  double sq = 0, lg = 0, vio = 0;
  for (int c = 0; c < n; c++) {
    double d = q[c];
    for (int r = 0; r < nc; r++) d += G[(size_t)r * n + c] * l[r];
    sq += d * d;
  }
  for (int r = 0; r < nc; r++) { lg += l[r] * g[r]; vio += g[r] - s[r]; }
  double phi = 0.5 * (sq + lg * lg);
  if (par.merit_function == DGSQP_MERIT_STAT_L1) phi += mu * vio;
  return phi;
}
static double f_dstat_norm(const Layout& L, const vec& du, const vec& l, const vec& dl, const vec& Q, const vec& q, const vec& G, const vec& g) {
  const int n = L.n, nc = L.nc;
  vec d(n), Qdu(n, 0.0), Gtdl(n, 0.0), Gdu(nc, 0.0);
  for (int c = 0; c < n; c++) { double t = q[c]; for (int r = 0; r < nc; r++) t += G[(size_t)r * n + c] * l[r]; d[c] = t; }
  for (int i = 0; i < n; i++) { double t = 0; for (int j = 0; j < n; j++) t += Q[(size_t)i * n + j] * du[j]; Qdu[i] = t; }
  for (int r = 0; r < nc; r++) { double t = 0; for (int c = 0; c < n; c++) { t += G[(size_t)r * n + c] * du[c]; Gtdl[c] += G[(size_t)r * n + c] * dl[r]; } Gdu[r] = t; }
  double a = 0, lg = 0, lGdu = 0, dlg = 0;
  for (int i = 0; i < n; i++) a += d[i] * (Qdu[i] + Gtdl[i]);
  for (int r = 0; r < nc; r++) { lg += l[r] * g[r]; lGdu += l[r] * Gdu[r]; dlg += dl[r] * g[r]; }
  return a + lg * (lGdu + dlg);
}
static double f_dphi(const Layout& L, const dgsqp_params_t& par, const vec& du, const vec& l, const vec& dl, const vec& s, const Lin& k, double mu) {
  double d = f_dstat_norm(L, du, l, dl, k.Q, k.q, k.G, k.g);
  if (par.merit_function == DGSQP_MERIT_STAT_L1) { double vio = 0; for (int r = 0; r < L.nc; r++) vio += k.g[r] - s[r]; d += -mu * vio; }
  return d;
}

// _get_mu (DGSQP.py:559-585).  A NaN directional derivative leaves `mu` unbound in the
// reference (UnboundLocalError); here it yields mu = 0 (only reachable after a failed QP,
// which this restatement reports as qp_fail before getting here).
static double get_mu(const Layout& L, const dgsqp_params_t& par, const vec& du, const vec& l, const vec& dl, const vec& s, const Lin& k) {
  if (par.merit_function != DGSQP_MERIT_STAT_L1) return 0.0;
  double vio = 0;
  for (int r = 0; r < L.nc; r++) vio += k.g[r] - s[r];
  const double d = f_dstat_norm(L, du, l, dl, k.Q, k.q, k.G, k.g);
  const double rho = 0.5;
  if (d < 0 && vio > 0) return -d / ((1 - rho) * vio);
  if (d >= 0 && vio > 0) return d / ((1 - rho) * vio);
  return 0.0;
}

struct Ctx {
  const dgsqp_problem_t& P;
  const dgsqp_params_t& par;
  const Layout& L;
  const double* x0;
  int literal;
  vec* trace;  // optional event log (code, value) pairs, compared event-by-event with the device trace
  mutable double osqp_rho = 0.1;  // par.osqp_rho_carry: the rho the previous OSQP call of this solve ended with (CasADi's plugin keeps its workspace)
  mutable bool qp_dead = false;   // qp_method = DGSQP_QP_OSQP: a QP inside the watchdog was primal / dual infeasible.  The reference carries the
                                  // NaN step on and raises in _get_mu at the next iteration (DGSQP.py:566-585): the solve ends with DGSQP_QP_FAIL
};
static inline void tr(const Ctx& c, int code, double v) { if (c.trace) { c.trace->push_back((double)code); c.trace->push_back(v); } }
static void eval_lin(const Ctx& c, const vec& u, const vec& l, bool hessian, Lin& out, vec* xout = nullptr) {
  Eval ev;
  evaluate(c.P, c.L, u.data(), l.data(), c.x0, hessian, c.literal, ev);
  out.q = ev.q; out.G = ev.G; out.g = ev.g;
  if (hessian) out.Q = ev.Q;
  if (xout) *xout = ev.x;
  if (c.par.variant == DGSQP_VARIANT_V2 && c.par.merit_function == DGSQP_MERIT_SUM_OBJ_L1) sum_obj_terms(c.P, c.L, u.data(), ev, out.obj, hessian ? &out.gobj : nullptr);
}
// _solve_qp (DGSQP.py:232-266); returns false on failure ("None in du")
static bool solve_qp(const Ctx& c, const Lin& k, vec& du, vec& lhat) {
  vec Qpd;
  nearest_pd(c.L.n, k.Q.data(), c.par.reg, Qpd, c.par.eig_floor);   // eig_floor: 1e-10 in the reference (DGSQP.py:1293)
  du.assign(c.L.n, 0.0); lhat.assign(c.L.nc, 0.0);
  if (c.par.qp_method == DGSQP_QP_OSQP) {   // OSQP's own arithmetic (oracle/osqp.hpp): uba = -g, NaN answer when infeasible
    vec uba(c.L.nc);
    for (int r = 0; r < c.L.nc; r++) uba[r] = -k.g[r];
    osqp_restate::Settings S;
    if (c.par.osqp_rho_carry) S.rho = c.osqp_rho;
    const osqp_restate::Info info = osqp_restate::conic(c.L.n, c.L.nc, Qpd.data(), k.q.data(), k.G.data(), uba.data(), du.data(), lhat.data(), S);
    if (c.par.osqp_rho_carry && std::isfinite(info.rho)) c.osqp_rho = info.rho;
    // a non-finite answer -- the NaNs OSQP stores for an infeasible QP, or an ADMM run that overflowed before its iteration limit --
    // is a NaN step: _get_mu raises on it (DGSQP.py:566-585)
    bool finite = true;
    for (double e : du) finite = finite && std::isfinite(e);
    for (double e : lhat) finite = finite && std::isfinite(e);
    return finite && !(info.status == osqp_restate::PRIMAL_INFEASIBLE || info.status == osqp_restate::DUAL_INFEASIBLE || info.status == osqp_restate::PRIMAL_INFEASIBLE_INACCURATE || info.status == osqp_restate::DUAL_INFEASIBLE_INACCURATE || info.status == osqp_restate::NAN_DATA);
  }
  if (qp_gi(c.L.n, c.L.nc, Qpd.data(), k.q.data(), k.G.data(), k.g.data(), du.data(), lhat.data()) != 0) return false;
  // NOT in the reference, default off (par.snap_active_bounds = 0).  The exact minimiser sits ON its active input bounds;
  // the active-set iterations leave them at rounding distance (+-1e-15); these rows are linear in u, the next iterate
  // inherits exactly that residual and _get_mu switches on its sign (DGSQP.py:566-585).  The knob puts du on the active bounds.
  for (int r = 0; c.par.snap_active_bounds && r < c.L.nc; r++) {
    const Row& R = c.L.rows[r];
    if (lhat[r] > 0 && R.type == R_IN_UB) du[c.L.col(R.a, R.k, R.idx)] = -k.g[r];
    else if (lhat[r] > 0 && R.type == R_IN_LB) du[c.L.col(R.a, R.k, R.idx)] = k.g[r];
  }
  return true;
}
static void step_vectors(const Layout& L, const Lin& k, const vec& du, const vec& l, const vec& lhat, vec& dl, vec& s, vec& ds) {
  dl.resize(L.nc); s.resize(L.nc); ds.resize(L.nc);
  for (int r = 0; r < L.nc; r++) {
    dl[r] = lhat[r] - l[r];
    s[r] = std::min(0.0, k.g[r]);  // DGSQP.py:414 (v1 uses min)
    double t = 0;
    for (int cidx = 0; cidx < L.n; cidx++) t += k.G[(size_t)r * L.n + cidx] * du[cidx];
    ds[r] = k.g[r] + t - s[r];
  }
}
static vec axpy(const vec& a, double al, const vec& b) { vec r(a.size()); for (size_t i = 0; i < a.size(); i++) r[i] = a[i] + al * b[i]; return r; }

// _line_search_3 (DGSQP.py:1057-1081): returns the LAST trial and its merit
static void line_search_3(const Ctx& c, double mu, const vec& u, const vec& du, const vec& l, const vec& dl, const vec& s, const vec& ds,
                          const Lin& k, vec& u_out, vec& l_out, double& phi_out) {
  const double phi = f_phi(c.L, c.par, l, s, k.q, k.G, k.g, mu);
  const double dphi = f_dphi(c.L, c.par, du, l, dl, s, k, mu);
  double alpha = 1.0;
  vec ut, lt, st; double phit = 0;
  for (int i = 0; i < c.par.line_search_iters; i++) {
    ut = axpy(u, alpha, du); lt = axpy(l, alpha, dl); st = axpy(s, alpha, ds);
    Lin tr;
    eval_lin(c, ut, lt, false, tr);
    phit = f_phi(c.L, c.par, lt, st, tr.q, tr.G, tr.g, mu);
    ::tr(c, 30, alpha); ::tr(c, 31, phit);
    if (phit <= phi + c.par.beta * alpha * dphi) break;
    alpha *= c.par.tau;
  }
  u_out = ut; l_out = lt; phi_out = phit;
}

// _watchdog_line_search_4 (DGSQP.py:1174-1288), state machine of SURVEY.md A.7
static void watchdog_4(const Ctx& c, double mu, const vec& u_k, const vec& du_k, const vec& l_k, const vec& dl_k, const vec& s_k, const vec& ds_k,
                       const Lin& lin_k, vec& u_out, vec& l_out, int& qp_solves) {
  qp_solves = 0;
  const int t_hat = 5;
  const double merit_max = 1e6;
  const double phi_k = f_phi(c.L, c.par, l_k, s_k, lin_k.q, lin_k.G, lin_k.g, mu);
  const double dphi_k = f_dphi(c.L, c.par, du_k, l_k, dl_k, s_k, lin_k, mu);
  vec u1 = axpy(u_k, 1.0, du_k), l1 = axpy(l_k, 1.0, dl_k), s1 = axpy(s_k, 1.0, ds_k);
  Lin tr;
  eval_lin(c, u1, l1, false, tr);
  const double phi1 = f_phi(c.L, c.par, l1, s1, tr.q, tr.G, tr.g, mu);
  ::tr(c, 20, phi1);
  if (phi1 <= phi_k + c.par.beta * dphi_k) { u_out = u1; l_out = l1; return; }
  bool fail = false;
  vec u_t = u1, l_t = l1, du, lhat, dl, s, ds, u_n, l_n;
  double phi_n = 0;
  Lin lt;
  const auto wd_start = std::chrono::steady_clock::now();    // start_time (:1205)
  for (int t = 0; t < t_hat; t++) {
    eval_lin(c, u_t, l_t, true, lt);
    bool ok = solve_qp(c, lt, du, lhat);
    qp_solves++;
    if (!ok && c.par.qp_method == DGSQP_QP_OSQP) { c.qp_dead = true; u_out = u_t; l_out = l_t; return; }
    if (!ok) { fail = true; break; }
    step_vectors(c.L, lt, du, l_t, lhat, dl, s, ds);
    u_n = axpy(u_t, 1.0, du); l_n = lhat; vec s_n = axpy(s, 1.0, ds);
    eval_lin(c, u_n, l_n, false, tr);
    phi_n = f_phi(c.L, c.par, l_n, s_n, tr.q, tr.G, tr.g, mu);
    ::tr(c, 21, phi_n);
    if (phi_n > merit_max) break;
    if (phi_n <= phi_k + c.par.beta * dphi_k) { u_out = u_n; l_out = l_n; return; }
    u_t = u_n; l_t = l_n;
    if (c.par.time_limit >= 0.0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - wd_start).count() > c.par.time_limit) { fail = true; break; }   // :1243-1247
  }
  // insist on merit decrease (:1250-1259)
  eval_lin(c, u_t, l_t, true, lt);
  bool ok = solve_qp(c, lt, du, lhat);
  qp_solves++;
  if (!ok && c.par.qp_method == DGSQP_QP_OSQP) { c.qp_dead = true; u_out = u_t; l_out = l_t; return; }
  if (!ok) fail = true;
  else {
    step_vectors(c.L, lt, du, l_t, lhat, dl, s, ds);
    line_search_3(c, mu, u_t, du, l_t, dl, s, ds, lt, u_n, l_n, phi_n);
    ::tr(c, 22, phi_n);
  }
  if (!fail) {
    if (phi_n <= phi_k + c.par.beta * dphi_k) { u_out = u_n; l_out = l_n; return; }
    else if (phi_n > phi_k) fail = true;
    else {
      Lin l2;
      eval_lin(c, u_n, l_n, true, l2);
      vec du2, lhat2;
      if (!solve_qp(c, l2, du2, lhat2)) {
        if (c.par.qp_method == DGSQP_QP_OSQP) { qp_solves++; c.qp_dead = true; u_out = u_n; l_out = l_n; return; }
        double ph; line_search_3(c, mu, u_k, du_k, l_k, dl_k, s_k, ds_k, lin_k, u_out, l_out, ph);
        return;
      }
      qp_solves++;
      vec dl2, s2, ds2;
      step_vectors(c.L, l2, du2, l_n, lhat2, dl2, s2, ds2);
      double ph; line_search_3(c, mu, u_n, du2, l_n, dl2, s2, ds2, l2, u_out, l_out, ph);
      return;
    }
  }
  double ph;
  line_search_3(c, mu, u_k, du_k, l_k, dl_k, s_k, ds_k, lin_k, u_out, l_out, ph);
}

// -----------------------------------------------------------------------------
// solve (DGSQP.py:302-507)
// -----------------------------------------------------------------------------
struct SolveOut {
  vec u, l, x, l_init;
  int status, iters, qp_solves;
  double cond[3];
  double cost[DGSQP_MAX_AGENTS];
};
static void solve_one(const dgsqp_problem_t& P, const dgsqp_params_t& par, const Layout& L, const double* x0, const double* u_ws, int literal, SolveOut& out, vec* trace = nullptr) {
  Ctx c{P, par, L, x0, literal, trace};
  const auto t_start = std::chrono::steady_clock::now();     // solve_start (DGSQP.py:304)
  vec u(u_ws, u_ws + L.n), l;
  {
    Eval ev;
    vec l0(L.nc, 0.0);
    evaluate(P, L, u.data(), l0.data(), x0, false, literal, ev);
    dual_init(par, L, ev, l);
  }
  out.l_init = l;
  int rel_tol_its = 0, sqp_it = 0, status = DGSQP_MAX_IT, total_qp = 0;
  double p_feas = 0, comp = 0, stat = 0;
  vec u_im1, Q_prev;
  while (true) {
    Lin k;
    if (sqp_it == 0 || !par.hessian_bfgs) {
      eval_lin(c, u, l, true, k);          // exact Hessian (DGSQP.py:353-355)
    } else {
      // damped BFGS (Nocedal & Wright, Procedure 18.2) on the projected Hessian of the previous iteration (:357-364, :535-557):
      // s = u - u_prev, y = d(u, l) - d(u_prev, l) with d = q + G^T l at the CURRENT multipliers
      eval_lin(c, u, l, false, k);
      Lin km;
      eval_lin(c, u_im1, l, false, km);
      const int n = L.n, nc = L.nc;
      vec s(n), y(n), Bm, Bs(n), r(n);
      for (int i = 0; i < n; i++) {
        double d = k.q[i], dm = km.q[i];
        for (int rr = 0; rr < nc; rr++) { d += k.G[(size_t)rr * n + i] * l[rr]; dm += km.G[(size_t)rr * n + i] * l[rr]; }
        s[i] = u[i] - u_im1[i]; y[i] = d - dm;
      }
      nearest_pd(n, Q_prev.data(), 0.0, Bm, par.eig_floor);           // self._nearestPD(Q_i): no reg (:364, :1290-1296)
      double sBs = 0, sy = 0;
      for (int i = 0; i < n; i++) { double t = 0; for (int j = 0; j < n; j++) t += Bm[(size_t)i * n + j] * s[j]; Bs[i] = t; sBs += s[i] * t; sy += s[i] * y[i]; }
      const double th = sy >= 0.2 * sBs ? 1.0 : 0.8 * sBs / (sBs - sy);
      double sr = 0;
      for (int i = 0; i < n; i++) { r[i] = th * y[i] + (1 - th) * Bs[i]; sr += s[i] * r[i]; }
      k.Q.assign((size_t)n * n, 0.0);
      for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) k.Q[(size_t)i * n + j] = Bm[(size_t)i * n + j] - Bs[i] * Bs[j] / sBs + r[i] * r[j] / sr;
    }
    Q_prev = k.Q;
    u_im1 = u;
    vec l_im1 = l;
    // convergence test (:368-398)
    p_feas = 0; comp = 0; stat = 0;
    double gmax = -INF;
    for (int r = 0; r < L.nc; r++) { gmax = std::max(gmax, k.g[r]); comp = std::max(comp, std::fabs(k.g[r] * l[r])); }
    p_feas = std::max(0.0, gmax);
    for (int cc = 0; cc < L.n; cc++) { double d = k.q[cc]; for (int r = 0; r < L.nc; r++) d += k.G[(size_t)r * L.n + cc] * l[r]; stat = std::max(stat, std::fabs(d)); }
    tr(c, 1, stat); tr(c, 2, p_feas); tr(c, 3, comp);
    const int qp_before = total_qp;    // trace code 40: QP solves of this iteration, at the reference's iter_data records (:386-451)
    if (stat > 1e5) { tr(c, 40, 0.0); status = DGSQP_DIVERGED; break; }
    if (p_feas < par.p_tol && comp < par.d_tol && stat < par.d_tol) { tr(c, 40, 0.0); status = DGSQP_CONV_ABS_TOL; break; }
    vec du, lhat;
    bool ok = solve_qp(c, k, du, lhat);
    total_qp++;
    if (!ok) { tr(c, 40, 1.0); status = DGSQP_QP_FAIL; break; }
    vec dl, s, ds;
    step_vectors(L, k, du, l, lhat, dl, s, ds);
    const double mu = get_mu(L, par, du, l, dl, s, k);
    { double d2 = 0; for (double e : du) d2 += e * e; tr(c, 10, d2); }
    tr(c, 11, mu);
    tr(c, 12, f_phi(L, par, l, s, k.q, k.G, k.g, mu));
    tr(c, 13, f_dphi(L, par, du, l, dl, s, k, mu));
    if (par.nonmono_ls) {
      int nqp = 0; vec un, ln;
      watchdog_4(c, mu, u, du, l, dl, s, ds, k, un, ln, nqp);
      u = un; l = ln; total_qp += nqp;
      if (c.qp_dead) { tr(c, 40, (double)(total_qp - qp_before)); status = DGSQP_QP_FAIL; break; }
    } else {
      vec un, ln; double ph;
      line_search_3(c, mu, u, du, l, dl, s, ds, k, un, ln, ph);
      u = un; l = ln;
    }
    // relative-tolerance exit (:454-462)
    double du2 = 0, dl2 = 0;
    for (int i = 0; i < L.n; i++) du2 += (u[i] - u_im1[i]) * (u[i] - u_im1[i]);
    for (int i = 0; i < L.nc; i++) dl2 += (l[i] - l_im1[i]) * (l[i] - l_im1[i]);
    tr(c, 40, (double)(total_qp - qp_before));
    if (std::sqrt(du2) < par.p_tol / 2 && std::sqrt(dl2) < par.d_tol / 2) {
      rel_tol_its++;
      if (rel_tol_its >= par.rel_tol_req && p_feas < par.p_tol) { status = DGSQP_CONV_REL_TOL; break; }
    } else rel_tol_its = 0;
    sqp_it++;
    if (sqp_it >= par.sqp_iters) { status = DGSQP_MAX_IT; break; }
    if (par.time_limit >= 0.0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() > par.time_limit) { status = DGSQP_TIME_LIMIT; break; }   // :470 (< 0: None -> inf, :64-67)
  }
  out.u = u; out.l = l;
  rollout(P, L, u.data(), x0, out.x);
  costs(P, L, u.data(), out.x, out.cost);
  out.status = status; out.iters = sqp_it; out.qp_solves = total_qp;
  out.cond[0] = p_feas; out.cond[1] = comp; out.cond[2] = stat;
}


// -----------------------------------------------------------------------------
// DG-SQP v2: solve (DGSQP_v2.py:322-720), _solve_qp (:253-284), _get_mu (:665-690), load_checkpoint (:692-712),
// line_search (:727-755), merit functions 'stat_l1' (:1141-1160).  Trace codes as in v1; 40 carries 1.0 per iteration.
// -----------------------------------------------------------------------------
struct V2Rec { vec u, du, l, dl; double mu = 0; bool has_step = false; };
static double v2_phi(const Layout& L, const vec& l, const vec& q, const vec& G, const vec& g, double mu) {
  // f_phi = 1/2 |q + G'l|^2 + mu sum(s), s = max(0, g) supplied by every caller as max(0, g) of the same point
  const int n = L.n, nc = L.nc;
  double sq = 0, vio = 0;
  for (int c = 0; c < n; c++) { double d = q[c]; for (int r = 0; r < nc; r++) d += G[(size_t)r * n + c] * l[r]; sq += d * d; }
  for (int r = 0; r < nc; r++) vio += std::max(0.0, g[r]);
  return 0.5 * sq + mu * vio;
}
static double v2_dstat(const Layout& L, const vec& du, const vec& l, const vec& dl, const Lin& k) {
  // d/d(u,l) [1/2 |stat|^2] . (du, dl) = d'(Q du + G' dl) with the raw game Hessian Q (rows a of Duu L^a) at (u, l)
  const int n = L.n, nc = L.nc;
  double a = 0;
  for (int i = 0; i < n; i++) {
    double d = k.q[i], t = 0;
    for (int r = 0; r < nc; r++) { d += k.G[(size_t)r * n + i] * l[r]; t += k.G[(size_t)r * n + i] * dl[r]; }
    for (int j = 0; j < n; j++) t += k.Q[(size_t)i * n + j] * du[j];
    a += d * t;
  }
  return a;
}
static bool v2_solve_qp(const Ctx& c, const Lin& k, double reg, vec& du, vec& lhat) {
  vec Qpd;
  nearest_pd(c.L.n, k.Q.data(), reg, Qpd, c.par.eig_floor);
  du.assign(c.L.n, 0.0); lhat.assign(c.L.nc, 0.0);
  if (c.par.qp_method == DGSQP_QP_OSQP) {
    vec uba(c.L.nc);
    for (int r = 0; r < c.L.nc; r++) uba[r] = -k.g[r];
    osqp_restate::Settings S;
    if (c.par.osqp_rho_carry) S.rho = c.osqp_rho;
    const osqp_restate::Info info = osqp_restate::conic(c.L.n, c.L.nc, Qpd.data(), k.q.data(), k.G.data(), uba.data(), du.data(), lhat.data(), S);
    if (c.par.osqp_rho_carry && std::isfinite(info.rho)) c.osqp_rho = info.rho;
    // a non-finite answer -- the NaNs OSQP stores for an infeasible QP, or an ADMM run that overflowed before its iteration limit --
    // is a NaN step: _get_mu raises on it (DGSQP.py:566-585)
    bool finite = true;
    for (double e : du) finite = finite && std::isfinite(e);
    for (double e : lhat) finite = finite && std::isfinite(e);
    return finite && !(info.status == osqp_restate::PRIMAL_INFEASIBLE || info.status == osqp_restate::DUAL_INFEASIBLE || info.status == osqp_restate::PRIMAL_INFEASIBLE_INACCURATE || info.status == osqp_restate::DUAL_INFEASIBLE_INACCURATE || info.status == osqp_restate::NAN_DATA);
  }
  return qp_gi(c.L.n, c.L.nc, Qpd.data(), k.q.data(), k.G.data(), k.g.data(), du.data(), lhat.data()) == 0;
}
// the merit of the chosen option at a point whose linearisation (q, G, g, obj) is `t`, multipliers lt; f_dphi_c along (du, dl) at k
static double v2_merit(const Ctx& c, const Lin& t, const vec& lt, double mu) {
  if (c.par.merit_function != DGSQP_MERIT_SUM_OBJ_L1) return v2_phi(c.L, lt, t.q, t.G, t.g, mu);
  double vio = 0;
  for (int r = 0; r < c.L.nc; r++) vio += std::max(0.0, t.g[r]);
  return t.obj + mu * vio;
}
static double v2_dmerit_c(const Ctx& c, const vec& du, const vec& l, const vec& dl, const Lin& k) {
  if (c.par.merit_function != DGSQP_MERIT_SUM_OBJ_L1) return v2_dstat(c.L, du, l, dl, k);
  double d = 0;
  for (int i = 0; i < c.L.n; i++) d += k.gobj[i] * du[i];
  return d;
}
static double v2_line_search(const Ctx& c, vec& u, const vec& du, vec& l, const vec& dl, double mu, const std::vector<double>& mem) {
  const dgsqp_params_t& par = c.par;
  const double sigma = par.merit_decrease;
  double phi_b = 0, dphi_b = 0, memmax = *std::max_element(mem.begin(), mem.end());
  if (par.merit_decrease_condition == DGSQP_DECREASE_ARMIJO) {
    Lin b;
    eval_lin(c, u, l, true, b);
    double vio = 0;
    for (int r = 0; r < c.L.nc; r++) vio += std::max(0.0, b.g[r]);
    phi_b = v2_merit(c, b, l, mu);                                   // s of the iterate = max(0, g) there
    dphi_b = v2_dmerit_c(c, du, l, dl, b) - mu * vio;
  }
  double a = 1.0, phi1 = 0;
  vec ut, lt;
  for (int i = 0; i < par.line_search_iters; i++) {
    ut = axpy(u, a, du); lt = axpy(l, a, dl);
    Lin t;
    eval_lin(c, ut, lt, false, t);
    const double phi = v2_merit(c, t, lt, mu);
    const double R = par.merit_decrease_condition == DGSQP_DECREASE_MAX ? (1 - sigma * a) * memmax : phi_b + sigma * a * dphi_b;
    ::tr(c, 30, a); ::tr(c, 31, phi);
    phi1 = v2_merit(c, t, lt, 1.0);
    if (phi <= R) break;
    a *= par.tau;
  }
  u = ut; l = lt;
  return phi1;
}
static void solve_one_v2(const dgsqp_problem_t& P, const dgsqp_params_t& par, const Layout& L, const double* x0, const double* u_ws, int literal, SolveOut& out, vec* trace = nullptr) {
  Ctx c{P, par, L, x0, literal, trace};
  const auto t_start = std::chrono::steady_clock::now();
  vec u(u_ws, u_ws + L.n), l;
  Eval ev0;
  {
    vec l0(L.nc, 0.0);
    evaluate(P, L, u.data(), l0.data(), x0, false, literal, ev0);
    dual_init(par, L, ev0, l);
  }
  out.l_init = l;
  vec u_im1 = u, l_im1 = l;
  std::vector<double> mem;                                     // deque(maxlen = nms_memory_size)
  const size_t mem_size = (size_t)std::max(1, std::min(16, par.nms_memory_size));
  auto mem_append = [&](double v) { if (mem.size() == mem_size) mem.erase(mem.begin()); mem.push_back(v); };
  if (par.merit_function == DGSQP_MERIT_SUM_OBJ_L1) {
    double obj0 = 0, vio0 = 0;
    sum_obj_terms(P, L, u.data(), ev0, obj0, nullptr);
    for (int r = 0; r < L.nc; r++) vio0 += std::max(0.0, ev0.g[r]);
    mem_append(obj0 + vio0);
  } else mem_append(v2_phi(L, l, ev0.q, ev0.G, ev0.g, 1.0));
  double reg = par.reg, delta = 0, ckpt_delta = 0, ckpt_reg = reg;
  int ckpt_counter = 0, ckpt_index = 0, sqp_it = 0, m_step_it = 0, rel_tol_its = 0, total_qp = 0, status = DGSQP_MAX_IT;
  bool finished = false;
  double p_feas = 0, comp = 0, stat = 0;
  std::vector<V2Rec> iter_data;
  while (true) {
    Lin k;
    eval_lin(c, u, l, true, k);
    p_feas = 0; comp = 0; stat = 0;
    double gmax = -INF;
    for (int r = 0; r < L.nc; r++) { gmax = std::max(gmax, k.g[r]); comp = std::max(comp, std::fabs(k.g[r] * l[r])); }
    p_feas = std::max(0.0, gmax);
    for (int cc = 0; cc < L.n; cc++) { double d = k.q[cc]; for (int r = 0; r < L.nc; r++) d += k.G[(size_t)r * L.n + cc] * l[r]; stat = std::max(stat, std::fabs(d)); }
    tr(c, 1, stat); tr(c, 2, p_feas); tr(c, 3, comp);
    if (stat > 1e10) { finished = true; status = DGSQP_DIVERGED; }
    if (p_feas < par.p_tol && comp < par.d_tol && stat < par.d_tol) { finished = true; status = DGSQP_CONV_ABS_TOL; }
    if (m_step_it >= par.sqp_iters) { finished = true; status = DGSQP_MAX_IT; }
    if (par.time_limit >= 0.0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() > par.time_limit) { finished = true; status = DGSQP_TIME_LIMIT; }
    if (finished) { tr(c, 40, 0.0); break; }
    V2Rec cur;
    cur.u = u; cur.l = l;
    vec du, lhat, dl;
    const bool ok = v2_solve_qp(c, k, reg, du, lhat);
    total_qp++;
    bool d_step = false, m_step = false;
    double mu = 0;
    if (!ok) {
      if (!par.nms || iter_data.empty()) { tr(c, 40, 1.0); status = DGSQP_QP_FAIL; break; }
      m_step = true;
      const int idx = std::min(ckpt_index, (int)iter_data.size() - 1);
      cur = iter_data[idx];
      u = cur.u; du = cur.du; l = cur.l; dl = cur.dl; mu = cur.mu;
    } else {
      dl.resize(L.nc);
      for (int r = 0; r < L.nc; r++) dl[r] = lhat[r] - l[r];
      double nrm = 0;
      for (double e : du) nrm += e * e;
      for (double e : dl) nrm += e * e;
      nrm = std::sqrt(nrm);
      if (sqp_it == 0) { delta = 20.0 * nrm; ckpt_delta = delta; }
      if (par.nms) {
        if (ckpt_counter >= par.nms_frequency) m_step = true;
        else if (nrm < delta) d_step = true;
        else m_step = true;
      }
      double vio = 0;
      for (int r = 0; r < L.nc; r++) vio += std::max(0.0, k.g[r]);
      if (par.merit_parameter < 0.0) {
        const double d = v2_dmerit_c(c, du, l, dl, k);
        mu = vio > 0 ? std::fabs(d) / (0.5 * vio) : 0.0;
      } else mu = par.merit_parameter;
      tr(c, 10, nrm * nrm); tr(c, 11, mu);
      cur.du = du; cur.dl = dl; cur.mu = mu; cur.has_step = true;
    }
    double phi_new = 0;
    if (d_step) {
      u = axpy(u, 1.0, du); l = axpy(l, 1.0, dl);
      delta *= par.delta_decay;
      ckpt_counter++;
    }
    if (m_step || (!d_step && !m_step)) {
      bool accept = false;
      if (m_step) {
        m_step_it++;
        vec un = axpy(u, 1.0, du), ln = axpy(l, 1.0, dl);
        Lin t;
        eval_lin(c, un, ln, false, t);
        const double phi = v2_merit(c, t, ln, 1.0);
        const double R = (1 - par.merit_decrease) * *std::max_element(mem.begin(), mem.end());
        tr(c, 20, phi);
        if (phi <= R) { accept = true; phi_new = phi; u = un; l = ln; }
        else if (ckpt_index <= (int)iter_data.size() - 1) {
          cur = iter_data[ckpt_index];
          u = cur.u; du = cur.du; l = cur.l; dl = cur.dl; mu = cur.mu;
          delta = ckpt_delta;
          reg = ckpt_reg;
        }
      }
      if (!accept) {
        phi_new = v2_line_search(c, u, du, l, dl, mu, mem);
        tr(c, 22, phi_new);
      }
      double du2 = 0, dl2 = 0;
      for (int i = 0; i < L.n; i++) du2 += (u[i] - u_im1[i]) * (u[i] - u_im1[i]);
      for (int i = 0; i < L.nc; i++) dl2 += (l[i] - l_im1[i]) * (l[i] - l_im1[i]);
      if (std::sqrt(du2) < par.p_tol && std::sqrt(dl2) < par.d_tol) {
        rel_tol_its++;
        if (rel_tol_its >= par.rel_tol_req && p_feas < par.p_tol) { finished = true; status = DGSQP_CONV_REL_TOL; }
      } else rel_tol_its = 0;
      u_im1 = u; l_im1 = l;
      reg *= par.reg_decay;
      mem_append(phi_new);
      if (m_step) { ckpt_counter = 0; ckpt_delta = delta; ckpt_reg = reg; ckpt_index = sqp_it + 1; }
    }
    tr(c, 40, 1.0);
    iter_data.push_back(cur);
    sqp_it++;
  }
  out.u = u; out.l = l;
  rollout(P, L, u.data(), x0, out.x);
  costs(P, L, u.data(), out.x, out.cost);
  out.status = status; out.iters = sqp_it; out.qp_solves = total_qp;
  out.cond[0] = p_feas; out.cond[1] = comp; out.cond[2] = stat;
}

static void solve_any(const dgsqp_problem_t& P, const dgsqp_params_t& par, const Layout& L, const double* x0, const double* u_ws, int literal, SolveOut& out, vec* trace = nullptr) {
  if (par.variant == DGSQP_VARIANT_V2) solve_one_v2(P, par, L, x0, u_ws, literal, out, trace);
  else solve_one(P, par, L, x0, u_ws, literal, out, trace);
}

// =============================================================================
// C entry points (tests / bench cpu_baseline only)
// =============================================================================
extern "C" {

int oracle_dims(const dgsqp_problem_t* P, int32_t* out /* M,N,nq,nu,n,nc */) {
  Layout L = make_layout(*P);
  out[0] = L.M; out[1] = L.N; out[2] = L.nq; out[3] = L.nu; out[4] = L.n; out[5] = L.nc;
  return 0;
}

// row table: [nc][5] = type,k,a,b,idx
int oracle_rows(const dgsqp_problem_t* P, int32_t* out) {
  Layout L = make_layout(*P);
  for (int r = 0; r < L.nc; r++) { out[5 * r] = L.rows[r].type; out[5 * r + 1] = L.rows[r].k; out[5 * r + 2] = L.rows[r].a; out[5 * r + 3] = L.rows[r].b; out[5 * r + 4] = L.rows[r].idx; }
  return 0;
}

// continuous + discrete dynamics of one agent, with first/second derivatives of fd
//   dq[nqa], qn[nqa], Jac[nqa][nqa+2], Hes[nqa][(nqa+2)^2]
int oracle_dynamics(const dgsqp_problem_t* P, int agent, const double* q, const double* u, double* dq, double* qn, double* Jac, double* Hes) {
  const dgsqp_agent_t& ag = P->agents[agent];
  const int nqa = model_nq(ag.model), nv = nqa + DGSQP_NUA;
  if (dq) fc<double>(*P, ag, q, u, dq);
  if (qn) fd<double>(*P, ag, nqa, q, u, qn);
  if (Jac || Hes) {
    Jet::nv = nv;
    Jet qj[DGSQP_MAX_NQA], uj[DGSQP_NUA], out[DGSQP_MAX_NQA];
    for (int i = 0; i < nqa; i++) qj[i] = Jet::var(q[i], i);
    for (int j = 0; j < DGSQP_NUA; j++) uj[j] = Jet::var(u[j], nqa + j);
    fd<Jet>(*P, ag, nqa, qj, uj, out);
    for (int i = 0; i < nqa; i++) {
      if (Jac) for (int j = 0; j < nv; j++) Jac[i * nv + j] = out[i].g[j];
      if (Hes) for (int j = 0; j < nv * nv; j++) Hes[i * nv * nv + j] = out[i].h[j];
    }
  }
  return 0;
}

int oracle_track(const dgsqp_problem_t* P, double s, double* curv, double* tangent, double* dtangent) {
  Jet::nv = 1;
  Jet sj = Jet::var(s, 0), psi, c;
  track_eval<Jet>(*P, sj, c, psi);
  *curv = c.v; *tangent = psi.v; if (dtangent) *dtangent = psi.g[0];
  return 0;
}

// DG-SQP v2, merit 'sum_obj_l1': sum of the agents' costs at u and its gradient w.r.t. all inputs (test hook of sum_obj_terms)
int oracle_sum_obj(const dgsqp_problem_t* P, const double* x0, const double* u, double* obj, double* grad) {
  Layout L = make_layout(*P);
  Eval ev;
  vec lz(L.nc, 0.0), g;
  evaluate(*P, L, u, lz.data(), x0, false, false, ev);
  sum_obj_terms(*P, L, u, ev, *obj, grad ? &g : nullptr);
  if (grad) std::copy(g.begin(), g.end(), grad);
  return 0;
}

// DGSQP._evaluate (hessian: 0 no, 1 stage-aggregated DP, 2 literal per-row DP)
int oracle_evaluate(const dgsqp_problem_t* P, const double* x0, const double* u, const double* l, int hessian,
                    double* q, double* g, double* G, double* Q, double* x, double* J) {
  Layout L = make_layout(*P);
  Eval ev;
  vec lz(L.nc, 0.0);
  evaluate(*P, L, u, l ? l : lz.data(), x0, hessian != 0, hessian == 2, ev);
  if (q) std::copy(ev.q.begin(), ev.q.end(), q);
  if (g) std::copy(ev.g.begin(), ev.g.end(), g);
  if (G) std::copy(ev.G.begin(), ev.G.end(), G);
  if (Q && hessian) std::copy(ev.Q.begin(), ev.Q.end(), Q);
  if (x) std::copy(ev.x.begin(), ev.x.end(), x);
  if (J) costs(*P, L, u, ev.x, J);
  return 0;
}

int oracle_dual_init(const dgsqp_problem_t* P, const dgsqp_params_t* par, const double* x0, const double* u, double* l0) {
  Layout L = make_layout(*P);
  Eval ev;
  vec lz(L.nc, 0.0), l;
  evaluate(*P, L, u, lz.data(), x0, false, 0, ev);
  dual_init(*par, L, ev, l);
  std::copy(l.begin(), l.end(), l0);
  return 0;
}

int oracle_nearest_pd2(int n, const double* Q, double reg, double eig_floor, double* out) {
  vec o;
  nearest_pd(n, Q, reg, o, eig_floor);
  std::memcpy(out, o.data(), sizeof(double) * (size_t)n * n);
  return 0;
}
int oracle_nearest_pd(int n, const double* Q, double reg, double* out) {
  vec o;
  nearest_pd(n, Q, reg, o);
  std::copy(o.begin(), o.end(), out);
  return 0;
}

int oracle_eigh(int n, const double* A, double* s, double* U) {
  vec a(A, A + (size_t)n * n), sv, Uv;
  jacobi_eigh(n, a, sv, Uv);
  std::copy(sv.begin(), sv.end(), s);
  std::copy(Uv.begin(), Uv.end(), U);
  return 0;
}

void oracle_set_qp_polish(int on) { qp_polish_enabled = on != 0; }
void oracle_qp_polish_stats(long* out4, int reset) { for (int i = 0; i < 4; i++) { out4[i] = qp_polish_stats[i]; if (reset) qp_polish_stats[i] = 0; } }
int oracle_qp(int n, int m, const double* H, const double* c, const double* G, const double* g, double* x, double* lam) {
  return qp_gi(n, m, H, c, G, g, x, lam);
}
// OSQP as the reference poses it (DGSQP.py:246): min 1/2 x'Hx + c'x s.t. G x <= -g.  info8 = {status, iters, polished, rho, rho updates,
// active rows of the polish, primal residual, dual residual of the ADMM iterate}
int oracle_osqp(int n, int m, const double* H, const double* c, const double* G, const double* g, double* x, double* lam, double* info8) {
  vec uba(m);
  for (int r = 0; r < m; r++) uba[r] = -g[r];
  const osqp_restate::Info info = osqp_restate::conic(n, m, H, c, G, uba.data(), x, lam);
  if (info8) { info8[0] = info.status; info8[1] = info.iters; info8[2] = info.polished; info8[3] = info.rho; info8[4] = info.rho_updates; info8[5] = info.n_active; info8[6] = info.pri_res; info8[7] = info.dua_res; }
  return info.status;
}

int oracle_lsqr(int m, int n, const double* A, const double* b, double atol, double btol, int iter_lim, double* x, int32_t* itn) {
  int it = 0;
  int istop = lsqr_dense(m, n, A, b, atol, btol, iter_lim, x, &it);
  if (itn) *itn = it;
  return istop;
}

// merit / step pieces for unit tests: phi and dphi at (l, s) for a given linearisation
int oracle_merit(const dgsqp_problem_t* P, const dgsqp_params_t* par, const double* Q, const double* q, const double* G, const double* g,
                 const double* l, const double* s, const double* du, const double* dl, double mu, double* phi, double* dphi, double* mu_out) {
  Layout L = make_layout(*P);
  Lin k;
  k.Q.assign(Q, Q + (size_t)L.n * L.n); k.q.assign(q, q + L.n); k.G.assign(G, G + (size_t)L.nc * L.n); k.g.assign(g, g + L.nc);
  vec lv(l, l + L.nc), sv(s, s + L.nc), duv(du, du + L.n), dlv(dl, dl + L.nc);
  if (mu_out) { *mu_out = get_mu(L, *par, duv, lv, dlv, sv, k); }
  if (phi) *phi = f_phi(L, *par, lv, sv, k.q, k.G, k.g, mu);
  if (dphi) *dphi = f_dphi(L, *par, duv, lv, dlv, sv, k, mu);
  return 0;
}

// DGSQP.solve() of one scenario with the event log used by the trace-parity tests
int oracle_solve_trace(const dgsqp_problem_t* P, const dgsqp_params_t* par, const double* x0, const double* u_ws, double* trace_out, int32_t max_pairs, int32_t* n_pairs) {
  Layout L = make_layout(*P);
  SolveOut o;
  vec t;
  solve_any(*P, *par, L, x0, u_ws, 0, o, &t);
  int np = (int)t.size() / 2;
  if (np > max_pairs) np = max_pairs;
  for (int i = 0; i < 2 * np; i++) trace_out[i] = t[i];
  *n_pairs = np;
  return 0;
}

// DGSQP.solve() for B scenarios on `nthreads` host threads
// bench.py's cpu_baseline: wall-clock seconds every scenario of the next oracle_solve_batch call took on its thread (null: off)
static double* g_scenario_seconds = nullptr;
void oracle_set_scenario_seconds(double* buf) { g_scenario_seconds = buf; }

int oracle_solve_batch(const dgsqp_problem_t* P, const dgsqp_params_t* par, int64_t B, const double* x0, const double* u_ws,
                       double* u_out, double* l_out, double* x_out, int32_t* status, int32_t* iters, int32_t* qp_solves,
                       double* cond, double* cost, double* l_init, int literal, int nthreads) {
  Layout L = make_layout(*P);
  double* const secs = g_scenario_seconds;
  std::atomic<int64_t> next{0};       // dynamic hand-out: iteration counts vary 1..50+, a static partition would time the unluckiest thread
  auto work = [&](int64_t, int64_t) {
    for (int64_t b = next++; b < B; b = next++) {
      SolveOut o;
      const auto t_scn = std::chrono::steady_clock::now();
      solve_any(*P, *par, L, x0 + b * L.nq, u_ws + b * L.n, literal, o);
      if (secs) secs[b] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_scn).count();
      if (u_out) std::copy(o.u.begin(), o.u.end(), u_out + b * L.n);
      if (l_out) std::copy(o.l.begin(), o.l.end(), l_out + b * L.nc);
      if (x_out) std::copy(o.x.begin(), o.x.end(), x_out + b * (int64_t)(L.N + 1) * L.nq);
      if (l_init) std::copy(o.l_init.begin(), o.l_init.end(), l_init + b * L.nc);
      if (status) status[b] = o.status;
      if (iters) iters[b] = o.iters;
      if (qp_solves) qp_solves[b] = o.qp_solves;
      if (cond) for (int i = 0; i < 3; i++) cond[b * 3 + i] = o.cond[i];
      if (cost) for (int a = 0; a < L.M; a++) cost[b * L.M + a] = o.cost[a];
    }
  };
  if (nthreads <= 1) { work(0, 1); return 0; }
  vector<std::thread> th;
  for (int t = 0; t < nthreads; t++) th.emplace_back(work, (int64_t)t, (int64_t)nthreads);
  for (auto& t : th) t.join();
  return 0;
}

}  // extern "C"
