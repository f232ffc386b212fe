// DGSQP._evaluate on device (reference DGSQP/solvers/DGSQP.py:509-533): rollout, dynamics
// Jacobians / Hessians, sensitivities, q, g, packed G and the raw game Hessian Q.
#pragma once
#include "dgsqp_device.h"

// ------------------------------------------------------------------------------------------------
// state-dependent part of agent a's cost (chicane.py:239-256, ablation.py:229-262): value and
// analytic joint gradient / Hessian.  Executed by ONE thread; Dx/Dxx are accumulated into.
// ------------------------------------------------------------------------------------------------
template <class XP, class DP>
__device__ inline double dev_state_cost(const DgProb& D, int a, XP xk, bool terminal, DP Dx, DP Dxx) {
  const dgsqp_agent_t& ag = D.P.agents[a];
  const int nq = D.nq, ia = D.qoff[a];
  double J = 0;
  // goal tracking 1/2 (q - goal)^T diag(w) (q - goal), terminal = mult x stage (merge.py:253-261)
  for (int i = 0; i < D.nqa[a]; i++)
    if (ag.w_goal[i] != 0.0) {
      const double w = (terminal ? ag.goal_term_mult : 1.0) * ag.w_goal[i], dd = xk[ia + i] - ag.goal[i];
      J += 0.5 * w * dd * dd;
      if (Dx) Dx[ia + i] += w * dd;
      if (Dxx) Dxx[(ia + i) * nq + ia + i] += w;
    }
  for (int b = 0; b < D.M; b++) {
    if (b == a) continue;
    const int ib = D.qoff[b];
    if (ag.w_block != 0.0) {
      const int ea = ia + D.eyidx[a], eb = ib + D.eyidx[b];
      const double dd = xk[ea] - xk[eb];
      J += 0.5 * ag.w_block * dd * dd;
      if (Dx) { Dx[ea] += ag.w_block * dd; Dx[eb] -= ag.w_block * dd; }
      if (Dxx) { Dxx[ea * nq + ea] += ag.w_block; Dxx[eb * nq + eb] += ag.w_block; Dxx[ea * nq + eb] -= ag.w_block; Dxx[eb * nq + ea] -= ag.w_block; }
    }
    if (ag.w_obs != 0.0) {
      const double dx = xk[ia] - xk[ib], dy = xk[ia + 1] - xk[ib + 1];
      const double r = sqrt(dx * dx + dy * dy);
      const double z = (ag.obs_cost_r + D.P.agents[b].obs_cost_r) - r;
      if (z > 0) {  // fmax(0,z)^2, derivative (z>0)
        J += 0.5 * ag.w_obs * z * z;
        const double ex = dx / r, ey = dy / r;
        if (Dx) {
          Dx[ia] -= ag.w_obs * z * ex; Dx[ia + 1] -= ag.w_obs * z * ey;
          Dx[ib] += ag.w_obs * z * ex; Dx[ib + 1] += ag.w_obs * z * ey;
        }
        if (Dxx) {
          // Hessian in the difference coordinates: w [ e e^T - z (I - e e^T)/r ]
          const double h[2][2] = {{ag.w_obs * (ex * ex - z * (1 - ex * ex) / r), ag.w_obs * (ex * ey + z * ex * ey / r)},
                                  {ag.w_obs * (ex * ey + z * ex * ey / r), ag.w_obs * (ey * ey - z * (1 - ey * ey) / r)}};
          for (int p = 0; p < 2; p++)
            for (int q2 = 0; q2 < 2; q2++) {
              Dxx[(ia + p) * nq + ia + q2] += h[p][q2]; Dxx[(ib + p) * nq + ib + q2] += h[p][q2];
              Dxx[(ia + p) * nq + ib + q2] -= h[p][q2]; Dxx[(ib + p) * nq + ia + q2] -= h[p][q2];
            }
        }
      }
    }
  }
  if (terminal) {
    const int sa = ia + D.sidx[a];
    J -= ag.w_prog * xk[sa];
    if (Dx) Dx[sa] -= ag.w_prog;
    for (int b = 0; b < D.M; b++) {
      if (b == a) continue;
      const int sb = D.qoff[b] + D.sidx[b];
      const double dl = xk[sb] - xk[sa];
      if (ag.comp_type == DGSQP_COMP_ATAN) {
        const double w = 1.0 + dl * dl;
        J += ag.w_comp * atan(dl);
        if (Dx) { Dx[sb] += ag.w_comp / w; Dx[sa] -= ag.w_comp / w; }
        if (Dxx) {
          const double f2 = -2.0 * ag.w_comp * dl / (w * w);
          Dxx[sa * nq + sa] += f2; Dxx[sb * nq + sb] += f2; Dxx[sa * nq + sb] -= f2; Dxx[sb * nq + sa] -= f2;
        }
      } else {
        J += ag.w_comp * dl;
        if (Dx) { Dx[sb] += ag.w_comp; Dx[sa] -= ag.w_comp; }
      }
    }
  }
  return J;
}

// ------------------------------------------------------------------------------------------------
// rollout x_{k+1} = f_d(x_k, u_k)   (evaluate_dynamics, DGSQP.py:597-601): one lane per agent
// ------------------------------------------------------------------------------------------------
// trial input u + alpha du, written the same way everywhere so that equal points give bit-identical trajectories
__device__ inline double step_u(double u, double alpha, double du) { return __builtin_fma(alpha, du, u); }
template <int NQA, bool SPL = false>
__device__ inline void dev_rollout_agent(const DgProb& D, int a, clptr ub, clptr du, double alpha, lptr x) {
  typedef Ty<0> T;
  const int nq = D.nq, qo = D.qoff[a];
  T q[NQA], u[2], qn[NQA];
  for (int i = 0; i < NQA; i++) q[i].c[0] = x[qo + i];
  for (int k = 0; k < D.N; k++) {
    const int i0 = am_col(D, a, k, 0);
    u[0].c[0] = du ? step_u(ub[i0], alpha, du[i0]) : ub[i0];
    u[1].c[0] = du ? step_u(ub[i0 + 1], alpha, du[i0 + 1]) : ub[i0 + 1];
    dev_fd<0, NQA, SPL>(D.P, D.P.agents[a], q, u, qn);
    for (int i = 0; i < NQA; i++) { q[i] = qn[i]; x[(k + 1) * nq + qo + i] = qn[i].c[0]; }
  }
}
// Dynamic-bicycle rollout on TWO lanes per agent.  The 40 f_c evaluations of one rk4 step (M = 10) are sequential and
// each costs ~9 fp64 transcendental calls; the front / rear tyre chains (atan2 -> atan -> sin) and the two angle sincos
// are the same instruction stream on different data, so lane 2a handles (front axle, e_psi) and lane 2a+1 (rear axle,
// e_psi + psi_t); results are exchanged inside the quad with DPP.  sincos(delta) is constant over the step and hoisted.
// Same arithmetic as dev_fc_dyn<0> (dynamics_models.py:2008-2062), evaluated redundantly on both lanes otherwise.
struct DynLane {
  // per-lane constants of the pair rollout, loaded once (agent parameters are lane-dependent, i.e. vector loads otherwise)
  double L_f, L_r, Bc, Cc, Dc, lin, c_da, c_dr, c_r, p_r, inv_mass, inv_Iz, fr, ff, L, invL;
  int role, simple_slip, pacejka, nsegs;
  double Ll, cdl, sdl, add;      // axle offset (+L_f / -L_r) and, per stage, the rotation into the wheel frame and the slip offset of this lane's axle
  // track segment currently containing s: [lo, hi), curvature, tangent angle at lo and its slope
  double lo, hi, curv, ang0, slope;
};
// Elementary functions of the scalar rollout.  Its ~10^3 sequential f_c evaluations run on a handful of lanes of ONE wavefront: the
// instruction count of the chain is the latency of the rollout (and of every line-search trial).  On a car that is not spinning
// every angle is small -- slip angles below atan(7/16), Pacejka arguments and heading errors below pi/4 -- and then the range
// reductions are identities: when EVERY active lane is in that range (wave-uniform test, no divergence) the table look-up, the
// quadrant selects and the unused cosine are skipped.  Same operations in the same order as the general path: bit-identical.
__device__ inline double roll_atan_core(double num, double rcp, double hi, double lo) {
  const double t = num * rcp, z = t * t;
  const double p = atan_poly(z);
  double e = __builtin_fma(-p, t * z, lo);
  e = __builtin_fma(num, rcp, e);
  return hi + e;
}
__device__ inline double roll_atan2(double y, double x) {
  const double ay = __builtin_fabs(y), ax = __builtin_fabs(x);
  const double y16 = 16.0 * ay;
  if (__all(x > 0.0 && y16 < 7.0 * ax)) {
    const double r = roll_atan_core(ay, fast_rcp(ax), 0.0, 0.0);
    return y < 0.0 ? -r : r;
  }
  const int rid = (int)(y16 >= 7.0 * ax) + (int)(y16 >= 11.0 * ax) + (int)(y16 >= 19.0 * ax) + (int)(y16 >= 39.0 * ax);
  clptr ta = LP(dg_prob.L.t_atan);
  const double hi = ta[rid], lo = ta[5 + rid];
  const bool big = rid == 4;
  const double kk = 0.5 * (double)(big ? 0 : rid);
  const double num = big ? -ax : __builtin_fma(-kk, ax, ay);
  double den = big ? ay : __builtin_fma(kk, ay, ax);
  den = den == 0.0 ? 1.0 : den;
  double r = roll_atan_core(num, fast_rcp(den), hi, lo);
  r = x < 0.0 ? (3.141592653589793 - r) + 1.2246467991473532e-16 : r;
  return y < 0.0 ? -r : r;
}
__device__ inline double roll_atan(double x) {
  const double ay = __builtin_fabs(x);
  if (__all(ay < 0.4375)) {
    const double r = roll_atan_core(ay, 1.0, 0.0, 0.0);
    return x < 0.0 ? -r : r;
  }
  const int rid = (int)(ay >= 0.4375) + (int)(ay >= 0.6875) + (int)(ay >= 1.1875) + (int)(ay >= 2.4375);
  clptr ta = LP(dg_prob.L.t_atan);
  const double hi = ta[rid], lo = ta[5 + rid];
  const bool big = rid == 4;
  const double kk = 0.5 * (double)(big ? 0 : rid);
  const double num = big ? -1.0 : ay - kk;
  const double den = big ? ay : __builtin_fma(kk, ay, 1.0);
  const double r = roll_atan_core(num, fast_rcp(den), hi, lo);
  return x < 0.0 ? -r : r;
}
__device__ inline double roll_sin(double x) {
  if (__all(__builtin_fabs(x) < 0.78)) { const double z = x * x; return __builtin_fma(x * z, sin_poly(z), x); }
  double s_, c_;
  dev_sincos(x, s_, c_);
  return s_;
}
__device__ inline void roll_sincos(double x, double& so, double& co) {
  if (__all(__builtin_fabs(x) < 0.78)) {
    const double z = x * x;
    so = __builtin_fma(x * z, sin_poly(z), x);
    co = __builtin_fma(z * z, cos_poly(z), __builtin_fma(-0.5, z, 1.0));
    return;
  }
  dev_sincos(x, so, co);
}
__device__ inline void dyn_lane_seek(DynLane& Z, double sbar) {
  constexpr int S1 = DGSQP_MAX_SEGS + 1;
  clptr tt = LP(dg_prob.L.t_track);
  int seg = 0;
  for (int i = 1; i < Z.nsegs; i++) seg += (sbar >= tt[i]) ? 1 : 0;
  Z.lo = tt[seg]; Z.hi = seg + 1 < Z.nsegs ? tt[seg + 1] : 1e300;
  Z.curv = tt[S1 + seg]; Z.ang0 = tt[2 * S1 + seg]; Z.slope = tt[3 * S1 + seg];
}
__device__ inline void dyn_fc_pair(DynLane& Z, const double* q, double ua, double us, double sd, double cd, double* dq) {
  const double vx = q[2], vy = q[3], w = q[4];
  const double sbar = wrap_s(q[6], Z.L, Z.invL);
  if (!(sbar >= Z.lo && sbar < Z.hi)) dyn_lane_seek(Z, sbar);     // rare: s crossed a segment boundary
  const double c = Z.curv;
  const double psit = (q[6] + (sbar - q[6] - Z.lo)) * Z.slope + Z.ang0;
  // role 0: front axle and e_psi ; role 1: rear axle and e_psi + psi_t.  Even / odd lanes differ only in per-lane constants: the axle
  // offset, the rotation into the wheel frame (identity except for a steered front axle with the exact slip formula) and the offset
  // of the simple slip formula -- no selects in the chain
  const bool front = Z.role == 0;
  const double vyl = __builtin_fma(w, Z.Ll, vy);
  const double ay_ = __builtin_fma(vyl, Z.cdl, -vx * Z.sdl), ax_ = __builtin_fma(vx, Z.cdl, vyl * Z.sdl);
  const double alpha = Z.add - roll_atan2(ay_, ax_);
  double F;
  if (Z.pacejka) F = Z.Dc * roll_sin(Z.Cc * roll_atan(Z.Bc * alpha));
  else F = alpha * Z.lin;
  double sa, ca;
  roll_sincos(front ? q[5] : q[5] + psit, sa, ca);
  // exchange inside the pair: quad_perm [0,0,2,2] takes the even lane's value, [1,1,3,3] the odd lane's
  const double fyf = dpp_f64<0xA0>(F), fyr = dpp_f64<0xF5>(F);
  const double se = dpp_f64<0xA0>(sa), ce = dpp_f64<0xA0>(ca), st = dpp_f64<0xF5>(sa), ct = dpp_f64<0xF5>(ca);
  const double avx = __builtin_fabs(vx);
  double Fx = vx * (-Z.c_da) - vx * avx * Z.c_dr;
  if (Z.c_r != 0.0) Fx = Fx - pow(avx, Z.p_r) * (vx / sqrt(vx * vx + 1e-6)) * Z.c_r;
  const double a_r = ua * Z.fr, a_f = ua * Z.ff;
  const double ax = a_r + a_f * cd + (Fx - fyf * sd) * Z.inv_mass;
  const double ay = a_f * sd + (fyf * cd + fyr) * Z.inv_mass;
  const double vlon = (vx * ce - vy * se) * fast_rcp(1.0 - q[7] * c);
  dq[0] = vx * ct - vy * st;
  dq[1] = vy * ct + vx * st;
  dq[2] = ax + w * vy;
  dq[3] = ay - w * vx;
  dq[4] = (fyf * cd * Z.L_f - fyr * Z.L_r) * Z.inv_Iz;
  dq[5] = w - vlon * c;
  dq[6] = vlon;
  dq[7] = vx * se + vy * ce;
}
typedef volatile __attribute__((address_space(3))) double vlds_d;
__device__ inline void dev_rollout_dyn_pair(const DgProb& D, int a, int role, clptr ub, clptr du, double alpha, lptr x, lptr prog = nullptr) {
  const dgsqp_problem_t& P = D.P;
  const dgsqp_agent_t& ag = P.agents[a];
  const int nq = D.nq, qo = D.qoff[a];
  DynLane Z;
  Z.role = role; Z.simple_slip = ag.simple_slip; Z.pacejka = ag.tire_model == 0; Z.nsegs = P.n_segs;
  Z.L_f = ag.L_f; Z.L_r = ag.L_r; Z.Ll = role == 0 ? ag.L_f : -ag.L_r;
  Z.Bc = role == 0 ? ag.pac_Bf : ag.pac_Br; Z.Cc = role == 0 ? ag.pac_Cf : ag.pac_Cr; Z.Dc = role == 0 ? ag.pac_Df : ag.pac_Dr;
  Z.lin = role == 0 ? ag.lin_Bf * ag.mass * ag.gravity * ag.L_r / (ag.L_f + ag.L_r) : ag.lin_Br * ag.mass * ag.gravity * ag.L_f / (ag.L_f + ag.L_r);
  Z.c_da = ag.c_da; Z.c_dr = ag.c_dr; Z.c_r = ag.c_r; Z.p_r = ag.p_r;
  Z.inv_mass = 1.0 / ag.mass; Z.inv_Iz = 1.0 / ag.I_z;
  Z.fr = ag.drive_wheels == 0 ? 0.5 : 1.0; Z.ff = ag.drive_wheels == 0 ? 0.5 : 0.0;
  Z.L = P.track_L; Z.invL = D.inv_track_L;
  Z.lo = 1.0; Z.hi = 0.0;   // empty interval: first use seeks
  double q[8], k1[8], k2[8], t[8];
  for (int i = 0; i < 8; i++) q[i] = x[qo + i];
  const double h = P.dt / P.substeps, h2 = 0.5 * h, h6 = h / 6.0;
  const int integ = P.integrator, nsub = integ == DGSQP_INT_EULER ? 1 : P.substeps;
  for (int k = 0; k < D.N; k++) {
    const int i0 = am_col(D, a, k, 0);
    const double ua = du ? step_u(ub[i0], alpha, du[i0]) : ub[i0], us = du ? step_u(ub[i0 + 1], alpha, du[i0 + 1]) : ub[i0 + 1];
    double sd, cd;
    dev_sincos(us, sd, cd);
    const bool rot = role == 0 && !ag.simple_slip;
    Z.cdl = rot ? cd : 1.0; Z.sdl = rot ? sd : 0.0; Z.add = (role == 0 && ag.simple_slip) ? us : 0.0;
    for (int m = 0; m < nsub; m++) {
      if (integ == DGSQP_INT_RK4) {
        dyn_fc_pair(Z, q, ua, us, sd, cd, k1);
        for (int i = 0; i < 8; i++) t[i] = __builtin_fma(k1[i], h2, q[i]);
        dyn_fc_pair(Z, t, ua, us, sd, cd, k2);
        for (int i = 0; i < 8; i++) { t[i] = __builtin_fma(k2[i], h2, q[i]); k1[i] = __builtin_fma(k2[i], 2.0, k1[i]); }
        dyn_fc_pair(Z, t, ua, us, sd, cd, k2);
        for (int i = 0; i < 8; i++) { t[i] = __builtin_fma(k2[i], h, q[i]); k1[i] = __builtin_fma(k2[i], 2.0, k1[i]); }
        dyn_fc_pair(Z, t, ua, us, sd, cd, k2);
        for (int i = 0; i < 8; i++) q[i] = __builtin_fma(k1[i] + k2[i], h6, q[i]);
      } else if (integ == DGSQP_INT_RK3) {
        double k3[8];
        dyn_fc_pair(Z, q, ua, us, sd, cd, k1);
        for (int i = 0; i < 8; i++) { k1[i] = k1[i] * h; t[i] = q[i] + k1[i] * 0.5; }
        dyn_fc_pair(Z, t, ua, us, sd, cd, k2);
        for (int i = 0; i < 8; i++) { k2[i] = k2[i] * h; t[i] = q[i] - k1[i] + k2[i] * 2.0; }
        dyn_fc_pair(Z, t, ua, us, sd, cd, k3);
        for (int i = 0; i < 8; i++) q[i] = q[i] + (k1[i] + k2[i] * 4.0 + k3[i] * h) / 6.0;
      } else if (integ == DGSQP_INT_RK2) {
        dyn_fc_pair(Z, q, ua, us, sd, cd, k1);
        for (int i = 0; i < 8; i++) t[i] = q[i] + k1[i] * h;
        dyn_fc_pair(Z, t, ua, us, sd, cd, k2);
        for (int i = 0; i < 8; i++) q[i] = q[i] + (k1[i] + k2[i]) * h2;
      } else {
        dyn_fc_pair(Z, q, ua, us, sd, cd, k1);
        for (int i = 0; i < 8; i++) q[i] = q[i] + k1[i] * P.dt;
      }
    }
    if (role == 0)
      for (int i = 0; i < 8; i++) x[(k + 1) * nq + qo + i] = q[i];
    if (prog) {      // fused rollout + derivative pass: x_{k+1} of every agent is in LDS, tell the other wavefronts
      __threadfence_block();
      if (threadIdx.x == 0) *(vlds_d*)prog = (double)(k + 1);
    }
  }
}
// K trajectories x^(j) = rollout(ub + alpha_j du), alpha_j = alpha0 tau^j, j < K, on K * (lanes per trajectory) lanes of
// wavefront 0 (one instruction stream: K trajectories cost the latency of one).  xs[j] has stride xstride doubles.
__device__ __noinline__ void dev_rollout_multi(const Ctx& c, clptr ub, clptr du, double alpha0, double tau, int K, lptr xs, int xstride,
                                              int K1 = 1 << 30, lptr xs2 = nullptr) {
  const DgProb& D = dg_prob;
  __syncthreads();
  // trajectory j lives at xs + j xstride for j < K1 and at xs2 + (j - K1) xstride beyond
  for (int i = TID; i < K * D.nq; i += NT) { const int j = i / D.nq; (j < K1 ? xs + j * xstride : xs2 + (j - K1) * xstride)[i % D.nq] = c.x0[i % D.nq]; }
  __syncthreads();
  bool all_dyn = D.P.track_kind == DGSQP_TRACK_ARCS;      // the pair rollout caches the arc segment of the track in registers
  for (int a = 0; a < D.M; a++) all_dyn = all_dyn && D.nqa[a] == 8;
  const int per = all_dyn ? 2 * D.M : D.M;
  if (TID < K * per) {
    const int j = TID / per, w = TID % per;
    double alpha = alpha0;
    for (int t = 0; t < j; t++) alpha *= tau;           // same products as the sequential alpha *= tau
    lptr x = j < K1 ? xs + j * xstride : xs2 + (j - K1) * xstride;
    if (all_dyn) dev_rollout_dyn_pair(D, w >> 1, w & 1, ub, du, alpha, x);
    else if (D.P.track_kind == DGSQP_TRACK_SPLINE && D.nqa[w] != 4) {
      if (D.nqa[w] == 8) dev_rollout_agent<8, true>(D, w, ub, du, alpha, x); else dev_rollout_agent<6, true>(D, w, ub, du, alpha, x);
    }
    else if (D.nqa[w] == 8) dev_rollout_agent<8>(D, w, ub, du, alpha, x);
    else if (D.nqa[w] == 4) dev_rollout_agent<4>(D, w, ub, du, alpha, x);
    else dev_rollout_agent<6>(D, w, ub, du, alpha, x);
  }
  __syncthreads();
}
__device__ inline void dev_rollout(const Ctx& c, clptr ue, lptr x) { dev_rollout_multi(c, ue, nullptr, 0.0, 1.0, 1, x, 0); }

// ------------------------------------------------------------------------------------------------
// derivatives of f_d at (x_k, u_k) by truncated Taylor propagation, one (agent, stage, direction)
// item per lane  (fAd/fBd dynamics_models.py:128-133, fEd/fFd/fGd :137-144)
// ------------------------------------------------------------------------------------------------
__device__ inline void dir_pair(int neff, int dir, int& i, int& j) {
  if (dir < neff) { i = j = dir; return; }
  int p = dir - neff;
  i = 0;
  while (p >= neff - 1 - i) { p -= neff - 1 - i; i++; }
  j = i + 1 + p;
}
template <int DEG, int NQA, int INTEG, bool SPL = false>
__device__ __forceinline__ void dev_taylor_item_impl(const Ctx& c, int a, int k, int dir, clptr ue) {
  const DgProb& D = dg_prob;
  const DgLds& L = D.L;
  typedef Ty<DEG> T;
  clptr x = LP(L.e_x + k * D.nq + D.qoff[a]);
  T q[NQA], u[2], out[NQA];
  for (int i = 0; i < NQA; i++) q[i] = ty_const<DEG>(x[i]);
  u[0] = ty_const<DEG>(ue[am_col(D, a, k, 0)]);
  u[1] = ty_const<DEG>(ue[am_col(D, a, k, 1)]);
  int ei, ej;
  dir_pair(D.neff[a], dir, ei, ej);
  {
    // seed direction e_i (+ e_j): written with static indices so that q / u stay in registers
    const int zi = D.effvar[a][ei], zj = ej != ei ? D.effvar[a][ej] : -1;
#pragma unroll
    for (int i = 0; i < NQA; i++) q[i].c[1] = (i == zi || i == zj) ? 1.0 : 0.0;
#pragma unroll
    for (int i = 0; i < 2; i++) u[i].c[1] = (NQA + i == zi || NQA + i == zj) ? 1.0 : 0.0;
  }
  dev_fd_t<DEG, NQA, INTEG, SPL>(D.P, D.P.agents[a], q, u, out);
  if (dir < D.neff[a]) {
    const int z = D.effvar[a][dir];
    if (z < NQA) {
      lptr A = LP(L.e_A[a] + k * NQA * NQA);
      for (int o = 0; o < NQA; o++) A[o * NQA + z] = out[o].c[1];
    } else {
      lptr B = LP(L.e_B[a] + k * NQA * 2);
      for (int o = 0; o < NQA; o++) B[o * 2 + (z - NQA)] = out[o].c[1];
    }
  }
  if constexpr (DEG >= 2) {
    gptr T2 = c.ws + D.ws_t2 + D.t2off[a] + (int64_t)k * D.t2k[a];
    const int nd = D.ndir[a];
    for (int o = 0; o < NQA; o++) T2[o * nd + dir] = out[o].c[2];
  }
}
// First-derivative items are compiled out of line, one instantiation per (model, integrator), which keeps their register
// allocation tight (no spills for euler); the second-order items are inlined into their caller.
template <int DEG, int NQA, int INTEG, bool SPL = false>
__device__ __noinline__ void dev_taylor_item_ool(const Ctx& c, int a, int k, int dir, clptr ue) { dev_taylor_item_impl<DEG, NQA, INTEG, SPL>(c, a, k, dir, ue); }
template <int DEG, int NQA, int INTEG>
__device__ __forceinline__ void dev_taylor_item(const Ctx& c, int a, int k, int dir, clptr ue) {
  // games on a spline track: their own out-of-line instantiations (the arc-track ones keep their register allocation)
  if (NQA != 4 && dg_prob.P.track_kind == DGSQP_TRACK_SPLINE) { dev_taylor_item_ool<DEG, NQA, INTEG, true>(c, a, k, dir, ue); return; }
  if constexpr (DEG >= 2 && NQA == 8 && INTEG != DGSQP_INT_EULER) dev_taylor_item_impl<DEG, NQA, INTEG>(c, a, k, dir, ue);
  else dev_taylor_item_ool<DEG, NQA, INTEG>(c, a, k, dir, ue);
}
template <int DEG, int INTEG>
__device__ __forceinline__ void dev_taylor_item_nqa(const Ctx& c, int nqa, int a, int k, int dir, clptr ue) {
  if (nqa == 8) dev_taylor_item<DEG, 8, INTEG>(c, a, k, dir, ue);
  else if (nqa == 4) dev_taylor_item<DEG, 4, INTEG>(c, a, k, dir, ue);
  else dev_taylor_item<DEG, 6, INTEG>(c, a, k, dir, ue);
}
template <int DEG>
__device__ __noinline__ void dev_dyn_derivs(const Ctx& c, clptr ue) {
  const DgProb& D = dg_prob;
  const DgLds& L = D.L;
  for (int a = 0; a < D.M; a++) {
    const int nqa = D.nqa[a];
    // columns of x, y: identity (they never enter fc)
    for (int it = TID; it < D.N * nqa; it += NT) {
      const int k = it / nqa, o = it % nqa;
      lptr A = LP(L.e_A[a] + k * nqa * nqa);
      A[o * nqa + 0] = (o == 0) ? 1.0 : 0.0;
      A[o * nqa + 1] = (o == 1) ? 1.0 : 0.0;
    }
  }
  // Work is handed out in wave-tasks of 64 items of ONE agent (the agent's constants then live in scalar registers),
  // round-robin over the wavefronts: all agents are in flight together instead of one 200-item pass per agent.
  // Measured on MI355X: the second-order pass of the dynamic bicycle under a multi-stage integrator spills ~1.2 KB per
  // lane and runs 1.6x faster when all wavefronts sweep one agent at a time; every other case prefers the wave-tasks below.
  bool heavy = DEG >= 2 && D.P.integrator != DGSQP_INT_EULER;
  for (int a = 0; a < D.M; a++) heavy = heavy && D.nqa[a] == 8;
  if (heavy) {
    for (int a = 0; a < D.M; a++) {
      const int nd = (DEG >= 2) ? D.ndir[a] : D.neff[a];
      for (int it = TID; it < D.N * nd; it += NT) {
        const int k = it / nd, dir = it % nd;
        switch (D.P.integrator) {
          case DGSQP_INT_EULER: dev_taylor_item_nqa<DEG, DGSQP_INT_EULER>(c, D.nqa[a], a, k, dir, ue); break;
          case DGSQP_INT_RK4: dev_taylor_item_nqa<DEG, DGSQP_INT_RK4>(c, D.nqa[a], a, k, dir, ue); break;
          case DGSQP_INT_RK3: dev_taylor_item_nqa<DEG, DGSQP_INT_RK3>(c, D.nqa[a], a, k, dir, ue); break;
          default: dev_taylor_item_nqa<DEG, DGSQP_INT_RK2>(c, D.nqa[a], a, k, dir, ue); break;
        }
      }
    }
  } else {
    const int wave = TID >> 6, lane = TID & 63;
    int ntask = 0;
    for (int a = 0; a < D.M; a++) ntask += (D.N * ((DEG >= 2) ? D.ndir[a] : D.neff[a]) + 63) >> 6;
    for (int t = wave; t < ntask; t += NT / 64) {
      int a = 0, rem = t;
      for (;; a++) {
        const int nt = (D.N * ((DEG >= 2) ? D.ndir[a] : D.neff[a]) + 63) >> 6;
        if (rem < nt) break;
        rem -= nt;
      }
      const int nd = (DEG >= 2) ? D.ndir[a] : D.neff[a];
      const int it = rem * 64 + lane;
      if (it < D.N * nd) {
        const int k = it / nd, dir = it % nd;
        switch (D.P.integrator) {   // one out-of-line instantiation per (model, integrator): registers are allocated per variant
          case DGSQP_INT_EULER: dev_taylor_item_nqa<DEG, DGSQP_INT_EULER>(c, D.nqa[a], a, k, dir, ue); break;
          case DGSQP_INT_RK4: dev_taylor_item_nqa<DEG, DGSQP_INT_RK4>(c, D.nqa[a], a, k, dir, ue); break;
          case DGSQP_INT_RK3: dev_taylor_item_nqa<DEG, DGSQP_INT_RK3>(c, D.nqa[a], a, k, dir, ue); break;
          default: dev_taylor_item_nqa<DEG, DGSQP_INT_RK2>(c, D.nqa[a], a, k, dir, ue); break;
        }
      }
    }
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// Fused rollout + derivative pass (dynamic bicycles under a multi-stage integrator).  The rollout is one dependent chain of
// N x substeps x stages f_c evaluations on a handful of lanes of wavefront 0 (~1.7 M cycles at N = 25, rk4, M = 10) during
// which the other seven wavefronts of the scenario would sit at a barrier; the derivative items of stage k need nothing but
// x_k.  Wavefront 0 publishes the number of finished stages in an LDS slot; the other wavefronts pull 64-item tasks in
// stage order from an LDS ticket and start on a task as soon as its last stage is there; wavefront 0 joins them when its
// rollout is done.  The second-order items also deliver A_k, B_k, so a point evaluated this way needs no further derivative
// pass (tag 2.0 in scal[DG_XVALID]) -- neither for a trial merit nor for the Hessian of the next linearisation at the same
// point (the Taylor tensor does not depend on the multipliers).  Same arithmetic as dev_rollout + dev_dyn_derivs<2>.
// ------------------------------------------------------------------------------------------------
#define DG_PROG 54
#define DG_TASK 53
__device__ inline bool dev_can_fuse_rollout() {
  const DgProb& D = dg_prob;
  bool ok = D.P.integrator != DGSQP_INT_EULER && 2 * D.M <= 64 && D.P.track_kind == DGSQP_TRACK_ARCS;
  for (int a = 0; a < D.M; a++) ok = ok && D.nqa[a] == 8 && D.ndir[a] == D.ndir[0];
  return ok;
}
__device__ __noinline__ void dev_rollout_with_derivs(const Ctx& c, clptr ue, lptr x) {
  const DgProb& D = dg_prob;
  const DgLds& L = D.L;
  lptr sc = LP(L.scal);
  __syncthreads();
  for (int i = TID; i < D.nq; i += NT) x[i] = c.x0[i];
  if (TID == 0) { sc[DG_PROG] = 0.0; sc[DG_TASK] = 0.0; }
  for (int a = 0; a < D.M; a++) {
    const int nqa = D.nqa[a];
    for (int it = TID; it < D.N * nqa; it += NT) {       // columns of x, y: identity (they never enter fc)
      const int k = it / nqa, o = it % nqa;
      lptr A = LP(L.e_A[a] + k * nqa * nqa);
      A[o * nqa + 0] = (o == 0) ? 1.0 : 0.0;
      A[o * nqa + 1] = (o == 1) ? 1.0 : 0.0;
    }
  }
  __syncthreads();
  const int wave = TID >> 6, lane = TID & 63;
  if (wave == 0 && lane < 2 * D.M) dev_rollout_dyn_pair(D, lane >> 1, lane & 1, ue, nullptr, 0.0, x, sc + DG_PROG);
  const int nd = D.ndir[0], per_agent = (D.N * nd + 63) >> 6, ntask = per_agent * D.M;
  __attribute__((address_space(3))) unsigned int* ticket = (__attribute__((address_space(3))) unsigned int*)(sc + DG_TASK);
  while (true) {
    unsigned int t = 0;
    if (lane == 0) t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    t = __builtin_amdgcn_readfirstlane(t);
    if ((int)t >= ntask) break;
    const int chunk = (int)t / D.M, a = (int)t % D.M;      // stage-major order, the agents alternate
    const int last = chunk * 64 + 63;
    const int kmax = last / nd < D.N - 1 ? last / nd : D.N - 1;
    while ((int)*(vlds_d*)(sc + DG_PROG) < kmax) __builtin_amdgcn_s_sleep(8);
    __threadfence_block();
    const int it = chunk * 64 + lane;
    if (it < D.N * nd) {
      const int k = it / nd, dir = it % nd;
      switch (D.P.integrator) {
        case DGSQP_INT_RK4: dev_taylor_item_nqa<2, DGSQP_INT_RK4>(c, 8, a, k, dir, ue); break;
        case DGSQP_INT_RK3: dev_taylor_item_nqa<2, DGSQP_INT_RK3>(c, 8, a, k, dir, ue); break;
        default: dev_taylor_item_nqa<2, DGSQP_INT_RK2>(c, 8, a, k, dir, ue); break;
      }
    }
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// sensitivity chains S^a_{k,t0}[:, j] = A_{k-1} ... A_{t0+1} B_{t0}[:, j]  (f_Du_x, DGSQP.py:642-650),
// consumed on the fly into the packed dense gradients (f_Du_C :823-826) and q (f_q :672-676, 898-899)
// ------------------------------------------------------------------------------------------------
// one entry of the packed constraint gradients (LDS, or the global scratch for games whose gradients exceed the arena)
__device__ inline void gd_store(const Ctx& c, int idx, double val) {
  if (dg_prob.gd_global) (c.ws + dg_prob.ws_gd)[idx] = val; else LP(dg_prob.L.gd)[idx] = val;
}
template <int NQA>
__device__ inline void dev_chain_item(const Ctx& c, clptr ue, int it);
__device__ __noinline__ void dev_chains(const Ctx& c, clptr ue) {
  const DgProb& D = dg_prob;
  const DgLds& L = D.L;
  lptr lds = LP(0);
  clptr x = lds + L.e_x;
  // own-state gradient of each agent's cost at every stage
  for (int it = TID; it < D.M * (D.N + 1); it += NT) {
    const int a = it / (D.N + 1), k = it % (D.N + 1);
    double Dx[DGSQP_MAX_AGENTS * DGSQP_MAX_NQA];
    for (int i = 0; i < D.nq; i++) Dx[i] = 0;
    dev_state_cost(D, a, x + k * D.nq, k == D.N, (double*)Dx, (double*)nullptr);
    for (int i = 0; i < D.nqa[a]; i++) lds[L.e_dJ + k * D.nq + D.qoff[a] + i] = Dx[D.qoff[a] + i];
  }
  __syncthreads();
  for (int it = TID; it < D.n; it += NT) {
    const int a = it / (D.N * DGSQP_NUA);
    if (D.nqa[a] == 8) dev_chain_item<8>(c, ue, it); else if (D.nqa[a] == 4) dev_chain_item<4>(c, ue, it); else dev_chain_item<6>(c, ue, it);
  }
  __syncthreads();
}
template <int NQA>
__device__ inline void dev_chain_item(const Ctx& c, clptr ue, int it) {
  const DgProb& D = dg_prob;
  const DgLds& L = D.L;
  lptr lds = LP(0);
  clptr x = lds + L.e_x;
  {
    const int a = it / (D.N * DGSQP_NUA), rem = it % (D.N * DGSQP_NUA), t0 = rem / DGSQP_NUA, j = rem % DGSQP_NUA;
    constexpr int nqa = NQA;
    const int qo = D.qoff[a];
    const dgsqp_agent_t& ag = D.P.agents[a];
    double v[NQA], w[NQA];
    clptr B = lds + L.e_B[a] + t0 * nqa * 2;
#pragma unroll
    for (int i = 0; i < nqa; i++) v[i] = B[i * 2 + j];
    // direct part of dJ^a/du^a_{t0,j}
    const double uk = ue[it], um = t0 > 0 ? ue[it - DGSQP_NUA] : 0.0;
    double qacc = ag.w_in[j] * uk + ag.w_rate[j] * (uk - um);
    if (t0 + 1 < D.N) qacc -= ag.w_rate[j] * (ue[it + DGSQP_NUA] - uk);
    int d = D.stage_dense0[t0 + 1];  // dense gradients are ordered by stage
    for (int k = t0 + 1; k <= D.N; k++) {
      clptr xk = x + k * D.nq;
      for (; d < D.stage_dense0[k + 1]; d++) {
        const DgDense dd = ld_dense(d);
        if (dd.kind == 0) {
          if (dd.a == a) {
            double val = 0.0;
#pragma unroll
            for (int i = 0; i < nqa; i++) val = (i == dd.idx) ? v[i] : val;  // keeps v[] in registers
            gd_store(c, dd.off + t0 * DGSQP_NUA + j, val);
          }
        } else if (dd.kind == 2) {
          if (dd.a == a) {     // lane half-plane: n(p_x) . d p_k / du, the normal is piecewise constant in p_x (merge.py:66-74)
            const auto& ln = ag.lane[dd.idx];
            const bool hi = xk[qo] >= ln.brk;
            gd_store(c, dd.off + t0 * DGSQP_NUA + j, (hi ? ln.n_hi[0] : ln.n_lo[0]) * v[0] + (hi ? ln.n_hi[1] : ln.n_lo[1]) * v[1]);
          }
        } else if (dd.a == a || dd.b == a) {
          const int ia = D.qoff[dd.a], ib = D.qoff[dd.b];
          const double dx = xk[ia] - xk[ib], dy = xk[ia + 1] - xk[ib + 1];
          const double s = 2.0 * (dx * v[0] + dy * v[1]);
          if (dd.a == a) gd_store(c, dd.off + t0 * DGSQP_NUA + j, -s);
          else gd_store(c, dd.off + 2 * k + t0 * DGSQP_NUA + j, s);
        }
      }
      clptr dJ = lds + L.e_dJ + k * D.nq + qo;
#pragma unroll
      for (int i = 0; i < nqa; i++) qacc += dJ[i] * v[i];
      if (k < D.N) {
        clptr A = lds + L.e_A[a] + k * nqa * nqa;
#pragma unroll
        for (int i = 0; i < nqa; i++) {
          double s = 0;
#pragma unroll
          for (int m = 0; m < nqa; m++) s += A[i * nqa + m] * v[m];
          w[i] = s;
        }
#pragma unroll
        for (int i = 0; i < nqa; i++) v[i] = w[i];
      }
    }
    lds[L.q + it] = qacc;
  }
}

// constraint values (f_Cxu, DGSQP.py:729-821, 911)
__device__ __noinline__ void dev_constraint_values(const Ctx& c, clptr ue) {
  const DgProb& D = dg_prob;
  const DgLds& L = D.L;
  clptr x = LP(L.e_x);
  for (int r = TID; r < D.nc; r += NT) {
    const DgRow R = ld_row(r);
    const dgsqp_agent_t& ag = D.P.agents[R.a];
    clptr xk = x + R.k * D.nq;
    double g;
    switch (R.type) {
      case DG_R_OBS: {
        const int ia = D.qoff[R.a], ib = D.qoff[R.b];
        const double dx = xk[ia] - xk[ib], dy = xk[ia + 1] - xk[ib + 1], dd = ag.radius + D.P.agents[R.b].radius;
        g = dd * dd - (dx * dx + dy * dy);
      } break;
      case DG_R_RATE_UB:
      case DG_R_RATE_LB: {
        const int col = am_col(D, R.a, R.k, R.idx);
        const double du = ue[col] - (R.k > 0 ? ue[col - DGSQP_NUA] : 0.0);
        g = R.type == DG_R_RATE_UB ? du - D.P.dt * ag.rate_ub[R.idx] : D.P.dt * ag.rate_lb[R.idx] - du;
      } break;
      case DG_R_IN_UB: g = ue[am_col(D, R.a, R.k, R.idx)] - ag.in_ub[R.idx]; break;
      case DG_R_IN_LB: g = ag.in_lb[R.idx] - ue[am_col(D, R.a, R.k, R.idx)]; break;
      case DG_R_ST_UB: g = xk[D.qoff[R.a] + R.idx] - ag.st_ub[R.idx]; break;
      case DG_R_LANE: {
        const auto& ln = ag.lane[R.idx];
        const int ia = D.qoff[R.a];
        const bool hi = xk[ia] >= ln.brk;
        const double nx = hi ? ln.n_hi[0] : ln.n_lo[0], ny = hi ? ln.n_hi[1] : ln.n_lo[1];
        g = nx * (xk[ia] - (ln.anchor[0] - ln.r * nx)) + ny * (xk[ia + 1] - (ln.anchor[1] - ln.r * ny));
      } break;
      default: g = ag.st_lb[R.idx] - xk[D.qoff[R.a] + R.idx]; break;
    }
    LP(L.g)[r] = g;
  }
  __syncthreads();
}

__device__ inline int dev_block_of(const DgProb& D, int xi) {
  int b = 0;
  while (b + 1 < D.M && xi >= D.qoff[b + 1]) b++;
  return b;
}

// generic (mixed-model) variant: run-time block offsets, arrays end up in scratch memory
// KG: the compact state-Hessian columns are read from the global scratch (DgProb.tab_const)
template <int NQA, bool KG = false>
__device__ __noinline__ void dev_hessian_row_generic(const Ctx& c, int row, int a, int k0, int j0) {
  const DgProb& D = dg_prob;
  const DgLds& L = D.L;
  lptr lds = LP(0);
  const int nq = D.nq, n = D.n, N = D.N, M = D.M;
  typename std::conditional<KG, cgptr, clptr>::type Kc;
  if constexpr (KG) Kc = c.ws + D.ws_K; else Kc = lds + L.e_K;
  cgptr Hg = c.ws + D.ws_H;
  gptr tang = c.ws + D.ws_tang;
  gptr Qrow = c.ws + D.ws_q + (int64_t)row * n;
  const dgsqp_agent_t& ag = D.P.agents[a];
  constexpr int EE = DG_MAXEFF * DG_MAXEFF;
  constexpr int NE = NQA;            // effective variables: states 2..NQA-1, then the two inputs
  const int qa = D.qoff[a];
  // forward tangent dx_t (block a only), t = k0+1 .. N, stored to the per-workgroup scratch (coalesced over lanes)
  double v[NQA], w[NQA];
  {
    clptr B = lds + L.e_B[a] + k0 * NQA * 2;
#pragma unroll
    for (int i = 0; i < NQA; i++) v[i] = B[i * 2 + j0];
    for (int t = k0 + 1; t <= N; t++) {
#pragma unroll
      for (int i = 0; i < NQA; i++) tang[((int64_t)t * DGSQP_MAX_NQA + i) * n + row] = v[i];
      if (t < N) {
        clptr A = lds + L.e_A[a] + t * NQA * NQA;
#pragma unroll
        for (int i = 0; i < NQA; i++) {
          double s = 0;
#pragma unroll
          for (int m = 0; m < NQA; m++) s += A[i * NQA + m] * v[m];
          w[i] = s;
        }
#pragma unroll
        for (int i = 0; i < NQA; i++) v[i] = w[i];
      }
    }
  }
  // backward second-order adjoint; mu is joint (n_q) because the state Hessian couples the agents
  double mu[DGSQP_MAX_AGENTS * DGSQP_MAX_NQA];
  // mu_N = L_xx,N dx_N  (v holds dx_N)
  for (int i = 0; i < nq; i++) mu[i] = 0.0;
  {
    const auto K = Kc + (a * (N + 1) + N) * M * 5;
    for (int b = 0; b < M; b++) {
      const int qb = D.qoff[b];
      mu[qb + 0] += K[b * 5 + 0] * v[0] + K[b * 5 + 1] * v[1];
      mu[qb + 1] += K[b * 5 + 1] * v[0] + K[b * 5 + 2] * v[1];
      mu[qb + D.eyidx[b]] += K[b * 5 + 3] * v[NQA - 1];
      mu[qb + D.sidx[b]] += K[b * 5 + 4] * v[NQA - 2];
    }
  }
  for (int t = N - 1; t >= 0; t--) {
    // dx_t (zero for t <= k0)
    if (t > k0) {
#pragma unroll
      for (int i = 0; i < NQA; i++) v[i] = tang[((int64_t)t * DGSQP_MAX_NQA + i) * n + row];
    } else {
#pragma unroll
      for (int i = 0; i < NQA; i++) v[i] = 0.0;
    }
    const double du = (t == k0) ? 1.0 : 0.0;
    // z = (dx_t effective part, du): effective variable e <-> state e+2 for e < NQA-2, inputs after
    cgptr Ha = Hg + (((int64_t)a * N + t) * M + a) * EE;
    double hz[NE];
#pragma unroll
    for (int e = 0; e < NE; e++) {
      double s = 0;
#pragma unroll
      for (int f = 0; f < NQA - 2; f++) s += Ha[e * DG_MAXEFF + f] * v[f + 2];
      s += Ha[e * DG_MAXEFF + (NQA - 2 + j0)] * du;
      hz[e] = s;
    }
    // row entries of stage t: B_t^T mu_{t+1} (+ H_u. z + L_uu du for agent a's own inputs)
    for (int b = 0; b < M; b++) {
      const int nqb = D.nqa[b], qb = D.qoff[b];
      clptr B = lds + L.e_B[b] + t * nqb * 2;
      for (int j = 0; j < DGSQP_NUA; j++) {
        double s = 0;
        for (int m = 0; m < nqb; m++) s += B[m * 2 + j] * mu[qb + m];
        if (b == a) {
          s += hz[NQA - 2 + j];
          if (j == j0) {
            if (t == k0) s += ag.w_in[j] + ag.w_rate[j] + (t + 1 < N ? ag.w_rate[j] : 0.0);
            else if (t == k0 + 1 || t == k0 - 1) s -= ag.w_rate[j];
          }
        }
        Qrow[am_col(D, b, t, j)] = s;
      }
    }
    if (t > 0) {
      // mu_t = A_t^T mu_{t+1} + L_xx,t dx_t + (H_xx dx_t + H_xu du) on block a
      double nm[DGSQP_MAX_AGENTS * DGSQP_MAX_NQA];
      for (int b = 0; b < M; b++) {
        const int nqb = D.nqa[b], qb = D.qoff[b];
        clptr A = lds + L.e_A[b] + t * nqb * nqb;
        for (int i = 0; i < nqb; i++) {
          double s = 0;
          for (int m = 0; m < nqb; m++) s += A[m * nqb + i] * mu[qb + m];
          nm[qb + i] = s;
        }
      }
      if (t > k0) {
        const auto K = Kc + (a * (N + 1) + t) * M * 5;
        for (int b = 0; b < M; b++) {
          const int qb = D.qoff[b];
          nm[qb + 0] += K[b * 5 + 0] * v[0] + K[b * 5 + 1] * v[1];
          nm[qb + 1] += K[b * 5 + 1] * v[0] + K[b * 5 + 2] * v[1];
          nm[qb + D.eyidx[b]] += K[b * 5 + 3] * v[NQA - 1];
          nm[qb + D.sidx[b]] += K[b * 5 + 4] * v[NQA - 2];
        }
      }
#pragma unroll
      for (int f = 0; f < NQA - 2; f++) nm[qa + f + 2] += hz[f];
      for (int i = 0; i < nq; i++) mu[i] = nm[i];
    }
  }
}

// uniform-model variant (every agent has NQA states): all register arrays are statically indexed
template <int NQA, int MM>
__device__ __noinline__ void dev_hessian_row(const Ctx& c, int row, int a, int k0, int j0) {
  const DgProb& D = dg_prob;
  const DgLds& L = D.L;
  lptr lds = LP(0);
  const int n = D.n, N = D.N;
  clptr Kc = lds + L.e_K;
  cgptr Hg = c.ws + D.ws_H;
  gptr tang = c.ws + D.ws_tang;
  gptr Qrow = c.ws + D.ws_q + (int64_t)row * n;
  const dgsqp_agent_t& ag = D.P.agents[a];
  constexpr int EE = DG_MAXEFF * DG_MAXEFF;
  constexpr int NU = DGSQP_NUA;
  const double wr = ag.w_rate[j0], win = ag.w_in[j0];
  double v[NQA], w[NQA];
  {
    clptr B = lds + L.e_B[a] + k0 * NQA * NU;
#pragma unroll
    for (int i = 0; i < NQA; i++) v[i] = B[i * NU + j0];
    for (int t = k0 + 1; t <= N; t++) {
#pragma unroll
      for (int i = 0; i < NQA; i++) tang[((int64_t)t * DGSQP_MAX_NQA + i) * n + row] = v[i];
      if (t < N) {
        clptr A = lds + L.e_A[a] + t * NQA * NQA;
#pragma unroll
        for (int i = 0; i < NQA; i++) {
          double s = 0;
#pragma unroll
          for (int m = 0; m < NQA; m++) s += A[i * NQA + m] * v[m];
          w[i] = s;
        }
#pragma unroll
        for (int i = 0; i < NQA; i++) v[i] = w[i];
      }
    }
  }
  double mu[MM][NQA];
  {
    clptr K = Kc + (a * (N + 1) + N) * MM * 5;
#pragma unroll
    for (int b = 0; b < MM; b++) {
#pragma unroll
      for (int i = 0; i < NQA; i++) mu[b][i] = 0.0;
      mu[b][0] = K[b * 5 + 0] * v[0] + K[b * 5 + 1] * v[1];
      mu[b][1] = K[b * 5 + 1] * v[0] + K[b * 5 + 2] * v[1];
      mu[b][NQA - 1] = K[b * 5 + 3] * v[NQA - 1];
      mu[b][NQA - 2] = K[b * 5 + 4] * v[NQA - 2];
    }
  }
  // own-block offset in the agent-major column order
  for (int t = N - 1; t >= 0; t--) {
    const bool live = t > k0;
#pragma unroll
    for (int i = 0; i < NQA; i++) v[i] = live ? tang[((int64_t)t * DGSQP_MAX_NQA + i) * n + row] : 0.0;
    const double du = (t == k0) ? 1.0 : 0.0;
    cgptr Ha = Hg + (((int64_t)a * N + t) * MM + a) * EE;
    double hz[NQA];
#pragma unroll
    for (int e = 0; e < NQA; e++) {
      double s = 0;
#pragma unroll
      for (int f = 0; f < NQA - 2; f++) s += Ha[e * DG_MAXEFF + f] * v[f + 2];
      s += Ha[e * DG_MAXEFF + (NQA - 2 + j0)] * du;
      hz[e] = s;
    }
    double dir0 = 0.0;   // agent a's own L_uu entry on input j0 at this stage
    if (t == k0) dir0 = win + wr + (t + 1 < N ? wr : 0.0);
    else if (t == k0 + 1 || t == k0 - 1) dir0 = -wr;
#pragma unroll
    for (int b = 0; b < MM; b++) {
      clptr B = lds + L.e_B[b] + t * NQA * NU;
#pragma unroll
      for (int j = 0; j < NU; j++) {
        double s = 0;
#pragma unroll
        for (int m = 0; m < NQA; m++) s += B[m * NU + j] * mu[b][m];
        if (b == a) s += (j == 0 ? hz[NQA - 2] : hz[NQA - 1]) + (j == j0 ? dir0 : 0.0);
        Qrow[(b * N + t) * NU + j] = s;
      }
    }
    if (t > 0) {
      clptr K = Kc + (a * (N + 1) + t) * MM * 5;
#pragma unroll
      for (int b = 0; b < MM; b++) {
        clptr A = lds + L.e_A[b] + t * NQA * NQA;
        double nm[NQA];
#pragma unroll
        for (int i = 0; i < NQA; i++) {
          double s = 0;
#pragma unroll
          for (int m = 0; m < NQA; m++) s += A[m * NQA + i] * mu[b][m];
          nm[i] = s;
        }
        // state-Hessian columns (v is zero when t <= k0, so no branch is needed)
        nm[0] += K[b * 5 + 0] * v[0] + K[b * 5 + 1] * v[1];
        nm[1] += K[b * 5 + 1] * v[0] + K[b * 5 + 2] * v[1];
        nm[NQA - 1] += K[b * 5 + 3] * v[NQA - 1];
        nm[NQA - 2] += K[b * 5 + 4] * v[NQA - 2];
        const double own = b == a ? 1.0 : 0.0;
#pragma unroll
        for (int f = 0; f < NQA - 2; f++) nm[f + 2] += own * hz[f];
#pragma unroll
        for (int i = 0; i < NQA; i++) mu[b][i] = nm[i];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Game Hessian by second-order adjoints.  Row i of Q (i an input of agent a) is the gradient of
// dL^a/du_i, i.e. one Hessian-vector product of agent a's Lagrangian L^a = J^a + l^T C with e_i:
//   costates      lam_k  = L_x,k + A_k^T lam_{k+1}                                    (once per agent)
//   H_k^b         = sum_o lam_{k+1}[b,o] grad^2 f^b_{k,o}   (Taylor tensor contracted with the costate, all (a,k,b) in parallel)
//   tangent       dx_{t+1} = A_t dx_t + B_t e_i                                        (one lane per i)
//   2nd adjoint   mu_t   = A_t^T mu_{t+1} + (L_xx,t + H_xx,t) dx_t + H_xu,t du_t
//   row entries   Q[i, (b,t,j)] = B_t^T mu_{t+1} + H_ux,t dx_t + (H_uu,t + L_uu) du_t
// Mathematically identical to the reference's backward DP (DGSQP.py:679-727, :828-877, f_Q :920-934) and to its
// own f_Duu_L (:937-941); here all N*n_u rows are independent lanes with 25 uniform steps each.
// ------------------------------------------------------------------------------------------------
// Costates of every agent's Lagrangian L^a = J^a + lm^T C along the current trajectory (multipliers lm in LDS):
//   lam^a_N = L^a_x,N ,  lam^a_k = L^a_x,k + A_k^T lam^a_{k+1}     (joint n_q vectors, e_lam[a][k][.])
// Stage 0 also leaves the compact state-Hessian columns (e_K) the second-order pass needs.  Used by the game Hessian
// and by the merit of a line-search trial point (gradient of the Lagrangians without forming G).
__device__ __noinline__ void dev_costates(const Ctx& c, clptr lm) {
  const DgProb& D = dg_prob;
  const DgLds& L = D.L;
  lptr lds = LP(0);
  const int nq = D.nq, N = D.N, M = D.M;
  clptr x = lds + L.e_x;
  lptr lam = lds + L.e_lam;   // [a][k][nq]
  lptr Dxs = lds + L.e_Dxs;   // [a][k][nq]   d/dx_k of (stage cost a + sum_r l_r c_r)
  lptr Kc = lds + L.e_K;      // [a][k][b][5] columns (block a) of the state Hessian: Kxx, Kxy, Kyy on positions, k_ey, k_s
  // ---- 0. first / second state derivatives of every agent's stage Lagrangian
  // (accumulated IN PLACE in their destinations -- every (agent, stage) belongs to one thread: as run-time indexed local arrays the two
  // vectors lived in scratch memory, 57 scratch accesses inside the loops of a function every line-search trial calls)
  auto stage = [&](int a, int k, lptr Dx, auto Kl) {
    for (int i = 0; i < nq; i++) Dx[i] = 0.0;
    for (int i = 0; i < M * 5; i++) Kl[i] = 0.0;
    clptr xk = x + k * nq;
    const dgsqp_agent_t& ag = D.P.agents[a];
    const int ia = D.qoff[a];
    // cost part (same terms as dev_state_cost, kept in the compact column form)
    for (int i = 0; i < D.nqa[a]; i++)
      if (ag.w_goal[i] != 0.0) {   // weights sit on x, y and the last two states only (checked at create)
        const double w = (k == N ? ag.goal_term_mult : 1.0) * ag.w_goal[i];
        Dx[ia + i] += w * (xk[ia + i] - ag.goal[i]);
        Kl[a * 5 + (i == 0 ? 0 : (i == 1 ? 2 : (i == D.nqa[a] - 1 ? 3 : 4)))] += w;
      }
    for (int b = 0; b < M; b++) {
      if (b == a) continue;
      const int ib = D.qoff[b];
      if (ag.w_block != 0.0) {
        const int ea = ia + D.eyidx[a], eb = ib + D.eyidx[b];
        const double dd = xk[ea] - xk[eb];
        Dx[ea] += ag.w_block * dd; Dx[eb] -= ag.w_block * dd;
        Kl[a * 5 + 3] += ag.w_block; Kl[b * 5 + 3] -= ag.w_block;
      }
      if (ag.w_obs != 0.0) {
        const double dx = xk[ia] - xk[ib], dy = xk[ia + 1] - xk[ib + 1];
        const double r = sqrt(dx * dx + dy * dy);
        const double z = (ag.obs_cost_r + D.P.agents[b].obs_cost_r) - r;
        if (z > 0) {
          const double ex = dx / r, ey = dy / r;
          Dx[ia] -= ag.w_obs * z * ex; Dx[ia + 1] -= ag.w_obs * z * ey; Dx[ib] += ag.w_obs * z * ex; Dx[ib + 1] += ag.w_obs * z * ey;
          const double hxx = ag.w_obs * (ex * ex - z * (1 - ex * ex) / r), hxy = ag.w_obs * (ex * ey + z * ex * ey / r), hyy = ag.w_obs * (ey * ey - z * (1 - ey * ey) / r);
          Kl[a * 5 + 0] += hxx; Kl[a * 5 + 1] += hxy; Kl[a * 5 + 2] += hyy;
          Kl[b * 5 + 0] -= hxx; Kl[b * 5 + 1] -= hxy; Kl[b * 5 + 2] -= hyy;
        }
      }
    }
    if (k == N) {
      const int sa = ia + D.sidx[a];
      Dx[sa] -= ag.w_prog;
      for (int b = 0; b < M; b++) {
        if (b == a) continue;
        const int sb = D.qoff[b] + D.sidx[b];
        const double dl = xk[sb] - xk[sa];
        if (ag.comp_type == DGSQP_COMP_ATAN) {
          const double w = 1.0 + dl * dl;
          Dx[sb] += ag.w_comp / w; Dx[sa] -= ag.w_comp / w;
          const double f2 = -2.0 * ag.w_comp * dl / (w * w);
          Kl[a * 5 + 4] += f2; Kl[b * 5 + 4] -= f2;
        } else { Dx[sb] += ag.w_comp; Dx[sa] -= ag.w_comp; }
      }
    }
    // constraint part: rows of stage k (obstacle and state rows only)
    for (int r = D.stage_row0[k]; r < D.stage_row0[k + 1]; r++) {
      const DgRow R = ld_row(r);
      if (R.dense < 0) continue;
      const double lr = lm[r];
      if (R.type == DG_R_OBS) {
        const int ip = D.qoff[R.a], iq = D.qoff[R.b];
        const double dx = xk[ip] - xk[iq], dy = xk[ip + 1] - xk[iq + 1];
        Dx[ip] -= 2 * lr * dx; Dx[ip + 1] -= 2 * lr * dy; Dx[iq] += 2 * lr * dx; Dx[iq + 1] += 2 * lr * dy;
        if (R.a == a || R.b == a) {
          const int other = R.a == a ? R.b : R.a;
          Kl[a * 5 + 0] -= 2 * lr; Kl[a * 5 + 2] -= 2 * lr;
          Kl[other * 5 + 0] += 2 * lr; Kl[other * 5 + 2] += 2 * lr;
        }
      } else if (R.type == DG_R_ST_UB) Dx[D.qoff[R.a] + R.idx] += lr;
      else if (R.type == DG_R_LANE) {
        const auto& ln = D.P.agents[R.a].lane[R.idx];
        const int ip = D.qoff[R.a];
        const bool hi = xk[ip] >= ln.brk;
        Dx[ip] += lr * (hi ? ln.n_hi[0] : ln.n_lo[0]); Dx[ip + 1] += lr * (hi ? ln.n_hi[1] : ln.n_lo[1]);
      } else Dx[D.qoff[R.a] + R.idx] -= lr;
    }
  };
  for (int it = TID; it < M * (N + 1); it += NT) {
    const int a = it / (N + 1), k = it % (N + 1);
    if (D.tab_const) stage(a, k, Dxs + (a * (N + 1) + k) * nq, (gptr)(c.ws + D.ws_K) + (a * (N + 1) + k) * M * 5);
    else stage(a, k, Dxs + (a * (N + 1) + k) * nq, Kc + (a * (N + 1) + k) * M * 5);
  }
  if (D.tab_const) __threadfence_block();
  __syncthreads();
  // ---- 1. costates: one wavefront per agent, lane = state component (in place where e_Dxs shares e_lam's slot)
  {
    const int a = TID >> 6, i = TID & 63;
    if (a < M) {
      int b = 0;
      if (i < nq) b = dev_block_of(D, i);
      const int nqa = D.nqa[b], qo = D.qoff[b];
      double cur = i < nq ? Dxs[(a * (N + 1) + N) * nq + i] : 0.0;
      if (i < nq) lam[(a * (N + 1) + N) * nq + i] = cur;
      for (int k = N - 1; k >= 1; k--) {
        // lam_k[i] = Dx_k[i] + sum_m A_k[m][i] lam_{k+1}[m]   (A block diagonal: m in the block of i)
        double s = i < nq ? Dxs[(a * (N + 1) + k) * nq + i] : 0.0;
        if (i < nq) {
          clptr A = lds + L.e_A[b] + k * nqa * nqa;
          clptr ln = lam + (a * (N + 1) + k + 1) * nq + qo;
          for (int m = 0; m < nqa; m++) s += A[m * nqa + (i - qo)] * ln[m];
          lam[(a * (N + 1) + k) * nq + i] = s;
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
  __syncthreads();
}

__device__ __noinline__ void dev_hessian_adjoint(const Ctx& c) {
  const DgProb& D = dg_prob;
  const DgLds& L = D.L;
  lptr lds = LP(0);
  const int nq = D.nq, n = D.n, N = D.N, M = D.M;
  clptr x = lds + L.e_x;
  lptr lam = lds + L.e_lam;   // [a][k][nq]
  lptr Dxs = lds + L.e_Dxs;   // [a][k][nq]   d/dx_k of (stage cost a + sum_r l_r c_r)
  lptr Kc = lds + L.e_K;      // [a][k][b][5] columns (block a) of the state Hessian: Kxx, Kxy, Kyy on positions, k_ey, k_s
  gptr Hg = c.ws + D.ws_H;    // [a][k][b][MAXEFF*MAXEFF]
  gptr tang = c.ws + D.ws_tang;  // [t][i][lane]
  gptr Qg = c.ws + D.ws_q;
  const gptr T2base = c.ws + D.ws_t2;
  constexpr int EE = DG_MAXEFF * DG_MAXEFF;
  PROF_BEGIN(ph0);
  dev_costates(c, lds + L.l);
  PROF_END(PH_H_COST, ph0);
  PROF_BEGIN(ph1);
  // ---- 2. H[a][k][b] = sum_o lam^a_{k+1}[b,o] * Hessian of f^b_{k,o} in effective variables (interpolated from the
  //         e_i / e_i+e_j Taylor coefficients: H_ii = 2 c_i, H_ij = c_ij - c_i - c_j)
  // (a) cv[a][k][b][dir] = sum_o lam^a_{k+1}[b,o] T2[b][k][o][dir], once per direction (coalesced over dir), kept in the
  //     LDS area of the speculative line-search trajectories (idle here)
  const int cv0 = D.lsqr_keeps_eval ? L.s_u : L.e_xs;      // (the dual start's vectors, where they sit right above the evaluation arrays, are idle too)
  lptr cvb = lds + cv0;
  const int cvcap = (L.o_du - cv0);
  int ndmax = 0;
  for (int b = 0; b < M; b++) ndmax = D.ndir[b] > ndmax ? D.ndir[b] : ndmax;
  const bool use_cv = M * N * M * ndmax <= cvcap;
  if (use_cv) {
    for (int it = TID; it < M * N * M * ndmax; it += NT) {
      const int dir = it % ndmax, b = (it / ndmax) % M, k = (it / (ndmax * M)) % N, a = it / (ndmax * M * N);
      double sv = 0.0;
      if (dir < D.ndir[b]) {
        cgptr T2 = T2base + D.t2off[b] + (int64_t)k * D.t2k[b];
        clptr lk = lam + (a * (N + 1) + k + 1) * nq + D.qoff[b];
        const int nd = D.ndir[b];
        for (int o = 0; o < D.nqa[b]; o++) sv += lk[o] * T2[o * nd + dir];
      }
      cvb[it] = sv;
    }
    __syncthreads();
  }
  // (b) Hessian entries from the contracted coefficients
  for (int it = TID; it < M * N * M * EE; it += NT) {
    const int e = it % EE, b = (it / EE) % M, k = (it / (EE * M)) % N, a = it / (EE * M * N);
    const int i = e / DG_MAXEFF, j = e % DG_MAXEFF, ne = D.neff[b];
    double h = 0.0;
    if (i < ne && j < ne) {
      const int lo = i < j ? i : j, hi = i < j ? j : i;
      const int dirp = ne + lo * (ne - 1) - lo * (lo - 1) / 2 + (hi - lo - 1);
      if (use_cv) {
        clptr cv = cvb + ((a * N + k) * M + b) * ndmax;
        h = i == j ? 2.0 * cv[i] : cv[dirp] - cv[lo] - cv[hi];
      } else {
        cgptr T2 = T2base + D.t2off[b] + (int64_t)k * D.t2k[b];
        clptr lk = lam + (a * (N + 1) + k + 1) * nq + D.qoff[b];
        const int nd = D.ndir[b];
        double sv = 0;
        if (i == j) { for (int o = 0; o < D.nqa[b]; o++) sv += lk[o] * T2[o * nd + i]; h = 2.0 * sv; }
        else { for (int o = 0; o < D.nqa[b]; o++) sv += lk[o] * (T2[o * nd + dirp] - T2[o * nd + lo] - T2[o * nd + hi]); h = sv; }
      }
    }
    Hg[it] = h;
  }
  __syncthreads();
  PROF_END(PH_H_CONTR, ph1);
  PROF_BEGIN(ph2);
  // ---- 3. one lane per row of Q
  for (int row = TID; row < n; row += NT) {
    const int a = row / (N * DGSQP_NUA), rem = row % (N * DGSQP_NUA), k0 = rem / DGSQP_NUA, j0 = rem % DGSQP_NUA;
    if (D.tab_const) {
      if (D.nqa[a] == 8) dev_hessian_row_generic<8, true>(c, row, a, k0, j0);
      else if (D.nqa[a] == 4) dev_hessian_row_generic<4, true>(c, row, a, k0, j0);
      else dev_hessian_row_generic<6, true>(c, row, a, k0, j0);
    } else if (!D.uniform_nqa) {
      if (D.nqa[a] == 8) dev_hessian_row_generic<8>(c, row, a, k0, j0);
      else if (D.nqa[a] == 4) dev_hessian_row_generic<4>(c, row, a, k0, j0);
      else dev_hessian_row_generic<6>(c, row, a, k0, j0);
    } else if (D.nqa[0] == 8) {
      switch (M) {
        case 1: dev_hessian_row<8, 1>(c, row, a, k0, j0); break;
        case 2: dev_hessian_row<8, 2>(c, row, a, k0, j0); break;
        case 3: dev_hessian_row<8, 3>(c, row, a, k0, j0); break;
        case 4: dev_hessian_row<8, 4>(c, row, a, k0, j0); break;
        default: dev_hessian_row_generic<8>(c, row, a, k0, j0); break;   // 5, 6 agents
      }
    } else if (D.nqa[0] == 4) {
      switch (M) {
        case 1: dev_hessian_row<4, 1>(c, row, a, k0, j0); break;
        case 2: dev_hessian_row<4, 2>(c, row, a, k0, j0); break;
        case 3: dev_hessian_row<4, 3>(c, row, a, k0, j0); break;
        case 4: dev_hessian_row<4, 4>(c, row, a, k0, j0); break;
        default: dev_hessian_row_generic<4>(c, row, a, k0, j0); break;   // 5, 6 agents
      }
    } else {
      switch (M) {
        case 1: dev_hessian_row<6, 1>(c, row, a, k0, j0); break;
        case 2: dev_hessian_row<6, 2>(c, row, a, k0, j0); break;
        case 3: dev_hessian_row<6, 3>(c, row, a, k0, j0); break;
        case 4: dev_hessian_row<6, 4>(c, row, a, k0, j0); break;
        default: dev_hessian_row_generic<6>(c, row, a, k0, j0); break;   // 5, 6 agents
      }
    }
  }
  __syncthreads();
  PROF_END(PH_H_ROWS, ph2);
}

// ------------------------------------------------------------------------------------------------
// _evaluate(u, l, x0, up=0, hessian)   (DGSQP.py:509-533).  `usrc` is copied into the EVAL scratch.
// Produces q, g, packed G (LDS) and, if hessian, raw Q in the global workspace.
// ------------------------------------------------------------------------------------------------
// The trajectory is reused when it is already known: `xsrc` (a speculative line-search rollout of exactly this point), or
// the EVAL scratch still holding the rollout of a bit-identical input (tag in scal[60], cleared by the phases that
// overwrite the scratch) -- e.g. the full evaluation that follows an accepted trial point.
#define DG_XVALID 60
#define DG_QP_NPREV 59   // scal slot: size of the saved active set (XL layout: 1 = the saved eigenvector basis is valid)
// stage 1: trial input, trajectory and constraint values g (no derivatives)
// `fuse`: the caller will (most probably) need the derivatives at this point -- run the rollout fused with the second-order
// derivative pass where that pays (dev_can_fuse_rollout).  Tag in scal[DG_XVALID]: 0 nothing valid, 1 trajectory valid,
// 2 trajectory AND A_k, B_k, Taylor tensor valid for the input held in the EVAL scratch.
__device__ inline void dev_evaluate_point(const Ctx& c, clptr usrc, double alpha, clptr dusrc, clptr xsrc = nullptr, bool fuse = false) {
  const DgProb& D = dg_prob;
  const DgLds& L = D.L;
  lptr ue = LP(L.e_ue);
  __syncthreads();
  const double tag = LP(L.scal)[DG_XVALID];
  const bool tagged = tag != 0.0;
  double newtag = 1.0;
  int differs = 0;
  for (int i = TID; i < D.n; i += NT) {
    const double v = dusrc ? step_u(usrc[i], alpha, dusrc[i]) : usrc[i];
    if (tagged && !xsrc) differs |= (__double_as_longlong(v) != __double_as_longlong(ue[i]));
    ue[i] = v;
  }
  if (xsrc) {
    for (int i = TID; i < (D.N + 1) * D.nq; i += NT) LP(L.e_x)[i] = xsrc[i];
    __syncthreads();
  } else if (!tagged || __syncthreads_or(differs)) {
    PROF_BEGIN(pt_);
    if (fuse && dev_can_fuse_rollout()) { dev_rollout_with_derivs(c, ue, LP(L.e_x)); newtag = 2.0; }
    else dev_rollout(c, ue, LP(L.e_x));
    PROF_END(PH_ROLLOUT, pt_);
  } else newtag = tag;          // same input as the one the scratch was computed for: everything that was valid still is
  __syncthreads();
  if (TID == 0) LP(L.scal)[DG_XVALID] = newtag;
  dev_constraint_values(c, ue);
}
// stage 2: derivatives at the point prepared by stage 1 -> q, packed G (and raw Q)
__device__ inline void dev_evaluate_derivs(const Ctx& c, bool hessian) {
  const DgLds& L = dg_prob.L;
  lptr ue = LP(L.e_ue);
  __syncthreads();
  const bool have = LP(L.scal)[DG_XVALID] == 2.0;       // A_k, B_k and the Taylor tensor of this very point are there already
  if (have) {}
  else if (hessian) {
    PROF_BEGIN(pt_); dev_dyn_derivs<2>(c, ue); PROF_END(PH_DERIV2, pt_);
    if (TID == 0 && LP(L.scal)[DG_XVALID] == 1.0) LP(L.scal)[DG_XVALID] = 2.0;
  }
  else { PROF_BEGIN(pt_); dev_dyn_derivs<1>(c, ue); PROF_END(PH_DERIV1, pt_); }
  { PROF_BEGIN(pt_); dev_chains(c, ue); PROF_END(PH_CHAINS, pt_); }
  if (hessian) {
    PROF_BEGIN(pt_);
    dev_hessian_adjoint(c);
    PROF_END(PH_DP, pt_);
  }
  __syncthreads();
}
// stage 2 of a line-search TRIAL point: only the dynamics Jacobians A_k, B_k (the merit gets the Lagrangian gradients from
// a costate sweep, dev_phi_trial; q and the packed G are NOT refreshed -- every accepted point is re-linearised in full)
__device__ inline void dev_evaluate_trial_derivs(const Ctx& c) {
  lptr ue = LP(dg_prob.L.e_ue);
  __syncthreads();
  if (LP(dg_prob.L.scal)[DG_XVALID] == 2.0) return;
  PROF_BEGIN(pt_); dev_dyn_derivs<1>(c, ue); PROF_END(PH_DERIV1, pt_);
}
__device__ inline void dev_evaluate(const Ctx& c, clptr usrc, double alpha, clptr dusrc, bool hessian, clptr xsrc = nullptr) {
  dev_evaluate_point(c, usrc, alpha, dusrc, xsrc, hessian);
  dev_evaluate_derivs(c, hessian);
}

// f_J (DGSQP.py:889-893): per-agent cost along the current rollout in the EVAL scratch
__device__ inline void dev_costs(const Ctx& c, clptr ue, double* Jout) {
  const DgProb& D = dg_prob;
  clptr x = LP(D.L.e_x);
  if (TID < D.M) {
    const int a = TID;
    const dgsqp_agent_t& ag = D.P.agents[a];
    double s = 0;
    for (int k = 0; k < D.N; k++) {
      for (int j = 0; j < DGSQP_NUA; j++) {
        const double uk = ue[am_col(D, a, k, j)], um = k > 0 ? ue[am_col(D, a, k - 1, j)] : 0.0;
        s += 0.5 * ag.w_in[j] * uk * uk + 0.5 * ag.w_rate[j] * (uk - um) * (uk - um);
      }
      s += dev_state_cost(D, a, x + k * D.nq, false, (double*)nullptr, (double*)nullptr);
    }
    s += dev_state_cost(D, a, x + D.N * D.nq, true, (double*)nullptr, (double*)nullptr);
    Jout[a] = s;
  }
}
