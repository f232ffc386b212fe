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
template <int NQA>
__device__ inline void dev_rollout_agent(const DgProb& D, int a, clptr ue, lptr x) {
  typedef Ty<0> T;
  const int nq = D.nq, qo = D.qoff[a];
  T q[NQA], u[2], qn[NQA];
  for (int i = 0; i < NQA; i++) q[i].c[0] = x[qo + i];
  for (int k = 0; k < D.N; k++) {
    u[0].c[0] = ue[am_col(D, a, k, 0)];
    u[1].c[0] = ue[am_col(D, a, k, 1)];
    dev_fd<0, NQA>(D.P, D.P.agents[a], q, u, qn);
    for (int i = 0; i < NQA; i++) { q[i] = qn[i]; x[(k + 1) * nq + qo + i] = qn[i].c[0]; }
  }
}
// Dynamic-bicycle rollout on TWO lanes per agent.  The 40 f_c evaluations of one rk4 step (M = 10) are sequential and
// each costs ~9 fp64 transcendental calls; the front / rear tyre chains (atan2 -> atan -> sin) and the two angle sincos
// are the same instruction stream on different data, so lane 2a handles (front axle, e_psi) and lane 2a+1 (rear axle,
// e_psi + psi_t); results are exchanged inside the quad with DPP.  sincos(delta) is constant over the step and hoisted.
// Same arithmetic as dev_fc_dyn<0> (dynamics_models.py:2008-2062), evaluated redundantly on both lanes otherwise.
__device__ inline void dyn_fc_pair(const dgsqp_problem_t& P, const dgsqp_agent_t& ag, int role, const double* q, double ua, double us,
                                   double sd, double cd, double* dq) {
  const double vx = q[2], vy = q[3], w = q[4];
  constexpr int S1 = DGSQP_MAX_SEGS + 1;
  clptr tt = LP(dg_prob.L.t_track);
  const double sbar = wrap_s(q[6], P.track_L);
  int seg = 0;
  for (int i = 1; i < P.n_segs; i++) seg += (sbar >= tt[i]) ? 1 : 0;
  const double c = tt[S1 + seg];
  const double psit = (q[6] + (sbar - q[6] - tt[seg])) * tt[3 * S1 + seg] + tt[2 * S1 + seg];
  const double vyf = vy + w * ag.L_f;
  // role 0: front axle and e_psi ; role 1: rear axle and e_psi + psi_t
  double ay_, ax_, add;
  if (role == 0) {
    if (ag.simple_slip) { ay_ = vyf; ax_ = vx; add = us; } else { ay_ = vyf * cd - vx * sd; ax_ = vx * cd + vyf * sd; add = 0.0; }
  } else { ay_ = vy - w * ag.L_r; ax_ = vx; add = 0.0; }
  const double alpha = add - atan2(ay_, ax_);
  double F;
  if (ag.tire_model == 0) {
    const double Bc = role == 0 ? ag.pac_Bf : ag.pac_Br, Cc = role == 0 ? ag.pac_Cf : ag.pac_Cr, Dc = role == 0 ? ag.pac_Df : ag.pac_Dr;
    F = Dc * sin(Cc * atan(Bc * alpha));
  } else {
    F = alpha * (role == 0 ? ag.lin_Bf * ag.mass * ag.gravity * ag.L_r / (ag.L_f + ag.L_r) : ag.lin_Br * ag.mass * ag.gravity * ag.L_f / (ag.L_f + ag.L_r));
  }
  double sa, ca;
  sincos(role == 0 ? q[5] : q[5] + psit, &sa, &ca);
  // exchange inside the pair: quad_perm [0,0,2,2] takes the even lane's value, [1,1,3,3] the odd lane's
  const double fyf = dpp_f64<0xA0>(F), fyr = dpp_f64<0xF5>(F);
  const double se = dpp_f64<0xA0>(sa), ce = dpp_f64<0xA0>(ca), st = dpp_f64<0xF5>(sa), ct = dpp_f64<0xF5>(ca);
  double Fx = vx * (-ag.c_da) - vx * (vx > 0 ? vx : -vx) * ag.c_dr;
  if (ag.c_r != 0.0) Fx = Fx - pow(vx > 0 ? vx : -vx, ag.p_r) * (vx / sqrt(vx * vx + 1e-6)) * ag.c_r;
  const double a_r = ag.drive_wheels == 0 ? ua * 0.5 : ua, a_f = ag.drive_wheels == 0 ? ua * 0.5 : 0.0;
  const double ax = a_r + a_f * cd + (Fx - fyf * sd) * (1.0 / ag.mass);
  const double ay = a_f * sd + (fyf * cd + fyr) * (1.0 / ag.mass);
  const double vlon = (vx * ce - vy * se) * (1.0 / (1.0 - q[7] * c));
  dq[0] = vx * ct - vy * st;
  dq[1] = vy * ct + vx * st;
  dq[2] = ax + w * vy;
  dq[3] = ay - w * vx;
  dq[4] = (fyf * cd * ag.L_f - fyr * ag.L_r) * (1.0 / ag.I_z);
  dq[5] = w - vlon * c;
  dq[6] = vlon;
  dq[7] = vx * se + vy * ce;
}
__device__ inline void dev_rollout_dyn_pair(const DgProb& D, int a, int role, clptr ue, lptr x) {
  const dgsqp_problem_t& P = D.P;
  const dgsqp_agent_t& ag = P.agents[a];
  const int nq = D.nq, qo = D.qoff[a];
  double q[8], k1[8], k2[8], k3[8], t[8];
  for (int i = 0; i < 8; i++) q[i] = x[qo + i];
  const double h = P.dt / P.substeps;
  for (int k = 0; k < D.N; k++) {
    const double ua = ue[am_col(D, a, k, 0)], us = ue[am_col(D, a, k, 1)];
    double sd, cd;
    sincos(us, &sd, &cd);
    if (P.integrator == DGSQP_INT_EULER) {
      dyn_fc_pair(P, ag, role, q, ua, us, sd, cd, k1);
      for (int i = 0; i < 8; i++) q[i] = q[i] + k1[i] * P.dt;
    } else {
      for (int m = 0; m < P.substeps; m++) {
        if (P.integrator == DGSQP_INT_RK4) {
          dyn_fc_pair(P, ag, role, q, ua, us, sd, cd, k1);
          for (int i = 0; i < 8; i++) t[i] = q[i] + k1[i] * (h / 2);
          dyn_fc_pair(P, ag, role, t, ua, us, sd, cd, k2);
          for (int i = 0; i < 8; i++) { t[i] = q[i] + k2[i] * (h / 2); k1[i] = k1[i] + k2[i] * 2.0; }
          dyn_fc_pair(P, ag, role, t, ua, us, sd, cd, k3);
          for (int i = 0; i < 8; i++) { t[i] = q[i] + k3[i] * h; k1[i] = k1[i] + k3[i] * 2.0; }
          dyn_fc_pair(P, ag, role, t, ua, us, sd, cd, k2);
          for (int i = 0; i < 8; i++) q[i] = q[i] + (k1[i] + k2[i]) * h / 6.0;
        } else if (P.integrator == DGSQP_INT_RK3) {
          dyn_fc_pair(P, ag, role, q, ua, us, sd, cd, k1);
          for (int i = 0; i < 8; i++) { k1[i] = k1[i] * h; t[i] = q[i] + k1[i] * 0.5; }
          dyn_fc_pair(P, ag, role, t, ua, us, sd, cd, k2);
          for (int i = 0; i < 8; i++) { k2[i] = k2[i] * h; t[i] = q[i] - k1[i] + k2[i] * 2.0; }
          dyn_fc_pair(P, ag, role, t, ua, us, sd, cd, k3);
          for (int i = 0; i < 8; i++) q[i] = q[i] + (k1[i] + k2[i] * 4.0 + k3[i] * h) / 6.0;
        } else {
          dyn_fc_pair(P, ag, role, q, ua, us, sd, cd, k1);
          for (int i = 0; i < 8; i++) t[i] = q[i] + k1[i] * h;
          dyn_fc_pair(P, ag, role, t, ua, us, sd, cd, k2);
          for (int i = 0; i < 8; i++) q[i] = q[i] + (k1[i] + k2[i]) * (h / 2);
        }
      }
    }
    if (role == 0)
      for (int i = 0; i < 8; i++) x[(k + 1) * nq + qo + i] = q[i];
  }
}
__device__ __noinline__ void dev_rollout(const Ctx& c, clptr ue, lptr x) {
  const DgProb& D = dg_prob;
  __syncthreads();
  for (int i = TID; i < D.nq; i += NT) x[i] = c.x0[i];
  __syncthreads();
  bool all_dyn = true;
  for (int a = 0; a < D.M; a++) all_dyn = all_dyn && D.nqa[a] == 8;
  if (all_dyn) {
    if (TID < 2 * D.M) dev_rollout_dyn_pair(D, TID >> 1, TID & 1, ue, x);
  } else if (TID < D.M) {
    if (D.nqa[TID] == 8) dev_rollout_agent<8>(D, TID, ue, x); else dev_rollout_agent<6>(D, TID, ue, x);
  }
  __syncthreads();
}

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
template <int DEG, int NQA>
__device__ inline void dev_taylor_item(const Ctx& c, int a, int k, int dir, clptr ue) {
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
    const int z = D.effvar[a][ei];
    if (z < NQA) q[z].c[1] += 1.0; else u[z - NQA].c[1] += 1.0;
  }
  if (ej != ei) {
    const int z = D.effvar[a][ej];
    if (z < NQA) q[z].c[1] += 1.0; else u[z - NQA].c[1] += 1.0;
  }
  dev_fd<DEG, NQA>(D.P, D.P.agents[a], q, u, out);
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
    const int nd = (DEG >= 2) ? D.ndir[a] : D.neff[a];
    for (int it = TID; it < D.N * nd; it += NT) {
      const int k = it / nd, dir = it % nd;
      if (nqa == 8) dev_taylor_item<DEG, 8>(c, a, k, dir, ue); else dev_taylor_item<DEG, 6>(c, a, k, dir, ue);
    }
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// sensitivity chains S^a_{k,t0}[:, j] = A_{k-1} ... A_{t0+1} B_{t0}[:, j]  (f_Du_x, DGSQP.py:642-650),
// consumed on the fly into the packed dense gradients (f_Du_C :823-826) and q (f_q :672-676, 898-899)
// ------------------------------------------------------------------------------------------------
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
    if (D.nqa[a] == 8) dev_chain_item<8>(c, ue, it); else dev_chain_item<6>(c, ue, it);
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
            lds[L.gd + dd.off + t0 * DGSQP_NUA + j] = val;
          }
        } else if (dd.a == a || dd.b == a) {
          const int ia = D.qoff[dd.a], ib = D.qoff[dd.b];
          const double dx = xk[ia] - xk[ib], dy = xk[ia + 1] - xk[ib + 1];
          const double s = 2.0 * (dx * v[0] + dy * v[1]);
          if (dd.a == a) lds[L.gd + dd.off + t0 * DGSQP_NUA + j] = -s;
          else lds[L.gd + dd.off + 2 * k + t0 * DGSQP_NUA + j] = s;
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
      default: g = ag.st_lb[R.idx] - xk[D.qoff[R.a] + R.idx]; break;
    }
    LP(L.g)[r] = g;
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// Hessian of agent a's Lagrangian L^a = J^a + l^T C w.r.t. the input sequence by ONE backward
// dynamic-programming sweep (same recursion as DGSQP.py:679-727 / :828-877, summed over rows by
// linearity, identical to the reference's own f_Duu_L :937-941).  Rows of agent a go to raw Q.
// ------------------------------------------------------------------------------------------------
// stage injection: d/dx_k and d2/dx_k^2 of [J^a_k + sum_r l_r c_r] for rows r of stage k
__device__ inline void dev_stage_injection(const Ctx& c, int a, int k, lptr inj) {
  const DgProb& D = dg_prob;
  const DgLds& L = D.L;
  const int nq = D.nq;
  for (int i = TID; i < nq + nq * nq; i += NT) inj[i] = 0.0;
  __syncthreads();
  if (TID == 0) {
    clptr xk = LP(L.e_x + k * nq);
    clptr l = LP(L.l);
    lptr Dx = inj;
    lptr Dxx = inj + nq;
    dev_state_cost(D, a, xk, k == D.N, Dx, Dxx);
    for (int r = D.stage_row0[k]; r < D.stage_row0[k + 1]; r++) {
      const DgRow R = ld_row(r);
      if (R.dense < 0) continue;  // rate / input-box rows are affine in u: no state derivatives
      const double lr = l[r];
      if (R.type == DG_R_OBS) {
        const int ia = D.qoff[R.a], ib = D.qoff[R.b];
        const double dx = xk[ia] - xk[ib], dy = xk[ia + 1] - xk[ib + 1];
        Dx[ia] -= 2 * lr * dx; Dx[ia + 1] -= 2 * lr * dy; Dx[ib] += 2 * lr * dx; Dx[ib + 1] += 2 * lr * dy;
        for (int p = 0; p < 2; p++) {
          Dxx[(ia + p) * nq + ia + p] -= 2 * lr; Dxx[(ib + p) * nq + ib + p] -= 2 * lr;
          Dxx[(ia + p) * nq + ib + p] += 2 * lr; Dxx[(ib + p) * nq + ia + p] += 2 * lr;
        }
      } else if (R.type == DG_R_ST_UB) Dx[D.qoff[R.a] + R.idx] += lr;
      else if (R.type == DG_R_ST_LB) Dx[D.qoff[R.a] + R.idx] -= lr;
    }
  }
  __syncthreads();
}

__device__ inline int dev_block_of(const DgProb& D, int xi) {
  int b = 0;
  while (b + 1 < D.M && xi >= D.qoff[b + 1]) b++;
  return b;
}

__device__ __noinline__ void dev_hessian_dp(const Ctx& c, int a) {
  const DgProb& D = dg_prob;
  const DgLds& L = D.L;
  lptr lds = LP(0);
  const int nq = D.nq, nu = D.nu, n = D.n, N = D.N;
  lds_d *Dx = lds + L.e_Dx, *Dxn = lds + L.e_Dx + nq, *Dxx = lds + L.e_Dxx, *nDxx = lds + L.e_nDxx;
  lds_d *tQA = lds + L.e_tQA, *tQB = lds + L.e_tQB, *A1 = lds + L.e_A1, *A2 = lds + L.e_A2;
  lds_d *Dxu = lds + L.e_Dxu, *Dxu2 = lds + L.e_Dxu + n * nq;  // double-buffered rows d2/du_t dx_k
  lds_d *Hc = lds + L.e_Hc, *cv = lds + L.e_cv, *inj = lds + L.e_inj;
  gptr Qg = c.ws + D.ws_q;
  const dgsqp_agent_t& ag = D.P.agents[a];

  dev_stage_injection(c, a, N, inj);
  for (int i = TID; i < nq; i += NT) Dx[i] = inj[i];
  for (int i = TID; i < nq * nq; i += NT) Dxx[i] = inj[nq + i];
  __syncthreads();

  for (int k = N - 1; k >= 0; k--) {
    // ---- phase 1: contraction of the Taylor tensor with the costate, Dxx*A, Dxx*B, stage injection
    for (int it = TID; it < D.M * DG_MAXDIR; it += NT) {
      const int b = it / DG_MAXDIR, dir = it % DG_MAXDIR;
      if (dir < D.ndir[b]) {
        cgptr T2 = c.ws + D.ws_t2 + D.t2off[b] + (int64_t)k * D.t2k[b];
        double s = 0;
        for (int o = 0; o < D.nqa[b]; o++) s += Dx[D.qoff[b] + o] * T2[o * D.ndir[b] + dir];
        cv[b * DG_MAXDIR + dir] = s;
      }
    }
    for (int it = TID; it < nq * (nq + nu); it += NT) {
      const int i = it / (nq + nu), jj = it % (nq + nu);
      if (jj < nq) {  // tQA[i][jj] = sum_l Dxx[i][l] A[l][jj], A block diagonal
        const int b = dev_block_of(D, jj), nqa = D.nqa[b], qo = D.qoff[b];
        clptr A = lds + L.e_A[b] + k * nqa * nqa;
        double s = 0;
        for (int m = 0; m < nqa; m++) s += Dxx[i * nq + qo + m] * A[m * nqa + (jj - qo)];
        tQA[i * nq + jj] = s;
      } else {
        const int cu = jj - nq, b = cu / DGSQP_NUA, j = cu % DGSQP_NUA;
        const int nqa = D.nqa[b], qo = D.qoff[b];
        clptr B = lds + L.e_B[b] + k * nqa * 2;
        double s = 0;
        for (int m = 0; m < nqa; m++) s += Dxx[i * nq + qo + m] * B[m * 2 + j];
        tQB[i * nu + cu] = s;
      }
    }
    if (k > 0) dev_stage_injection(c, a, k, inj); else __syncthreads();
    // Hc[b][i][j]: Hessian (in effective variables) of  costate . f_d  for agent block b
    for (int it = TID; it < D.M * DG_MAXEFF * DG_MAXEFF; it += NT) {
      const int b = it / (DG_MAXEFF * DG_MAXEFF), i = (it / DG_MAXEFF) % DG_MAXEFF, j = it % DG_MAXEFF;
      const int ne = D.neff[b];
      if (i < ne && j < ne) {
        clptr cb = cv + b * DG_MAXDIR;
        double h;
        if (i == j) h = 2.0 * cb[i];
        else {
          const int lo = i < j ? i : j, hi = i < j ? j : i;
          const int dir = ne + lo * (ne - 1) - lo * (lo - 1) / 2 + (hi - lo - 1);
          h = cb[dir] - cb[lo] - cb[hi];
        }
        Hc[it] = h;
      }
    }
    __syncthreads();
    // ---- phase 2: A1, A2, and the rows t>k of Dxu (emit B1, propagate into the other buffer)
    for (int it = TID; it < nu * (nu + nq); it += NT) {
      const int c1 = it / (nu + nq), jj = it % (nu + nq);
      const int b1 = c1 / DGSQP_NUA, j1 = c1 % DGSQP_NUA, nqa1 = D.nqa[b1], qo1 = D.qoff[b1];
      clptr B = lds + L.e_B[b1] + k * nqa1 * 2;
      if (jj < nu) {  // A1[c1][c2] = Duu_J + B^T Dxx B + sum_i Dx_i F_i      (DGSQP.py:698-700)
        const int c2 = jj, b2 = c2 / DGSQP_NUA, j2 = c2 % DGSQP_NUA;
        double s = 0;
        for (int m = 0; m < nqa1; m++) s += B[m * 2 + j1] * tQB[(qo1 + m) * nu + c2];
        if (b1 == b2) s += Hc[(b1 * DG_MAXEFF + (D.neff[b1] - 2 + j1)) * DG_MAXEFF + (D.neff[b1] - 2 + j2)];
        if (c1 == c2 && b1 == a) s += ag.w_in[j1] + ag.w_rate[j1] + (k + 1 < N ? ag.w_rate[j1] : 0.0);
        A1[c1 * nu + c2] = s;
      } else {        // A2[c1][x] = B^T Dxx A + sum_i Dx_i G_i                 (DGSQP.py:708-710)
        const int xi = jj - nu;
        double s = 0;
        for (int m = 0; m < nqa1; m++) s += B[m * 2 + j1] * tQA[(qo1 + m) * nq + xi];
        const int li = xi - qo1;
        if (li >= 2 && li < nqa1) s += Hc[(b1 * DG_MAXEFF + (D.neff[b1] - 2 + j1)) * DG_MAXEFF + (li - 2)];
        A2[c1 * nq + xi] = s;
      }
    }
    {
      const int nrows = (N - 1 - k) * nu, row0 = (k + 1) * nu, per = nu + nq;
      for (int it = TID; it < nrows * per; it += NT) {
        const int row = row0 + it / per, jj = it % per;
        clptr old = Dxu + row * nq;
        if (jj < nu) {   // B1 = Dxu_Q[-1] @ B_k (+ d2J/du_{k+1}du_k)                 (DGSQP.py:704-706)
          const int t = row / nu, ju = row % nu, ar = ju / DGSQP_NUA;
          const int cu = jj, b = cu / DGSQP_NUA, j = cu % DGSQP_NUA, nqa = D.nqa[b], qo = D.qoff[b];
          clptr B = lds + L.e_B[b] + k * nqa * 2;
          double s = 0;
          for (int m = 0; m < nqa; m++) s += old[qo + m] * B[m * 2 + j];
          if (t == k + 1 && cu == ju && b == a) s -= ag.w_rate[j];
          const int ri = am_col(D, ar, t, ju % DGSQP_NUA), ci = am_col(D, b, k, j);
          if (ar == a) Qg[(int64_t)ri * n + ci] = s;
          if (b == a) Qg[(int64_t)ci * n + ri] = s;
        } else {         // Dxu_Q[-1] @ A_k                                            (DGSQP.py:714)
          const int jx = jj - nu, b = dev_block_of(D, jx), nqa = D.nqa[b], qo = D.qoff[b];
          clptr A = lds + L.e_A[b] + k * nqa * nqa;
          double s = 0;
          for (int m = 0; m < nqa; m++) s += old[qo + m] * A[m * nqa + (jx - qo)];
          Dxu2[row * nq + jx] = s;
        }
      }
    }
    __syncthreads();
    // ---- phase 3: emit A1, store A2 as rows of stage k, update Dxx / Dx  (DGSQP.py:696, 717-719)
    for (int it = TID; it < nu * nu; it += NT) {
      const int c1 = it / nu, c2 = it % nu;
      if (c1 / DGSQP_NUA == a)
        Qg[(int64_t)am_col(D, a, k, c1 % DGSQP_NUA) * n + am_col(D, c2 / DGSQP_NUA, k, c2 % DGSQP_NUA)] = A1[it];
    }
    for (int it = TID; it < nu * nq; it += NT) Dxu2[k * nu * nq + it] = A2[it];
    if (k > 0) {
      for (int it = TID; it < nq * nq; it += NT) {
        const int i = it / nq, jx = it % nq;
        const int b = dev_block_of(D, i), nqa = D.nqa[b], qo = D.qoff[b];
        clptr A = lds + L.e_A[b] + k * nqa * nqa;
        double s = inj[nq + it];
        for (int m = 0; m < nqa; m++) s += A[m * nqa + (i - qo)] * tQA[(qo + m) * nq + jx];
        const int li = i - qo, lj = jx - qo;
        if (li >= 2 && lj >= 2 && lj < nqa) s += Hc[(b * DG_MAXEFF + (li - 2)) * DG_MAXEFF + (lj - 2)];
        nDxx[it] = s;
      }
      for (int jx = TID; jx < nq; jx += NT) {
        const int b = dev_block_of(D, jx), nqa = D.nqa[b], qo = D.qoff[b];
        clptr A = lds + L.e_A[b] + k * nqa * nqa;
        double s = inj[jx];
        for (int m = 0; m < nqa; m++) s += Dx[qo + m] * A[m * nqa + (jx - qo)];
        Dxn[jx] = s;
      }
    }
    __syncthreads();
    if (k > 0) {
      for (int it = TID; it < nq * nq; it += NT) Dxx[it] = nDxx[it];
      for (int it = TID; it < nq; it += NT) Dx[it] = Dxn[it];
    }
    { lptr t = Dxu; Dxu = Dxu2; Dxu2 = t; }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// _evaluate(u, l, x0, up=0, hessian)   (DGSQP.py:509-533).  `usrc` is copied into the EVAL scratch.
// Produces q, g, packed G (LDS) and, if hessian, raw Q in the global workspace.
// ------------------------------------------------------------------------------------------------
__device__ inline void dev_evaluate(const Ctx& c, clptr usrc, double alpha, clptr dusrc, bool hessian) {
  const DgProb& D = dg_prob;
  const DgLds& L = D.L;
  lptr ue = LP(L.e_ue);
  __syncthreads();
  for (int i = TID; i < D.n; i += NT) ue[i] = dusrc ? usrc[i] + alpha * dusrc[i] : usrc[i];
  { PROF_BEGIN(pt_); dev_rollout(c, ue, LP(L.e_x)); PROF_END(PH_ROLLOUT, pt_); }
  if (hessian) { PROF_BEGIN(pt_); dev_dyn_derivs<2>(c, ue); PROF_END(PH_DERIV2, pt_); }
  else { PROF_BEGIN(pt_); dev_dyn_derivs<1>(c, ue); PROF_END(PH_DERIV1, pt_); }
  { PROF_BEGIN(pt_); dev_chains(c, ue); dev_constraint_values(c, ue); PROF_END(PH_CHAINS, pt_); }
  if (hessian) {
    PROF_BEGIN(pt_);
    for (int a = 0; a < D.M; a++) dev_hessian_dp(c, a);
    PROF_END(PH_DP, pt_);
  }
  __syncthreads();
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
