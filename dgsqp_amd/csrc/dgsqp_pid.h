// PID lane-follower warm start and collision rejection of the Monte-Carlo scripts
// (scripts/DGSQP_ALGAMES_monte_carlo_chicane.py:411-447 with DGSQP/solvers/PID.py; check_collision chicane.py:38-43):
// what every Monte-Carlo sample runs before DGSQP.solve().  One lane per (scenario, agent); the plant is the agent's own
// continuous model integrated with rk4 (10 sub-steps per dt by default), the same f_c the solver differentiates.
#pragma once

template <int NQA, bool SPL = false>
__device__ inline void dev_pid_agent(const DgProb& D, int a, cgptr q0, const dgsqp_pid_t& pid, gptr u_out, gptr q_out) {
  typedef Ty<0> T;
  const dgsqp_problem_t& P = D.P;
  const dgsqp_agent_t& ag = P.agents[a];
  constexpr int V = 2, EPSI = NQA == 8 ? 5 : 3, EY = NQA - 1;
  T q[NQA], k1[NQA], k2[NQA], t[NQA], u[2];
#pragma unroll
  for (int i = 0; i < NQA; i++) q[i].c[0] = q0[i];
  if (q_out)
    for (int i = 0; i < NQA; i++) q_out[i] = q0[i];
  const double v_ref = q0[V], lat_ref = q0[EY];
  double ei = 0.0, up0 = 0.0, up1 = 0.0;
  const double dt = P.dt, h = dt / pid.substeps;
  for (int k = 0; k < D.N; k++) {
    // PID.solve: speed P controller, steering PI on ey_gain (e_y - e_y0) + e_psi; rate saturation first, then magnitude
    double ua = -(pid.kp_v * (q[V].c[0] - v_ref));
    const double e = pid.ey_gain * (q[EY].c[0] - lat_ref) + q[EPSI].c[0];
    ei = fmin(fmax(ei + e * dt, -pid.ei_max), pid.ei_max);
    double us = -(pid.kp_s * e + pid.ki_s * ei);
    ua = fmin(fmax(fmin(fmax(ua - up0, -pid.du_max[0]), pid.du_max[0]) + up0, -pid.u_max[0]), pid.u_max[0]);
    us = fmin(fmax(fmin(fmax(us - up1, -pid.du_max[1]), pid.du_max[1]) + up1, -pid.u_max[1]), pid.u_max[1]);
    up0 = ua; up1 = us;
    u_out[am_col(D, a, k, 0)] = ua;
    u_out[am_col(D, a, k, 1)] = us;
    u[0].c[0] = ua; u[1].c[0] = us;
    FcPre<0> pre;
    if constexpr (NQA == 8) dev_fc_pre_dyn<0>(ag, u, pre); else dev_fc_pre_kin<0>(ag, u, pre);
    for (int m = 0; m < pid.substeps; m++) {
      dev_fc<0, NQA, SPL>(P, ag, q, u, pre, k1);
#pragma unroll
      for (int i = 0; i < NQA; i++) t[i] = q[i] + k1[i] * (h / 2);
      dev_fc<0, NQA, SPL>(P, ag, t, u, pre, k2);
#pragma unroll
      for (int i = 0; i < NQA; i++) { t[i] = q[i] + k2[i] * (h / 2); k1[i] = k1[i] + k2[i] * 2.0; }
      dev_fc<0, NQA, SPL>(P, ag, t, u, pre, k2);
#pragma unroll
      for (int i = 0; i < NQA; i++) { t[i] = q[i] + k2[i] * h; k1[i] = k1[i] + k2[i] * 2.0; }
      dev_fc<0, NQA, SPL>(P, ag, t, u, pre, k2);
#pragma unroll
      for (int i = 0; i < NQA; i++) q[i] = q[i] + (k1[i] + k2[i]) * (h / 6.0);
    }
    if (q_out)
      for (int i = 0; i < NQA; i++) q_out[(int64_t)(k + 1) * D.nq + i] = q[i].c[0];
  }
}

// u_ws [B][n] agent-major, q_ws [B][(N+1) n_q]
__global__ void __launch_bounds__(DG_BLOCK)
dg_pid_kernel(int64_t B, const double* __restrict__ q0, dgsqp_pid_t pid, double* __restrict__ u_ws, double* __restrict__ q_ws) {
  const DgProb& D = dg_prob;
  dev_load_tables();    // f_c reads the track tables from LDS
  for (int64_t it = (int64_t)blockIdx.x * DG_BLOCK + TID; it < B * D.M; it += (int64_t)gridDim.x * DG_BLOCK) {
    const int64_t b = it / D.M;
    const int a = (int)(it % D.M);
    cgptr q0a = (cgptr)q0 + b * D.nq + D.qoff[a];
    gptr uo = (gptr)u_ws + b * D.n;
    gptr qo = q_ws ? (gptr)q_ws + b * (int64_t)(D.N + 1) * D.nq + D.qoff[a] : nullptr;
    if (D.P.track_kind == DGSQP_TRACK_SPLINE) { if (D.nqa[a] == 8) dev_pid_agent<8, true>(D, a, q0a, pid, uo, qo); else dev_pid_agent<6, true>(D, a, q0a, pid, uo, qo); }
    else if (D.nqa[a] == 8) dev_pid_agent<8>(D, a, q0a, pid, uo, qo); else dev_pid_agent<6>(D, a, q0a, pid, uo, qo);
  }
}
// collide[b] = 1 if any two agents come closer than r_i + r_j at any stage of the warm-start trajectories
__global__ void dg_collide_kernel(int64_t B, const double* __restrict__ q_ws, int32_t* __restrict__ collide) {
  const DgProb& D = dg_prob;
  for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < B; b += (int64_t)gridDim.x * blockDim.x) {
    const double* q = q_ws + b * (int64_t)(D.N + 1) * D.nq;
    int hit = 0;
    for (int k = 0; k <= D.N; k++)
      for (int i = 0; i < D.M; i++)
        for (int j = i + 1; j < D.M; j++) {
          const double dx = q[k * D.nq + D.qoff[i]] - q[k * D.nq + D.qoff[j]], dy = q[k * D.nq + D.qoff[i] + 1] - q[k * D.nq + D.qoff[j] + 1];
          if (sqrt(dx * dx + dy * dy) < D.P.agents[i].radius + D.P.agents[j].radius) hit = 1;
        }
    collide[b] = hit;
  }
}
