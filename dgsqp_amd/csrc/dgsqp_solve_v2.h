// DG-SQP v2 state machine (reference DGSQP/solvers/DGSQP_v2.py:322-720) on the evaluation / _nearestPD / QP kernels of v1:
// non-monotone strategy with relaxed "d-steps" inside a decaying radius and merit-checked "m-steps" that fall back to the last
// checkpoint and a backtracking line search (watchdog), regularisation that decays after every m-step, a merit memory, merit
// 1/2 |q + G'l|^2 + mu sum(max(0, g)) ('stat_l1', DGSQP_v2.py:1141-1160).  SURVEY.md section 8 row (f3); it is the solver the
// reference pairs with the dynamic-bicycle game (scripts/comparison_study_barc/exact_dgsqp.py).
#pragma once

#define DG_V2_MEM0 16        // scalar slots 16..31: merit memory (deque(maxlen = nms_memory_size), DGSQP_v2.py:343)

// one iteration record of the reference's iter_data that load_checkpoint (DGSQP_v2.py:692-712) may need again:
// primal / dual iterate at the start of the iteration, the QP step, the merit parameter.  (slack_iterate = max(0, g) at that
// iterate is recomputed with the re-linearisation the line search performs anyway, DGSQP_v2.py:735.)
struct V2Rec { gptr p; };
__device__ inline int v2_rec_doubles(const DgProb& D) { return 2 * D.n + 2 * D.nc + 2; }
__device__ inline void v2_rec_store(const Ctx& c, gptr rec, cgptr ubase, cgptr lbase, double mu) {   // (u, l) of the iteration start; du, lhat from LDS
  const DgProb& D = dg_prob;
  const DgLds& L = D.L;
  __syncthreads();
  for (int i = TID; i < D.n; i += NT) { rec[i] = ubase[i]; rec[D.n + i] = LP(0)[L.o_du + i]; }
  for (int r = TID; r < D.nc; r += NT) { rec[2 * D.n + r] = lbase[r]; rec[2 * D.n + D.nc + r] = LP(0)[L.o_lhat + r]; }
  if (TID == 0) rec[2 * D.n + 2 * D.nc] = mu;
  __threadfence_block();
  __syncthreads();
}
__device__ inline double v2_rec_load(const Ctx& c, cgptr rec) {      // -> u, du, l, lhat in LDS; returns mu
  const DgProb& D = dg_prob;
  const DgLds& L = D.L;
  __syncthreads();
  for (int i = TID; i < D.n; i += NT) { LP(0)[L.u + i] = rec[i]; LP(0)[L.o_du + i] = rec[D.n + i]; }
  for (int r = TID; r < D.nc; r += NT) { LP(0)[L.l + r] = rec[2 * D.n + r]; LP(0)[L.o_lhat + r] = rec[2 * D.n + D.nc + r]; }
  const double mu = rec[2 * D.n + 2 * D.nc];
  __syncthreads();
  return mu;
}
__device__ inline void v2_rec_copy(const Ctx& c, gptr dst, cgptr src) {
  __syncthreads();
  for (int i = TID; i < v2_rec_doubles(dg_prob); i += NT) dst[i] = src[i];
  __threadfence_block();
  __syncthreads();
}

// _solve_qp with the current regularisation (DGSQP_v2.py:253-284) at the linearisation held in LDS
// v = Q^T d for the directional derivative of 1/2 |d|^2 ('stat_l1'), or the gradient of the summed objective ('sum_obj_l1')
__device__ inline void v2_merit_vector(const Ctx& c) {
  if (dg_prob.par.merit_function == DGSQP_MERIT_SUM_OBJ_L1) (void)dev_v2_sum_obj(c, true); else dev_qt_mul(c);
}
__device__ inline int v2_qp(const Ctx& c) {
  const DgProb& D = dg_prob;
  v2_merit_vector(c);
  if (D.osqp && D.big == 2) { dev_xl_psd(c, nullptr); return dev_qp_osqp_xl(c); }
  if (D.osqp) { dev_psd_inverse(c, c.ws + D.ws_xM, false); return dev_qp_osqp(c); }
  if (D.big == 2) { dev_xl_psd(c, nullptr); return dev_xl_qp(c); }
  // the explicit-inverse kernels need eig_floor + reg >= 1e-8 (dgsqp_layout.h); reg decays towards 0 during a v2 solve
  if (D.classic_qp && D.eig_floor + dev_reg() < 1e-8) {
    if (TID == 0) LP(D.L.scal)[DG_QP_NPREV] = 0.0;      // the classical kernels reuse the QP scratch: no saved active set afterwards
    dev_psd_inverse(c, c.ws + D.ws_xM, false);
    return dev_xl_qp(c);
  }
  dev_psd_inverse(c, nullptr);
  return dev_qp(c);
}

// line_search (DGSQP_v2.py:727-755) from the base (u, du, l, lhat) in LDS; phi_b / dphi_b are the merit and its directional
// derivative at the base ('armijo') and memmax the largest remembered merit ('max').  On return u and l hold the LAST trial;
// returns that trial's merit with mu = 1 (what is appended to the merit memory).
__device__ inline double v2_line_search(const Ctx& c, double mu, double phi_b, double dphi_b, double memmax) {
  const DgProb& D = dg_prob;
  const DgLds& L = D.L;
  lptr lds = LP(0);
  const double sigma = D.par.merit_decrease;
  double alpha = 1.0;
  const int K = D.ls_spec, xsz = ((D.N + 1) * D.nq + 1) & ~1;
  for (int i = 0; i < D.par.line_search_iters; i++) {
    if (K > 1) {
      if (i % K == 0) {
        const int left = D.par.line_search_iters - i;
        dev_rollout_multi(c, lds + L.u, lds + L.o_du, alpha, D.par.tau, left < K ? left : K, lds + L.e_xs, xsz, D.ls_spec1, lds + L.e_xs2);
      }
      const int jt = i % K;
      dev_evaluate_point(c, lds + L.u, alpha, lds + L.o_du, jt < D.ls_spec1 ? lds + L.e_xs + jt * xsz : lds + L.e_xs2 + (jt - D.ls_spec1) * xsz);
    } else dev_evaluate_point(c, lds + L.u, alpha, lds + L.o_du);
    const double phit = dev_trial_merit(c, alpha, 0.0, mu);
    const double R = D.par.merit_decrease_condition == DGSQP_DECREASE_MAX ? (1.0 - sigma * alpha) * memmax : phi_b + sigma * alpha * dphi_b;
    dev_tr(c, 30, alpha); dev_tr(c, 31, phit);
    if (phit <= R) break;
    if (i + 1 < D.par.line_search_iters) alpha *= D.par.tau;      // (the reference multiplies once more after the last trial; _a is not used again)
  }
  const double phi1 = 0.5 * lds[L.scal + DG_V2_DD] + lds[L.scal + DG_V2_VIO];     // f_phi(..., mu = 1) at the last trial (DGSQP_v2.py:754)
  __syncthreads();
  for (int i = TID; i < D.n; i += NT) lds[L.u + i] = step_u(lds[L.u + i], alpha, lds[L.o_du + i]);
  for (int r = TID; r < D.nc; r += NT) lds[L.l + r] += alpha * (lds[L.o_lhat + r] - lds[L.l + r]);
  __syncthreads();
  return phi1;
}

// Returns true when the scenario was DEFERRED (dev_solve's contract; round 4: v2's loop variables travel in DgParkEntry.xd / .xi, its
// iteration records, merit memory and previous iterate are part of the stored scratch / LDS image).
__device__ inline bool dev_solve_v2(const Ctx& c, cgptr u_ws, int64_t b, const SolveOutPtrs& O, const DgParkEntry* resume = nullptr,
                                    long long ticket = 0, unsigned long long ticks0 = 0ull, int* iters_out = nullptr) {
  const DgProb& D = dg_prob;
  const DgLds& L = D.L;
  lptr lds = LP(0);
  lptr sc = lds + L.scal;
  const int n = D.n, nc = D.nc;
  const dgsqp_params_t& par = D.par;
  gptr rec_ckpt = c.ws + D.ws_v2, rec_last = rec_ckpt + v2_rec_doubles(D), rec_cur = rec_last + v2_rec_doubles(D);
  gptr im1 = rec_cur + v2_rec_doubles(D);          // u_im1 [n], l_im1 [nc]
  __syncthreads();
  const bool timed = par.time_limit >= 0.0;
  double t_start = 0.0, obj0 = 0.0;
  if (!resume) {
    if (TID == 0) { sc[DG_XVALID] = 0.0; sc[DG_QP_NPREV] = 0.0; sc[DG_REG] = par.reg; sc[DG_ITREC] = 0.0; sc[DG_PSD_PD] = 0.0; sc[DG_OSQP_RHO] = 0.1; }
    for (int i = TID; i < n; i += NT) lds[L.u + i] = u_ws[i];
    for (int r = TID; r < nc; r += NT) lds[L.l + r] = 0.0;
    __syncthreads();
    t_start = timed ? dev_block_clock() : 0.0;
    // dual warm start and the first entry of the merit memory (DGSQP_v2.py:333-343)
    dev_evaluate(c, lds + L.u, 0.0, nullptr, false);
    obj0 = par.merit_function == DGSQP_MERIT_SUM_OBJ_L1 ? dev_v2_sum_obj(c, false) : 0.0;      // (the dual start reuses the EVAL scratch)
    dev_dual_init(c);
    dev_log_iterate(c);
    dev_stat_vector(c, lds + L.l, lds + L.d);
  }
  int mem_n = 0, mem_head = 0;                       // ring: entries sc[DG_V2_MEM0 + (mem_head + k) % size], k < mem_n
  const int mem_size = par.nms_memory_size < 1 ? 1 : (par.nms_memory_size > 16 ? 16 : par.nms_memory_size);
  auto mem_append = [&](double v) {
    __syncthreads();
    if (mem_n < mem_size) { if (TID == 0) sc[DG_V2_MEM0 + (mem_head + mem_n) % mem_size] = v; mem_n++; }
    else { if (TID == 0) sc[DG_V2_MEM0 + mem_head] = v; mem_head = (mem_head + 1) % mem_size; }
    __syncthreads();
  };
  auto mem_max = [&]() { double m = -INFINITY; for (int k = 0; k < mem_n; k++) m = fmax(m, sc[DG_V2_MEM0 + (mem_head + k) % mem_size]); return m; };
  if (!resume) {
    double dd = 0, vio = 0;
    for (int i = TID; i < n; i += NT) dd += lds[L.d + i] * lds[L.d + i];
    for (int r = TID; r < nc; r += NT) vio += fmax(lds[L.g + r], 0.0);
    dd = block_sum(dd, lds + L.red); vio = block_sum(vio, lds + L.red);
    mem_append((par.merit_function == DGSQP_MERIT_SUM_OBJ_L1 ? obj0 : 0.5 * dd) + vio);      // nms_initial_reference_factor = 1 (DGSQP_v2.py:213)
    for (int i = TID; i < n; i += NT) im1[i] = lds[L.u + i];
    for (int r = TID; r < nc; r += NT) im1[n + r] = lds[L.l + r];
  }
  double reg = par.reg, delta = 0.0, ckpt_delta = 0.0, ckpt_reg = reg;
  int ckpt_counter = 0, ckpt_index = 0;
  int sqp_it = 0, m_step_it = 0, rel_tol_its = 0, total_qp = 0, status = DGSQP_MAX_IT;
  if (resume) {
    sqp_it = resume->sqp_it; rel_tol_its = resume->rel_tol_its; total_qp = resume->total_qp;
    reg = resume->xd[0]; delta = resume->xd[1]; ckpt_delta = resume->xd[2]; ckpt_reg = resume->xd[3];
    m_step_it = resume->xi[0]; ckpt_counter = resume->xi[1]; ckpt_index = resume->xi[2]; mem_n = resume->xi[3]; mem_head = resume->xi[4];
    if (timed) t_start = dev_block_clock() - resume->xd[4];      // the time it spent set aside does not count towards time_limit
  }
  bool finished = false;
  double cond[3] = {0, 0, 0};
  while (true) {
    if (TID == 0) sc[DG_REG] = reg;
    dev_evaluate(c, lds + L.u, 0.0, nullptr, true);
    dev_stat_vector(c, lds + L.l, lds + L.d);
    {
      double gm = -INFINITY, cm = 0, sm = 0;
      for (int r = TID; r < nc; r += NT) { gm = fmax(gm, lds[L.g + r]); cm = fmax(cm, fabs(lds[L.g + r] * lds[L.l + r])); }
      for (int i = TID; i < n; i += NT) sm = fmax(sm, fabs(lds[L.d + i]));
      cond[0] = fmax(0.0, block_max(gm, lds + L.red));
      cond[1] = block_max(cm, lds + L.red);
      cond[2] = block_max(sm, lds + L.red);
    }
    dev_tr(c, 1, cond[2]); dev_tr(c, 2, cond[0]); dev_tr(c, 3, cond[1]);
    // termination tests in the reference's order -- later ones overwrite the message of earlier ones (DGSQP_v2.py:389-414)
    if (cond[2] > 1e10) { finished = true; status = DGSQP_DIVERGED; }
    if (cond[0] < par.p_tol && cond[1] < par.d_tol && cond[2] < par.d_tol) { finished = true; status = DGSQP_CONV_ABS_TOL; }
    if (m_step_it >= par.sqp_iters) { finished = true; status = DGSQP_MAX_IT; }
    if (timed && (dev_block_clock() - t_start) * 1e-8 > par.time_limit) { finished = true; status = DGSQP_TIME_LIMIT; }
    if (finished) { dev_tr(c, 40, 0.0); dev_log_iterate(c); break; }
    // the iterate this iteration starts from (iter_data[sqp_it].primal_iterate / dual_iterate)
    gptr ubase = c.ws + D.ws_base, lbase = ubase + 2 * n;            // (the v1 watchdog's backup area is free in v2)
    for (int i = TID; i < n; i += NT) ubase[i] = lds[L.u + i];
    for (int r = TID; r < nc; r += NT) lbase[r] = lds[L.l + r];
    __threadfence_block();
    const int flag = v2_qp(c);
    total_qp++;
    bool d_step = false, m_step = false, loaded = false;
    double mu = 0.0;
    LinScal S;
    S.dstat = S.vio = S.phi = 0.0;
    if (flag != 0) {
      if (!par.nms || sqp_it == 0) { dev_tr(c, 40, 1.0); dev_log_iterate(c); status = DGSQP_QP_FAIL; break; }     // :432-465
      m_step = true;
      mu = v2_rec_load(c, ckpt_index <= sqp_it - 1 ? rec_ckpt : rec_last);     // _idx = min(checkpoint_index, len(iter_data) - 1)
      v2_rec_copy(c, rec_cur, ckpt_index <= sqp_it - 1 ? rec_ckpt : rec_last);
      loaded = true;
    } else {
      double nrm = 0;
      for (int i = TID; i < n; i += NT) nrm += lds[L.o_du + i] * lds[L.o_du + i];
      for (int r = TID; r < nc; r += NT) { const double t = lds[L.o_lhat + r] - lds[L.l + r]; nrm += t * t; }
      nrm = sqrt(block_sum(nrm, lds + L.red));
      if (sqp_it == 0) { delta = 20.0 * nrm; ckpt_delta = delta; }       // nms_initial_step_size_factor (DGSQP_v2.py:212,469-472)
      if (par.nms) {
        if (ckpt_counter >= par.nms_frequency) m_step = true;
        else if (nrm < delta) d_step = true;
        else m_step = true;
      }
      dev_step_scalars(c, S);
      if (par.merit_parameter < 0.0) mu = S.vio > 0 ? fabs(S.dstat) / (0.5 * S.vio) : 0.0;       // _get_mu (DGSQP_v2.py:665-690)
      else mu = par.merit_parameter;
      dev_tr(c, 10, nrm * nrm); dev_tr(c, 11, mu);
      v2_rec_store(c, rec_cur, ubase, lbase, mu);
    }
    double phi_new = 0.0;
    bool searched = false;
    if (d_step) {                                   // relaxed step (DGSQP_v2.py:503-510)
      dev_take_full_step(c);
      delta *= par.delta_decay;
      ckpt_counter++;
    }
    if (m_step || (!d_step && !m_step)) {
      bool accept = false;
      if (m_step) {
        m_step_it++;
        dev_evaluate_point(c, lds + L.u, 1.0, lds + L.o_du, nullptr, true);
        const double phi = dev_trial_merit(c, 1.0, 0.0, 1.0);
        const double R = (1.0 - par.merit_decrease) * mem_max();
        dev_tr(c, 20, phi);
        if (phi <= R) { accept = true; phi_new = phi; dev_take_full_step(c); }
        else if (ckpt_index <= sqp_it - 1) {        // watchdog: back to the checkpoint (DGSQP_v2.py:533-544)
          mu = v2_rec_load(c, rec_ckpt);
          v2_rec_copy(c, rec_cur, rec_ckpt);
          loaded = true;
          delta = ckpt_delta;
          reg = ckpt_reg;
        }
      }
      if (!accept) {
        double phi_b = S.phi + mu * S.vio, dphi_b = S.dstat - mu * S.vio;
        if (loaded && par.merit_decrease_condition == DGSQP_DECREASE_ARMIJO) {
          // base values of the Armijo test at the loaded point: _evaluate(hessian=True) there (DGSQP_v2.py:734-737)
          dev_evaluate(c, lds + L.u, 0.0, nullptr, true);
          dev_stat_vector(c, lds + L.l, lds + L.d);
          v2_merit_vector(c);
          LinScal Sb;
          dev_step_scalars(c, Sb);
          phi_b = Sb.phi + mu * Sb.vio; dphi_b = Sb.dstat - mu * Sb.vio;
        }
        phi_new = v2_line_search(c, mu, phi_b, dphi_b, mem_max());
        searched = true;
        dev_tr(c, 22, phi_new);
      }
      // relative-tolerance exit, bookkeeping of an m-step / line-search step (DGSQP_v2.py:548-572, :575-594)
      double du2 = 0, dl2 = 0;
      for (int i = TID; i < n; i += NT) { const double t = lds[L.u + i] - im1[i]; du2 += t * t; }
      for (int r = TID; r < nc; r += NT) { const double t = lds[L.l + r] - im1[n + r]; dl2 += t * t; }
      du2 = block_sum(du2, lds + L.red); dl2 = block_sum(dl2, lds + L.red);
      if (sqrt(du2) < par.p_tol && sqrt(dl2) < par.d_tol) {
        rel_tol_its++;
        if (rel_tol_its >= par.rel_tol_req && cond[0] < par.p_tol) { finished = true; status = DGSQP_CONV_REL_TOL; }
      } else rel_tol_its = 0;
      __syncthreads();
      for (int i = TID; i < n; i += NT) im1[i] = lds[L.u + i];
      for (int r = TID; r < nc; r += NT) im1[n + r] = lds[L.l + r];
      __threadfence_block();
      reg *= par.reg_decay;
      mem_append(phi_new);
      if (m_step) { ckpt_counter = 0; ckpt_delta = delta; ckpt_reg = reg; ckpt_index = sqp_it + 1; }
    }
    (void)searched;
    dev_tr(c, 40, 1.0);
    dev_log_iterate(c);
    // iter_data.append(_data): this iteration's record becomes the latest one, and the checkpoint's if it is the checkpoint iteration
    v2_rec_copy(c, rec_last, rec_cur);
    if (ckpt_index == sqp_it) v2_rec_copy(c, rec_ckpt, rec_cur);
    sqp_it++;
    // deferral of long scenarios (dev_solve): set aside here, resumed at the top of the loop.  Unlike v1, also with a wall-clock limit (the
    // v2 study always sets one, 600 s: comparison_study_barc/globals.py:40): the entry carries the time spent solving so far
    if (!resume && !finished) {
      const long long slot = dev_park_reserve(c, sqp_it, ticks0);
      if (slot >= 0) {
        const unsigned long long now = dev_bcast_u64(TID == 0 ? wall_clock64() : 0ull);
        const double sm = cond[2] == cond[2] ? fmin(cond[2], 1e30) : 1e30;
        const unsigned long long key = (unsigned long long)((double)(now - ticks0) * (1.0 + 2.0 * log10(1.0 + sm)));
        const double xd[6] = {reg, delta, ckpt_delta, ckpt_reg, timed ? dev_block_clock() - t_start : 0.0, 0.0};
        const int xi[6] = {m_step_it, ckpt_counter, ckpt_index, mem_n, mem_head, 0};
        dev_park_store(c, (unsigned int)slot, sqp_it, rel_tol_its, total_qp, ticket, key, cond, xd, xi);
        return true;
      }
    }
  }
  if (iters_out) *iters_out = sqp_it;
  // outputs
  __syncthreads();
  lds_d* ue = lds + L.e_ue;
  dev_final_rollout(c);
  if (O.cost) dev_costs(c, ue, O.cost + b * D.M);
  if (O.u) for (int i = TID; i < n; i += NT) O.u[b * n + i] = lds[L.u + i];
  if (O.l) for (int r = TID; r < nc; r += NT) O.l[b * nc + r] = lds[L.l + r];
  if (O.x) for (int i = TID; i < (D.N + 1) * D.nq; i += NT) O.x[b * (int64_t)(D.N + 1) * D.nq + i] = lds[L.e_x + i];
  if (TID == 0) {
    if (O.status) O.status[b] = status;
    if (O.iters) O.iters[b] = sqp_it;
    if (O.qp_solves) O.qp_solves[b] = total_qp;
    if (O.cond) for (int i = 0; i < 3; i++) O.cond[b * 3 + i] = cond[i];
  }
  __syncthreads();
  return false;
}
