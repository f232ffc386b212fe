// QP sub-problem of one SQP iteration.
//
// _solve_qp (DGSQP.py:232-266):  min 1/2 x'Bx + q'x  s.t.  G x <= -g,  with P = B^-1 packed in LDS.
// Dual active-set method (Goldfarb & Idnani 1983) in range-space form.  The problem is small (n <= 128 unknowns, up to n
// active rows) and a step is a chain of short dependent operations, so the work is split by its nature:
//   * block-wide phases (all wavefronts): the primal point x = x_u - Y lam, the scan for the most violated row, y = P a_p, dots
//     with the dense gradients;
//   * wavefront-0 sections without any block barrier: the step direction in the multipliers, r = S^-1 A_W y through the inverse
//     Cholesky factor T of the Schur complement S = A_W P A_W^T (two lane-parallel products, qpt_solve), step lengths, updates of T
//     (a bordering column per added row, column rotations per removed one, qpt_drop).
// Vector element i lives in lane i & 63 of wavefront 0 (two registers per lane for n > 64).  For every active row j the
// vector y_j = P a_j is kept (global scratch, L2 resident).  The primal point is a function of the multipliers,
// x = -P (q + A_W^T lam) = x_u - Y lam: inside the wavefront-0 section only the violation of row p is tracked (it falls by t delta
// per primal step), x itself is rebuilt block-wide once per added row -- a thrashing solve takes tens of thousands of steps, and a
// product with Y per step (one wavefront reading m columns from L2) was three quarters of their cost.
// The result is the KKT point OSQP(polish=True) returns when its polish succeeds.
#pragma once

// Dots of all distinct dense constraint gradients with the vector held in LDS at v (all wavefronts).  One chunk
// (<= DG_CHUNK contiguous entries of a gradient and of v) per thread, all 2 x DG_CHUNK reads issued together; the chunk
// sums of a gradient are then added in a fixed order (bitwise reproducible, no atomics).  Two barriers.
template <class GP>
__device__ inline void qp_dense_dots(const DgProb& D, GP gd, clptr v, lptr part, lptr out) {
  for (int t = TID; t < D.ntask; t += NT) {
    const DgTask T = ld_task(t);
    const GP p = gd + T.p0;
    clptr w = v + T.v0;
    double pv[DG_CHUNK], wv[DG_CHUNK];
#pragma unroll
    for (int i = 0; i < DG_CHUNK; i++) { pv[i] = p[i]; wv[i] = w[i]; }   // reads past the chunk stay inside the LDS arena
    double s[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < DG_CHUNK; i++) s[i & 3] += i < T.len ? pv[i] * wv[i] : 0.0;
    part[t] = (s[0] + s[1]) + (s[2] + s[3]);
  }
  __syncthreads();
  for (int d = TID; d < D.ndense; d += NT) {
    const DgDense dd = ld_dense(d);
    const int ts = dd.t0lo + 256 * dd.t0hi;
    double s = 0;
    for (int i = 0; i < dd.nt; i++) s += part[ts + i];
    out[d] = s;
  }
  __syncthreads();
}
// a_r . v with the dense part taken from precomputed dense dots.  Branch-free: every row is  s1 v[c1] + s0 v[c0] + sd dd[di]
// with zero coefficients for the parts it does not have, so the three LDS reads are unconditional and independent.
__device__ inline double qpw_row_dot(const DgProb& D, const DgRow R, clptr v, clptr dd) {
  const bool dense = R.dense >= 0;
  const bool rate = R.type == DG_R_RATE_UB || R.type == DG_R_RATE_LB;
  const bool pos = R.type == DG_R_IN_UB || R.type == DG_R_RATE_UB;
  const int c1 = dense ? 0 : am_col(D, R.a, R.k, R.idx);
  const bool has0 = rate && R.k > 0;
  const int c0 = has0 ? c1 - DGSQP_NUA : c1;
  const double s1 = dense ? 0.0 : (pos ? 1.0 : -1.0);
  const double s0 = has0 ? -s1 : 0.0;
  const double sd = dense ? (double)R.sgn : 0.0;
  const int di = dense ? R.dense : 0;
  const double v1 = v[c1], v0 = v[c0], vd = dd[di];
  return s1 * v1 + s0 * v0 + sd * vd;
}

// out = sum_j coef_j Y[slot_j]  (lane-distributed), coefficients in LDS at cf, slots in LDS at ys.  Y lives in the
// workgroup's global scratch (L2): eight columns per iteration so that the loads overlap.
__device__ inline void qpw_ymul(cgptr Y, int n, int m, int lane, clptr cf, const lds_i_t* ys, double& o0, double& o1) {
  double a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
  const bool okA = lane < n, okB = lane + 64 < n;
  const int la = okA ? lane : 0, lb = okB ? lane + 64 : 0;
  int j = 0;
  for (; j + 7 < m; j += 8) {
    double ya[8], yb[8], cc[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int64_t base = (int64_t)ys[j + k] * n;
      cc[k] = cf[j + k];
      ya[k] = Y[base + la];
      yb[k] = n > 64 ? Y[base + lb] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < 8; k++) { a[k & 3] = __builtin_fma(ya[k], cc[k], a[k & 3]); b[k & 3] = __builtin_fma(yb[k], cc[k], b[k & 3]); }
  }
  for (; j < m; j++) {
    const int64_t base = (int64_t)ys[j] * n;
    const double cc = cf[j];
    a[0] = __builtin_fma(Y[base + la], cc, a[0]);
    if (n > 64) b[0] = __builtin_fma(Y[base + lb], cc, b[0]);
  }
  o0 = okA ? (a[0] + a[1]) + (a[2] + a[3]) : 0.0; o1 = okB ? (b[0] + b[1]) + (b[2] + b[3]) : 0.0;
}

// state of the active-set iteration kept in wavefront 0's registers between the block-wide phases
struct QpwState {
  double x0, x1;   // primal point (elements lane, lane + 64): only where a section updates it itself (warm start, polish)
  int m, nfree;    // active rows, free Y slots
  int ill;         // a row was accepted or rejected on a curvature below 1e-9 of its unprojected value (reg = 0 regime)
};
struct QpPtrs {
  lptr xv, R, lam, cvec, wv, rv, rd, yv, tv, ddx, ddy, dpart, scal, prevlam, xu, part;   // (R: the slot of the inverse factor T)
  lds_i_t *alist, *yslot, *yfree, *prev;
  lds_b_t* act;
  clptr gd, g, Pp;
  cgptr gdG;   // packed gradients in the global scratch (DgProb.gd_global), else null
  cgptr PpG;   // packed P in the global scratch (big layout), else unused
  gptr Y;
};
__device__ inline QpPtrs qp_ptrs(const Ctx& c) {
  const DgProb& D = dg_prob;
  const DgLds& L = D.L;
  lptr lds = LP(0);
  QpPtrs q;
  q.xv = lds + L.o_du; q.R = lds + L.p_R; q.lam = lds + L.p_lam; q.cvec = lds + L.p_c; q.wv = lds + L.p_w; q.rv = lds + L.p_r;
  q.rd = lds + L.p_rd; q.yv = lds + L.p_y; q.tv = lds + L.p_t; q.ddx = lds + L.yd; q.ddy = lds + L.p_yd2; q.dpart = lds + L.p_dpart;
  q.scal = lds + L.scal; q.prevlam = lds + L.w_prevlam; q.xu = lds + L.p_xu; q.part = lds + L.p_part;
  q.alist = (lds_i_t*)(lds + L.p_alist); q.yslot = (lds_i_t*)(lds + L.p_yslot); q.yfree = (lds_i_t*)(lds + L.p_yfree);
  q.prev = (lds_i_t*)(lds + L.w_prev); q.act = (lds_b_t*)(lds + L.p_act);
  q.gd = lds + L.gd; q.gdG = D.gd_global ? c.ws + D.ws_gd : nullptr; q.g = lds + L.g; q.Pp = lds + (D.big ? 0 : L.g_Bp); q.PpG = c.ws + D.ws_P;
  q.Y = c.ws + D.ws_Y;
  return q;
}
// The packed gradients live in LDS or -- DgProb.gd_global: XL layouts, and the half-arena build's n = 100 games -- in the workgroup's scratch
__device__ inline double qp_gcoef(const DgProb& D, const QpPtrs& q, int p, int col) {
  return q.gdG ? g_row_coef<cgptr>(D, q.gdG, p, col) : g_row_coef<clptr>(D, q.gd, p, col);
}
__device__ inline void qp_gd_dots(const DgProb& D, const QpPtrs& q, clptr v, lptr part, lptr out) {
  if (q.gdG) qp_dense_dots<cgptr>(D, q.gdG, v, part, out); else qp_dense_dots<clptr>(D, q.gd, v, part, out);
}
// a_i . y_j and |a_i|^2 of a dense row's packed gradient (one or two agents' blocks) against a column y_j of Y
template <class GP>
__device__ inline void qp_dense_pair(const DgProb& D, const DgDense dd, GP gp, cgptr yj, double& dot, double& nrm2) {
  const int len = DGSQP_NUA * dd.k;
  double s0 = 0, s1 = 0, n0 = 0, n1 = 0;
  for (int part = 0; part < (dd.kind == 1 ? 2 : 1); part++) {
    cgptr yy = yj + (part == 0 ? dd.a : dd.b) * D.N * DGSQP_NUA;
    GP gg = gp + part * len;
    int e = 0;
    for (; e + 1 < len; e += 2) {
      const double g0 = gg[e], g1 = gg[e + 1];
      s0 = __builtin_fma(g0, yy[e], s0); s1 = __builtin_fma(g1, yy[e + 1], s1);
      n0 = __builtin_fma(g0, g0, n0); n1 = __builtin_fma(g1, g1, n1);
    }
    if (e < len) { const double g0 = gg[e]; s0 = __builtin_fma(g0, yy[e], s0); n0 = __builtin_fma(g0, g0, n0); }
  }
  dot = s0 + s1; nrm2 = n0 + n1;
}
// All wavefronts: a_p into tv, y = P a_p into yv (a column copy for box / rate rows), dense-gradient dots of y into ddy.
__device__ inline void qp_row_products(const Ctx& c, const QpPtrs& q, int p) {
  const DgProb& D = dg_prob;
  const int n = D.n;
  const DgRow R = ld_row(p);
  __syncthreads();   // previous readers of tv / yv are done
  if (R.dense < 0) {
    const int c1 = am_col(D, R.a, R.k, R.idx);
    const bool has0 = (R.type == DG_R_RATE_UB || R.type == DG_R_RATE_LB) && R.k > 0;
    const double sgn = (R.type == DG_R_IN_UB || R.type == DG_R_RATE_UB) ? 1.0 : -1.0;
    for (int i = TID; i < n; i += NT) {
      double pv = D.big ? q.PpG[tri(i, c1)] : q.Pp[tri(i, c1)], av = i == c1 ? 1.0 : 0.0;
      if (has0) { pv -= D.big ? q.PpG[tri(i, c1 - DGSQP_NUA)] : q.Pp[tri(i, c1 - DGSQP_NUA)]; if (i == c1 - DGSQP_NUA) av = -1.0; }
      q.yv[i] = sgn * pv; q.tv[i] = sgn * av;
    }
    __syncthreads();
  } else {
    for (int col = TID; col < n; col += NT) q.tv[col] = qp_gcoef(D, q, p, col);
    dev_p_mul(c, q.tv, q.yv, 1.0);
  }
  qp_gd_dots(D, q, q.yv, q.dpart, q.ddy);
}
// All wavefronts: xv = x_u - Y lam for the m active rows (slots in yslot).  Thread = element i x quarter of the columns: every
// thread has its loads from L2 in flight together; the four partial sums meet in LDS.
__device__ inline void qp_x_from_lambda(const QpPtrs& q, int m) {
  const DgProb& D = dg_prob;
  const int n = D.n;
  constexpr int NSEG = NT / 128;
  const int i = TID & 127, sg = TID >> 7;
  __syncthreads();
  if (i < n) {
    const int len = (m + NSEG - 1) / NSEG;
    const int j0 = sg * len, j1 = (j0 + len < m) ? j0 + len : m;
    double a[4] = {0, 0, 0, 0};
    int j = j0;
    for (; j + 7 < j1; j += 8) {
      double yv[8], lv[8];
#pragma unroll
      for (int k = 0; k < 8; k++) { yv[k] = q.Y[(int64_t)q.yslot[j + k] * n + i]; lv[k] = q.lam[j + k]; }
#pragma unroll
      for (int k = 0; k < 8; k++) a[k & 3] = __builtin_fma(yv[k], lv[k], a[k & 3]);
    }
    for (; j < j1; j++) a[0] = __builtin_fma(q.Y[(int64_t)q.yslot[j] * n + i], q.lam[j], a[0]);
    q.part[sg * n + i] = (a[0] + a[1]) + (a[2] + a[3]);
  }
  __syncthreads();
  if (TID < n) {
    double s = 0;
#pragma unroll
    for (int g = 0; g < NSEG; g++) s += q.part[g * n + TID];
    q.xv[TID] = q.xu[TID] - s;
  }
  __syncthreads();
}
// All wavefronts: most violated inactive row at the point held in xv (lowest index on ties); NONE if the point is feasible.
__device__ inline int qp_scan(const QpPtrs& q, double tol) {
  const DgProb& D = dg_prob;
  const int NONE = 0x7fffffff;
  qp_gd_dots(D, q, q.xv, q.dpart, q.ddx);
  double best = -tol;
  int bi = NONE;
  for (int r = TID; r < D.nc; r += NT) {
    if (q.act[r]) continue;
    const double s = -(q.g[r] + qpw_row_dot(D, ld_row(r), q.xv, q.ddx));
    if (s < best) { best = s; bi = r; }     // increasing r per thread: the first minimum is kept
  }
  double bv; int p;
  block_argmin(best, bi, LP(D.L.red), bv, p);
  return p;
}
// wavefront 0: append row p (whose y is in yv, a_p in tv, w in wv) to the factorisation
__device__ inline void qpw_append(const QpPtrs& q, QpwState& S, int n, int lane, int p, double delta, double lam_p) {
  // (r = T w of the step that made p active is in rv: the new column of T is (-r / rho; 1 / rho), rho^2 = delta)
  const int m = S.m;
  const double ird = 1.0 / sqrt(delta);
  qpt_put_column(q.R, m, lane, lane < m ? -q.rv[lane] * ird : 0.0, lane + 64 < m ? -q.rv[lane + 64] * ird : 0.0, ird);
  const int slot = q.yfree[S.nfree - 1];
  if (lane < n) q.Y[(int64_t)slot * n + lane] = q.yv[lane];
  if (lane + 64 < n) q.Y[(int64_t)slot * n + lane + 64] = q.yv[lane + 64];
  if (lane == 0) { q.alist[m] = p; q.lam[m] = lam_p; q.act[p] = 1; q.yslot[m] = slot; }
  S.nfree--; S.m++;
}
__device__ inline void qpw_remove(const QpPtrs& q, QpwState& S, int lane, int jd) {
  const int freed = q.yslot[jd];
  qpt_drop(q.R, q.alist, q.yslot, q.lam, q.act, S.m, jd, lane, q.wv, q.rv);
  if (lane == 0) q.yfree[S.nfree] = freed;
  S.nfree++; S.m--;
}
// wavefront 0: steps 2a-2c of the dual method for the row p prepared by qp_row_products, until p is active (returns 0)
// or the QP is found infeasible (returns 1)
__device__ inline int qpw_add_constraint(const QpPtrs& q, QpwState& S, int lane, int p) {
  const DgProb& D = dg_prob;
  const int n = D.n, npk = n * (n + 1) / 2;
  const int NONE = 0x7fffffff;
  const bool okA = lane < n, okB = lane + 64 < n;
  const double t0 = okA ? q.tv[lane] : 0.0, t1 = okB ? q.tv[lane + 64] : 0.0;
  const double y0 = okA ? q.yv[lane] : 0.0, y1 = okB ? q.yv[lane + 64] : 0.0;
  const double app = wave_sum(t0 * y0 + t1 * y1), apap = wave_sum(t0 * t0 + t1 * t1);
  const double gp = q.g[p];
  double lp = 0.0;
  (void)npk;
  // a_p.x - b_p  (b = -g), > 0, at the point the scan looked at; every primal step x += t z lowers it by t delta (a_p.z = -delta)
  double viol = wave_sum(t0 * (okA ? q.xv[lane] : 0.0) + t1 * (okB ? q.xv[lane + 64] : 0.0)) + gp;
  for (int inner = 0; inner < 4 * (n + D.nc); inner++) {
    // ---- step 2a: directions.  c = A_A y ; w = T^T c ; r = T w
    PROF_BEGIN(pq3);
    const int m = S.m;
    for (int j = lane; j < m; j += 64) q.cvec[j] = qpw_row_dot(D, ld_row(q.alist[j]), q.yv, q.ddy);
    double r0, r1;
    const double ww = qpt_solve(q.R, m, lane, q.cvec, q.wv, q.rv, r0, r1);
    // ---- step 2b: step lengths.  t1 keeps the multipliers >= 0, t2 makes constraint p active
    double ta = INFINITY; int jd = NONE;
    if (lane < m && r0 > 0) { ta = q.lam[lane] / r0; jd = lane; }
    if (lane + 64 < m && r1 > 0) { const double tt = q.lam[lane + 64] / r1; if (tt < ta) { ta = tt; jd = lane + 64; } }
    wave_argmin(ta, jd);
    const double delta = app - ww;  // a_p^T (P - P A^T S^-1 A P) a_p >= 0
    // Is a_p independent of the active rows (n active rows span everything)?  delta is a difference of numbers of size app:
    // above 1e-9 app it is trusted.  With reg = 0 the projected Hessian keeps eigenvalues of 1e-10, P spans ten decades and
    // legitimate rows come down to delta ~ 1e-14 app, next to the rounding noise an exactly dependent row leaves.  There
    // the explicit primal direction z = Y r - y decides: for an independent row -a_p.z reproduces delta, for a dependent
    // one both are unrelated noise.
    bool indep = m < n && delta > 3e-15 * app && delta > 1e-18 * apap;
    if (indep && !(delta > 1e-9 * app)) {
      double d0, d1;
      qpw_ymul(q.Y, n, m, lane, q.rv, q.yslot, d0, d1);
      const double z0 = d0 - y0, z1 = d1 - y1;
      S.ill = 1;
      const double dz = -wave_sum(t0 * z0 + t1 * z1);
      indep = __builtin_fabs(delta - dz) <= 0.3 * delta;
    }
    const double tb = indep ? viol / delta : INFINITY;
    const double t = fmin(ta, tb);
    PROF_END(PH_Q_DIR, pq3);
    if (!(t < INFINITY)) return 1;
    PROF_BEGIN(pq4);
    if (indep) viol -= t * delta;                    // x += t z, z = Y r - y
    if (lane < m) q.lam[lane] -= t * r0;
    if (lane + 64 < m) q.lam[lane + 64] -= t * r1;
    lp += t;
    PROF_END(PH_Q_STEP, pq4);
    PROF_BEGIN(pq5);
    if (indep && !(ta < tb)) {   // full step: constraint p becomes active
      qpw_append(q, S, n, lane, p, delta, lp);
      PROF_END(PH_Q_UPD, pq5);
      return 0;
    }
    // partial / dual-only step: drop blocking constraint jd (column deletion + Givens)
#ifdef DG_PROF
    if (lane == 0) { atomicAdd(&dg_prof[2 * PH_SWEEP + 1], 1ULL); }
#endif
    qpw_remove(q, S, lane, jd);
    PROF_END(PH_Q_UPD, pq5);
  }
  return 1;
}

// ------------------------------------------------------------------------------------------------
// _solve_qp core (DGSQP.py:246).  Out: du (L.o_du), lhat (L.o_lhat).  Returns 0 ok, 1 infeasible, 2 iteration limit.
// Block-wide phases (scan, P a_p, dense dots) alternate with wavefront-0 sections; the other wavefronts wait at the
// barrier that ends each section.
// ------------------------------------------------------------------------------------------------
__device__ __noinline__ int dev_qp(const Ctx& c) {
  const DgProb& D = dg_prob;
  const DgLds& L = D.L;
  lptr lds = LP(0);
  const int n = D.n, nc = D.nc, npk = n * (n + 1) / 2;
  const QpPtrs q = qp_ptrs(c);
  lptr lhat = lds + L.o_lhat;
  const int lane = TID & 63;
  const bool w0 = TID < 64;
  const bool okA = lane < n, okB = lane + 64 < n;
  const double TOL = 1e-10;
  const int NONE = 0x7fffffff;
  __syncthreads();
  PROF_BEGIN(pt_qp);
  for (int r = TID; r < nc; r += NT) { q.act[r] = 0; lhat[r] = 0.0; }
  if (TID == 0) q.scal[2] = 0.0;   // set when the active-set loop meets the ill-conditioned regime
  for (int i = TID; i < n; i += NT) q.yfree[i] = n - 1 - i;     // stack of free Y slots (top = lowest index)
  dev_p_mul(c, lds + L.q, q.xu, -1.0);  // unconstrained minimiser x_u = -P q
  for (int i = TID; i < n; i += NT) q.xv[i] = q.xu[i];
  __syncthreads();
  QpwState S;
  S.m = 0; S.nfree = n; S.ill = 0;
  S.x0 = 0.0; S.x1 = 0.0;

  // ---- warm start from the final active set W of this scenario's previous QP.  (x(W'), W') with x(W') the minimiser on
  // the rows W' held as equalities and multipliers >= 0 is a valid S-pair for any independent subset W' of W, so the
  // dual method continues from it and reaches the same (unique) minimiser; only the path is shorter.
  const int nprev = D.par.qp_warm_start ? (int)q.scal[DG_QP_NPREV] : 0;
  if (nprev > 0) {
    PROF_BEGIN(pqw);
    // (a) y_j = P a_j for the first NB guessed rows, straight into Y slot j: box / rate rows are column copies (one
    //     wavefront per row), dense rows go through the block-wide product.
    int NB = nprev < 48 ? nprev : 48;
    while (NB > 0 && dg_tcol(NB) + NB * (NB + 1) / 2 > dg_tcol(n)) NB--;
    lptr Sb = q.R + dg_tcol(NB);           // S = A Y (packed by rows) lives behind the part of T the batch can fill
    for (int jj = TID >> 6; jj < NB; jj += NT / 64) {
      const DgRow Rw = ld_row(q.prev[jj]);
      if (Rw.dense >= 0) continue;
      const int c1 = am_col(D, Rw.a, Rw.k, Rw.idx);
      const bool has0 = (Rw.type == DG_R_RATE_UB || Rw.type == DG_R_RATE_LB) && Rw.k > 0;
      const double sgn = (Rw.type == DG_R_IN_UB || Rw.type == DG_R_RATE_UB) ? 1.0 : -1.0;
      for (int i = lane; i < n; i += 64) {
        double pv = D.big ? q.PpG[tri(i, c1)] : q.Pp[tri(i, c1)];
        if (has0) pv -= D.big ? q.PpG[tri(i, c1 - DGSQP_NUA)] : q.Pp[tri(i, c1 - DGSQP_NUA)];
        q.Y[(int64_t)jj * n + i] = sgn * pv;
      }
    }
    for (int jj = 0; jj < NB; jj++) {
      const int p = q.prev[jj];
      if (ld_row(p).dense < 0) continue;          // uniform
      __syncthreads();
      for (int col = TID; col < n; col += NT) q.tv[col] = qp_gcoef(D, q, p, col);
      dev_p_mul(c, q.tv, q.yv, 1.0);
      for (int i = TID; i < n; i += NT) q.Y[(int64_t)jj * n + i] = q.yv[i];
    }
    __threadfence_block();
    __syncthreads();
    // (b) S_ij = a_i . y_j for i <= j < NB, one thread per pair; |a_j|^2 into tv[j]
    for (int t = TID; t < NB * (NB + 1) / 2; t += NT) {
      int j = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
      while ((j + 1) * (j + 2) / 2 <= t) j++;
      while (j * (j + 1) / 2 > t) j--;
      const int i = t - j * (j + 1) / 2;
      const DgRow Ri = ld_row(q.prev[i]);
      cgptr yj = q.Y + (int64_t)j * n;
      double sv, a2;
      if (Ri.dense < 0) {
        const int c1 = am_col(D, Ri.a, Ri.k, Ri.idx);
        const bool has0 = (Ri.type == DG_R_RATE_UB || Ri.type == DG_R_RATE_LB) && Ri.k > 0;
        const double sgn = (Ri.type == DG_R_IN_UB || Ri.type == DG_R_RATE_UB) ? 1.0 : -1.0;
        sv = yj[c1];
        if (has0) sv -= yj[c1 - DGSQP_NUA];
        sv *= sgn;
        a2 = has0 ? 2.0 : 1.0;
      } else {
        const DgDense dd = ld_dense(Ri.dense);
        double dot;
        if (q.gdG) qp_dense_pair<cgptr>(D, dd, q.gdG + dd.off, yj, dot, a2); else qp_dense_pair<clptr>(D, dd, q.gd + dd.off, yj, dot, a2);
        sv = Ri.sgn * dot;
      }
      Sb[t] = sv;
      if (i == j) q.tv[j] = a2;
    }
    __syncthreads();
    // (c) bordering Cholesky of S in wavefront 0 (rows that are numerically dependent on the accepted ones are skipped);
    //     yslot[k] is both the Y slot and the batch index of accepted row k
    if (w0) {
      for (int j = 0; j < NB; j++) {
        const int m = S.m;
        const int rowj = j * (j + 1) / 2;
        for (int k = lane; k < m; k += 64) q.cvec[k] = Sb[rowj + q.yslot[k]];
        const double app = Sb[rowj + j], apap = q.tv[j];
        double ra, rb;
        const double ww = qpt_solve(q.R, m, lane, q.cvec, q.wv, q.rv, ra, rb);
        const double delta = app - ww;
        if (m < n && delta > 1e-11 * app && delta > 1e-18 * apap) {   // conservative: a skipped guess is found again by the main loop
          const double ird = 1.0 / sqrt(delta);
          qpt_put_column(q.R, m, lane, -ra * ird, -rb * ird, ird);
          if (lane == 0) {
            const int p = q.prev[j];
            q.alist[m] = p; q.lam[m] = q.prevlam[j]; q.act[p] = 1; q.yslot[m] = j;
          }
          S.m++;
        }
      }
      // free Y slots: everything not taken by an accepted row
      int base = 0;
      for (int h = 0; h < 2; h++) {
        const int sl = lane + 64 * h;
        bool fr = sl < n;
        for (int k = 0; k < S.m; k++) fr = fr && q.yslot[k] != sl;
        const unsigned long long mask = __ballot(fr);
        if (fr) q.yfree[base + __popcll(mask & ((1ull << lane) - 1ull))] = sl;
        base += __popcll(mask);
      }
      S.nfree = base;
    }
    // (d) guesses beyond the batch (rare): one at a time
    for (int jj = NB; jj < nprev; jj++) {
      const int p = q.prev[jj];
      qp_row_products(c, q, p);
      if (w0) {
        const int m = S.m;
        for (int j = lane; j < m; j += 64) q.cvec[j] = qpw_row_dot(D, ld_row(q.alist[j]), q.yv, q.ddy);
        const double t0 = okA ? q.tv[lane] : 0.0, t1 = okB ? q.tv[lane + 64] : 0.0;
        const double y0 = okA ? q.yv[lane] : 0.0, y1 = okB ? q.yv[lane + 64] : 0.0;
        const double app = wave_sum(t0 * y0 + t1 * y1), apap = wave_sum(t0 * t0 + t1 * t1);
        double ra, rb;
        const double ww = qpt_solve(q.R, m, lane, q.cvec, q.wv, q.rv, ra, rb);
        const double delta = app - ww;
        if (m < n && delta > 1e-11 * app && delta > 1e-18 * apap) qpw_append(q, S, n, lane, p, delta, q.prevlam[jj]);
      }
    }
    __syncthreads();
    qp_gd_dots(D, q, q.xv, q.dpart, q.ddx);     // xv still holds the unconstrained minimiser
    if (w0) {
      PROF_COUNT(PH_C_NPREV, nprev); PROF_COUNT(PH_C_MBUILD, S.m);
      // Multipliers: primal active-set steps on the dual problem restricted to W (min 1/2 l'Sl - v'l, l >= 0), from the
      // previous QP's multipliers towards the equality solution l_eq = S^-1 v, v = A x_unc - b; the first multiplier to
      // reach zero leaves W and l_eq is recomputed.
      for (int j = lane; j < S.m; j += 64) q.cvec[j] = q.g[q.alist[j]] + qpw_row_dot(D, ld_row(q.alist[j]), q.xv, q.ddx);
      while (S.m > 0) {
        const int m = S.m;
        double r0, r1;
        (void)qpt_solve(q.R, m, lane, q.cvec, q.wv, q.rv, r0, r1);   // rv = l_eq
        const double l0 = lane < m ? q.lam[lane] : 0.0, l1 = lane + 64 < m ? q.lam[lane + 64] : 0.0;
        double t = INFINITY; int jd = NONE;
        if (lane < m && r0 < 0.0) { t = l0 > 0.0 ? l0 / (l0 - r0) : 0.0; jd = lane; }
        if (lane + 64 < m && r1 < 0.0) { const double t1 = l1 > 0.0 ? l1 / (l1 - r1) : 0.0; if (t1 < t) { t = t1; jd = lane + 64; } }
        wave_argmin(t, jd);
        if (jd == NONE) {
          if (lane < m) q.lam[lane] = r0;
          if (lane + 64 < m) q.lam[lane + 64] = r1;
          break;
        }
        if (lane < m) q.lam[lane] = l0 + t * (r0 - l0);
        if (lane + 64 < m) q.lam[lane + 64] = l1 + t * (r1 - l1);
        const double ca = (lane >= jd && lane + 1 < m) ? q.cvec[lane + 1] : 0.0, cb = (lane + 64 >= jd && lane + 65 < m) ? q.cvec[lane + 65] : 0.0;
        qpw_remove(q, S, lane, jd);
        if (lane >= jd && lane + 1 < m) q.cvec[lane] = ca;
        if (lane + 64 >= jd && lane + 65 < m) q.cvec[lane + 64] = cb;
      }
      PROF_COUNT(PH_C_MWARM, S.m);
      if (lane == 0) q.scal[1] = (double)S.m;
    }
    __syncthreads();
    qp_x_from_lambda(q, (int)q.scal[1]);     // x = x_u - Y l
    PROF_END(PH_Q_WARM, pqw);
  }

  int ret = 2;
  const int max_outer = 4 * (n + nc);
  int iter = 0;
  // The active-set loop ends at a point the scan finds feasible; the polish below (what OSQP's polish does with
  // polish_refine_iter) then moves it, so the loop is entered again until a polished point passes the scan unchanged.
  for (int round = 0; round < 4; round++) {
    int added = 0;
    ret = 2;
    for (; iter < max_outer; iter++) {
      // ---- step 1: most violated inactive constraint (lowest index on ties)
      PROF_BEGIN(pq1);
      const int p = qp_scan(q, TOL);
      PROF_END(PH_Q_SCAN, pq1);
      if (p == NONE) { ret = 0; break; }
      added++;
#ifdef DG_PROF
      if (TID == 0) { atomicAdd(&dg_prof[2 * PH_SWEEP], 1ULL); }
#endif
      PROF_BEGIN(pq2);
      qp_row_products(c, q, p);
      PROF_END(PH_Q_Y, pq2);
      if (w0) {
        const int st = qpw_add_constraint(q, S, lane, p);
        if (lane == 0) { q.scal[0] = (double)st; q.scal[1] = (double)S.m; q.scal[2] = (double)S.ill; }
      }
      __syncthreads();
      if (q.scal[0] != 0.0) { ret = 1; break; }
      PROF_BEGIN(pq7);
      qp_x_from_lambda(q, (int)q.scal[1]);
      PROF_END(PH_Q_STEP, pq7);
    }
    if (ret != 0 || (round > 0 && added == 0)) break;
    PROF_BEGIN(pq6);
    // (x sits on the stationarity manifold of the multipliers, x = -P (q + A^T lam), by construction: qp_x_from_lambda)
    if (w0) { S.x0 = okA ? q.xv[lane] : 0.0; S.x1 = okB ? q.xv[lane + 64] : 0.0; }
    // Iterative refinement on the active set: P is an explicit inverse, so the active rows hold to ~1e-12 only; two
    // projection steps  x <- x - Y S^-1 (A x - b),  lam <- lam + S^-1 (A x - b)  (Y = P A^T) bring them to rounding level.
    for (int pass = 0; pass < 2; pass++) {
      qp_gd_dots(D, q, q.xv, q.dpart, q.ddx);
      if (w0 && S.m > 0) {
        const int m = S.m;
        for (int j = lane; j < m; j += 64) q.cvec[j] = q.g[q.alist[j]] + qpw_row_dot(D, ld_row(q.alist[j]), q.xv, q.ddx);
        double r0, r1;
        (void)qpt_solve(q.R, m, lane, q.cvec, q.wv, q.rv, r0, r1);
        double d0, d1;
        qpw_ymul(q.Y, n, m, lane, q.rv, q.yslot, d0, d1);
        S.x0 -= d0; S.x1 -= d1;
        if (lane < m) q.lam[lane] += r0;
        if (lane + 64 < m) q.lam[lane + 64] += r1;
        if (okA) q.xv[lane] = S.x0;
        if (okB) q.xv[lane + 64] = S.x1;
      }
      __syncthreads();
    }
    PROF_END(PH_Q_REFINE, pq6);
    if (q.scal[2] == 0.0) break;   // well-conditioned QP: the polish moves x by rounding errors only, nothing to re-verify
  }
  if (TID == 0) q.scal[1] = (double)S.m;
  __syncthreads();
  const int m = (int)q.scal[1];
  // The exact minimiser sits ON its active input bounds; the polished point holds them to rounding distance (to ~1e-9 in
  // the reg = 0 regime).  These rows are linear in u, the next linearisation inherits exactly that residual and _get_mu
  // switches on its sign (DGSQP.py:566-585).  par.snap_active_bounds (default 0 = literal) puts du on the active bounds.
  if (ret == 0 && D.par.snap_active_bounds) {
    for (int j = TID; j < m; j += NT) {
      const int r = q.alist[j];
      const DgRow Rw = ld_row(r);
      if (Rw.type == DG_R_IN_UB) q.xv[am_col(D, Rw.a, Rw.k, Rw.idx)] = -q.g[r];
      else if (Rw.type == DG_R_IN_LB) q.xv[am_col(D, Rw.a, Rw.k, Rw.idx)] = q.g[r];
    }
    __syncthreads();
  }
  PROF_COUNT(PH_C_MFINAL, m);
  if (ret == 0)
    for (int j = TID; j < m; j += NT) { lhat[q.alist[j]] = q.lam[j]; q.prev[j] = q.alist[j]; q.prevlam[j] = q.lam[j]; }
  __syncthreads();
  if (TID == 0) q.scal[DG_QP_NPREV] = ret == 0 ? (double)m : 0.0;
  __syncthreads();
  PROF_END(PH_QP, pt_qp);
  return ret;
}
