// QP sub-problem of one SQP iteration with OSQP's own arithmetic (dgsqp_params_t.qp_method = DGSQP_QP_OSQP).
//
// The reference solves  min 1/2 x'Mx + q'x  s.t.  G x <= -g  with  ca.conic('qp', 'osqp', ..., {polish: True})
// (DGSQP/solvers/DGSQP.py:183-201) called as  solver(h=Q, g=q, a=G, uba=-g, x0=0)  (DGSQP.py:246-249).  OSQP is a third-party
// dependency of the reference (setup.py:15, unpinned; absent from /root/reference): this file restates its published algorithm --
// Stellato, Banjac, Goulart, Bemporad, Boyd, "OSQP: an operator splitting solver for quadratic programs", Math. Prog. Comp. 12
// (2020): Algorithm 1 (ADMM), 3.4 (termination, infeasibility certificates), 4 (polish), 5.1 (Ruiz equilibration), 5.2 (rho) --
// with the OSQP 0.6 defaults (rho 0.1, sigma 1e-6, alpha 1.6, eps_abs = eps_rel 1e-3, eps_prim_inf = eps_dual_inf 1e-4, max_iter 4000,
// scaling 10, adaptive rho with tolerance 5, check_termination 25, polish delta 1e-6 with 3 refinement steps), exactly as
// oracle/osqp_restate.py (numpy) and oracle/osqp.hpp (C++) do; the parity tests compare this kernel with those.  The conic plugin poses
// the problem with identity rows for the (absent) variable bounds ABOVE the G rows:  l <= [I; G] x <= u,  l = -inf, u = [inf; -g].
// Stated deviations (shared with the two CPU restatements): adaptive-rho interval fixed at 25 iterations (OSQP derives it from the
// wall-clock time of its first factorisation), every call starts from rho = 0.1 (inside CasADi's plugin the adapted rho persists).
//
// One workgroup per QP; what differs from the literal algorithm is algebra that is exact in exact arithmetic
// (tools/osqp_reduced_proto.py checks this formulation against the literal one on QPs harvested from SQP runs):
//   * nothing is ever scaled in place.  Ruiz equilibration carries D (n), E_I (identity rows), E (G rows) and c; column / row norms
//     of  c D M D  and  E G D  are taken on the fly through the packed constraint gradients (dense gradients shared by ub / lb rows);
//   * the quasi-definite ADMM system  [Ps + sigma I, As'; As, -diag(1/rho)] (xt, nu) = (sigma x - qs, z - y / rho)  is solved in its
//     reduced form  K xt = sigma x - qs + As' (rho z - y),  zt = As xt,  K = Ps + sigma I + rho_I (E_I D)^2 + rho W,  W = Gs' Gs, through the
//     EXPLICIT inverse of the n x n matrix K (register-resident Gauss-Jordan sweep, rebuilt when rho changes; W is formed once): an
//     ADMM iteration is then two structured products with G and one n x n product -- ~10 barriers, no dependent chains;
//   * the identity rows never clip (|x| << 1e30) and carry y = 0, z = E_I D x: they enter K's diagonal (rho_I = 1e-6: "loose" rows), the
//     right-hand side and the norms of the stopping tests, and are not stored;
//   * the polish runs in UNSCALED variables:  [c M, A'; A, 0] (x, nu) = (-c q, b)  with the regularised matrix
//     [c M + delta D^-2, A'; A, -delta E^-2] -- OSQP's scaled system after the change of variables x = D xs, nu = E nus -- in range-space
//     form: Hu^-1 explicit (same sweep), Schur complement S = A Hu^-1 A' + delta E^-2 held through the inverse Cholesky factor T that
//     the dual active-set QP (dgsqp_qp.h) borders row by row, three refinement steps against the unregularised residual.
#pragma once

#define DG_OSQP_INFO 32   // scal slots 32..39: status, iterations, polished, rho, rho updates, active rows of the polish, primal / dual residual of the ADMM iterate
#define OSQP_INFTY 1e30
#define OSQP_MIN_SCALING 1e-4
#define OSQP_MAX_SCALING 1e4
#define OSQP_RHO_MIN 1e-6
#define OSQP_RHO_MAX 1e6
enum { OSQP_SOLVED = 1, OSQP_SOLVED_INACCURATE = 2, OSQP_MAX_ITER = -2, OSQP_PRIMAL_INFEASIBLE = -3, OSQP_DUAL_INFEASIBLE = -4, OSQP_NAN_DATA = -10 };

__device__ inline double osqp_limit(double v) { v = v < OSQP_MIN_SCALING ? 1.0 : v; return fmin(v, OSQP_MAX_SCALING); }

struct OsqpPtrs {
  lptr x, Dv, EI, rhs, xt, tmp, dx;   // n-vectors: scaled iterate, column scaling, scaling of the identity rows, three work vectors, x - x_prev
  lptr z, y, E, dy, w;                // n_c-vectors (G rows): ADMM's z and y, row scaling, delta y, work vector
  lptr part, ddx, dpart, yd, red, scal;
  clptr gd, q, g;
  cgptr M;                            // projected + regularised Hessian, row-major n x n (written by dev_psd_inverse)
  gptr W;                             // Gs' Gs, row-major n x n (the slot of the active-set QP's Y)
};
__device__ inline OsqpPtrs osqp_ptrs(const Ctx& c) {
  const DgProb& D = dg_prob;
  const DgLds& L = D.L;
  lptr lds = LP(0);
  OsqpPtrs o;
  o.x = lds + L.p_lam; o.Dv = lds + L.p_c; o.EI = lds + L.p_w; o.rhs = lds + L.p_r; o.xt = lds + L.p_y; o.tmp = lds + L.p_t; o.dx = lds + L.p_rd;
  o.z = lds + L.a_z; o.y = lds + L.a_y; o.E = lds + L.a_E; o.dy = lds + L.a_dy; o.w = lds + L.a_w;
  o.part = lds + L.p_part; o.ddx = lds + L.p_yd2; o.dpart = lds + L.p_dpart; o.yd = lds + L.yd; o.red = lds + L.red; o.scal = lds + L.scal;
  o.gd = lds + L.gd; o.q = lds + L.q; o.g = lds + L.g;
  o.M = c.ws + D.ws_xM; o.W = c.ws + D.ws_Y;
  return o;
}

// eight maxima over the workgroup at once (two barriers); red: 64 doubles
__device__ inline void block_max8(double (&v)[8], lptr red) {
#pragma unroll
  for (int k = 0; k < 8; k++) v[k] = wave_max(v[k]);
  __syncthreads();
  if ((TID & 63) == 0) {
#pragma unroll
    for (int k = 0; k < 8; k++) red[(TID >> 6) * 8 + k] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 8; k++) {
    double t = red[k];
#pragma unroll
    for (int w = 1; w < NT / 64; w++) t = fmax(t, red[w * 8 + k]);
    v[k] = t;
  }
}

// One pass over the symmetric matrix M (row-major in the L2-resident scratch): out_i = sum_j M_ij v_j, or with ABSMAX
// out_i = max_j |M_ij| v_j (v >= 0).  Thread = row i x quarter of the columns; M_ij is read as M[j][i] (coalesced over i).
template <bool ABSMAX>
__device__ inline void osqp_m_pass(cgptr M, int n, clptr v, lptr part, lptr out) {
  constexpr int NSEG = NT / 128;
  const int i = TID & 127, sg = TID >> 7;
  __syncthreads();
  if (i < n) {
    const int len = (n + NSEG - 1) / NSEG;
    const int j0 = sg * len, j1 = (j0 + len < n) ? j0 + len : n;
    double a[4] = {0, 0, 0, 0};
    int j = j0;
    for (; j + 3 < j1; j += 4) {
      double m[4], t[4];
#pragma unroll
      for (int k = 0; k < 4; k++) { m[k] = M[(int64_t)(j + k) * n + i]; t[k] = v[j + k]; }
#pragma unroll
      for (int k = 0; k < 4; k++) a[k] = ABSMAX ? fmax(a[k], __builtin_fabs(m[k]) * t[k]) : __builtin_fma(m[k], t[k], a[k]);
    }
    for (; j < j1; j++) { const double m = M[(int64_t)j * n + i]; a[0] = ABSMAX ? fmax(a[0], __builtin_fabs(m) * v[j]) : __builtin_fma(m, v[j], a[0]); }
    part[sg * n + i] = ABSMAX ? fmax(fmax(a[0], a[1]), fmax(a[2], a[3])) : (a[0] + a[1]) + (a[2] + a[3]);
  }
  __syncthreads();
  if (TID < n) {
    double s = part[TID];
#pragma unroll
    for (int g = 1; g < NSEG; g++) s = ABSMAX ? fmax(s, part[g * n + TID]) : s + part[g * n + TID];
    out[TID] = s;
  }
  __syncthreads();
}

// out[d] = max over the entries of dense gradient d of |gd_p| Dv[column(p)]  (same chunk tasks as qp_dense_dots)
__device__ inline void osqp_dense_absmax(const DgProb& D, clptr gd, clptr Dv, lptr part, lptr out) {
  __syncthreads();
  for (int t = TID; t < D.ntask; t += NT) {
    const DgTask T = ld_task(t);
    clptr p = gd + T.p0;
    clptr w = Dv + T.v0;
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < DG_CHUNK; i++) { const double pv = p[i], wv = w[i]; s = i < T.len ? fmax(s, __builtin_fabs(pv) * wv) : s; }
    part[t] = s;
  }
  __syncthreads();
  for (int d = TID; d < D.ndense; d += NT) {
    const DgDense dd = ld_dense(d);
    const int ts = dd.t0lo + 256 * dd.t0hi;
    double s = 0;
    for (int i = 0; i < dd.nt; i++) s = fmax(s, part[ts + i]);
    out[d] = s;
  }
  __syncthreads();
}
// out[col] = max_r E_r |G_r,col|  (structure of gt_mul_t with max for the sum)
__device__ inline void osqp_gt_absmax(const DgProb& D, clptr gd, clptr E, lptr yd, lptr out) {
  __syncthreads();
  for (int d = TID; d < D.ndense; d += NT) {
    const DgDense dd = ld_dense(d);
    yd[d] = fmax(dd.r_pos >= 0 ? E[dd.r_pos] : 0.0, dd.r_neg >= 0 ? E[dd.r_neg] : 0.0);
  }
  __syncthreads();
  for (int it = TID; it < 4 * D.n; it += NT) {
    const int col = it >> 2, part = it & 3;
    const int a = col / (D.N * DGSQP_NUA), rem = col % (D.N * DGSQP_NUA), t = rem / DGSQP_NUA, j = rem % DGSQP_NUA;
    double s = 0;
    if (part == 0) {
      int r;
      if ((r = D.r_in_ub[a][t][j]) >= 0) s = fmax(s, E[r]);
      if ((r = D.r_in_lb[a][t][j]) >= 0) s = fmax(s, E[r]);
      if ((r = D.r_rate_ub[a][t][j]) >= 0) s = fmax(s, E[r]);
      if ((r = D.r_rate_lb[a][t][j]) >= 0) s = fmax(s, E[r]);
      if (t + 1 < D.N) {
        if ((r = D.r_rate_ub[a][t + 1][j]) >= 0) s = fmax(s, E[r]);
        if ((r = D.r_rate_lb[a][t + 1][j]) >= 0) s = fmax(s, E[r]);
      }
    }
    for (int d = D.stage_dense0[t + 1] + part; d < D.ndense; d += 4) {
      const DgDense dd = ld_dense(d);
      if (dd.a == a) s = fmax(s, yd[d] * __builtin_fabs(gd[dd.off + t * DGSQP_NUA + j]));
      else if (dd.kind == 1 && dd.b == a) s = fmax(s, yd[d] * __builtin_fabs(gd[dd.off + 2 * dd.k + t * DGSQP_NUA + j]));
    }
    s = fmax(s, dpp_f64<0xB1>(s));
    s = fmax(s, dpp_f64<0x4E>(s));
    if (part == 0) out[col] = s;
  }
  __syncthreads();
}
// entry of dense gradient dd at column (a, t, j), 0 outside its support
__device__ inline double osqp_dense_entry(const DgDense dd, clptr gd, int a, int t, int j) {
  if (t >= dd.k) return 0.0;
  if (dd.a == a) return gd[dd.off + t * DGSQP_NUA + j];
  if (dd.kind == 1 && dd.b == a) return gd[dd.off + 2 * dd.k + t * DGSQP_NUA + j];
  return 0.0;
}
// W = Gs' Gs = D G' E^2 G D, row-major n x n into the scratch: dense gradients contribute outer products on their supports (weight:
// the E^2 of the rows sharing them), box rows the diagonal, rate rows the diagonal and the (t, t - 1) entries
__device__ inline void osqp_build_w(const DgProb& D, const OsqpPtrs& o) {
  const int n = D.n;
  __syncthreads();
  for (int d = TID; d < D.ndense; d += NT) {
    const DgDense dd = ld_dense(d);
    const double ep = dd.r_pos >= 0 ? o.E[dd.r_pos] : 0.0, en = dd.r_neg >= 0 ? o.E[dd.r_neg] : 0.0;
    o.yd[d] = ep * ep + en * en;
  }
  __syncthreads();
  for (int e = TID; e < n * n; e += NT) {
    const int i = e / n, j = e - i * n;
    if (j > i) continue;
    const int ai = i / (D.N * DGSQP_NUA), ri = i % (D.N * DGSQP_NUA), ti = ri / DGSQP_NUA, ji = ri % DGSQP_NUA;
    const int aj = j / (D.N * DGSQP_NUA), rj = j % (D.N * DGSQP_NUA), tj = rj / DGSQP_NUA, jj = rj % DGSQP_NUA;
    double s = 0;
    for (int d = D.stage_dense0[(ti > tj ? ti : tj) + 1]; d < D.ndense; d++) {
      const DgDense dd = ld_dense(d);
      const double gi = osqp_dense_entry(dd, o.gd, ai, ti, ji);
      if (gi == 0.0) continue;
      s = __builtin_fma(o.yd[d] * gi, osqp_dense_entry(dd, o.gd, aj, tj, jj), s);
    }
    if (ai == aj && ji == jj) {
      auto e2 = [&](int r) { const double ev = r >= 0 ? o.E[r] : 0.0; return ev * ev; };
      if (ti == tj) {
        s += e2(D.r_in_ub[ai][ti][ji]) + e2(D.r_in_lb[ai][ti][ji]) + e2(D.r_rate_ub[ai][ti][ji]) + e2(D.r_rate_lb[ai][ti][ji]);
        if (ti + 1 < D.N) s += e2(D.r_rate_ub[ai][ti + 1][ji]) + e2(D.r_rate_lb[ai][ti + 1][ji]);
      } else if (ti - tj == 1) {
        s -= e2(D.r_rate_ub[ai][ti][ji]) + e2(D.r_rate_lb[ai][ti][ji]);     // (rate rows of stage t: +1 at t, -1 at t - 1)
      }
    }
    s *= o.Dv[i] * o.Dv[j];
    o.W[(int64_t)i * n + j] = s;
    o.W[(int64_t)j * n + i] = s;
  }
  __threadfence_block();
  __syncthreads();
}

// Explicit inverse of the SPD matrix  sM dI M dI + sW W + diag(dg)  (dI = diag(di), di == nullptr: identity) into the packed-P slot
// (LDS, or the scratch in the big layout): register-resident Gauss-Jordan sweep (spd_sweep_regs).  Returns false when a pivot was not
// positive / finite (the matrix is not numerically SPD).  tws: 2 (NH RPT + 4) doubles of LDS.
template <int RPT>
__device__ __noinline__ bool dev_osqp_inverse_t(const Ctx& c, cgptr M, cgptr W, clptr di, clptr dg, double sM, double sW, lptr tws) {
  const DgProb& D = dg_prob;
  const int n = D.n;
  constexpr int NH = DG_NH;
  const int jc = TID & 127, hf = TID >> 7;
  const bool colok = jc < n;
  __syncthreads();
  double Br[RPT];
#pragma unroll
  for (int r = 0; r < RPT; r++) {
    const int i = hf + NH * r;
    double a = 0.0;
    if (colok && i < n) {
      a = sM * (di ? di[i] * di[jc] : 1.0) * M[(int64_t)i * n + jc];
      if (sW != 0.0) a = __builtin_fma(sW, W[(int64_t)i * n + jc], a);
      if (i == jc) a += dg[i];
    }
    Br[r] = a;
  }
  __syncthreads();
  spd_sweep_regs<RPT>(Br, tws, n);
  int bad = 0;
  if (D.big) {
    gptr Pp = c.ws + D.ws_P;
#pragma unroll
    for (int r = 0; r < RPT; r++) {
      const int i = hf + NH * r;
      if (colok && i < n && i >= jc) Pp[i * (i + 1) / 2 + jc] = -Br[r];
      if (colok && i == jc && !(-Br[r] > 0.0 && -Br[r] < 1e300)) bad = 1;
    }
    __threadfence_block();
  } else {
    lptr Pp = LP(D.L.g_Bp);
#pragma unroll
    for (int r = 0; r < RPT; r++) {
      const int i = hf + NH * r;
      if (colok && i < n && i >= jc) Pp[i * (i + 1) / 2 + jc] = -Br[r];
      if (colok && i == jc && !(-Br[r] > 0.0 && -Br[r] < 1e300)) bad = 1;
    }
  }
  return !__syncthreads_or(bad);
}
__device__ inline bool dev_osqp_inverse(const Ctx& c, const OsqpPtrs& o, clptr di, clptr dg, double sM, double sW) {
  const int n = dg_prob.n;
  lptr tws = LP(dg_prob.L.a_sw);
  if (n <= 32) return dev_osqp_inverse_t<32 / DG_NH>(c, o.M, o.W, di, dg, sM, sW, tws);
  if (n <= 64) return dev_osqp_inverse_t<64 / DG_NH>(c, o.M, o.W, di, dg, sM, sW, tws);
  if (n <= 100) return dev_osqp_inverse_t<100 / DG_NH>(c, o.M, o.W, di, dg, sM, sW, tws);
  return dev_osqp_inverse_t<128 / DG_NH>(c, o.M, o.W, di, dg, sM, sW, tws);
}

// out_r = E_r (G (D v))_r for every G row.  Leaves D v in o.tmp and its dense dots in o.ddx.
__device__ inline void osqp_gs_mul(const OsqpPtrs& o, clptr v, lptr out) {
  const DgProb& D = dg_prob;
  __syncthreads();
  for (int j = TID; j < D.n; j += NT) o.tmp[j] = o.Dv[j] * v[j];
  __syncthreads();
  qp_dense_dots(D, o.gd, o.tmp, o.dpart, o.ddx);
  for (int r = TID; r < D.nc; r += NT) out[r] = o.E[r] * qpw_row_dot(D, ld_row(r), o.tmp, o.ddx);
  __syncthreads();
}
// out = D G' (E w);  sc: an n_c-vector of scratch (may be w itself)
__device__ inline void osqp_gst_mul(const Ctx& c, const OsqpPtrs& o, clptr w, lptr sc, lptr out) {
  const DgProb& D = dg_prob;
  __syncthreads();
  for (int r = TID; r < D.nc; r += NT) sc[r] = o.E[r] * w[r];
  gt_mul(c, sc, out);
  for (int j = TID; j < D.n; j += NT) out[j] *= o.Dv[j];
  __syncthreads();
}
// out = sum_j nu_j a_{alist[j]} over the m rows of the polish (no n_c-vector needed): four lanes per column
__device__ inline void osqp_at_active(const DgProb& D, clptr gd, const lds_i_t* alist, clptr nu, int m, lptr out) {
  __syncthreads();
  for (int it = TID; it < 4 * D.n; it += NT) {
    const int col = it >> 2, part = it & 3;
    double s = 0;
    for (int j = part; j < m; j += 4) s = __builtin_fma(nu[j], g_row_coef(D, gd, alist[j], col), s);
    s += dpp_f64<0xB1>(s);
    s += dpp_f64<0x4E>(s);
    if (part == 0) out[col] = s;
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// _solve_qp core with OSQP's arithmetic.  In: M (scratch, ws_xM), q, g, packed G.  Out: du (L.o_du), lhat (L.o_lhat).
// Returns 0 when OSQP hands back a point (solved, solved inaccurate, or the iteration limit: the reference continues from whatever
// OSQP returns), 1 when it reports primal / dual infeasibility or non-finite data, or when the point is not finite (a NaN step:
// DGSQP.py:566-585 raises).
// ------------------------------------------------------------------------------------------------
__device__ __noinline__ int dev_qp_osqp(const Ctx& c) {
  const DgProb& D = dg_prob;
  const DgLds& L = D.L;
  lptr lds = LP(0);
  const int n = D.n, nc = D.nc;
  const OsqpPtrs o = osqp_ptrs(c);
  lptr du = lds + L.o_du, lhat = lds + L.o_lhat;
  const int lane = TID & 63;
  const bool w0 = TID < 64;
  const double sigma = 1e-6, alpha = 1.6, eps_abs = 1e-3, eps_rel = 1e-3, eps_inf = 1e-4, delta = 1e-6;
  const int max_iter = 4000, check_every = 25;
  __syncthreads();
  PROF_BEGIN(pt_qp);
  if (TID == 0) { o.scal[DG_QP_NPREV] = 0.0; o.scal[DG_XVALID] = 0.0; }
  // ---- data must be finite (the conic plugin returns NaN otherwise)
  {
    int bad = 0;
    for (int e = TID; e < n * n; e += NT) bad |= !(__builtin_fabs(o.M[e]) < INFINITY);
    for (int j = TID; j < n; j += NT) bad |= !(__builtin_fabs(o.q[j]) < INFINITY);
    for (int r = TID; r < nc; r += NT) bad |= (o.g[r] != o.g[r]);
    for (int p = TID; p < D.ngd; p += NT) bad |= !(__builtin_fabs(o.gd[p]) < INFINITY);
    if (__syncthreads_or(bad)) {
      if (TID == 0) { o.scal[DG_OSQP_INFO] = OSQP_NAN_DATA; o.scal[DG_OSQP_INFO + 1] = 0; o.scal[DG_OSQP_INFO + 2] = 0; }
      __syncthreads();
      return 1;
    }
  }
  // ---- Ruiz equilibration (section 5.1; OSQP scale_data()): 10 passes
  PROF_BEGIN(po1);
  for (int j = TID; j < n; j += NT) { o.Dv[j] = 1.0; o.EI[j] = 1.0; }
  for (int r = TID; r < nc; r += NT) o.E[r] = 1.0;
  double cc = 1.0;
  for (int it = 0; it < 10; it++) {
    osqp_m_pass<true>(o.M, n, o.Dv, o.part, o.tmp);                 // tmp_j = max_i |M_ij| D_i
    osqp_dense_absmax(D, o.gd, o.Dv, o.dpart, o.ddx);               // ddx_d = max_p |gd_p| D_col(p)
    for (int r = TID; r < nc; r += NT) {
      const DgRow R = ld_row(r);
      double rm;
      if (R.dense >= 0) rm = o.ddx[R.dense];
      else {
        const int c1 = am_col(D, R.a, R.k, R.idx);
        rm = o.Dv[c1];
        if ((R.type == DG_R_RATE_UB || R.type == DG_R_RATE_LB) && R.k > 0) rm = fmax(rm, o.Dv[c1 - DGSQP_NUA]);
      }
      o.w[r] = 1.0 / sqrt(osqp_limit(o.E[r] * rm));
    }
    osqp_gt_absmax(D, o.gd, o.E, o.yd, o.xt);                       // xt_j = max_r E_r |G_rj|
    for (int j = TID; j < n; j += NT) {
      const double dj = o.Dv[j], aI = o.EI[j] * dj;
      const double dn = fmax(cc * dj * o.tmp[j], fmax(aI, dj * o.xt[j]));
      o.Dv[j] = dj * (1.0 / sqrt(osqp_limit(dn)));
      o.EI[j] *= 1.0 / sqrt(osqp_limit(aI));
    }
    for (int r = TID; r < nc; r += NT) o.E[r] *= o.w[r];
    osqp_m_pass<true>(o.M, n, o.Dv, o.part, o.tmp);                 // with the new D (barriers inside)
    double cm = 0, qn = 0;
    for (int j = TID; j < n; j += NT) { cm += cc * o.Dv[j] * o.tmp[j]; qn = fmax(qn, __builtin_fabs(cc * o.Dv[j] * o.q[j])); }
    cm = block_sum(cm, o.red);
    qn = block_max(qn, o.red);
    const double ct = osqp_limit(cm / n);
    qn = qn < OSQP_MIN_SCALING ? 1.0 : fmin(qn, OSQP_MAX_SCALING);
    cc *= 1.0 / fmax(ct, qn);
  }
  const double cinv = 1.0 / cc;
  PROF_END(PH_O_SCALE, po1);
  // ---- W = Gs' Gs (once), K(rho) and its inverse
  PROF_BEGIN(po2);
  osqp_build_w(D, o);
  PROF_END(PH_O_W, po2);
  double rho = 0.1;
  int rho_updates = 0;
  auto rho_I = [&](int j, double r) { return o.EI[j] * OSQP_INFTY > OSQP_INFTY * OSQP_MIN_SCALING ? OSQP_RHO_MIN : r; };   // "loose" row: both bounds beyond 1e26 after scaling
  auto build_kinv = [&](double r) -> bool {
    __syncthreads();
    for (int j = TID; j < n; j += NT) { const double aI = o.EI[j] * o.Dv[j]; o.tmp[j] = sigma + rho_I(j, r) * aI * aI; }
    __syncthreads();
    PROF_BEGIN(po3);
    const bool okk = dev_osqp_inverse(c, o, o.Dv, o.tmp, cc, r);
    PROF_END(PH_O_KINV, po3);
    return okk;
  };
  bool spd = build_kinv(rho);
  for (int j = TID; j < n; j += NT) { o.x[j] = 0.0; o.dx[j] = 0.0; }
  for (int r = TID; r < nc; r += NT) { o.z[r] = 0.0; o.y[r] = 0.0; o.dy[r] = 0.0; }
  __syncthreads();
  // The G rows have  l = -inf -> -1e30 E_r,  u = E_r min(-g_r, 1e30):  never equalities (rho_vec = rho on all of them), never "loose"
  // unless -g_r >= 1e26 / E_r (then OSQP gives the row rho_min; not reproduced: no game produces such a row)
  int status = OSQP_MAX_ITER, iters = 0;
  double pri_res = INFINITY, dua_res = INFINITY;
  bool stopped = !spd;
  if (!spd) status = OSQP_NAN_DATA;
  auto residual_vectors = [&]() {       // Ax (G rows) -> w, Px -> rhs, A'y -> xt
    osqp_gs_mul(o, o.x, o.w);
    osqp_m_pass<false>(o.M, n, o.tmp, o.part, o.rhs);               // (o.tmp = D x after osqp_gs_mul)
    for (int j = TID; j < n; j += NT) o.rhs[j] *= cc * o.Dv[j];
    __syncthreads();
    osqp_gst_mul(c, o, o.y, o.dy, o.xt);                            // (dy is free here: its last use was the infeasibility test)
  };
  double eps_p = 0, eps_d = 0, ad_pr = 0, ad_dr = 0;
  auto residuals = [&]() {
    double v[8] = {0, 0, 0, 0, 0, 0, 0, 0}, u[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int r = TID; r < nc; r += NT) {
      const double ei = 1.0 / o.E[r], ax = o.w[r], zz = o.z[r];
      v[0] = fmax(v[0], __builtin_fabs(ei * (ax - zz))); v[1] = fmax(v[1], __builtin_fabs(ei * zz)); v[2] = fmax(v[2], __builtin_fabs(ei * ax));
      v[3] = fmax(v[3], __builtin_fabs(ax - zz)); v[4] = fmax(v[4], __builtin_fabs(zz)); v[5] = fmax(v[5], __builtin_fabs(ax));
    }
    for (int j = TID; j < n; j += NT) {
      const double di = 1.0 / o.Dv[j], px = o.rhs[j], aty = o.xt[j], qs = cc * o.Dv[j] * o.q[j];
      v[6] = fmax(v[6], __builtin_fabs(o.Dv[j] * o.x[j]));                    // identity rows: |z / E| = |Ax / E| = |D x|
      v[7] = fmax(v[7], __builtin_fabs(o.EI[j] * o.Dv[j] * o.x[j]));          // ... and |z| = |Ax| scaled
      u[0] = fmax(u[0], __builtin_fabs(di * (px + qs + aty))); u[1] = fmax(u[1], __builtin_fabs(di * qs)); u[2] = fmax(u[2], __builtin_fabs(di * aty));
      u[3] = fmax(u[3], __builtin_fabs(di * px)); u[4] = fmax(u[4], __builtin_fabs(px + qs + aty)); u[5] = fmax(u[5], __builtin_fabs(qs));
      u[6] = fmax(u[6], __builtin_fabs(aty)); u[7] = fmax(u[7], __builtin_fabs(px));
    }
    block_max8(v, o.red);
    block_max8(u, o.red);
    pri_res = v[0];
    dua_res = cinv * u[0];
    eps_p = eps_abs + eps_rel * fmax(fmax(v[1], v[6]), fmax(v[2], v[6]));
    eps_d = eps_abs + eps_rel * cinv * fmax(u[1], fmax(u[2], u[3]));
    ad_pr = v[3] / (fmax(fmax(v[4], v[7]), fmax(v[5], v[7])) + 1e-10);
    ad_dr = u[4] / (fmax(u[5], fmax(u[6], u[7])) + 1e-10);
  };
  PROF_BEGIN(po4);
  for (int it = 1; !stopped && it <= max_iter; it++) {
    iters = it;
    // (1) right-hand side and the reduced solve
    for (int r = TID; r < nc; r += NT) o.w[r] = o.E[r] * (rho * o.z[r] - o.y[r]);
    gt_mul(c, o.w, o.tmp);
    for (int j = TID; j < n; j += NT) {
      const double dj = o.Dv[j], aI = o.EI[j] * dj, xj = o.x[j];
      o.rhs[j] = sigma * xj - cc * dj * o.q[j] + dj * o.tmp[j] + aI * rho_I(j, rho) * (aI * xj);
    }
    dev_p_mul(c, o.rhs, o.xt, 1.0);
    // (2) zt = As xt, relaxation, projection, dual update
    osqp_gs_mul(o, o.xt, o.w);
    for (int r = TID; r < nc; r += NT) {
      const double er = o.E[r], zp = o.z[r], yr = o.y[r];
      const double zr = alpha * o.w[r] + (1.0 - alpha) * zp;
      const double us = er * fmin(-o.g[r], OSQP_INFTY), ls = -OSQP_INFTY * er;
      const double zn = fmin(fmax(zr + yr / rho, ls), us);
      const double dyr = rho * (zr - zn);
      o.z[r] = zn; o.dy[r] = dyr; o.y[r] = yr + dyr;
    }
    for (int j = TID; j < n; j += NT) { const double xp = o.x[j], xn = alpha * o.xt[j] + (1.0 - alpha) * xp; o.x[j] = xn; o.dx[j] = xn - xp; }
    __syncthreads();
    if (it % check_every != 0) continue;
    // ---- termination (section 3.4) every 25 iterations; the same products serve the rho adaptation (section 5.2)
    PROF_BEGIN(po5);
    // primal infeasibility certificate (uses delta y, which the residual products overwrite)
    bool pinf = false;
    {
      double nrm = 0, lhs = 0;
      for (int r = TID; r < nc; r += NT) {
        const double er = o.E[r], us = er * fmin(-o.g[r], OSQP_INFTY), ls = -OSQP_INFTY * er;
        const bool inf_u = us > OSQP_INFTY * OSQP_MIN_SCALING, inf_l = ls < -OSQP_INFTY * OSQP_MIN_SCALING;
        double v = o.dy[r];
        v = (inf_u && inf_l) ? 0.0 : (inf_u ? fmin(v, 0.0) : (inf_l ? fmax(v, 0.0) : v));
        o.w[r] = v;
        nrm = fmax(nrm, __builtin_fabs(er * v));
        if (!inf_u) lhs += us * fmax(v, 0.0);
        if (!inf_l) lhs += ls * fmin(v, 0.0);
      }
      nrm = block_max(nrm, o.red);
      lhs = block_sum(lhs, o.red);
      if (nrm > 1.0 / OSQP_INFTY && lhs < -eps_inf * nrm) {
        osqp_gst_mul(c, o, o.w, o.w, o.xt);                          // As' dy; the test divides by D again
        double mx = 0;
        for (int j = TID; j < n; j += NT) mx = fmax(mx, __builtin_fabs(o.xt[j] / o.Dv[j]));
        mx = block_max(mx, o.red);
        pinf = mx < eps_inf * nrm;
      }
    }
    residual_vectors();
    residuals();
    PROF_END(PH_O_CHECK, po5);
    if (pri_res <= eps_p && dua_res <= eps_d) { status = OSQP_SOLVED; break; }
    if (pinf) { status = OSQP_PRIMAL_INFEASIBLE; break; }
    {
      // dual infeasibility certificate
      double nrm = 0, qdx = 0;
      for (int j = TID; j < n; j += NT) { nrm = fmax(nrm, __builtin_fabs(o.Dv[j] * o.dx[j])); qdx += cc * o.Dv[j] * o.q[j] * o.dx[j]; }
      nrm = block_max(nrm, o.red);
      qdx = block_sum(qdx, o.red);
      bool dinf = false;
      if (nrm > 1.0 / OSQP_INFTY && qdx < -cc * eps_inf * nrm) {
        osqp_gs_mul(o, o.dx, o.w);                                   // w = As dx; o.tmp = D dx
        osqp_m_pass<false>(o.M, n, o.tmp, o.part, o.rhs);
        double mx = 0;
        for (int j = TID; j < n; j += NT) mx = fmax(mx, __builtin_fabs(cc * o.rhs[j]));      // |Dinv (Ps dx)| = |c M D dx|
        mx = block_max(mx, o.red);
        if (mx < cc * eps_inf * nrm) {
          int viol = 0;
          for (int r = TID; r < nc; r += NT) {
            const double er = o.E[r], us = er * fmin(-o.g[r], OSQP_INFTY), ls = -OSQP_INFTY * er, adx = o.w[r] / er;
            const bool ok_u = us > OSQP_INFTY * OSQP_MIN_SCALING || adx < eps_inf * nrm;
            const bool ok_l = ls < -OSQP_INFTY * OSQP_MIN_SCALING || adx > -eps_inf * nrm;
            viol |= !(ok_u && ok_l);
          }
          for (int j = TID; j < n; j += NT) {
            const bool inf_b = o.EI[j] * OSQP_INFTY > OSQP_INFTY * OSQP_MIN_SCALING;
            const double adx = o.Dv[j] * o.dx[j];
            viol |= !((inf_b || adx < eps_inf * nrm) && (inf_b || adx > -eps_inf * nrm));
          }
          dinf = !__syncthreads_or(viol);
        }
      }
      if (dinf) { status = OSQP_DUAL_INFEASIBLE; break; }
    }
    // rho adaptation (interval fixed at 25)
    {
      const double rho_new = fmin(fmax(rho * sqrt(ad_pr / (ad_dr + 1e-10)), OSQP_RHO_MIN), OSQP_RHO_MAX);
      if (rho_new > rho * 5.0 || rho_new < rho / 5.0) {
        rho = rho_new;
        rho_updates++;
        if (!build_kinv(rho)) { status = OSQP_NAN_DATA; break; }
      }
    }
  }
  PROF_END(PH_O_ADMM, po4);
  PROF_COUNT(PH_O_ITERS, iters);
  if (status == OSQP_MAX_ITER) {      // iteration limit: OSQP re-checks with 10x the tolerances ("solved inaccurate")
    residual_vectors();
    residuals();
    if (pri_res <= 10.0 * eps_p && dua_res <= 10.0 * eps_d) status = OSQP_SOLVED_INACCURATE;
  }
  __syncthreads();
  // ---- the ADMM iterate, unscaled, is the answer unless the polish improves on it
  for (int j = TID; j < n; j += NT) du[j] = o.Dv[j] * o.x[j];
  for (int r = TID; r < nc; r += NT) lhat[r] = cinv * o.E[r] * o.y[r];
  int polished = 0, na = 0;
  if (status == OSQP_SOLVED) {
    // ---- polish (section 4).  Active rows in row order:  upper  (u - z) < y,  lower  (z - l) < -y  (l = -1e30 E: never)
    const QpPtrs q = qp_ptrs(c);
    lptr xs = lds + L.a_tail, nu = xs + ((n + 1) & ~1), e1 = nu + ((n + 1) & ~1), e2 = e1 + ((n + 1) & ~1), regd = e2 + ((n + 1) & ~1);
    if (w0) {
      int cnt = 0;
      for (int base = 0; base < nc; base += 64) {
        const int r = base + lane;
        bool act = false;
        double er = 1.0;
        if (r < nc) {
          er = o.E[r];
          const double us = er * fmin(-o.g[r], OSQP_INFTY), ls = -OSQP_INFTY * er;
          act = ((us - o.z[r]) < o.y[r]) || ((o.z[r] - ls) < -o.y[r]);
        }
        const unsigned long long mask = __ballot(act);
        const int pos = cnt + __popcll(mask & ((1ull << lane) - 1ull));
        if (act && pos < D.osqp_namax) { q.prev[pos] = r; regd[pos] = delta / (er * er); }
        cnt += __popcll(mask);
      }
      if (lane == 0) o.scal[1] = (double)cnt;
    }
    __syncthreads();
    na = (int)o.scal[1];
    bool ok = na <= D.osqp_namax;
    if (ok) {
      // Hu^-1 = (c M + delta D^-2)^-1 into the packed-P slot (the ADMM's K^-1 is dead)
      for (int j = TID; j < n; j += NT) o.tmp[j] = delta / (o.Dv[j] * o.Dv[j]);
      __syncthreads();
      PROF_BEGIN(po6);
      ok = dev_osqp_inverse(c, o, nullptr, o.tmp, cc, 0.0);
      PROF_END(PH_O_PINV, po6);
    }
    PROF_COUNT(PH_O_NACT, na);
    PROF_BEGIN(po7);
    if (ok) {
      // inverse Cholesky factor T of S = A Hu^-1 A' + delta E^-2, one bordering step per active row (dgsqp_qp.h machinery)
      for (int r = TID; r < nc; r += NT) q.act[r] = 0;
      for (int i = TID; i < n; i += NT) q.yfree[i] = n - 1 - i;
      __syncthreads();
      QpwState S;
      S.m = 0; S.nfree = n; S.ill = 0; S.x0 = 0.0; S.x1 = 0.0;
      const bool okA = lane < n, okB = lane + 64 < n;
      for (int k = 0; k < na && ok; k++) {
        const int p = q.prev[k];
        qp_row_products(c, q, p);
        if (w0) {
          const int m = S.m;
          for (int j = lane; j < m; j += 64) q.cvec[j] = qpw_row_dot(D, ld_row(q.alist[j]), q.yv, q.ddy);
          const double t0 = okA ? q.tv[lane] : 0.0, t1 = okB ? q.tv[lane + 64] : 0.0;
          const double y0 = okA ? q.yv[lane] : 0.0, y1 = okB ? q.yv[lane + 64] : 0.0;
          const double app = wave_sum(t0 * y0 + t1 * y1);
          double ra, rb;
          const double ww = qpt_solve(q.R, m, lane, q.cvec, q.wv, q.rv, ra, rb);
          const double dlt = app - ww + regd[k];
          const bool good = dlt > 0.0 && dlt < INFINITY;
          if (good) qpw_append(q, S, n, lane, p, dlt, 0.0);
          if (lane == 0) o.scal[0] = good ? 0.0 : 1.0;
        }
        __syncthreads();
        ok = o.scal[0] == 0.0;
      }
      __threadfence_block();
      __syncthreads();
    }
    PROF_END(PH_O_PROWS, po7);
    PROF_BEGIN(po8);
    if (ok) {
      const int m = na;
      QpPtrs q2 = q;
      q2.xv = lds + L.p_t;      // (tv: only qp_row_products needs it) the solve's x part
      // (dx, dnu) = Kreg^-1 (r1, r2):  t = Hu^-1 r1,  dnu = S^-1 (A t - r2),  dx = t - Y dnu.   r1 in e1, r2 in e2; dnu -> q.lam, dx -> q2.xv
      auto kkt_solve = [&]() {
        dev_p_mul(c, e1, q.xu, 1.0);
        qp_dense_dots(D, q.gd, q.xu, q.dpart, q.ddx);
        if (w0) {
          for (int j = lane; j < m; j += 64) q.cvec[j] = qpw_row_dot(D, ld_row(q.alist[j]), q.xu, q.ddx) - e2[j];
          double ra, rb;
          (void)qpt_solve(q.R, m, lane, q.cvec, q.wv, q.rv, ra, rb);
          if (lane < m) q.lam[lane] = ra;
          if (lane + 64 < m) q.lam[lane + 64] = rb;
        }
        __syncthreads();
        qp_x_from_lambda(q2, m);
      };
      for (int j = TID; j < n; j += NT) e1[j] = -cc * o.q[j];
      for (int k = TID; k < m; k += NT) e2[k] = fmin(-o.g[q.alist[k]], OSQP_INFTY);
      __syncthreads();
      kkt_solve();
      for (int j = TID; j < n; j += NT) xs[j] = q2.xv[j];
      for (int k = TID; k < m; k += NT) nu[k] = q.lam[k];
      __syncthreads();
      for (int rf = 0; rf < 3; rf++) {
        // residual of the UNREGULARISED system:  e1 = -c q - (c M xs + A' nu),  e2 = b - A xs
        osqp_m_pass<false>(o.M, n, xs, o.part, o.rhs);
        osqp_at_active(D, o.gd, q.alist, nu, m, o.xt);
        qp_dense_dots(D, q.gd, xs, q.dpart, q.ddx);
        for (int j = TID; j < n; j += NT) e1[j] = -cc * o.q[j] - (cc * o.rhs[j] + o.xt[j]);
        for (int k = TID; k < m; k += NT) e2[k] = fmin(-o.g[q.alist[k]], OSQP_INFTY) - qpw_row_dot(D, ld_row(q.alist[k]), xs, q.ddx);
        __syncthreads();
        kkt_solve();
        for (int j = TID; j < n; j += NT) xs[j] += q2.xv[j];
        for (int k = TID; k < m; k += NT) nu[k] += q.lam[k];
        __syncthreads();
      }
      // acceptance on the residuals alone (multiplier signs are not looked at)
      osqp_m_pass<false>(o.M, n, xs, o.part, o.rhs);
      osqp_at_active(D, o.gd, q.alist, nu, m, o.xt);
      qp_dense_dots(D, q.gd, xs, q.dpart, q.ddx);
      double pr_p = 0, dr_p = 0;
      int nonfin = 0;
      for (int r = TID; r < nc; r += NT) pr_p = fmax(pr_p, fmax(0.0, qpw_row_dot(D, ld_row(r), xs, q.ddx) - fmin(-o.g[r], OSQP_INFTY)));
      for (int j = TID; j < n; j += NT) { dr_p = fmax(dr_p, __builtin_fabs(cc * o.rhs[j] + cc * o.q[j] + o.xt[j])); nonfin |= !(__builtin_fabs(xs[j]) < INFINITY); }
      for (int k = TID; k < m; k += NT) nonfin |= !(__builtin_fabs(nu[k]) < INFINITY);
      pr_p = block_max(pr_p, o.red);
      dr_p = cinv * block_max(dr_p, o.red);
      nonfin = __syncthreads_or(nonfin);
      const bool better = (pr_p < pri_res && dr_p < dua_res) || (pr_p < pri_res && dua_res < 1e-10) || (dr_p < dua_res && pri_res < 1e-10);
      if (better && !nonfin) {
        for (int j = TID; j < n; j += NT) du[j] = xs[j];
        for (int r = TID; r < nc; r += NT) lhat[r] = 0.0;
        __syncthreads();
        for (int k = TID; k < m; k += NT) lhat[q.alist[k]] = cinv * nu[k];
        polished = 1;
      } else polished = -1;
    } else polished = -1;
    PROF_END(PH_O_PSOLVE, po8);
  }
  __syncthreads();
  if (TID == 0) {
    o.scal[DG_OSQP_INFO] = (double)status; o.scal[DG_OSQP_INFO + 1] = (double)iters; o.scal[DG_OSQP_INFO + 2] = (double)polished; o.scal[DG_OSQP_INFO + 3] = rho;
    o.scal[DG_OSQP_INFO + 4] = (double)rho_updates; o.scal[DG_OSQP_INFO + 5] = (double)na; o.scal[DG_OSQP_INFO + 6] = pri_res; o.scal[DG_OSQP_INFO + 7] = dua_res;
  }
  // a non-finite answer (an ADMM run that overflowed before its iteration limit) is a NaN step as well
  int nonfinite = 0;
  for (int j = TID; j < n; j += NT) nonfinite |= !(__builtin_fabs(du[j]) < INFINITY);
  for (int r = TID; r < nc; r += NT) nonfinite |= !(__builtin_fabs(lhat[r]) < INFINITY);
  nonfinite = __syncthreads_or(nonfinite);
  PROF_END(PH_QP, pt_qp);
  return (nonfinite || status == OSQP_PRIMAL_INFEASIBLE || status == OSQP_DUAL_INFEASIBLE || status == OSQP_NAN_DATA) ? 1 : 0;
}
