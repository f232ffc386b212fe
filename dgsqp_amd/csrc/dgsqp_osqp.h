// QP sub-problem of one SQP iteration with OSQP's own arithmetic (dgsqp_params_t.qp_method = DGSQP_QP_OSQP).
//
// The reference solves  min 1/2 x'Mx + q'x  s.t.  G x <= -g  with  ca.conic('qp', 'osqp', ..., {polish: True})
// (DGSQP/solvers/DGSQP.py:183-201) called as  solver(h=Q, g=q, a=G, uba=-g, x0=0)  (DGSQP.py:246-249).  OSQP is a third-party
// dependency of the reference (setup.py:15, unpinned; absent from /root/reference): this file restates its published algorithm --
// Stellato, Banjac, Goulart, Bemporad, Boyd, "OSQP: an operator splitting solver for quadratic programs", Math. Prog. Comp. 12
// (2020): Algorithm 1 (ADMM), 3.4 (termination, infeasibility certificates), 4 (polish), 5.1 (Ruiz equilibration), 5.2 (rho) --
// with the OSQP 0.6 defaults (rho 0.1, sigma 1e-6, alpha 1.6, eps_abs = eps_rel 1e-3, eps_prim_inf = eps_dual_inf 1e-4, max_iter 4000,
// scaling 10, adaptive rho with tolerance 5, check_termination 25, polish delta 1e-6 with 3 refinement steps), exactly as
// the test infrastructure's two CPU restatements (numpy: osqp_restate, C++: osqp.hpp) do; the parity tests compare this kernel with those.  The conic plugin poses
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
enum { OSQP_SOLVED = 1, OSQP_SOLVED_INACCURATE = 2, OSQP_PRIMAL_INFEASIBLE_INACCURATE = 3, OSQP_DUAL_INFEASIBLE_INACCURATE = 4, OSQP_MAX_ITER = -2, OSQP_PRIMAL_INFEASIBLE = -3, OSQP_DUAL_INFEASIBLE = -4, OSQP_NAN_DATA = -10 };

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
  // one thread per pair of the lower triangle, enumerated in STAGE-MAJOR column order (column c' = t M nu + a nu + j): for i' >= j' the
  // later stage is i''s, so the gradients a pair sums over -- those of the stages after max(t_i, t_j) -- depend on i' alone and the lanes of
  // a wavefront (consecutive pairs: mostly one i') loop over the same range.  (Round 5 walked all n^2 entries and skipped the upper half:
  // half the lanes idle, loop lengths mixed within a wavefront -- 0.52 Mcycles per QP at n = 100.)  Same sums in the same order per pair.
  const int NU = DGSQP_NUA, MN = D.M * NU;
  for (int p = TID; p < n * (n + 1) / 2; p += NT) {
    int ip = (int)((sqrt(8.0 * p + 1.0) - 1.0) * 0.5);
    while ((ip + 1) * (ip + 2) / 2 <= p) ip++;
    while (ip * (ip + 1) / 2 > p) ip--;
    const int jp = p - ip * (ip + 1) / 2;                     // stage-major indices, jp <= ip
    const int ti = ip / MN, ai = (ip - ti * MN) / NU, ji = ip % NU;
    const int tj = jp / MN, aj = (jp - tj * MN) / NU, jj = jp % NU;
    int i = ai * D.N * NU + ti * NU + ji, j = aj * D.N * NU + tj * NU + jj;      // agent-major columns (the decision vector's order)
    if (j > i) { const int t_ = i; i = j; j = t_; }           // (the pair is stored at both places; the summation below is symmetric in its two entries
    const int Ai = i / (D.N * NU), Ti = (i % (D.N * NU)) / NU, Ji = i % NU;        //  only up to the order of the two factors of a product: keep round 5's (i >= j))
    const int Aj = j / (D.N * NU), Tj = (j % (D.N * NU)) / NU, Jj = j % NU;
    double s = 0;
    for (int d = D.stage_dense0[(Ti > Tj ? Ti : Tj) + 1]; d < D.ndense; d++) {
      const DgDense dd = ld_dense(d);
      const double gi = osqp_dense_entry(dd, o.gd, Ai, Ti, Ji);
      if (gi == 0.0) continue;
      s = __builtin_fma(o.yd[d] * gi, osqp_dense_entry(dd, o.gd, Aj, Tj, Jj), s);
    }
    if (Ai == Aj && Ji == Jj) {
      auto e2 = [&](int r) { const double ev = r >= 0 ? o.E[r] : 0.0; return ev * ev; };
      if (Ti == Tj) {
        s += e2(D.r_in_ub[Ai][Ti][Ji]) + e2(D.r_in_lb[Ai][Ti][Ji]) + e2(D.r_rate_ub[Ai][Ti][Ji]) + e2(D.r_rate_lb[Ai][Ti][Ji]);
        if (Ti + 1 < D.N) s += e2(D.r_rate_ub[Ai][Ti + 1][Ji]) + e2(D.r_rate_lb[Ai][Ti + 1][Ji]);
      } else if (Ti - Tj == 1) {
        s -= e2(D.r_rate_ub[Ai][Ti][Ji]) + e2(D.r_rate_lb[Ai][Ti][Ji]);     // (rate rows of stage t: +1 at t, -1 at t - 1)
      }
    }
    s *= o.Dv[i] * o.Dv[j];
    o.W[(int64_t)i * n + j] = s;
    o.W[(int64_t)j * n + i] = s;
  }
  __threadfence_block();
  __syncthreads();
}

// Explicit inverse of the SPD matrix  sM dI M dI + sW W + diag(dg)  (dI = diag(di), di == nullptr: identity) by the register-resident
// Gauss-Jordan sweep (spd_sweep_regs): on return Br holds MINUS the inverse, thread (jc = TID & 127, hf = TID >> 7) its column jc at
// rows hf + NH r.  With `store` the inverse also goes to the packed-P slot (LDS, or the scratch in the big layout) for dev_p_mul.
// Returns false when a pivot was not positive / finite (the matrix is not numerically SPD).  tws: 4 (NH RPT + 4) doubles of LDS.
template <int RPT>
__device__ __forceinline__ bool osqp_inverse_regs(const Ctx& c, cgptr M, cgptr W, clptr di, clptr dg, double sM, double sW, lptr tws, double (&Br)[RPT], bool store) {
  const DgProb& D = dg_prob;
  const int n = D.n;
  constexpr int NH = DG_NH;
  const int jc = TID & 127, hf = TID >> 7;
  const bool colok = jc < n;
  __syncthreads();
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
#pragma unroll
  for (int r = 0; r < RPT; r++) {
    const int i = hf + NH * r;
    if (colok && i == jc && !(-Br[r] > 0.0 && -Br[r] < 1e300)) bad = 1;
  }
  if (store) {
    if (D.big) {
      gptr Pp = c.ws + D.ws_P;
#pragma unroll
      for (int r = 0; r < RPT; r++) { const int i = hf + NH * r; if (colok && i < n && i >= jc) Pp[i * (i + 1) / 2 + jc] = -Br[r]; }
      __threadfence_block();
    } else {
      lptr Pp = LP(D.L.g_Bp);
#pragma unroll
      for (int r = 0; r < RPT; r++) { const int i = hf + NH * r; if (colok && i < n && i >= jc) Pp[i * (i + 1) / 2 + jc] = -Br[r]; }
    }
  }
  return !__syncthreads_or(bad);
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


// ---- the ADMM iteration's own products.  Per QP (the T slot is free during the ADMM) three index tables are built from the static
// structure of G: for every column the six box / rate rows that touch it, and the list of (dense gradient, entry) pairs covering it --
// the packed gradients transposed by index.  G' w is then one pass with four lanes per column over independent, unrollable loads; the
// generic gt_mul (row / gradient tables looked up per column, row indices from constant memory) took 14 of an iteration's 29 kcycles.
struct OsqpTabs {
  const __attribute__((address_space(3))) short* colrows;            // [n][8]: r_in_ub, r_in_lb, r_rate_ub, r_rate_lb, r_rate_ub(t+1), r_rate_lb(t+1), -, -   (-1: none)
  __attribute__((address_space(3))) unsigned short* cstart;          // [n + 1]
  __attribute__((address_space(3))) unsigned int* pairT;             // [ngd]: (dense gradient << 16) | offset of the entry inside the packed gradients ... offsets < 65536
};
__device__ inline OsqpTabs osqp_tabs() {
  const DgProb& D = dg_prob;
  lptr base = LP(D.L.a_tab);
  OsqpTabs t;
  t.colrows = (const __attribute__((address_space(3))) short*)base;
  t.cstart = (__attribute__((address_space(3))) unsigned short*)(base + 2 * D.n);
  t.pairT = (__attribute__((address_space(3))) unsigned int*)(base + 2 * D.n + (D.n + 8) / 4);
  return t;
}
__device__ inline void osqp_build_tables(const DgProb& D, const OsqpTabs& T) {
  const int n = D.n;
  __attribute__((address_space(3))) short* cr = (__attribute__((address_space(3))) short*)T.colrows;
  __syncthreads();
  for (int col = TID; col < n; col += NT) {
    const int a = col / (D.N * DGSQP_NUA), rem = col % (D.N * DGSQP_NUA), t = rem / DGSQP_NUA, j = rem % DGSQP_NUA;
    cr[8 * col + 0] = D.r_in_ub[a][t][j]; cr[8 * col + 1] = D.r_in_lb[a][t][j];
    cr[8 * col + 2] = D.r_rate_ub[a][t][j]; cr[8 * col + 3] = D.r_rate_lb[a][t][j];
    cr[8 * col + 4] = t + 1 < D.N ? D.r_rate_ub[a][t + 1][j] : (short)-1; cr[8 * col + 5] = t + 1 < D.N ? D.r_rate_lb[a][t + 1][j] : (short)-1;
    cr[8 * col + 6] = -1; cr[8 * col + 7] = -1;
    int cnt = 0;
    for (int d = D.stage_dense0[t + 1]; d < D.ndense; d++) { const DgDense dd = ld_dense(d); cnt += (dd.a == a) || (dd.kind == 1 && dd.b == a); }
    T.cstart[col + 1] = (unsigned short)cnt;
  }
  __syncthreads();
  if (TID == 0) { unsigned int s = 0; T.cstart[0] = 0; for (int col = 0; col < n; col++) { s += T.cstart[col + 1]; T.cstart[col + 1] = (unsigned short)s; } }
  __syncthreads();
  for (int col = TID; col < n; col += NT) {
    const int a = col / (D.N * DGSQP_NUA), rem = col % (D.N * DGSQP_NUA), t = rem / DGSQP_NUA, j = rem % DGSQP_NUA;
    int k = T.cstart[col];
    for (int d = D.stage_dense0[t + 1]; d < D.ndense; d++) {
      const DgDense dd = ld_dense(d);
      if (dd.a == a) T.pairT[k++] = ((unsigned int)d << 16) | (unsigned int)(dd.off + t * DGSQP_NUA + j);
      else if (dd.kind == 1 && dd.b == a) T.pairT[k++] = ((unsigned int)d << 16) | (unsigned int)(dd.off + 2 * dd.k + t * DGSQP_NUA + j);
    }
  }
  __syncthreads();
}
// ------------------------------------------------------------------------------------------------
// _solve_qp core with OSQP's arithmetic.  In: M (scratch, ws_xM), q, g, packed G.  Out: du (L.o_du), lhat (L.o_lhat).
// Returns 0 when OSQP hands back a point (solved, solved inaccurate, or the iteration limit: the reference continues from whatever
// OSQP returns), 1 when it reports primal / dual infeasibility or non-finite data, or when the point is not finite (a NaN step:
// DGSQP.py:566-585 raises).
// ------------------------------------------------------------------------------------------------
// Setup of one OSQP call (no registers shared with the ADMM loop: its own function keeps that loop's register allocation clean):
// finite-data check, Ruiz equilibration (section 5.1; OSQP scale_data(): 10 passes) into o.Dv / o.EI / o.E, W = Gs' Gs, the index tables.
// Returns the cost scaling c, or a NaN when the data is not finite.
__device__ __noinline__ double osqp_setup(const Ctx& c) {
  const DgProb& D = dg_prob;
  const int n = D.n, nc = D.nc;
  const OsqpPtrs o = osqp_ptrs(c);
  // ---- data must be finite (the conic plugin returns NaN otherwise)
  {
    int bad = 0;
    for (int e = TID; e < n * n; e += NT) bad |= !(__builtin_fabs(o.M[e]) < INFINITY);
    for (int j = TID; j < n; j += NT) bad |= !(__builtin_fabs(o.q[j]) < INFINITY);
    for (int r = TID; r < nc; r += NT) bad |= (o.g[r] != o.g[r]);
    for (int p = TID; p < D.ngd; p += NT) bad |= !(__builtin_fabs(o.gd[p]) < INFINITY);
    if (__syncthreads_or(bad)) return __builtin_nan("");
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
  PROF_END(PH_O_SCALE, po1);
  // ---- W = Gs' Gs (once per call)
  PROF_BEGIN(po2);
  osqp_build_w(D, o);
  PROF_END(PH_O_W, po2);
  osqp_build_tables(D, osqp_tabs());
  return cc;
}

// Polish (section 4) of the ADMM point held in o.z / o.y (scaled), in unscaled variables (see the header).  On acceptance du / lhat are
// overwritten with the polished point.  Returns 1 accepted, -1 rejected (or not attempted: more active rows than the T slot holds, a
// pivot that was not positive); *na_out = active rows.
template <int RPT>
__device__ __noinline__ int osqp_polish(const Ctx& c, double cc, double pri_res, double dua_res, int* na_out) {
  const DgProb& D = dg_prob;
  const DgLds& L = D.L;
  lptr lds = LP(0);
  const int n = D.n, nc = D.nc;
  const OsqpPtrs o = osqp_ptrs(c);
  lptr du = lds + L.o_du, lhat = lds + L.o_lhat;
  const int lane = TID & 63;
  const bool w0 = TID < 64;
  const double delta = 1e-6, cinv = 1.0 / cc;
  int polished = 0, na = 0;
  double Kr[RPT];
  {
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
      ok = osqp_inverse_regs<RPT>(c, o.M, o.W, nullptr, o.tmp, cc, 0.0, LP(L.a_sw), Kr, true);
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
  *na_out = na;
  return polished;
}

// out_i = sum_j P_ij t_j for the packed symmetric P (LDS, or the scratch in the big layout): dev_p_mul_t without its leading barrier; the
// caller supplies the final phase (what to do with row i's sum)
template <class PT, class F>
__device__ inline void osqp_pmul_fused(PT Pp, clptr t, lptr part, int n, F&& fin) {
  constexpr int NSEG = NT / 128;
  const int i = TID & 127, sg = TID >> 7;
  if (i < n) {
    const int len = (n + NSEG - 1) / NSEG;
    const int j0 = sg * len, j1 = (j0 + len < n) ? j0 + len : n;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    const int split = (i + 1 < j1) ? ((i + 1 > j0) ? i + 1 : j0) : j1;
    const PT row = Pp + i * (i + 1) / 2;
    int j = j0;
    for (; j + 3 < split; j += 4) { a0 += row[j] * t[j]; a1 += row[j + 1] * t[j + 1]; a2 += row[j + 2] * t[j + 2]; a3 += row[j + 3] * t[j + 3]; }
    for (; j < split; j++) a0 += row[j] * t[j];
    for (; j + 3 < j1; j += 4) {
      a0 += Pp[j * (j + 1) / 2 + i] * t[j]; a1 += Pp[(j + 1) * (j + 2) / 2 + i] * t[j + 1];
      a2 += Pp[(j + 2) * (j + 3) / 2 + i] * t[j + 2]; a3 += Pp[(j + 3) * (j + 4) / 2 + i] * t[j + 3];
    }
    for (; j < j1; j++) a0 += Pp[j * (j + 1) / 2 + i] * t[j];
    part[sg * n + i] = (a0 + a1) + (a2 + a3);
  }
  __syncthreads();
  if (TID < n) {
    double s = 0;
#pragma unroll
    for (int g = 0; g < NSEG; g++) s += part[g * n + TID];
    fin(TID, s);
  }
  __syncthreads();
}

#define DG_OSQP_CHK 40    // scal slots 40..46: what osqp_check hands back -- pri_res, dua_res, eps_pri, eps_dua, the two ratios of the rho rule, flags (1 primal, 2 dual infeasible)
__device__ inline double osqp_rho_I(const OsqpPtrs& o, int j, double rho) { return o.EI[j] * OSQP_INFTY > OSQP_INFTY * OSQP_MIN_SCALING ? OSQP_RHO_MIN : rho; }   // "loose" identity row: both bounds beyond 1e26 after scaling

// K(rho)^-1 = (Ps + sigma I + rho_I (E_I D)^2 + rho W)^-1 into the packed-P slot.  Its own function (as are the iteration, the check and
// the polish below): each piece gets a register allocation of its own -- inlined into one function the sweep's 150 live registers
// pushed the iteration's addresses into scratch, a memory round trip per barrier phase.
template <int RPT>
__device__ __noinline__ bool osqp_build_kinv(const Ctx& c, double rho, double cc) {
  const DgProb& D = dg_prob;
  const OsqpPtrs o = osqp_ptrs(c);
  __syncthreads();
  for (int j = TID; j < D.n; j += NT) { const double aI = o.EI[j] * o.Dv[j]; o.tmp[j] = 1e-6 + osqp_rho_I(o, j, rho) * aI * aI; }
  __syncthreads();
  PROF_BEGIN(po3);
  double Br[RPT];
  const bool ok = osqp_inverse_regs<RPT>(c, o.M, o.W, o.Dv, o.tmp, cc, rho, LP(D.L.a_sw), Br, true);
  PROF_END(PH_O_KINV, po3);
  return ok;
}

// `count` ADMM iterations (Algorithm 1 with relaxation alpha = 1.6) on the state in LDS: x, z, y, w = E (rho z - y); leaves delta x, delta y of
// the last one.  The iterations between two termination checks run inside ONE call so that every thread keeps its slice of K^-1 -- row
// TID & 127, a quarter of the columns: LEN <= 32 values -- in registers across them: the product K^-1 rhs then reads only the right-hand
// side from LDS (wave-uniform addresses), 1.3 kcycles instead of 4.5 k with the packed triangle read from LDS in every iteration.
template <int LEN>
__device__ __noinline__ void osqp_iterate_block(const Ctx& c, double rho, double cc, int count) {
  const DgProb& D = dg_prob;
  const int n = D.n, nc = D.nc;
  const OsqpPtrs o = osqp_ptrs(c);
  const OsqpTabs tabs = osqp_tabs();
  const double sigma = 1e-6, alpha = 1.6, irho = 1.0 / rho;
  auto rho_I = [&](int j, double r) { return osqp_rho_I(o, j, r); };
  constexpr int NSEG = NT / 128;
  const int pi = TID & 127, sg = TID >> 7;
  const int plen = (n + NSEG - 1) / NSEG, pj0 = sg * plen, pj1 = (pj0 + plen < n) ? pj0 + plen : n;
  double pr[LEN];
#pragma unroll
  for (int k = 0; k < LEN; k++) {
    const int j = pj0 + k;
    const bool valid = pi < n && j < pj1;
    const int hi = pi > j ? pi : j, lo = pi > j ? j : pi;
    const int idx = valid ? hi * (hi + 1) / 2 + lo : 0;
    const double v = D.big ? (c.ws + D.ws_P)[idx] : LP(D.L.g_Bp)[idx];
    pr[k] = valid ? v : 0.0;
  }
  // ... and the packed gradient entries it multiplies in every iteration: the 16 values of its first dense-dot task (phase 5) and the
  // first GTR (value, gradient index) pairs of its quarter of a column of G' (phase 2) -- both orders of the same numbers
  constexpr int GTR = 14;
  const bool task0 = TID < D.ntask;
  const DgTask T0 = task0 ? ld_task(TID) : DgTask{0, 0, 0, 0};
  double pv0[DG_CHUNK];
#pragma unroll
  for (int i = 0; i < DG_CHUNK; i++) pv0[i] = (task0 && i < T0.len) ? o.gd[T0.p0 + i] : 0.0;
  const bool gcol = TID < 4 * n;
  const int gk0 = gcol ? tabs.cstart[TID >> 2] + (TID & 3) : 0, gk1 = gcol ? tabs.cstart[(TID >> 2) + 1] : 0;
  double gtv[GTR];
  int gti[GTR];
#pragma unroll
  for (int m = 0; m < GTR; m++) {
    const int k = gk0 + 4 * m;
    const bool valid = k < gk1;
    const unsigned int pa = valid ? tabs.pairT[k] : 0u;
    gtv[m] = valid ? o.gd[pa & 0xffffu] : 0.0;
    gti[m] = valid ? (int)(pa >> 16) : 0;
  }
  int gcnt = gcol && gk1 > gk0 ? (gk1 - gk0 + 3) / 4 : 0;               // pairs of this lane held in registers ...
  gcnt = gcnt < GTR ? gcnt : GTR;
  const int gwave = (int)wave_max((double)gcnt);                          // ... and the most any lane of the wavefront holds (uniform loop bound)
  for (int rep = 0; rep < count; rep++) {
    // One ADMM iteration in six barrier phases.  Carried between iterations: w = E (rho z - y)  (rebuilt after a check, which uses w).
    // (1) yd_d = w[r+] - w[r-]: what every dense gradient contributes to G' w
    PROF_BEGIN(pa1);
    for (int d = TID; d < D.ndense; d += NT) {
      const DgDense dd = ld_dense(d);
      o.yd[d] = (dd.r_pos >= 0 ? o.w[dd.r_pos] : 0.0) - (dd.r_neg >= 0 ? o.w[dd.r_neg] : 0.0);
    }
    __syncthreads();
    // (2) rhs = sigma x - qs + D G' w + a_I rho_I (a_I x): four lanes per column over the transposed index table
    for (int it4 = TID; it4 < 4 * n; it4 += NT) {
      const int col = it4 >> 2, part = it4 & 3;
      double s0 = 0, s1 = 0;
      if (part == 0) {
        const __attribute__((address_space(3))) short* cr = tabs.colrows + 8 * col;
        const int r0 = cr[0], r1 = cr[1], r2 = cr[2], r3 = cr[3], r4 = cr[4], r5 = cr[5];
        const double w0 = o.w[r0 >= 0 ? r0 : 0], w1 = o.w[r1 >= 0 ? r1 : 0], w2 = o.w[r2 >= 0 ? r2 : 0], w3 = o.w[r3 >= 0 ? r3 : 0], w4 = o.w[r4 >= 0 ? r4 : 0], w5 = o.w[r5 >= 0 ? r5 : 0];
        s0 = ((r0 >= 0 ? w0 : 0.0) - (r1 >= 0 ? w1 : 0.0)) + ((r2 >= 0 ? w2 : 0.0) - (r3 >= 0 ? w3 : 0.0));
        s1 = (r5 >= 0 ? w5 : 0.0) - (r4 >= 0 ? w4 : 0.0);
      }
      const int k1 = tabs.cstart[col + 1];
      int k = tabs.cstart[col] + part;
      if (it4 == TID) {        // this thread's own column quarter: values and indices from registers (zeros beyond its end)
#pragma unroll
        for (int m = 0; m < GTR; m += 2)
          if (m < gwave) { s0 = __builtin_fma(o.yd[gti[m]], gtv[m], s0); s1 = __builtin_fma(o.yd[gti[m + 1]], gtv[m + 1], s1); }
        k += 4 * GTR;
      }
      for (; k + 4 < k1; k += 8) {
        const unsigned int pa = tabs.pairT[k], pb = tabs.pairT[k + 4];
        const double ga = o.gd[pa & 0xffffu], ya = o.yd[pa >> 16], gb = o.gd[pb & 0xffffu], yb = o.yd[pb >> 16];
        s0 = __builtin_fma(ya, ga, s0); s1 = __builtin_fma(yb, gb, s1);
      }
      if (k < k1) { const unsigned int pa = tabs.pairT[k]; s0 = __builtin_fma(o.yd[pa >> 16], o.gd[pa & 0xffffu], s0); }
      double sm = s0 + s1;
      sm += dpp_f64<0xB1>(sm);
      sm += dpp_f64<0x4E>(sm);
      if (part == 0) {
        const double dj = o.Dv[col], aI = o.EI[col] * dj, xj = o.x[col];
        o.rhs[col] = sigma * xj - cc * dj * o.q[col] + dj * sm + aI * rho_I(col, rho) * (aI * xj);
      }
    }
    __syncthreads();
    PROF_END(PH_O_GT, pa1);
    // (3, 4) xt = K^-1 rhs; relaxation of x; tmp = D xt for the product with G
    PROF_BEGIN(pa2);
    {
      double a[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int k = 0; k < LEN; k++) {
        const int j = pj0 + k < n ? pj0 + k : n - 1;          // (beyond the slice: pr = 0 times a finite entry)
        a[k & 3] = __builtin_fma(pr[k], o.rhs[j], a[k & 3]);
      }
      if (pi < n) o.part[sg * n + pi] = (a[0] + a[1]) + (a[2] + a[3]);
      __syncthreads();
      if (TID < n) {
        double xt = 0;
#pragma unroll
        for (int g = 0; g < NSEG; g++) xt += o.part[g * n + TID];
        const double xp = o.x[TID], xn = alpha * xt + (1.0 - alpha) * xp;
        o.x[TID] = xn; o.dx[TID] = xn - xp; o.tmp[TID] = o.Dv[TID] * xt;
      }
      __syncthreads();
    }
    PROF_END(PH_O_PMUL, pa2);
    // (5) chunk sums of the dense gradients' dots with D xt
    PROF_BEGIN(pa3);
    if (task0) {           // the thread's first task: gradient values from registers
      clptr wv = o.tmp + T0.v0;
      double wq[DG_CHUNK];
#pragma unroll
      for (int i = 0; i < DG_CHUNK; i++) wq[i] = wv[i];
      double sa[4] = {0, 0, 0, 0};
#pragma unroll
      for (int i = 0; i < DG_CHUNK; i++) sa[i & 3] += i < T0.len ? pv0[i] * wq[i] : 0.0;
      o.dpart[TID] = (sa[0] + sa[1]) + (sa[2] + sa[3]);
    }
    for (int t = TID + NT; t < D.ntask; t += NT) {
      const DgTask T = ld_task(t);
      clptr p = o.gd + T.p0;
      clptr wv = o.tmp + T.v0;
      double pv[DG_CHUNK], wq[DG_CHUNK];
#pragma unroll
      for (int i = 0; i < DG_CHUNK; i++) { pv[i] = p[i]; wq[i] = wv[i]; }
      double sa[4] = {0, 0, 0, 0};
#pragma unroll
      for (int i = 0; i < DG_CHUNK; i++) sa[i & 3] += i < T.len ? pv[i] * wq[i] : 0.0;
      o.dpart[t] = (sa[0] + sa[1]) + (sa[2] + sa[3]);
    }
    __syncthreads();
    PROF_END(PH_O_GS, pa3);
    // (6) zt = E G (D xt) row by row, relaxation, projection onto [l, u], dual update, next iteration's w
    PROF_BEGIN(pa4);
    for (int r = TID; r < nc; r += NT) {
      const DgRow R = ld_row(r);
      double zt;
      if (R.dense >= 0) {
        const DgDense dd = ld_dense(R.dense);
        const int ts = dd.t0lo + 256 * dd.t0hi;
        double sd = 0;
        for (int i = 0; i < dd.nt; i++) sd += o.dpart[ts + i];
        zt = (double)R.sgn * sd;
      } else {
        const bool rate = R.type == DG_R_RATE_UB || R.type == DG_R_RATE_LB, pos = R.type == DG_R_IN_UB || R.type == DG_R_RATE_UB;
        const int c1 = am_col(D, R.a, R.k, R.idx);
        const bool has0 = rate && R.k > 0;
        const double v1 = o.tmp[c1], v0 = o.tmp[has0 ? c1 - DGSQP_NUA : c1];
        zt = (pos ? 1.0 : -1.0) * v1 + (has0 ? (pos ? -1.0 : 1.0) : 0.0) * v0;
      }
      const double er = o.E[r], zp = o.z[r], yr = o.y[r];
      const double zr = alpha * (er * zt) + (1.0 - alpha) * zp;
      const double us = er * fmin(-o.g[r], OSQP_INFTY), ls = -OSQP_INFTY * er;
      const double zn = fmin(fmax(__builtin_fma(yr, irho, zr), ls), us);        // (y / rho as a multiplication by 1 / rho: a 20-instruction IEEE division per row otherwise)
      const double dyr = rho * (zr - zn), yn = yr + dyr;
      o.z[r] = zn; o.dy[r] = dyr; o.y[r] = yn;
      o.w[r] = er * (rho * zn - yn);
    }
    __syncthreads();
    PROF_END(PH_O_UPD, pa4);
  }
}

// The termination tests of a check iteration (section 3.4) and the ratios of the rho rule (5.2); results in scal[DG_OSQP_CHK ..].
// `approx`: the check at the iteration limit -- OSQP then repeats check_termination with every tolerance times ten
// (check_termination(work, approximate = 1)); the infeasibility certificates at 10 x eps_inf come out of the same products: flags 4, 8.
__device__ __noinline__ void osqp_check(const Ctx& c, double cc, bool approx) {
  const DgProb& D = dg_prob;
  const int n = D.n, nc = D.nc;
  const OsqpPtrs o = osqp_ptrs(c);
  const double eps_abs = 1e-3, eps_rel = 1e-3, eps_inf = 1e-4, cinv = 1.0 / cc;
  PROF_BEGIN(po5);
    // primal infeasibility certificate (uses delta y, which the residual products overwrite)
    bool pinf = false, pinf10 = false, dinf10 = false;
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
        pinf10 = approx && lhs < -10.0 * eps_inf * nrm && mx < 10.0 * eps_inf * nrm;
      }
    }
    // Ax (G rows) -> w, Px -> rhs, A'y -> xt
    osqp_gs_mul(o, o.x, o.w);
    osqp_m_pass<false>(o.M, n, o.tmp, o.part, o.rhs);               // (o.tmp = D x after osqp_gs_mul)
    for (int j = TID; j < n; j += NT) o.rhs[j] *= cc * o.Dv[j];
    __syncthreads();
    osqp_gst_mul(c, o, o.y, o.dy, o.xt);                            // (dy is free here: its last use was the infeasibility test)
    double pri_res, dua_res, eps_p, eps_d, ad_pr, ad_dr;
    {
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
    }
    bool dinf = false;
    if (!(pri_res <= eps_p && dua_res <= eps_d) && !pinf)
    {
      // dual infeasibility certificate
      double nrm = 0, qdx = 0;
      for (int j = TID; j < n; j += NT) { nrm = fmax(nrm, __builtin_fabs(o.Dv[j] * o.dx[j])); qdx += cc * o.Dv[j] * o.q[j] * o.dx[j]; }
      nrm = block_max(nrm, o.red);
      qdx = block_sum(qdx, o.red);
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
        if (approx && qdx < -cc * 10.0 * eps_inf * nrm && mx < cc * 10.0 * eps_inf * nrm) {      // block-uniform
          const double e10 = 10.0 * eps_inf;
          int viol = 0;
          for (int r = TID; r < nc; r += NT) {
            const double er = o.E[r], us = er * fmin(-o.g[r], OSQP_INFTY), ls = -OSQP_INFTY * er, adx = o.w[r] / er;
            const bool ok_u = us > OSQP_INFTY * OSQP_MIN_SCALING || adx < e10 * nrm;
            const bool ok_l = ls < -OSQP_INFTY * OSQP_MIN_SCALING || adx > -e10 * nrm;
            viol |= !(ok_u && ok_l);
          }
          for (int j = TID; j < n; j += NT) {
            const bool inf_b = o.EI[j] * OSQP_INFTY > OSQP_INFTY * OSQP_MIN_SCALING;
            const double adx = o.Dv[j] * o.dx[j];
            viol |= !((inf_b || adx < e10 * nrm) && (inf_b || adx > -e10 * nrm));
          }
          dinf10 = !__syncthreads_or(viol);
        }
      }
    }
    __syncthreads();
    if (TID == 0) {
      o.scal[DG_OSQP_CHK] = pri_res; o.scal[DG_OSQP_CHK + 1] = dua_res; o.scal[DG_OSQP_CHK + 2] = eps_p; o.scal[DG_OSQP_CHK + 3] = eps_d;
      o.scal[DG_OSQP_CHK + 4] = ad_pr; o.scal[DG_OSQP_CHK + 5] = ad_dr; o.scal[DG_OSQP_CHK + 6] = (pinf ? 1.0 : 0.0) + (dinf ? 2.0 : 0.0) + (pinf10 ? 4.0 : 0.0) + (dinf10 ? 8.0 : 0.0);
    }
    __syncthreads();
    PROF_END(PH_O_CHECK, po5);
}

template <int RPT>
__device__ __noinline__ int dev_qp_osqp_t(const Ctx& c) {
  const DgProb& D = dg_prob;
  const DgLds& L = D.L;
  lptr lds = LP(0);
  const int n = D.n, nc = D.nc;
  const OsqpPtrs o = osqp_ptrs(c);
  lptr du = lds + L.o_du, lhat = lds + L.o_lhat;
  const int max_iter = 4000, check_every = 25;
  __syncthreads();
  PROF_BEGIN(pt_qp);
  if (TID == 0) { o.scal[DG_QP_NPREV] = 0.0; o.scal[DG_XVALID] = 0.0; }
  // ---- setup: finite data, Ruiz equilibration, W = Gs' Gs, index tables
  const double cc = osqp_setup(c);
  if (cc != cc) {        // non-finite data: the conic plugin returns NaN
    if (TID == 0) { o.scal[DG_OSQP_INFO] = OSQP_NAN_DATA; o.scal[DG_OSQP_INFO + 1] = 0; o.scal[DG_OSQP_INFO + 2] = 0; }
    __syncthreads();
    return 1;
  }
  const double cinv = 1.0 / cc;
  double rho = D.par.osqp_rho_carry ? o.scal[DG_OSQP_RHO] : 0.1;      // (carried from the scenario's previous call: include/dgsqp.h)
  int rho_updates = 0;
  for (int j = TID; j < n; j += NT) { o.x[j] = 0.0; o.dx[j] = 0.0; }
  for (int r = TID; r < nc; r += NT) { o.z[r] = 0.0; o.y[r] = 0.0; o.dy[r] = 0.0; o.w[r] = 0.0; }
  __syncthreads();
  // The G rows have  l = -inf -> -1e30 E_r,  u = E_r min(-g_r, 1e30):  never equalities (rho_vec = rho on all of them), never "loose"
  // unless -g_r >= 1e26 / E_r (then OSQP gives the row rho_min; not reproduced: no game produces such a row)
  int status = OSQP_MAX_ITER, iters = 0, approx_flags = 0;
  double pri_res = INFINITY, dua_res = INFINITY, eps_p = 0, eps_d = 0;
  bool need_kinv = true;
  PROF_BEGIN(po4);
  static_assert(4000 % 25 == 0, "the iteration limit falls on a termination check");
  constexpr int PLEN = (RPT * DG_NH + NT / 128 - 1) / (NT / 128);      // columns of K^-1 per thread (RPT * DG_NH = the size class of n)
  for (int it = check_every; it <= max_iter; it += check_every) {       // one block of iterations, then a termination check
    if (need_kinv) {      // first iteration, or rho was changed by the previous check
      need_kinv = false;
      if (!osqp_build_kinv<RPT>(c, rho, cc)) { status = OSQP_NAN_DATA; break; }
      // (the sweep's column buffers lie over delta y and w)
      for (int r = TID; r < nc; r += NT) { o.dy[r] = 0.0; o.w[r] = o.E[r] * (rho * o.z[r] - o.y[r]); }
      __syncthreads();
    }
    osqp_iterate_block<PLEN>(c, rho, cc, check_every);
    iters = it;
    // ---- termination (section 3.4) every 25 iterations; the same products serve the rho adaptation (section 5.2)
    osqp_check(c, cc, it == max_iter);
    pri_res = o.scal[DG_OSQP_CHK]; dua_res = o.scal[DG_OSQP_CHK + 1]; eps_p = o.scal[DG_OSQP_CHK + 2]; eps_d = o.scal[DG_OSQP_CHK + 3];
    const double ad_pr = o.scal[DG_OSQP_CHK + 4], ad_dr = o.scal[DG_OSQP_CHK + 5];
    const int flags = (int)o.scal[DG_OSQP_CHK + 6];
    if (pri_res <= eps_p && dua_res <= eps_d) { status = OSQP_SOLVED; break; }
    if (flags & 1) { status = OSQP_PRIMAL_INFEASIBLE; break; }
    if (flags & 2) { status = OSQP_DUAL_INFEASIBLE; break; }
    approx_flags = flags;
    // rho adaptation (interval fixed at 25)
    {
      const double rho_new = fmin(fmax(rho * sqrt(ad_pr / (ad_dr + 1e-10)), OSQP_RHO_MIN), OSQP_RHO_MAX);
      if (rho_new > rho * 5.0 || rho_new < rho / 5.0) {
        rho = rho_new;
        rho_updates++;
        need_kinv = true;
      }
    }
    // (the check used w; the next iteration needs w = E (rho z - y), with the rho just chosen)
    for (int r = TID; r < nc; r += NT) o.w[r] = o.E[r] * (rho * o.z[r] - o.y[r]);
    __syncthreads();
  }
  PROF_END(PH_O_ADMM, po4);
  PROF_COUNT(PH_O_ITERS, iters);
  // iteration limit: OSQP re-checks with 10x the tolerances ("solved inaccurate"); iteration 4000 is a check iteration, its residuals are at hand
  // ... and the two infeasibility certificates with 10x their tolerance, in OSQP's order (flags 4, 8 of the last check): an "inaccurate"
  // infeasibility verdict returns NaN like an accurate one (osqp's store_solution), i.e. the failing flag here
  if (status == OSQP_MAX_ITER && iters == max_iter) {
    if (pri_res <= 10.0 * eps_p && dua_res <= 10.0 * eps_d) status = OSQP_SOLVED_INACCURATE;
    else if (approx_flags & 4) status = OSQP_PRIMAL_INFEASIBLE_INACCURATE;
    else if (approx_flags & 8) status = OSQP_DUAL_INFEASIBLE_INACCURATE;
  }
  __syncthreads();
  // ---- the ADMM iterate, unscaled, is the answer unless the polish improves on it
  for (int j = TID; j < n; j += NT) du[j] = o.Dv[j] * o.x[j];
  for (int r = TID; r < nc; r += NT) lhat[r] = cinv * o.E[r] * o.y[r];
  int polished = 0, na = 0;
  if (status == OSQP_SOLVED) polished = osqp_polish<RPT>(c, cc, pri_res, dua_res, &na);
  __syncthreads();
  if (TID == 0) {
    o.scal[DG_OSQP_INFO] = (double)status; o.scal[DG_OSQP_INFO + 1] = (double)iters; o.scal[DG_OSQP_INFO + 2] = (double)polished; o.scal[DG_OSQP_INFO + 3] = rho;
    atomicAdd(&dg_osqp_count[0], 1ULL); atomicAdd(&dg_osqp_count[1], (unsigned long long)iters);
    if (D.par.osqp_rho_carry) o.scal[DG_OSQP_RHO] = rho;
    o.scal[DG_OSQP_INFO + 4] = (double)rho_updates; o.scal[DG_OSQP_INFO + 5] = (double)na; o.scal[DG_OSQP_INFO + 6] = pri_res; o.scal[DG_OSQP_INFO + 7] = dua_res;
  }
  // a non-finite answer (an ADMM run that overflowed before its iteration limit) is a NaN step as well
  int nonfinite = 0;
  for (int j = TID; j < n; j += NT) nonfinite |= !(__builtin_fabs(du[j]) < INFINITY);
  for (int r = TID; r < nc; r += NT) nonfinite |= !(__builtin_fabs(lhat[r]) < INFINITY);
  nonfinite = __syncthreads_or(nonfinite);
  PROF_END(PH_QP, pt_qp);
  return (nonfinite || status == OSQP_PRIMAL_INFEASIBLE || status == OSQP_DUAL_INFEASIBLE || status == OSQP_PRIMAL_INFEASIBLE_INACCURATE ||
          status == OSQP_DUAL_INFEASIBLE_INACCURATE || status == OSQP_NAN_DATA) ? 1 : 0;
}
__device__ inline int dev_qp_osqp(const Ctx& c) {
  const int n = dg_prob.n;
  if (n <= 32) return dev_qp_osqp_t<32 / DG_NH>(c);
  if (n <= 64) return dev_qp_osqp_t<64 / DG_NH>(c);
  if (n <= 100) return dev_qp_osqp_t<100 / DG_NH>(c);
  return dev_qp_osqp_t<128 / DG_NH>(c);
}
