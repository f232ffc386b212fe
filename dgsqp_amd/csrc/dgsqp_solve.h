// PSD projection (_nearestPD), the QP sub-problem (_solve_qp), the dual initialisation (LSQR),
// the merit / line-search / watchdog logic and the SQP outer loop of DGSQP.solve() on device.
#pragma once
#include "dgsqp_eval.h"

__device__ inline int tri(int i, int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }

// ------------------------------------------------------------------------------------------------
// _nearestPD + reg (DGSQP.py:1290-1296, :238-239) by parallel-order cyclic Jacobi in LDS.
// In : raw Q (global workspace).  Out: P = (nearestPD(Q) + reg I)^-1 packed in L.g_Bp.
// The projected matrix itself is never needed by the QP below, only its inverse.
// If Qpd != nullptr the projected+regularised matrix is also written there (test hook).
// ------------------------------------------------------------------------------------------------
__device__ inline void dev_psd_inverse(const Ctx& c, double* Qpd) {
  const DgProb& D = *c.D;
  const DgLds& L = D.L;
  double* lds = c.lds;
  const int n = D.n, npk = n * (n + 1) / 2;
  const int np = (n + 1) & ~1, m = np / 2;
  double* Bp = lds + L.g_Bp;
  double* Vt = lds + L.g_V;  // Vt[k*n + i] = component i of eigenvector k
  double* rc = lds + L.g_rot;
  double* rs = rc + m + 1;
  int* rp = (int*)(rs + m + 1);
  int* rq = rp + m + 1;
  double* wts = lds + L.g_rot + 4 * ((n + 1) / 2 + 1);
  double* sev = wts + n + 2;
  double* red = lds + L.red;
  const double* Qg = c.ws + D.ws_q;
  __syncthreads();
  PROF_BEGIN(pt_j);
  for (int t = TID; t < npk; t += NT) {
    int i = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while ((i + 1) * (i + 2) / 2 <= t) i++;
    while (i * (i + 1) / 2 > t) i--;
    const int j = t - i * (i + 1) / 2;
    Bp[t] = 0.5 * (Qg[(int64_t)i * n + j] + Qg[(int64_t)j * n + i]);  // B = (A + A^T)/2
  }
  for (int t = TID; t < n * n; t += NT) Vt[t] = (t / n == t % n) ? 1.0 : 0.0;
  __syncthreads();
  bool final_sweep = false;
  for (int sweep = 0; sweep < 40; sweep++) {
    double off = 0, dg = 0;
    for (int t = TID; t < npk; t += NT) {
      int i = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
      while ((i + 1) * (i + 2) / 2 <= t) i++;
      while (i * (i + 1) / 2 > t) i--;
      const double v = Bp[t];
      if (t - i * (i + 1) / 2 == i) dg += v * v; else off += v * v;
    }
    off = block_sum(off, red);
    dg = block_sum(dg, red);
    if (final_sweep || off <= 1e-31 * dg || off < 1e-300) break;
    // Jacobi converges quadratically: once |off| <= 1e-9 |B|_F one more sweep reaches the rounding floor
    if (off <= 1e-18 * dg) final_sweep = true;
    PROF_BEGIN(pt_s);
    for (int rd = 0; rd < np - 1; rd++) {
      // rotation angles of the m disjoint pairs of this round
      if (TID < m) {
        int a, b;
        if (TID == 0) { a = np - 1; b = rd; }
        else { a = (rd + TID) % (np - 1); b = (rd - TID + (np - 1)) % (np - 1); }
        const int p = a < b ? a : b, q = a < b ? b : a;
        double cs = 1.0, sn = 0.0;
        if (q < n) {
          const double apq = Bp[tri(q, p)];
          if (apq != 0.0) {
            const double app = Bp[tri(p, p)], aqq = Bp[tri(q, q)];
            const double theta = (aqq - app) / (2.0 * apq);
            const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            cs = 1.0 / sqrt(t * t + 1.0);
            sn = t * cs;
          }
        }
        rc[TID] = cs; rs[TID] = sn; rp[TID] = p; rq[TID] = q;
      }
      __syncthreads();
      // B <- J^T B J, one 2x2 block per task
      for (int t = TID; t < m * (m + 1) / 2; t += NT) {
        int r2 = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
        while ((r2 + 1) * (r2 + 2) / 2 <= t) r2++;
        while (r2 * (r2 + 1) / 2 > t) r2--;
        const int r1 = t - r2 * (r2 + 1) / 2;
        const int p1 = rp[r1], q1 = rq[r1], p2 = rp[r2], q2 = rq[r2];
        if (q1 >= n || q2 >= n) continue;
        const double c1 = rc[r1], s1 = rs[r1], c2 = rc[r2], s2 = rs[r2];
        if (r1 == r2) {
          const int ipp = tri(p1, p1), iqq = tri(q1, q1), ipq = tri(q1, p1);
          const double apq = Bp[ipq];
          if (s1 != 0.0) {
            const double t2 = s1 / c1;
            Bp[ipp] -= t2 * apq; Bp[iqq] += t2 * apq; Bp[ipq] = 0.0;
          }
        } else {
          const int i00 = tri(p1, p2), i01 = tri(p1, q2), i10 = tri(q1, p2), i11 = tri(q1, q2);
          const double x00 = Bp[i00], x01 = Bp[i01], x10 = Bp[i10], x11 = Bp[i11];
          const double y00 = c2 * x00 - s2 * x01, y01 = s2 * x00 + c2 * x01;
          const double y10 = c2 * x10 - s2 * x11, y11 = s2 * x10 + c2 * x11;
          Bp[i00] = c1 * y00 - s1 * y10; Bp[i01] = c1 * y01 - s1 * y11;
          Bp[i10] = s1 * y00 + c1 * y10; Bp[i11] = s1 * y01 + c1 * y11;
        }
      }
      // V <- V J (rows of Vt)
      for (int t = TID; t < m * n; t += NT) {
        const int r = t / n, i = t % n;
        const int p = rp[r], q = rq[r];
        if (q >= n) continue;
        const double cs = rc[r], sn = rs[r];
        const double vp = Vt[p * n + i], vq = Vt[q * n + i];
        Vt[p * n + i] = cs * vp - sn * vq;
        Vt[q * n + i] = sn * vp + cs * vq;
      }
      __syncthreads();
    }
    PROF_END(PH_SWEEP, pt_s);
  }
  PROF_END(PH_JACOBI, pt_j);
  PROF_BEGIN(pt_p);
  // eigenvalues: negative -> 1e-10 (DGSQP.py:1294), then + reg; P = V diag(1/.) V^T
  for (int i = TID; i < n; i += NT) {
    double s = Bp[tri(i, i)];
    if (s < 0) s = 1e-10;
    sev[i] = s;
    wts[i] = 1.0 / (s + (D.par.reg > 0 ? D.par.reg : 0.0));
  }
  __syncthreads();
  if (Qpd) {
    for (int t = TID; t < n * n; t += NT) {
      const int i = t / n, j = t % n;
      double a = 0;
      for (int k = 0; k < n; k++) a += Vt[k * n + i] * sev[k] * Vt[k * n + j];
      if (i == j && D.par.reg > 0) a += D.par.reg;
      Qpd[t] = a;
    }
  }
  for (int t = TID; t < npk; t += NT) {
    int i = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while ((i + 1) * (i + 2) / 2 <= t) i++;
    while (i * (i + 1) / 2 > t) i--;
    const int j = t - i * (i + 1) / 2;
    double a = 0;
    for (int k = 0; k < n; k++) a += Vt[k * n + i] * wts[k] * Vt[k * n + j];
    Bp[t] = a;
  }
  __syncthreads();
  PROF_END(PH_PFORM, pt_p);
}

// out = P t  (P packed symmetric in LDS)
__device__ inline void dev_p_mul(const Ctx& c, const double* t, double* out, double scale) {
  const DgProb& D = *c.D;
  const double* Pp = c.lds + D.L.g_Bp;
  const int n = D.n;
  __syncthreads();
  for (int i = TID; i < n; i += NT) {
    double s = 0;
    const double* row = Pp + i * (i + 1) / 2;
    for (int j = 0; j <= i; j++) s += row[j] * t[j];
    for (int j = i + 1; j < n; j++) s += Pp[j * (j + 1) / 2 + i] * t[j];
    out[i] = scale * s;
  }
  __syncthreads();
}

// coefficient of constraint row r at column col
__device__ inline double g_row_coef(const DgProb& D, const double* gd, int r, int col) {
  const DgRow R = D.rows[r];
  const int a = col / (D.N * DGSQP_NUA), rem = col % (D.N * DGSQP_NUA), t = rem / DGSQP_NUA, j = rem % DGSQP_NUA;
  switch (R.type) {
    case DG_R_IN_UB: return (a == R.a && t == R.k && j == R.idx) ? 1.0 : 0.0;
    case DG_R_IN_LB: return (a == R.a && t == R.k && j == R.idx) ? -1.0 : 0.0;
    case DG_R_RATE_UB:
    case DG_R_RATE_LB: {
      if (a != R.a || j != R.idx) return 0.0;
      double v = 0;
      if (t == R.k) v = 1.0; else if (t == R.k - 1) v = -1.0;
      return R.type == DG_R_RATE_UB ? v : -v;
    }
    default: {
      const DgDense dd = D.dense[R.dense];
      if (t >= dd.k) return 0.0;
      if (a == dd.a) return R.sgn * gd[dd.off + t * DGSQP_NUA + j];
      if (dd.kind == 1 && a == dd.b) return R.sgn * gd[dd.off + 2 * dd.k + t * DGSQP_NUA + j];
      return 0.0;
    }
  }
}

// wavefront-0 helper: solve (R^T R) r = c for the packed upper-triangular factor R of order m (<= 128).
// Lane j keeps entries j and j+64 in registers; pivots are broadcast with shuffles, so there is no LDS
// hazard inside the substitution loops.  Writes w = R^-T c to wv, r to rv and returns |w|^2 to every lane.
__device__ inline double qp_wave_solve(const double* R, int m, int lane, const double* cvec, double* wv, double* rv,
                                       double& r0_out, double& r1_out) {
  double c0 = lane < m ? cvec[lane] : 0.0, c1 = lane + 64 < m ? cvec[lane + 64] : 0.0;
  double w0 = 0, w1 = 0;
  for (int i = 0; i < m; i++) {
    const double ci = __shfl(i < 64 ? c0 : c1, i & 63);
    const double wi = ci / R[tri(i, i)];
    if (lane == (i & 63)) { if (i < 64) w0 = wi; else w1 = wi; }
    if (lane > i && lane < m) c0 -= R[tri(lane, i)] * wi;
    if (lane + 64 > i && lane + 64 < m) c1 -= R[tri(lane + 64, i)] * wi;
  }
  double ww = w0 * w0 + w1 * w1;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ww += __shfl_xor(ww, o);
  if (lane < m) wv[lane] = w0;
  if (lane + 64 < m) wv[lane + 64] = w1;
  double r0 = 0, r1 = 0;
  for (int j = m - 1; j >= 0; j--) {
    const double wj = __shfl(j < 64 ? w0 : w1, j & 63);
    const double rj = wj / R[tri(j, j)];
    if (lane == (j & 63)) { if (j < 64) r0 = rj; else r1 = rj; }
    if (lane < j) w0 -= R[tri(j, lane)] * rj;
    if (lane + 64 < j) w1 -= R[tri(j, lane + 64)] * rj;
  }
  if (lane < m) rv[lane] = r0;
  if (lane + 64 < m) rv[lane + 64] = r1;
  r0_out = r0; r1_out = r1;
  return ww;
}

// ------------------------------------------------------------------------------------------------
// _solve_qp core (DGSQP.py:246):  min 1/2 x'Bx + q'x  s.t.  G x <= -g   with P = B^-1 in LDS.
// Dual active-set method (Goldfarb-Idnani 1983) in range-space form: the Cholesky factor R of the
// Schur complement G_A P G_A^T is kept packed in LDS and updated by wavefront 0 with shuffles.
// This is the KKT point OSQP(polish=True) returns when its polish succeeds.
// Out: du (L.o_du), lhat (L.o_lhat).  Returns 0 ok, 1 infeasible, 2 iteration limit.
// ------------------------------------------------------------------------------------------------
__device__ inline int dev_qp(const Ctx& c) {
  const DgProb& D = *c.D;
  const DgLds& L = D.L;
  double* lds = c.lds;
  const int n = D.n, nc = D.nc;
  double *x = lds + L.o_du, *lhat = lds + L.o_lhat;
  double *R = lds + L.p_R, *lam = lds + L.p_lam, *cvec = lds + L.p_c, *wv = lds + L.p_w, *rv = lds + L.p_r;
  double *y = lds + L.p_y, *z = lds + L.p_z, *tv = lds + L.p_t;
  int* alist = (int*)(lds + L.p_alist);
  unsigned char* act = (unsigned char*)(lds + L.p_act);
  double* red = lds + L.red;
  double* scal = lds + L.scal;
  const double* gd = lds + L.gd;
  const double* g = lds + L.g;
  const double* q = lds + L.q;
  const double TOL = 1e-10;
  const int lane = TID & 63;
  const int NONE = 0x7fffffff;

  __syncthreads();
  PROF_BEGIN(pt_qp);
  for (int r = TID; r < nc; r += NT) act[r] = 0;
  dev_p_mul(c, q, x, -1.0);  // unconstrained minimiser x = -P q
  int m = 0;
  int ret = 2;
  const int max_outer = 4 * (n + nc);
  for (int iter = 0; iter < max_outer; iter++) {
    // ---- step 1: most violated inactive constraint (lowest index on ties)
    double best = -TOL;
    int bi = NONE;
    for (int r = TID; r < nc; r += NT) {
      if (act[r]) continue;
      const double s = -(g[r] + g_row_dot(D, gd, r, x));
      if (s < best || (s == best && r < bi)) { best = s; bi = r; }
    }
    double bv; int p;
    block_argmin(best, bi, red, bv, p);
    if (p == NONE) { ret = 0; break; }
    for (int col = TID; col < n; col += NT) tv[col] = g_row_coef(D, gd, p, col);
    dev_p_mul(c, tv, y, 1.0);  // y = P a_p
    double part = 0, part2 = 0;
    for (int i = TID; i < n; i += NT) { part += tv[i] * y[i]; part2 += tv[i] * tv[i]; }
    const double app = block_sum(part, red);
    const double apap = block_sum(part2, red);
    double lp = 0.0;
    bool infeasible = false;
    for (int inner = 0; inner < 4 * (n + nc); inner++) {
      // ---- step 2a: directions.  c = A_A y ; R^T w = c ; r = R^-1 w
      for (int j = TID; j < m; j += NT) cvec[j] = g_row_dot(D, gd, alist[j], y);
      double pv = 0;
      for (int i = TID; i < n; i += NT) pv += tv[i] * x[i];
      const double viol = block_sum(pv, red) + g[p];  // a_p.x - b_p  (b = -g), > 0
      if (TID < 64) {
        double r0, r1;
        const double ww = qp_wave_solve(R, m, lane, cvec, wv, rv, r0, r1);
        // ---- step 2b: step lengths.  t1 keeps the multipliers >= 0, t2 makes constraint p active
        double t1 = INFINITY; int jd = NONE;
        if (lane < m && r0 > 0) { t1 = lam[lane] / r0; jd = lane; }
        if (lane + 64 < m && r1 > 0) { const double tt = lam[lane + 64] / r1; if (tt < t1) { t1 = tt; jd = lane + 64; } }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          const double t2 = __shfl_xor(t1, o); const int j2 = __shfl_xor(jd, o);
          if (t2 < t1 || (t2 == t1 && j2 < jd)) { t1 = t2; jd = j2; }
        }
        if (lane == 0) {
          const double delta = app - ww;  // a_p^T (P - P A^T S^-1 A P) a_p >= 0
          const bool indep = delta > 1e-11 * app && delta > 1e-18 * apap;
          const double t2 = indep ? viol / delta : INFINITY;
          scal[0] = fmin(t1, t2);
          scal[1] = (double)jd;
          scal[2] = delta;
          scal[3] = indep ? 0.0 : 1.0;          // 1: dual step only
          scal[4] = (t1 < t2) ? 1.0 : 0.0;      // 1: partial step (blocking multiplier reaches 0 first)
        }
      }
      __syncthreads();
      const double t = scal[0];
      const int jd = (int)scal[1];
      const double delta = scal[2];
      const bool dual_only = scal[3] != 0.0;
      const bool partial = scal[4] != 0.0;
      if (!(t < INFINITY)) { infeasible = true; break; }
      if (!dual_only) {
        // primal direction z = -(y - P A_A^T r);  x += t z
        for (int col = TID; col < n; col += NT) {
          double s = 0;
          for (int j = 0; j < m; j++) s += rv[j] * g_row_coef(D, gd, alist[j], col);
          z[col] = s;
        }
        dev_p_mul(c, z, cvec, 1.0);
        for (int i = TID; i < n; i += NT) x[i] += t * (cvec[i] - y[i]);
      }
      for (int j = TID; j < m; j += NT) lam[j] -= t * rv[j];
      lp += t;
      __syncthreads();
      if (!dual_only && !partial) {
        // ---- full step: constraint p becomes active, append column [w ; sqrt(delta)] to R
        for (int i = TID; i < m; i += NT) R[tri(m, i)] = wv[i];
        if (TID == 0) { R[tri(m, m)] = sqrt(delta); alist[m] = p; lam[m] = lp; act[p] = 1; }
        m++;
        __syncthreads();
        break;
      }
      // ---- partial / dual-only step: drop blocking constraint jd (column deletion + Givens)
      if (TID < 64) {
        const int mn = m - 1;
        const int ca = lane, cb = lane + 64;
        const bool sa = ca >= jd && ca < mn, sb = cb >= jd && cb < mn;
        if (lane == 0) act[alist[jd]] = 0;
        // old diagonals become the sub-diagonal of the shifted columns
        double suba = sa ? R[tri(ca + 1, ca + 1)] : 0.0, subb = sb ? R[tri(cb + 1, cb + 1)] : 0.0;
        const int ala = sa ? alist[ca + 1] : 0, alb = sb ? alist[cb + 1] : 0;
        const double lma = sa ? lam[ca + 1] : 0.0, lmb = sb ? lam[cb + 1] : 0.0;
        for (int i = 0; i < mn; i++) {  // row-synchronous shift: new column cc <- old column cc+1, rows 0..cc
          double ta = 0, tb = 0;
          const bool da = sa && i <= ca, db = sb && i <= cb;
          if (da) ta = R[tri(ca + 1, i)];
          if (db) tb = R[tri(cb + 1, i)];
          if (da) R[tri(ca, i)] = ta;
          if (db) R[tri(cb, i)] = tb;
        }
        if (sa) { alist[ca] = ala; lam[ca] = lma; }
        if (sb) { alist[cb] = alb; lam[cb] = lmb; }
        for (int k = jd; k < mn; k++) {
          double dk = 0;
          if (lane == (k & 63)) dk = R[tri(k, k)];
          dk = __shfl(dk, k & 63);
          const double sub = __shfl(k < 64 ? suba : subb, k & 63);
          const double h = hypot(dk, sub);
          const double cs = h > 0 ? dk / h : 1.0, sn = h > 0 ? sub / h : 0.0;
          if (lane == (k & 63)) R[tri(k, k)] = h;
          if (ca > k && ca < mn) {
            const double ra = R[tri(ca, k)];
            if (ca == k + 0) {}
            const double rb = R[tri(ca, k + 1)];
            R[tri(ca, k)] = cs * ra + sn * rb;
            R[tri(ca, k + 1)] = -sn * ra + cs * rb;
          }
          if (cb > k && cb < mn) {
            const double ra = R[tri(cb, k)], rb = R[tri(cb, k + 1)];
            R[tri(cb, k)] = cs * ra + sn * rb;
            R[tri(cb, k + 1)] = -sn * ra + cs * rb;
          }
        }
      }
      m--;
      __syncthreads();
    }
    if (infeasible) { ret = 1; break; }
  }
  __syncthreads();
  // Iterative refinement on the final active set (what OSQP's polish does with polish_refine_iter):
  // P is an explicit inverse, so the active rows hold to ~1e-12 only; two projection steps
  //   x <- x - P A^T S^-1 (A x - b),  lam <- lam + S^-1 (A x - b)
  // bring them to rounding level.
  if (ret == 0 && m > 0) {
    for (int pass = 0; pass < 2; pass++) {
      for (int j = TID; j < m; j += NT) cvec[j] = g[alist[j]] + g_row_dot(D, gd, alist[j], x);
      __syncthreads();
      if (TID < 64) { double r0, r1; (void)qp_wave_solve(R, m, lane, cvec, wv, rv, r0, r1); }
      __syncthreads();
      for (int col = TID; col < n; col += NT) {
        double s = 0;
        for (int j = 0; j < m; j++) s += rv[j] * g_row_coef(D, gd, alist[j], col);
        z[col] = s;
      }
      dev_p_mul(c, z, cvec, 1.0);
      for (int i = TID; i < n; i += NT) x[i] -= cvec[i];
      for (int j = TID; j < m; j += NT) lam[j] += rv[j];
      __syncthreads();
    }
  }
  for (int r = TID; r < nc; r += NT) lhat[r] = 0.0;
  __syncthreads();
  if (ret == 0)
    for (int j = TID; j < m; j += NT) lhat[alist[j]] = lam[j];
  __syncthreads();
  PROF_END(PH_QP, pt_qp);
  return ret;
}

// ------------------------------------------------------------------------------------------------
// dual initialisation  l = max(0, -lsqr(G G^T, G q))   (DGSQP.py:320-327).
// LSQR restated from scipy 1.15.3 scipy/sparse/linalg/_isolve/lsqr.py (Paige & Saunders 1982) with
// its defaults damp=0, atol=btol=1e-6, conlim=1e8, iter_lim=2*n_c; the operator G G^T is applied as
// G (G^T v) through the packed constraint gradients instead of being assembled.
// ------------------------------------------------------------------------------------------------
__device__ inline void dev_ggt_mul(const Ctx& c, const double* vin, double* vout, double* tmpn) {
  const DgProb& D = *c.D;
  gt_mul(c, vin, tmpn);
  for (int r = TID; r < D.nc; r += NT) vout[r] = g_row_dot(D, c.lds + D.L.gd, r, tmpn);
  __syncthreads();
}
__device__ inline void dev_sym_ortho(double a, double b, double& cs, double& sn, double& r) {
  if (b == 0) { cs = (a > 0) - (a < 0); sn = 0; r = fabs(a); }
  else if (a == 0) { cs = 0; sn = (b > 0) - (b < 0); r = fabs(b); }
  else if (fabs(b) > fabs(a)) { const double tau = a / b; sn = ((b > 0) - (b < 0)) / sqrt(1 + tau * tau); cs = sn * tau; r = b / sn; }
  else { const double tau = b / a; cs = ((a > 0) - (a < 0)) / sqrt(1 + tau * tau); sn = cs * tau; r = a / cs; }
}
__device__ inline void dev_dual_init(const Ctx& c) {
  const DgProb& D = *c.D;
  const DgLds& L = D.L;
  double* lds = c.lds;
  const int nc = D.nc;
  double *u = lds + L.s_u, *v = lds + L.s_v, *w = lds + L.s_w, *x = lds + L.s_x, *tmp = lds + L.s_t;
  double* tn = lds + L.d;  // n-vector scratch (d is recomputed afterwards)
  double* red = lds + L.red;
  double* l = lds + L.l;
  PROF_BEGIN(pt_l);
  const double eps = 2.220446049250313e-16;
  const double atol = D.par.lsqr_atol, btol = D.par.lsqr_btol, ctol = 1e-8;
  const int iter_lim = D.par.lsqr_iter_lim > 0 ? D.par.lsqr_iter_lim : 2 * nc;
  // b = G q
  __syncthreads();
  double p = 0;
  for (int r = TID; r < nc; r += NT) { const double b = g_row_dot(D, lds + L.gd, r, lds + L.q); u[r] = b; x[r] = 0.0; p += b * b; }
  const double bnorm = sqrt(block_sum(p, red));
  double beta = bnorm, alfa = 0;
  if (beta > 0) {
    for (int r = TID; r < nc; r += NT) u[r] *= 1 / beta;
    dev_ggt_mul(c, u, v, tn);
    p = 0;
    for (int r = TID; r < nc; r += NT) p += v[r] * v[r];
    alfa = sqrt(block_sum(p, red));
  } else {
    for (int r = TID; r < nc; r += NT) v[r] = 0.0;
  }
  if (alfa > 0) for (int r = TID; r < nc; r += NT) v[r] *= 1 / alfa;
  for (int r = TID; r < nc; r += NT) w[r] = v[r];
  __syncthreads();
  double rhobar = alfa, phibar = beta;
  double anorm = 0, ddnorm = 0, res2 = 0, xnorm = 0, xxnorm = 0, zz = 0, cs2 = -1, sn2 = 0;
  int itn = 0;
  if (alfa * beta != 0) {
    while (itn < iter_lim) {
      itn++;
      dev_ggt_mul(c, v, tmp, tn);
      p = 0;
      for (int r = TID; r < nc; r += NT) { const double t = tmp[r] - alfa * u[r]; u[r] = t; p += t * t; }
      beta = sqrt(block_sum(p, red));
      if (beta > 0) {
        for (int r = TID; r < nc; r += NT) u[r] *= 1 / beta;
        anorm = sqrt(anorm * anorm + alfa * alfa + beta * beta);
        dev_ggt_mul(c, u, tmp, tn);
        p = 0;
        for (int r = TID; r < nc; r += NT) { const double t = tmp[r] - beta * v[r]; v[r] = t; p += t * t; }
        alfa = sqrt(block_sum(p, red));
        if (alfa > 0) for (int r = TID; r < nc; r += NT) v[r] *= 1 / alfa;
      }
      double cs, sn, rho;
      dev_sym_ortho(rhobar, beta, cs, sn, rho);
      const double theta = sn * alfa;
      rhobar = -cs * alfa;
      const double phi = cs * phibar;
      phibar = sn * phibar;
      const double tau = sn * phi;
      const double t1 = phi / rho, t2 = -theta / rho;
      p = 0;
      __syncthreads();
      for (int r = TID; r < nc; r += NT) {
        const double wr = w[r], dk = (1 / rho) * wr;
        p += dk * dk;
        x[r] = x[r] + t1 * wr;
        w[r] = v[r] + t2 * wr;
      }
      ddnorm += block_sum(p, red);
      const double delta = sn2 * rho, gambar = -cs2 * rho, rhs = phi - delta * zz, zbar = rhs / gambar;
      xnorm = sqrt(xxnorm + zbar * zbar);
      const double gamma = sqrt(gambar * gambar + theta * theta);
      cs2 = gambar / gamma; sn2 = theta / gamma; zz = rhs / gamma;
      xxnorm += zz * zz;
      const double acond = anorm * sqrt(ddnorm);
      const double rnorm = sqrt(phibar * phibar + res2);
      const double arnorm = alfa * fabs(tau);
      const double test1 = rnorm / bnorm, test2 = arnorm / (anorm * rnorm + eps), test3 = 1 / (acond + eps);
      const double tt1 = test1 / (1 + anorm * xnorm / bnorm), rtol = btol + atol * anorm * xnorm / bnorm;
      int istop = 0;
      if (itn >= iter_lim) istop = 7;
      if (1 + test3 <= 1) istop = 6;
      if (1 + test2 <= 1) istop = 5;
      if (1 + tt1 <= 1) istop = 4;
      if (test3 <= ctol) istop = 3;
      if (test2 <= atol) istop = 2;
      if (test1 <= rtol) istop = 1;
      if (istop != 0) break;
    }
  }
  __syncthreads();
  for (int r = TID; r < nc; r += NT) l[r] = fmax(0.0, -x[r]);
  __syncthreads();
  PROF_END(PH_LSQR, pt_l);
}

// ------------------------------------------------------------------------------------------------
// quantities of one SQP linearisation needed by the merit function (DGSQP.py:949-979)
// ------------------------------------------------------------------------------------------------
struct LinScal {
  double phi, dphi;   // merit and its directional derivative at the base point (with the caller's mu)
  double S0, S1;      // sum(s), sum(ds) with s = min(0,g), ds = g + G du - s   (DGSQP.py:414-415)
  double dstat, vio;  // f_dstat_norm and sum(g - s)
};

// d = q + G^T l  (also the stationarity vector of the convergence test, DGSQP.py:368)
__device__ inline void dev_stat_vector(const Ctx& c, const double* lvec, double* dout) {
  const DgProb& D = *c.D;
  gt_mul(c, lvec, dout);
  for (int i = TID; i < D.n; i += NT) dout[i] += c.lds[D.L.q + i];
  __syncthreads();
}
// v = Qraw^T d
__device__ inline void dev_qt_mul(const Ctx& c) {
  const DgProb& D = *c.D;
  const double* Qg = c.ws + D.ws_q;
  const double* d = c.lds + D.L.d;
  __syncthreads();
  PROF_BEGIN(pt_q);
  for (int j = TID; j < D.n; j += NT) {
    double s = 0;
    for (int i = 0; i < D.n; i++) s += Qg[(int64_t)i * D.n + j] * d[i];
    c.lds[D.L.v + j] = s;
  }
  __syncthreads();
  PROF_END(PH_QTMUL, pt_q);
}
// after a QP solve at the current linearisation: everything phi / dphi / mu need
__device__ inline void dev_step_scalars(const Ctx& c, LinScal& S) {
  const DgProb& D = *c.D;
  const DgLds& L = D.L;
  double* lds = c.lds;
  const double *q = lds + L.q, *g = lds + L.g, *l = lds + L.l, *d = lds + L.d, *v = lds + L.v;
  const double *du = lds + L.o_du, *lhat = lds + L.o_lhat;
  double* t = lds + L.p_t;  // G^T lhat
  double* red = lds + L.red;
  PROF_BEGIN(pt_m);
  gt_mul(c, lhat, t);
  double a1 = 0, a2 = 0, lGdu = 0, dd = 0;
  for (int i = TID; i < D.n; i += NT) {
    a1 += v[i] * du[i];                    // d^T Q du  (raw Q, DGSQP.py:964)
    a2 += d[i] * (t[i] - (d[i] - q[i]));   // d^T G^T dl
    lGdu += (d[i] - q[i]) * du[i];         // l^T G du
    dd += d[i] * d[i];
  }
  double lg = 0, lhg = 0, vio = 0, s0 = 0, ssum = 0;
  for (int r = TID; r < D.nc; r += NT) {
    const double gr = g[r];
    lg += l[r] * gr;
    lhg += lhat[r] * gr;
    vio += fmax(gr, 0.0);                  // g - min(0,g)
    s0 += fmin(gr, 0.0);
    ssum += gr + g_row_dot(D, lds + L.gd, r, du);  // s + ds
  }
  a1 = block_sum(a1, red); a2 = block_sum(a2, red); lGdu = block_sum(lGdu, red); dd = block_sum(dd, red);
  lg = block_sum(lg, red); lhg = block_sum(lhg, red); vio = block_sum(vio, red); s0 = block_sum(s0, red); ssum = block_sum(ssum, red);
  S.dstat = a1 + a2 + lg * (lGdu + (lhg - lg));
  S.vio = vio;
  S.S0 = s0;
  S.S1 = ssum - s0;
  S.phi = 0.5 * (dd + lg * lg);  // + mu*vio added by the caller
  S.dphi = S.dstat;
  PROF_END(PH_MERIT, pt_m);
}
// merit of a trial point: current (q, g, G) in LDS belong to the trial u; multipliers l + alpha (lhat - l)
__device__ inline double dev_phi_trial(const Ctx& c, double alpha, double sum_s, double mu) {
  const DgProb& D = *c.D;
  const DgLds& L = D.L;
  double* lds = c.lds;
  double* lt = lds + L.s_x;
  double* dt = lds + L.s_t;
  double* red = lds + L.red;
  const double *l = lds + L.l, *lhat = lds + L.o_lhat, *g = lds + L.g;
  __syncthreads();
  PROF_BEGIN(pt_m);
  double lg = 0, sg = 0;
  for (int r = TID; r < D.nc; r += NT) {
    const double v = l[r] + alpha * (lhat[r] - l[r]);
    lt[r] = v;
    lg += v * g[r];
    sg += g[r];
  }
  dev_stat_vector(c, lt, dt);
  double dd = 0;
  for (int i = TID; i < D.n; i += NT) dd += dt[i] * dt[i];
  dd = block_sum(dd, red); lg = block_sum(lg, red); sg = block_sum(sg, red);
  double phi = 0.5 * (dd + lg * lg);
  if (D.par.merit_function == DGSQP_MERIT_STAT_L1) phi += mu * (sg - sum_s);
  PROF_END(PH_MERIT, pt_m);
  return phi;
}

// _line_search_3 (DGSQP.py:1057-1081) from the base (u, du, l, lhat) held in LDS.  On return u and l hold
// the LAST trial point; returns its merit.
__device__ inline double dev_line_search(const Ctx& c, double mu, double phi, double dphi, double S0, double S1) {
  const DgProb& D = *c.D;
  const DgLds& L = D.L;
  double* lds = c.lds;
  double alpha = 1.0, phit = 0.0;
  for (int i = 0; i < D.par.line_search_iters; i++) {
    dev_evaluate(c, lds + L.u, alpha, lds + L.o_du, false);
    phit = dev_phi_trial(c, alpha, S0 + alpha * S1, mu);
    dev_tr(c, 30, alpha); dev_tr(c, 31, phit);
    if (phit <= phi + D.par.beta * alpha * dphi) break;
    if (i + 1 < D.par.line_search_iters) alpha *= D.par.tau;
  }
  __syncthreads();
  for (int i = TID; i < D.n; i += NT) lds[L.u + i] += alpha * lds[L.o_du + i];
  for (int r = TID; r < D.nc; r += NT) lds[L.l + r] += alpha * (lds[L.o_lhat + r] - lds[L.l + r]);
  __syncthreads();
  return phit;
}

// full linearisation + QP at the current (u, l): _evaluate(hessian=True) followed by _solve_qp.
// Returns the QP flag (0 ok).  Leaves d, v, du, lhat and P in LDS.
__device__ inline int dev_linearize_and_qp(const Ctx& c, bool do_qp, double* cond3, double* Qpd) {
  const DgProb& D = *c.D;
  const DgLds& L = D.L;
  double* lds = c.lds;
  dev_evaluate(c, lds + L.u, 0.0, nullptr, true);
  dev_stat_vector(c, lds + L.l, lds + L.d);
  if (cond3) {  // convergence measures (DGSQP.py:376-378)
    double gm = -INFINITY, cm = 0, sm = 0;
    for (int r = TID; r < D.nc; r += NT) { gm = fmax(gm, lds[L.g + r]); cm = fmax(cm, fabs(lds[L.g + r] * lds[L.l + r])); }
    for (int i = TID; i < D.n; i += NT) sm = fmax(sm, fabs(lds[L.d + i]));
    cond3[0] = fmax(0.0, block_max(gm, lds + L.red));
    cond3[1] = block_max(cm, lds + L.red);
    cond3[2] = block_max(sm, lds + L.red);
  }
  if (!do_qp) return 0;
  dev_qt_mul(c);
  dev_psd_inverse(c, Qpd);
  return dev_qp(c);
}

__device__ inline void dev_save_base(const Ctx& c) {
  const DgProb& D = *c.D;
  const DgLds& L = D.L;
  double* b = c.ws + D.ws_base;
  for (int i = TID; i < D.n; i += NT) { b[i] = c.lds[L.u + i]; b[D.n + i] = c.lds[L.o_du + i]; }
  for (int r = TID; r < D.nc; r += NT) { b[2 * D.n + r] = c.lds[L.l + r]; b[2 * D.n + D.nc + r] = c.lds[L.o_lhat + r]; }
  __syncthreads();
}
__device__ inline void dev_restore_base(const Ctx& c) {
  const DgProb& D = *c.D;
  const DgLds& L = D.L;
  const double* b = c.ws + D.ws_base;
  __syncthreads();
  for (int i = TID; i < D.n; i += NT) { c.lds[L.u + i] = b[i]; c.lds[L.o_du + i] = b[D.n + i]; }
  for (int r = TID; r < D.nc; r += NT) { c.lds[L.l + r] = b[2 * D.n + r]; c.lds[L.o_lhat + r] = b[2 * D.n + D.nc + r]; }
  __syncthreads();
}
__device__ inline void dev_take_full_step(const Ctx& c) {  // u += du ; l = lhat
  const DgProb& D = *c.D;
  const DgLds& L = D.L;
  __syncthreads();
  for (int i = TID; i < D.n; i += NT) c.lds[L.u + i] += c.lds[L.o_du + i];
  for (int r = TID; r < D.nc; r += NT) c.lds[L.l + r] = c.lds[L.o_lhat + r];
  __syncthreads();
}

// _watchdog_line_search_4 (DGSQP.py:1174-1288; branch order of SURVEY.md A.7).  Base point and step
// (u_k, du_k, l_k, lhat_k) are in LDS on entry; returns the number of extra QP solves.
__device__ inline int dev_watchdog(const Ctx& c, double mu, const LinScal& Sk) {
  const DgProb& D = *c.D;
  const DgLds& L = D.L;
  double* lds = c.lds;
  const double beta = D.par.beta;
  const double phi_k = Sk.phi, dphi_k = Sk.dphi;
  int nqp = 0;
  dev_save_base(c);
  // relaxed (full) step
  dev_evaluate(c, lds + L.u, 1.0, lds + L.o_du, false);
  const double phi1 = dev_phi_trial(c, 1.0, Sk.S0 + Sk.S1, mu);
  dev_tr(c, 20, phi1);
  if (phi1 <= phi_k + beta * dphi_k) { dev_take_full_step(c); return 0; }
  dev_take_full_step(c);  // (u_t, l_t) = (u_k + du_k, l_k + dl_k)
  bool fail = false;
  LinScal S;
  double phi_n = 0;
  for (int t = 0; t < 5; t++) {
    const int flag = dev_linearize_and_qp(c, true, nullptr, nullptr);
    nqp++;
    if (flag != 0) { fail = true; break; }
    dev_step_scalars(c, S);
    dev_evaluate(c, lds + L.u, 1.0, lds + L.o_du, false);
    phi_n = dev_phi_trial(c, 1.0, S.S0 + S.S1, mu);
    dev_tr(c, 21, phi_n);
    if (phi_n > 1e6) break;                                   // merit_max; (u_t, l_t) not advanced
    if (phi_n <= phi_k + beta * dphi_k) { dev_take_full_step(c); return nqp; }
    dev_take_full_step(c);
  }
  // insist on merit decrease
  {
    const int flag = dev_linearize_and_qp(c, true, nullptr, nullptr);
    nqp++;
    if (flag != 0) fail = true;
    else {
      dev_step_scalars(c, S);
      const double phi_b = S.phi + (D.par.merit_function == DGSQP_MERIT_STAT_L1 ? mu * S.vio : 0.0);
      const double dphi_b = S.dstat - (D.par.merit_function == DGSQP_MERIT_STAT_L1 ? mu * S.vio : 0.0);
      phi_n = dev_line_search(c, mu, phi_b, dphi_b, S.S0, S.S1);
      dev_tr(c, 22, phi_n);
    }
  }
  if (!fail) {
    if (phi_n <= phi_k + beta * dphi_k) return nqp;
    else if (phi_n > phi_k) fail = true;
    else {
      const int flag = dev_linearize_and_qp(c, true, nullptr, nullptr);
      if (flag != 0) {
        dev_restore_base(c);
        dev_line_search(c, mu, phi_k, dphi_k, Sk.S0, Sk.S1);
        return nqp;
      }
      nqp++;
      dev_step_scalars(c, S);
      const double phi_b = S.phi + (D.par.merit_function == DGSQP_MERIT_STAT_L1 ? mu * S.vio : 0.0);
      const double dphi_b = S.dstat - (D.par.merit_function == DGSQP_MERIT_STAT_L1 ? mu * S.vio : 0.0);
      dev_line_search(c, mu, phi_b, dphi_b, S.S0, S.S1);
      return nqp;
    }
  }
  dev_restore_base(c);
  dev_line_search(c, mu, phi_k, dphi_k, Sk.S0, Sk.S1);
  return nqp;
}

// ------------------------------------------------------------------------------------------------
// DGSQP.solve() for one scenario (DGSQP.py:302-507)
// ------------------------------------------------------------------------------------------------
struct SolveOutPtrs {
  double *u, *l, *x, *cond, *cost;
  int32_t *status, *iters, *qp_solves;
};
__device__ inline void dev_solve(const Ctx& c, const double* u_ws, int64_t b, const SolveOutPtrs& O) {
  const DgProb& D = *c.D;
  const DgLds& L = D.L;
  double* lds = c.lds;
  const int n = D.n, nc = D.nc;
  __syncthreads();
  for (int i = TID; i < n; i += NT) lds[L.u + i] = u_ws[i];
  for (int r = TID; r < nc; r += NT) lds[L.l + r] = 0.0;
  __syncthreads();
  // dual warm start
  dev_evaluate(c, lds + L.u, 0.0, nullptr, false);
  dev_dual_init(c);
  int sqp_it = 0, rel_tol_its = 0, status = DGSQP_MAX_IT, total_qp = 0;
  double cond[3] = {0, 0, 0};
  const bool l1 = D.par.merit_function == DGSQP_MERIT_STAT_L1;
  while (true) {
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
    if (cond[2] > 1e5) { status = DGSQP_DIVERGED; break; }
    if (cond[0] < D.par.p_tol && cond[1] < D.par.d_tol && cond[2] < D.par.d_tol) { status = DGSQP_CONV_ABS_TOL; break; }
    dev_qt_mul(c);
    dev_psd_inverse(c, nullptr);
    const int flag = dev_qp(c);
    total_qp++;
    if (flag != 0) { status = DGSQP_QP_FAIL; break; }
    LinScal S;
    dev_step_scalars(c, S);
    // _get_mu (DGSQP.py:559-585)
    double mu = 0.0;
    if (l1 && S.vio > 0) mu = (S.dstat < 0 ? -S.dstat : S.dstat) / (0.5 * S.vio);
    if (l1) { S.phi += mu * S.vio; S.dphi = S.dstat - mu * S.vio; }
    {
      double d2 = 0;
      for (int i = TID; i < n; i += NT) d2 += lds[L.o_du + i] * lds[L.o_du + i];
      d2 = block_sum(d2, lds + L.red);
      dev_tr(c, 10, d2); dev_tr(c, 11, mu); dev_tr(c, 12, S.phi); dev_tr(c, 13, S.dphi);
    }
    dev_save_base(c);  // also u_im1 / l_im1 of the relative-tolerance test
    if (D.par.nonmono_ls) total_qp += dev_watchdog(c, mu, S);
    else dev_line_search(c, mu, S.phi, S.dphi, S.S0, S.S1);
    // relative-tolerance exit (DGSQP.py:454-462)
    double du2 = 0, dl2 = 0;
    const double* bk = c.ws + D.ws_base;
    for (int i = TID; i < n; i += NT) { const double t = lds[L.u + i] - bk[i]; du2 += t * t; }
    for (int r = TID; r < nc; r += NT) { const double t = lds[L.l + r] - bk[2 * n + r]; dl2 += t * t; }
    du2 = block_sum(du2, lds + L.red);
    dl2 = block_sum(dl2, lds + L.red);
    if (sqrt(du2) < D.par.p_tol / 2 && sqrt(dl2) < D.par.d_tol / 2) {
      rel_tol_its++;
      if (rel_tol_its >= D.par.rel_tol_req && cond[0] < D.par.p_tol) { status = DGSQP_CONV_REL_TOL; break; }
    } else rel_tol_its = 0;
    sqp_it++;
    if (sqp_it >= D.par.sqp_iters) { status = DGSQP_MAX_IT; break; }
  }
  // outputs: q_pred = evaluate_dynamics(u, x0) (DGSQP.py:476), cost = f_J (:492)
  __syncthreads();
  double* ue = lds + L.e_ue;
  for (int i = TID; i < n; i += NT) ue[i] = lds[L.u + i];
  dev_rollout(c, ue, lds + L.e_x);
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
}
