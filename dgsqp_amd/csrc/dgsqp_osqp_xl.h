// OSQP's arithmetic (dgsqp_osqp.h: the restated ADMM + polish of ca.conic('qp', 'osqp', {polish: True}), DGSQP.py:183-201, 246-249) for
// the XL layout: 128 < n <= 320 decision variables, BASELINE configs[2], [3], [4] (n = 150, 200, 300; up to 1,587 rows).
//
// Same algorithm, same algebra as dgsqp_osqp.h (Ruiz equilibration carried as D, E_I, E, c; the reduced ADMM system
// K xt = sigma x - qs + As' (rho z - y), K = Ps + sigma I + rho_I (E_I D)^2 + rho W, W = Gs' Gs; identity rows not stored; polish in unscaled
// variables in range-space form) -- what changes is where the data lives and how K is inverted:
//   * the n x n matrices (M, W, the factor J of K / Hu, K^-1, the polish's Y and Schur complement) sit in the workgroup's L2 scratch.  K is
//     factored by the blocked elimination of dgsqp_xl.h (xl_eliminate_blocked: 16 pivots per pass on the matrix cores) and K^-1 = J J' is
//     then formed EXPLICITLY, again on the matrix cores (ox_inverse_from_factor): an ADMM iteration streams one n x n matrix, not two;
//   * the packed constraint gradients are transposed once per QP into the scratch (ox_build_tables: per column the list of (gradient,
//     entry) pairs and the entries' values in that order): G' w is one pass of four lanes per column over streamed values, and W = G' E^2 G
//     is built column by column from the same table -- the per-QP index tables of the LDS path, which do not fit next to 1,587-row vectors;
//   * LDS holds x, y (in the QP's output slots), z, w and seven n-vectors; the row scaling E and delta y live in the scratch (coalesced row
//     loops); the slot of w later holds the polish's vectors and, with the dense-dot partials behind it, the elimination's multipliers;
//   * the polish builds Y = Ju' A_W' row by row (a box or rate row is a row of Ju), the Schur complement S = Y Y' + delta E^-2, factors it
//     with the same elimination and runs OSQP's three refinement steps against the unregularised residual; a polish with more active rows
//     than variables is reported unsuccessful.
// Checked QP by QP against the test infrastructure's C++ restatement of OSQP (osqp.hpp; tests/test_gpu.py::test_xl_device_osqp_matches_the_cpu_restatement, ::test_xl_osqp_sizes_
// between_the_configs: same status, ADMM iteration count, rho, polish verdict and active rows on 60 of 60 QPs).
#pragma once

struct OxPtrs {
  lptr x, y, z, w, Dv, EI, rhs, xt, tmp, dx, tv, part, dpart, yd2, ddx, red, scal, Mm;
  lds_i_t* alist;
  clptr q, g, gdL;
  cgptr gdG;
  gptr E, dy;               // n_c-vectors in the scratch
  cgptr M;                  // projected + regularised Hessian (dev_xl_psd), row-major n x n
  gptr W, J, Y, S;          // Gs' Gs; factor of K / Hu; polish: Y (one row of n per active row), Schur complement / its factor
};
__device__ inline OxPtrs ox_ptrs(const Ctx& c) {
  const DgProb& D = dg_prob;
  const DgLds& L = D.L;
  lptr lds = LP(0);
  OxPtrs o;
  o.x = lds + L.o_du; o.y = lds + L.o_lhat; o.z = lds + L.ox_z; o.w = lds + L.ox_w; o.Mm = lds + L.ox_w;
  o.dpart = lds + L.ox_dpart; o.yd2 = lds + L.ox_yd2; o.part = lds + L.ox_part;
  o.Dv = lds + L.ox_nv; o.EI = o.Dv + L.ox_np; o.rhs = o.EI + L.ox_np; o.xt = o.rhs + L.ox_np; o.tmp = o.xt + L.ox_np; o.dx = o.tmp + L.ox_np; o.tv = o.dx + L.ox_np;
  o.alist = (lds_i_t*)(lds + L.ox_alist);
  o.ddx = lds + L.yd; o.red = lds + L.red; o.scal = lds + L.scal;
  o.q = lds + L.q; o.g = lds + L.g; o.gdL = lds + L.gd; o.gdG = c.ws + D.ws_gd;
  o.E = c.ws + D.wsx_E; o.dy = c.ws + D.wsx_dy;
  o.M = c.ws + D.ws_R; o.W = c.ws + D.ws_V; o.J = c.ws + D.ws_P; o.Y = c.ws + D.wsx_Y; o.S = c.ws + D.wsx_S;
  return o;
}

// out_i = sum_j M_ij v_j  (ABSMAX: max_j |M_ij| v_j) for the symmetric matrix in the scratch; thread (g, i): column i of the g-th part of the rows
template <bool ABSMAX>
__device__ inline void ox_m_pass(cgptr M, int n, clptr v, lptr part, lptr out) {
  const XlSplit S = xl_split(n);
  __syncthreads();
  if (S.g < S.G && S.i < n) {
    const int ja = (S.g * n) / S.G, jb = ((S.g + 1) * n) / S.G;
    double a[4] = {0, 0, 0, 0};
    int j = ja;
    for (; j + 7 < jb; j += 8) {
      double m[8], t[8];
#pragma unroll
      for (int k = 0; k < 8; k++) { m[k] = M[(int64_t)(j + k) * n + S.i]; t[k] = v[j + k]; }
#pragma unroll
      for (int k = 0; k < 8; k++) a[k & 3] = ABSMAX ? fmax(a[k & 3], __builtin_fabs(m[k]) * t[k]) : __builtin_fma(m[k], t[k], a[k & 3]);
    }
    for (; j < jb; j++) { const double m = M[(int64_t)j * n + S.i]; a[0] = ABSMAX ? fmax(a[0], __builtin_fabs(m) * v[j]) : __builtin_fma(m, v[j], a[0]); }
    part[S.g * n + S.i] = ABSMAX ? fmax(fmax(a[0], a[1]), fmax(a[2], a[3])) : (a[0] + a[1]) + (a[2] + a[3]);
  }
  __syncthreads();
  if (TID < n) {
    double s = part[TID];
    for (int g = 1; g < S.G; g++) s = ABSMAX ? fmax(s, part[g * n + TID]) : s + part[g * n + TID];
    out[TID] = s;
  }
  __syncthreads();
}

template <class GP>
__device__ inline void ox_dense_absmax(const DgProb& D, GP gd, clptr Dv, lptr part, lptr out) {
  __syncthreads();
  for (int t = TID; t < D.ntask; t += NT) {
    const DgTask T = ld_task(t);
    const GP p = gd + T.p0;
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
// out[col] = max_r E_r |G_r,col|
template <class GP>
__device__ inline void ox_gt_absmax(const DgProb& D, GP gd, cgptr E, lptr yd, lptr out) {
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
// W = Gs' Gs = D G' E^2 G D into the scratch, column by column through the transposed table (ox_build_tables, below): column j's
// (gradient, value) pairs are scattered into an LDS vector indexed by gradient -- weighted with the E^2 of the rows sharing the gradient --,
// then every row i >= j gathers its own pairs against it (four lanes per row, independent loads).  Box rows add to the diagonal, rate
// rows to the diagonal and the (t, t - 1) entries.  (The first version visited every gradient of the later stages per ENTRY through the
// constant-memory table: 47 Mcycles per QP at n = 300, a tenth of the QP; this one takes 2.)
struct OxTabs;
__device__ inline OxTabs ox_tabs(const Ctx& c);
template <class GP>
__device__ __noinline__ void ox_build_w(const Ctx& c, const OxPtrs& o, GP gd);

// out_r = E_r (G (D v))_r for every G row; leaves D v in o.tmp and its dense dots in o.ddx
template <class GP>
__device__ inline void ox_gs_mul(const OxPtrs& o, GP gd, clptr v, lptr out) {
  const DgProb& D = dg_prob;
  __syncthreads();
  for (int j = TID; j < D.n; j += NT) o.tmp[j] = o.Dv[j] * v[j];
  __syncthreads();
  qp_dense_dots<GP>(D, gd, o.tmp, o.dpart, o.ddx);
  for (int r = TID; r < D.nc; r += NT) out[r] = o.E[r] * qpw_row_dot(D, ld_row(r), o.tmp, o.ddx);
  __syncthreads();
}
// ---- G' w through a transposed index table.  The generic gt_mul looks every dense gradient up per column (837 table entries per column
// at n = 300, most of them not covering it: 0.3 Mcycles per product, half of an ADMM iteration).  Per QP the covering (gradient, entry)
// pairs of every column are listed once in the workgroup's scratch -- the packed gradients transposed by index, as the LDS path does in
// LDS (osqp_build_tables) -- and a product is one pass of four lanes per column over independent loads.
typedef __attribute__((address_space(1))) unsigned int glb_u32;
struct OxTabs { const glb_u32* cstart; const glb_u32* pairT; cgptr gdT; };      // cstart[n + 1]; pairT[k] = (gradient << 22) | offset inside the packed gradients;
                                                                                // gdT[k]: that entry's VALUE -- the packed gradients transposed, streamed instead of gathered
__device__ inline OxTabs ox_tabs(const Ctx& c) {
  const DgProb& D = dg_prob;
  OxTabs T;
  T.cstart = (const glb_u32*)(c.ws + D.wsx_tab);
  T.pairT = T.cstart + ((D.n + 2) & ~1);
  T.gdT = c.ws + D.wsx_gdT;
  return T;
}
// The same values in WAVE-INTERLEAVED order, for the ADMM iteration (ox_iterate_block).  The columns are taken in the order DgProb.ox_perm
// (sorted by length, longest first: neighbours in that order have nearly equal lengths), sixteen to a group; task it4 = 4 s + quarter of
// sorted column s is lane it4 & 63 of group it4 >> 6; entry m of a lane -- the column's entry quarter + 4 m -- sits at
// ox_gstart[group] + 64 m + lane, the group padded with zeros to its longest quarter column rounded up to eight entries: a wavefront
// load reads ONE 512-byte run where the column-major table gave it 16 cache lines, eight such loads are in flight per lane.  The
// multiplier of entry e of column (a, t, j) is element (ybase[a] + cnt_a - cnt_col + e) of the per-agent lists ylist (agent a's covering
// gradients in table order: a column covers the LAST cnt_col of them) -- gathered once per iteration into LDS, so the loop reads values
// only, no index.
struct OxTabI { const glb_u32* ybase; const glb_u32* ylist; cgptr gdI; };
__device__ inline OxTabI ox_tabi(const Ctx& c) {
  const DgProb& D = dg_prob;
  OxTabI T;
  T.ybase = (const glb_u32*)(c.ws + D.wsx_tabI);
  T.ylist = T.ybase + (DGSQP_MAX_AGENTS + 1);
  T.gdI = c.ws + D.wsx_gdI;
  return T;
}
template <class GP>
__device__ __noinline__ void ox_build_tables(const Ctx& c, const OxPtrs& o, GP gd) {
  const DgProb& D = dg_prob;
  const int n = D.n;
  glb_u32* cs = (glb_u32*)(c.ws + D.wsx_tab);
  glb_u32* pt = cs + ((n + 2) & ~1);
  __syncthreads();
  for (int col = TID; col < n; col += NT) {
    const int a = col / (D.N * DGSQP_NUA), rem = col % (D.N * DGSQP_NUA), t = rem / DGSQP_NUA;
    int cnt = 0;
    for (int d = D.stage_dense0[t + 1]; d < D.ndense; d++) { const DgDense dd = ld_dense(d); cnt += (dd.a == a) || (dd.kind == 1 && dd.b == a); }
    o.tmp[col] = (double)cnt;
  }
  __syncthreads();
  if (TID == 0) { double sacc = 0; for (int col = 0; col < n; col++) { const double cn = o.tmp[col]; o.tmp[col] = sacc; sacc += cn; } o.tmp[n] = sacc; }     // (n + 1 entries: tmp is followed by dx, dead here)
  __syncthreads();
  for (int col = TID; col <= n; col += NT) cs[col] = (unsigned int)o.tmp[col];
  for (int col = TID; col < n; col += NT) {
    const int a = col / (D.N * DGSQP_NUA), rem = col % (D.N * DGSQP_NUA), t = rem / DGSQP_NUA, j = rem % DGSQP_NUA;
    int k = (int)o.tmp[col];
    for (int d = D.stage_dense0[t + 1]; d < D.ndense; d++) {
      const DgDense dd = ld_dense(d);
      if (dd.a == a) pt[k++] = ((unsigned int)d << 22) | (unsigned int)(dd.off + t * DGSQP_NUA + j);
      else if (dd.kind == 1 && dd.b == a) pt[k++] = ((unsigned int)d << 22) | (unsigned int)(dd.off + 2 * dd.k + t * DGSQP_NUA + j);
    }
  }
  XSYNC();
  gptr gT = c.ws + D.wsx_gdT;
  const int tot = (int)o.tmp[n];
  for (int k = TID; k < tot; k += NT) gT[k] = gd[pt[k] & 0x3fffffu];
  XSYNC();
  // ---- the wave-interleaved copy (OxTabI): read by ox_iterate_block<GP, true>, n > 176
  if (n <= 176) return;
  const int NG = (4 * n + 63) / 64;
  glb_u32* yb = (glb_u32*)(c.ws + D.wsx_tabI);
  glb_u32* yl = yb + (DGSQP_MAX_AGENTS + 1);
  if (TID == 64) {        // per-agent lists: the gradients covering agent a, in table (= increasing gradient) order
    unsigned int acc = 0;
    for (int a = 0; a < D.M; a++) {
      yb[a] = acc;
      for (int d = 0; d < D.ndense; d++) { const DgDense dd = ld_dense(d); if (dd.a == a || (dd.kind == 1 && dd.b == a)) yl[acc++] = (unsigned int)d; }
    }
    for (int a = D.M; a <= DGSQP_MAX_AGENTS; a++) yb[a] = acc;
  }
  gptr gI = c.ws + D.wsx_gdI;
  for (int G = 0; G < NG; G++) {
    const int base = D.ox_gstart[G], len = (D.ox_gstart[G + 1] - base) >> 6;
    for (int e = TID; e < 64 * len; e += NT) {
      const int m = e >> 6, it4 = 64 * G + (e & 63), sc = it4 >> 2, idx = (it4 & 3) + 4 * m;
      double v = 0.0;
      if (sc < n) { const int col = D.ox_perm[sc], k0 = (int)cs[col], cn = (int)cs[col + 1] - k0; if (idx < cn) v = gT[k0 + idx]; }
      gI[base + e] = v;
    }
  }
  XSYNC();
}
// out = G' w  (w: an n_c-vector in LDS that already carries the row scaling)
template <class GP>
__device__ inline void ox_gt_mul(const Ctx& c, const OxPtrs& o, GP gd, clptr w, lptr out) {
  const DgProb& D = dg_prob;
  const int n = D.n;
  const OxTabs T = ox_tabs(c);
  lptr yd = o.ddx;
  __syncthreads();
  for (int d = TID; d < D.ndense; d += NT) {
    const DgDense dd = ld_dense(d);
    yd[d] = (dd.r_pos >= 0 ? w[dd.r_pos] : 0.0) - (dd.r_neg >= 0 ? w[dd.r_neg] : 0.0);
  }
  __syncthreads();
  for (int it4 = TID; it4 < 4 * n; it4 += NT) {
    const int col = it4 >> 2, part = it4 & 3;
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    if (part == 0) {
      const int a = col / (D.N * DGSQP_NUA), rem = col % (D.N * DGSQP_NUA), t = rem / DGSQP_NUA, j = rem % DGSQP_NUA;
      int r;
      if ((r = D.r_in_ub[a][t][j]) >= 0) s0 += w[r];
      if ((r = D.r_in_lb[a][t][j]) >= 0) s0 -= w[r];
      if ((r = D.r_rate_ub[a][t][j]) >= 0) s1 += w[r];
      if ((r = D.r_rate_lb[a][t][j]) >= 0) s1 -= w[r];
      if (t + 1 < D.N) {
        if ((r = D.r_rate_ub[a][t + 1][j]) >= 0) s2 -= w[r];
        if ((r = D.r_rate_lb[a][t + 1][j]) >= 0) s2 += w[r];
      }
    }
    const int k1 = (int)T.cstart[col + 1];
    int k = (int)T.cstart[col] + part;
    for (; k + 12 < k1; k += 16) {
      const unsigned int pa = T.pairT[k], pb = T.pairT[k + 4], pc = T.pairT[k + 8], pd = T.pairT[k + 12];
      const double ga = T.gdT[k], gb = T.gdT[k + 4], gc = T.gdT[k + 8], gg = T.gdT[k + 12];
      s0 = __builtin_fma(yd[pa >> 22], ga, s0); s1 = __builtin_fma(yd[pb >> 22], gb, s1); s2 = __builtin_fma(yd[pc >> 22], gc, s2); s3 = __builtin_fma(yd[pd >> 22], gg, s3);
    }
    for (; k < k1; k += 4) { const unsigned int pa = T.pairT[k]; s0 = __builtin_fma(yd[pa >> 22], T.gdT[k], s0); }
    double sm = (s0 + s1) + (s2 + s3);
    sm += dpp_f64<0xB1>(sm);
    sm += dpp_f64<0x4E>(sm);
    if (part == 0) out[col] = sm;
  }
  __syncthreads();
}
template <class GP>
__device__ __noinline__ void ox_build_w(const Ctx& c, const OxPtrs& o, GP gd) {
  const DgProb& D = dg_prob;
  const int n = D.n, nd = D.ndense;
  const OxTabs T = ox_tabs(c);
  lptr colj = o.w, wt = o.yd2;          // (w's slot holds n_c >= ndense doubles; the ADMM has not started)
  __syncthreads();
  for (int d = TID; d < nd; d += NT) {
    const DgDense dd = ld_dense(d);
    const double ep = dd.r_pos >= 0 ? o.E[dd.r_pos] : 0.0, en = dd.r_neg >= 0 ? o.E[dd.r_neg] : 0.0;
    wt[d] = ep * ep + en * en;
  }
  __syncthreads();
  auto e2 = [&](int r) { const double ev = r >= 0 ? o.E[r] : 0.0; return ev * ev; };
  for (int j = 0; j < n; j++) {
    for (int d = TID; d < nd; d += NT) colj[d] = 0.0;
    __syncthreads();
    for (int k = (int)T.cstart[j] + TID; k < (int)T.cstart[j + 1]; k += NT) { const unsigned int d = T.pairT[k] >> 22; colj[d] = wt[d] * T.gdT[k]; }
    __syncthreads();
    const int aj = j / (D.N * DGSQP_NUA), rj = j % (D.N * DGSQP_NUA), tj = rj / DGSQP_NUA, jj = rj % DGSQP_NUA;
    const double dj = o.Dv[j];
    for (int it4 = TID; it4 < 4 * (n - j); it4 += NT) {
      const int i = j + (it4 >> 2), part = it4 & 3;
      const int k1 = (int)T.cstart[i + 1];
      int k = (int)T.cstart[i] + part;
      double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
      for (; k + 12 < k1; k += 16) {
        const unsigned int pa = T.pairT[k], pb = T.pairT[k + 4], pc = T.pairT[k + 8], pd = T.pairT[k + 12];
        const double ga = T.gdT[k], gb = T.gdT[k + 4], gc = T.gdT[k + 8], gg = T.gdT[k + 12];
        s0 = __builtin_fma(ga, colj[pa >> 22], s0); s1 = __builtin_fma(gb, colj[pb >> 22], s1); s2 = __builtin_fma(gc, colj[pc >> 22], s2); s3 = __builtin_fma(gg, colj[pd >> 22], s3);
      }
      for (; k < k1; k += 4) s0 = __builtin_fma(T.gdT[k], colj[T.pairT[k] >> 22], s0);
      double sm = (s0 + s1) + (s2 + s3);
      sm += dpp_f64<0xB1>(sm);
      sm += dpp_f64<0x4E>(sm);
      if (part == 0) {
        const int ai = i / (D.N * DGSQP_NUA), ri = i % (D.N * DGSQP_NUA), ti = ri / DGSQP_NUA, ji = ri % DGSQP_NUA;
        if (ai == aj && ji == jj) {
          if (ti == tj) {
            sm += e2(D.r_in_ub[ai][ti][ji]) + e2(D.r_in_lb[ai][ti][ji]) + e2(D.r_rate_ub[ai][ti][ji]) + e2(D.r_rate_lb[ai][ti][ji]);
            if (ti + 1 < D.N) sm += e2(D.r_rate_ub[ai][ti + 1][ji]) + e2(D.r_rate_lb[ai][ti + 1][ji]);
          } else if (ti - tj == 1) {
            sm -= e2(D.r_rate_ub[ai][ti][ji]) + e2(D.r_rate_lb[ai][ti][ji]);
          }
        }
        sm *= o.Dv[i] * dj;
        o.W[(int64_t)i * n + j] = sm;
        o.W[(int64_t)j * n + i] = sm;
      }
    }
    __syncthreads();
  }
  XSYNC();
}
// out = D G' (E w): E w goes to the LDS vector sc (may be w itself), then the transposed product
template <class GP>
__device__ inline void ox_gst_mul(const Ctx& c, const OxPtrs& o, GP gd, clptr w, lptr sc, lptr out) {
  const DgProb& D = dg_prob;
  __syncthreads();
  for (int r = TID; r < D.nc; r += NT) sc[r] = o.E[r] * w[r];
  ox_gt_mul<GP>(c, o, gd, sc, out);
  for (int j = TID; j < D.n; j += NT) out[j] *= o.Dv[j];
  __syncthreads();
}

// J = L^-T of the SPD matrix whose lower triangle `fill(i, k)` (k <= i) describes: A^-1 = J J'.  J: m x m, row stride m, in the scratch;
// LDS: Mm (16 m), tab (>= 512), tv (>= m).  Returns false when a pivot is not positive (A is not numerically SPD).
template <class F>
__device__ inline bool ox_factor(gptr J, int m, lptr Mm, lptr tab, lptr tv, F&& fill) {
  __syncthreads();
  for (int e = TID; e < m * m; e += NT) { const int i = e / m, k = e - i * m; J[e] = k <= i ? fill(i, k) : 0.0; }
  XSYNC();
  const bool bad = xl_eliminate_blocked(J, m, m, Mm, tab, tv);
  __syncthreads();
  if (bad) return false;
  int nf = 0;
  for (int j = TID; j < m; j += NT) { const double dj = J[(int64_t)j * m + j]; nf |= !(dj > 0.0 && dj < 1e300); tv[j] = 1.0 / sqrt(dj); }
  if (__syncthreads_or(nf)) return false;
  const int tr = TID >> 4, tc = TID & 15;
  for (int i = tr; i < m; i += NT / 16) {
    const double ri = tv[i];
    gptr Ji = J + (int64_t)i * m;
    for (int k = tc; k < i; k += 16) { J[(int64_t)k * m + i] = Ji[k] * ri; Ji[k] = 0.0; }
  }
  for (int j = TID; j < m; j += NT) J[(int64_t)j * m + j] = tv[j];
  XSYNC();
  return true;
}
// out = J (J' v)  (= A^-1 v); t: an n-vector of LDS scratch (may not alias v or out)
__device__ inline void ox_solve(cgptr J, int m, clptr v, lptr t, lptr out, lptr part) {
  const XlSplit S = xl_split(m);
  xl_jt_mul<cgptr>(J, m, m, S, 0, m, v, t, part);
  xl_j_mul<cgptr>(J, m, m, S, 0, m, t, out, part);
}

// A^-1 = J J' (full, symmetric, row-major m x m) from the factor J = L^-T, on the matrix cores: tile (ti, tj <= ti) of the result is the
// sum over 16-column chunks k >= 16 ti of J[i][k] J[j][k] (J is upper triangular: earlier chunks are zero and skipped), mirrored on
// store.  An ADMM iteration then needs ONE pass over an n x n matrix (ox_m_pass) instead of two over J -- the product is bound by the
// bytes it streams from L2 / Infinity Cache (two thirds of an iteration at n = 300 otherwise).
template <class OT>
__device__ inline void ox_inverse_from_factor(cgptr J, OT* Ainv, int m) {
  const int lane = TID & 63, wave = TID >> 6, li = lane & 15, h = lane >> 4;
  const int T = (m + 15) >> 4, ntile = T * (T + 1) / 2;
  __syncthreads();
  for (int t = wave; t < ntile; t += NT / 64) {
    int ti = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
    while (ti * (ti + 1) / 2 > t) ti--;
    while ((ti + 1) * (ti + 2) / 2 <= t) ti++;
    const int tj = t - ti * (ti + 1) / 2;
    const int rowa = 16 * ti + li, rowb = 16 * tj + li;       // A operand: row of the result tile; B operand: its column
    xl_v4d acc = {0.0, 0.0, 0.0, 0.0};
    for (int kc = ti; kc < T; kc += 2) {                       // two chunks per pass: eight loads per operand in flight
      double av[8], bv[8];
#pragma unroll
      for (int u = 0; u < 2; u++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int kk = 16 * (kc + u) + 4 * q + h;
          const bool on = kc + u < T && kk < m;
          av[4 * u + q] = (on && rowa < m) ? J[(int64_t)rowa * m + kk] : 0.0;
          bv[4 * u + q] = (on && rowb < m) ? J[(int64_t)rowb * m + kk] : 0.0;
        }
      }
#pragma unroll
      for (int q = 0; q < 8; q++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], bv[q], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = 16 * ti + h + 4 * r, col = 16 * tj + li;
      if (row < m && col < m) { Ainv[(int64_t)row * m + col] = (OT)acc[r]; if (ti != tj) Ainv[(int64_t)col * m + row] = (OT)acc[r]; }
    }
  }
  XSYNC();
}

__device__ inline double ox_rho_I(const OxPtrs& o, int j, double rho) { return o.EI[j] * OSQP_INFTY > OSQP_INFTY * OSQP_MIN_SCALING ? OSQP_RHO_MIN : rho; }

// out_i = sum_j Kinv_ij v_j for the symmetric matrix held in fp32 (dgsqp_params_t.mixed_precision): fp64 accumulation, half the bytes
typedef __attribute__((address_space(1))) float glb_f;
__device__ inline void ox_m_pass_f32(const glb_f* M, int n, clptr v, lptr part, lptr out) {
  const XlSplit S = xl_split(n);
  __syncthreads();
  if (S.g < S.G && S.i < n) {
    const int ja = (S.g * n) / S.G, jb = ((S.g + 1) * n) / S.G;
    double a[4] = {0, 0, 0, 0};
    int j = ja;
    for (; j + 15 < jb; j += 16) {        // sixteen 4-byte loads in flight per lane = the bytes in flight of the fp64 pass
      float m[16];
#pragma unroll
      for (int k = 0; k < 16; k++) m[k] = M[(int64_t)(j + k) * n + S.i];
#pragma unroll
      for (int k = 0; k < 16; k++) a[k & 3] = __builtin_fma((double)m[k], v[j + k], a[k & 3]);
    }
    for (; j < jb; j++) a[0] = __builtin_fma((double)M[(int64_t)j * n + S.i], v[j], a[0]);
    part[S.g * n + S.i] = (a[0] + a[1]) + (a[2] + a[3]);
  }
  __syncthreads();
  if (TID < n) {
    double s = part[TID];
    for (int g = 1; g < S.G; g++) s += part[g * n + TID];
    out[TID] = s;
  }
  __syncthreads();
}

// Setup: finite-data check, Ruiz equilibration (10 passes), W.  Returns c, or NaN for non-finite data.
template <class GP>
__device__ __noinline__ double ox_setup(const Ctx& c, GP gd) {
  const DgProb& D = dg_prob;
  const int n = D.n, nc = D.nc;
  const OxPtrs o = ox_ptrs(c);
  {
    int bad = 0;
    for (int e = TID; e < n * n; e += NT) bad |= !(__builtin_fabs(o.M[e]) < INFINITY);
    for (int j = TID; j < n; j += NT) bad |= !(__builtin_fabs(o.q[j]) < INFINITY);
    for (int r = TID; r < nc; r += NT) bad |= (o.g[r] != o.g[r]);
    for (int p = TID; p < D.ngd; p += NT) bad |= !(__builtin_fabs(gd[p]) < INFINITY);
    if (__syncthreads_or(bad)) return __builtin_nan("");
  }
  PROF_BEGIN(po1);
  for (int j = TID; j < n; j += NT) { o.Dv[j] = 1.0; o.EI[j] = 1.0; }
  for (int r = TID; r < nc; r += NT) o.E[r] = 1.0;
  XSYNC();
  double cc = 1.0;
  for (int it = 0; it < 10; it++) {
    ox_m_pass<true>(o.M, n, o.Dv, o.part, o.tmp);
    ox_dense_absmax<GP>(D, gd, o.Dv, o.dpart, o.ddx);
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
    ox_gt_absmax<GP>(D, gd, o.E, o.yd2, o.xt);
    for (int j = TID; j < n; j += NT) {
      const double dj = o.Dv[j], aI = o.EI[j] * dj;
      const double dn = fmax(cc * dj * o.tmp[j], fmax(aI, dj * o.xt[j]));
      o.Dv[j] = dj * (1.0 / sqrt(osqp_limit(dn)));
      o.EI[j] *= 1.0 / sqrt(osqp_limit(aI));
    }
    for (int r = TID; r < nc; r += NT) o.E[r] *= o.w[r];
    XSYNC();
    ox_m_pass<true>(o.M, n, o.Dv, o.part, o.tmp);
    double cm = 0, qn = 0;
    for (int j = TID; j < n; j += NT) { cm += cc * o.Dv[j] * o.tmp[j]; qn = fmax(qn, __builtin_fabs(cc * o.Dv[j] * o.q[j])); }
    cm = block_sum(cm, o.red);
    qn = block_max(qn, o.red);
    const double ct = osqp_limit(cm / n);
    qn = qn < OSQP_MIN_SCALING ? 1.0 : fmin(qn, OSQP_MAX_SCALING);
    cc *= 1.0 / fmax(ct, qn);
  }
  PROF_END(PH_O_SCALE, po1);
  ox_build_tables<GP>(c, o, gd);
  PROF_BEGIN(po2);
  ox_build_w<GP>(c, o, gd);
  PROF_END(PH_O_W, po2);
  return cc;
}

// factor of K(rho) = Ps + sigma I + rho_I (E_I D)^2 + rho W into o.J
__device__ __noinline__ bool ox_build_k(const Ctx& c, double rho, double cc) {
  const DgProb& D = dg_prob;
  const int n = D.n;
  const OxPtrs o = ox_ptrs(c);
  __syncthreads();
  for (int j = TID; j < n; j += NT) { const double aI = o.EI[j] * o.Dv[j]; o.tmp[j] = 1e-6 + ox_rho_I(o, j, rho) * aI * aI; }
  __syncthreads();
  PROF_BEGIN(po3);
  const bool ok = ox_factor(o.J, n, o.Mm, o.part, o.tv, [&](int i, int k) {
    double a = cc * (o.Dv[i] * o.Dv[k]) * o.M[(int64_t)i * n + k];
    a = __builtin_fma(rho, o.W[(int64_t)i * n + k], a);
    if (i == k) a += o.tmp[i];
    return a;
  });
  if (ok) {        // K^-1 itself, in the slot of the polish's Schur complement (unused until then)
    // dgsqp_params_t.mixed_precision: fp32 storage where the QP's Hessian is regularised.  ADMM run with a rounded inverse converges to the
    // QP whose Hessian is off by K (K^-1 - fl32(K^-1)) K: below the 1e-3 of the curve and circuit games' reg (same solutions on 20 of 20
    // solves converged in both, tests/test_gpu.py::test_xl_osqp_mixed_precision_against_fp64), but at reg = 0 the flat directions of P
    // (eigenvalues 1e-10 after _nearestPD) drown in it: the six-car merge converged on 42 % instead of 88 % of 256 scenarios, with 53 instead
    // of 36 QPs per solve.  Hence fp64 whenever reg < 1e-4 (v2 decays reg along a solve: decided per factorisation).
    const bool f32 = D.par.mixed_precision && dev_reg() >= 1e-4;
    if (D.par.mixed_precision) {
      if (TID == 0) o.scal[DG_OSQP_F32] = f32 ? 1.0 : 0.0;
      __syncthreads();
    }
    if (f32) ox_inverse_from_factor<glb_f>(o.J, (glb_f*)o.S, n);
    else ox_inverse_from_factor<glb_d>(o.J, o.S, n);
  }
  PROF_END(PH_O_KINV, po3);
  return ok;
}

// `count` ADMM iterations (Algorithm 1, alpha = 1.6): x, z, y, w = E (rho z - y) in LDS; leaves delta x (LDS) and delta y (scratch) of the
// last one.  The iterations between two termination checks run inside ONE call: every thread keeps the first OX_RC values of its slice of
// K^-1 (column TID of its row group: n / G values, G = 3, 2, 1 at n = 150, 200, 300) in registers across them -- the iteration is bound by
// the bytes it streams, and K^-1 is most of them: all but 2 of 50 values at n = 150, half of them at n = 200, a sixth at n = 300.  Same
// products in the same order as ox_m_pass / ox_m_pass_f32 (the register prefix covers whole groups of the unrolled loop): bit-identical.
#define OX_RC 48
// IL: G' w from the wave-interleaved table (n > 176: BASELINE configs[3], [4]) or, IL = false, from the column-major one (the packed-matrix
// layouts up to n = 176: there all but two values of each thread's K^-1 slice sit in registers, the phase is short, and the interleaved
// loop's extra live registers cost the K^-1 product more than the table gains -- profiles/r06_xl_admm_iteration.txt).  Two instantiations,
// two register allocations.
template <class GP, bool IL>
__device__ __noinline__ void ox_iterate_block(const Ctx& c, GP gd, double rho, double cc, int count) {
  const DgProb& D = dg_prob;
  const int n = D.n, nc = D.nc;
  const OxPtrs o = ox_ptrs(c);
  const double sigma = 1e-6, alpha = 1.6, irho = 1.0 / rho;
  const bool f32 = D.par.mixed_precision && o.scal[DG_OSQP_F32] != 0.0;
  const XlSplit S = xl_split(n);
  const bool act = S.g < S.G && S.i < n;
  const int ja = act ? (S.g * n) / S.G : 0, jb = act ? ((S.g + 1) * n) / S.G : 0;
  const bool regs = n / S.G >= OX_RC;                      // (block-uniform: every row group then holds at least OX_RC rows)
  double kc[OX_RC];
#pragma unroll
  for (int k = 0; k < OX_RC; k++) {
    const bool valid = regs && act;                        // ja + k < jb by the line above
    const int64_t idx = valid ? (int64_t)(ja + k) * n + S.i : 0;
    kc[k] = valid ? (f32 ? (double)((const glb_f*)o.S)[idx] : o.S[idx]) : 0.0;
  }
  // ... and the six box / rate rows of the columns it sums in G' w (a table walk through DgProb per column and iteration otherwise)
  constexpr int GR = (4 * DG_NVARMAX + NT - 1) / NT;      // rounds of (column, quarter) tasks
  const OxTabs T = ox_tabs(c);
  int rr[GR][6];
#pragma unroll
  for (int g = 0; g < GR; g++) {
    const int it4 = TID + g * NT;
    const bool lead = it4 < 4 * n && (it4 & 3) == 0;
    const int col = lead ? (IL ? (int)D.ox_perm[it4 >> 2] : it4 >> 2) : 0;           // (IL: tasks run over the columns in the wave-interleaved table's order, sorted by length)
    const int a = lead ? col / (D.N * DGSQP_NUA) : 0, rem = lead ? col % (D.N * DGSQP_NUA) : 0, t = rem / DGSQP_NUA, j = rem % DGSQP_NUA;
    const bool nxt = lead && t + 1 < D.N;
    rr[g][0] = lead ? D.r_in_ub[a][t][j] : -1; rr[g][1] = lead ? D.r_in_lb[a][t][j] : -1;
    rr[g][2] = lead ? D.r_rate_ub[a][t][j] : -1; rr[g][3] = lead ? D.r_rate_lb[a][t][j] : -1;
    rr[g][4] = nxt ? D.r_rate_ub[a][nxt ? t + 1 : t][j] : -1; rr[g][5] = nxt ? D.r_rate_lb[a][nxt ? t + 1 : t][j] : -1;
  }
  // ... and, per (column, quarter) task, where its run of the wave-interleaved table starts, how long the group's runs are, how many
  // entries are its own and where its multipliers start in the per-agent lists; per thread, the gradients whose multipliers it gathers
  const OxTabI TI = ox_tabi(c);
  const int NGI = (4 * n + 63) / 64;
  int gbase[GR], glen[GR], gcnt[GR], gyo[GR], gcol[GR];
#pragma unroll
  for (int g = 0; g < GR; g++) {
    const int it4 = TID + g * NT, G = it4 >> 6, sc = it4 >> 2, part = it4 & 3;
    const bool in = IL && G < NGI, has = IL && sc < n;
    const int col = has ? D.ox_perm[sc] : 0;
    gcol[g] = has && part == 0 ? col : -1;
    gbase[g] = in ? D.ox_gstart[G] : 0;
    glen[g] = in ? (D.ox_gstart[G + 1] - gbase[g]) >> 6 : 0;
    const int cn = has ? (int)(T.cstart[col + 1] - T.cstart[col]) : 0;
    gcnt[g] = cn > part ? (cn - part + 3) >> 2 : 0;
    const int a = has ? col / (D.N * DGSQP_NUA) : 0;
    gyo[g] = has ? (int)TI.ybase[a + 1] - cn + part : 0;      // ybase[a] + cnt_a - cnt_col + quarter
  }
  // (measured, profiles/r06_xl_admm_iteration.txt: with the group starts as scalar loads inside the loop and the list re-read per iteration
  // the phase is 36 kcycles at n = 300 instead of 22 k -- these few registers are worth it)
  constexpr int YR = (2 * DG_NDMAX + NT - 1) / NT;
  int yld[YR];
#pragma unroll
  for (int g = 0; g < YR; g++) yld[g] = IL && TID + g * NT < D.ox_nya ? (int)TI.ylist[TID + g * NT] : -1;
  for (int rep = 0; rep < count; rep++) {
    PROF_BEGIN(pa1);
    if constexpr (IL)
    {                                                       // G' w as ox_gt_mul forms it (w already carries E), the row indices from registers,
      lptr ya = o.dpart;                                    // the gradient values from the wave-interleaved table (same entries, same order of the sums)
      clptr w = o.w;
      __syncthreads();
#pragma unroll
      for (int g = 0; g < YR; g++) {                        // the multipliers in per-agent list order
        if (yld[g] >= 0) { const DgDense dd = ld_dense(yld[g]); ya[TID + g * NT] = (dd.r_pos >= 0 ? w[dd.r_pos] : 0.0) - (dd.r_neg >= 0 ? w[dd.r_neg] : 0.0); }
      }
      __syncthreads();
#pragma unroll
      for (int g = 0; g < GR; g++) {
        const int it4 = TID + g * NT;
        if (it4 < 64 * NGI) {                               // (wave-uniform: whole groups)
          double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
          if (rr[g][0] >= 0) s0 += w[rr[g][0]];
          if (rr[g][1] >= 0) s0 -= w[rr[g][1]];
          if (rr[g][2] >= 0) s1 += w[rr[g][2]];
          if (rr[g][3] >= 0) s1 -= w[rr[g][3]];
          if (rr[g][4] >= 0) s2 -= w[rr[g][4]];
          if (rr[g][5] >= 0) s2 += w[rr[g][5]];
          cgptr gp = TI.gdI + gbase[g] + (it4 & 63);
          const int cnt = gcnt[g], nfull = cnt & ~3, yo = gyo[g];
          for (int m = 0; m < glen[g]; m += 8) {            // glen: a multiple of eight; eight 512-byte runs in flight per wavefront
            double gv[8], yv[8];
#pragma unroll
            for (int j = 0; j < 8; j++) gv[j] = gp[64 * (m + j)];
#pragma unroll
            for (int j = 0; j < 8; j++) { const bool v = m + j < cnt; const double y = ya[v ? yo + 4 * (m + j) : 0]; yv[j] = v ? y : 0.0; }
            // same sums in the same order as the column-major loop: entries below nfull (whole quads of this lane) go to accumulator
            // (entry & 3), the last one to three to s0 in order; entries past the lane's count are zero times zero
#pragma unroll
            for (int j = 0; j < 8; j++) {
              if ((j & 3) == 0) s0 = __builtin_fma(yv[j], gv[j], s0);
              else {
                const bool tl = m + j >= nfull;
                double& sk = (j & 3) == 1 ? s1 : ((j & 3) == 2 ? s2 : s3);
                const double r = __builtin_fma(yv[j], gv[j], tl ? s0 : sk);
                s0 = tl ? r : s0; sk = tl ? sk : r;
              }
            }
          }
          double sm = (s0 + s1) + (s2 + s3);
          sm += dpp_f64<0xB1>(sm);
          sm += dpp_f64<0x4E>(sm);
          if (gcol[g] >= 0) o.xt[gcol[g]] = sm;
        }
      }
      __syncthreads();
    }
    else
    {                                                       // G' w as ox_gt_mul forms it (w already carries E), the row indices from registers
      lptr yd = o.ddx;
      clptr w = o.w;
      __syncthreads();
      for (int d = TID; d < D.ndense; d += NT) {
        const DgDense dd = ld_dense(d);
        yd[d] = (dd.r_pos >= 0 ? w[dd.r_pos] : 0.0) - (dd.r_neg >= 0 ? w[dd.r_neg] : 0.0);
      }
      __syncthreads();
#pragma unroll
      for (int g = 0; g < GR; g++) {
        const int it4 = TID + g * NT;
        if (it4 < 4 * n) {
          const int col = it4 >> 2, part = it4 & 3;
          double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
          if (rr[g][0] >= 0) s0 += w[rr[g][0]];
          if (rr[g][1] >= 0) s0 -= w[rr[g][1]];
          if (rr[g][2] >= 0) s1 += w[rr[g][2]];
          if (rr[g][3] >= 0) s1 -= w[rr[g][3]];
          if (rr[g][4] >= 0) s2 -= w[rr[g][4]];
          if (rr[g][5] >= 0) s2 += w[rr[g][5]];
          const int k1 = (int)T.cstart[col + 1];
          int k = (int)T.cstart[col] + part;
          for (; k + 12 < k1; k += 16) {
            const unsigned int pa = T.pairT[k], pb = T.pairT[k + 4], pc = T.pairT[k + 8], pd = T.pairT[k + 12];
            const double ga = T.gdT[k], gb = T.gdT[k + 4], gc = T.gdT[k + 8], gg = T.gdT[k + 12];
            s0 = __builtin_fma(yd[pa >> 22], ga, s0); s1 = __builtin_fma(yd[pb >> 22], gb, s1); s2 = __builtin_fma(yd[pc >> 22], gc, s2); s3 = __builtin_fma(yd[pd >> 22], gg, s3);
          }
          for (; k < k1; k += 4) { const unsigned int pa = T.pairT[k]; s0 = __builtin_fma(yd[pa >> 22], T.gdT[k], s0); }
          double sm = (s0 + s1) + (s2 + s3);
          sm += dpp_f64<0xB1>(sm);
          sm += dpp_f64<0x4E>(sm);
          if (part == 0) o.xt[col] = sm;
        }
      }
      __syncthreads();
    }
    for (int j = TID; j < n; j += NT) {
      const double dj = o.Dv[j], aI = o.EI[j] * dj, xj = o.x[j];
      o.rhs[j] = sigma * xj - cc * dj * o.q[j] + dj * o.xt[j] + aI * ox_rho_I(o, j, rho) * (aI * xj);
    }
    __syncthreads();
    PROF_END(PH_O_GT, pa1);
    PROF_BEGIN(pa2);
    if (!regs) {
      if (f32) ox_m_pass_f32((const glb_f*)o.S, n, o.rhs, o.part, o.xt);
      else ox_m_pass<false>(o.S, n, o.rhs, o.part, o.xt);     // xt = K^-1 rhs (explicit inverse: one pass over n x n)
    } else {
      if (act) {
        double a[4] = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < OX_RC; k++) a[k & 3] = __builtin_fma(kc[k], o.rhs[ja + k], a[k & 3]);
        int j = ja + OX_RC;
        if (f32) {
          const glb_f* M = (const glb_f*)o.S;
          for (; j + 15 < jb; j += 16) {
            float m[16];
#pragma unroll
            for (int k = 0; k < 16; k++) m[k] = M[(int64_t)(j + k) * n + S.i];
#pragma unroll
            for (int k = 0; k < 16; k++) a[k & 3] = __builtin_fma((double)m[k], o.rhs[j + k], a[k & 3]);
          }
          for (; j < jb; j++) a[0] = __builtin_fma((double)M[(int64_t)j * n + S.i], o.rhs[j], a[0]);
        } else {
          cgptr M = o.S;
          for (; j + 7 < jb; j += 8) {
            double m[8], t[8];
#pragma unroll
            for (int k = 0; k < 8; k++) { m[k] = M[(int64_t)(j + k) * n + S.i]; t[k] = o.rhs[j + k]; }
#pragma unroll
            for (int k = 0; k < 8; k++) a[k & 3] = __builtin_fma(m[k], t[k], a[k & 3]);
          }
          for (; j < jb; j++) a[0] = __builtin_fma(M[(int64_t)j * n + S.i], o.rhs[j], a[0]);
        }
        o.part[S.g * n + S.i] = (a[0] + a[1]) + (a[2] + a[3]);
      }
      __syncthreads();
      if (TID < n) {
        double sm = o.part[TID];
        for (int g = 1; g < S.G; g++) sm += o.part[g * n + TID];
        o.xt[TID] = sm;
      }
      __syncthreads();
    }
    for (int i = TID; i < n; i += NT) {
      const double xt = o.xt[i], xp = o.x[i], xn = alpha * xt + (1.0 - alpha) * xp;
      o.x[i] = xn; o.dx[i] = xn - xp; o.tmp[i] = o.Dv[i] * xt;
    }
    __syncthreads();
    PROF_END(PH_O_PMUL, pa2);
    PROF_BEGIN(pa3);
    qp_dense_dots<GP>(D, gd, o.tmp, o.dpart, o.ddx);
    PROF_END(PH_O_GS, pa3);
    PROF_BEGIN(pa4);
    for (int r = TID; r < nc; r += NT) {
      const double zt = qpw_row_dot(D, ld_row(r), o.tmp, o.ddx);
      const double er = o.E[r], zp = o.z[r], yr = o.y[r];
      const double zr = alpha * (er * zt) + (1.0 - alpha) * zp;
      const double us = er * fmin(-o.g[r], OSQP_INFTY), ls = -OSQP_INFTY * er;
      const double zn = fmin(fmax(__builtin_fma(yr, irho, zr), ls), us);
      const double dyr = rho * (zr - zn), yn = yr + dyr;
      o.z[r] = zn; o.dy[r] = dyr; o.y[r] = yn;
      o.w[r] = er * (rho * zn - yn);
    }
    XSYNC();
    PROF_END(PH_O_UPD, pa4);
  }
}

// Termination tests of a check iteration and the ratios of the rho rule (osqp_check of dgsqp_osqp.h; delta y is read from the scratch and
// survives, w is used as work vector and rebuilt by the caller)
template <class GP>
__device__ __noinline__ void ox_check(const Ctx& c, GP gd, double cc, bool approx) {
  const DgProb& D = dg_prob;
  const int n = D.n, nc = D.nc;
  const OxPtrs o = ox_ptrs(c);
  const double eps_abs = 1e-3, eps_rel = 1e-3, eps_inf = 1e-4, cinv = 1.0 / cc;
  PROF_BEGIN(po5);
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
      ox_gst_mul<GP>(c, o, gd, o.w, o.w, o.xt);
      double mx = 0;
      for (int j = TID; j < n; j += NT) mx = fmax(mx, __builtin_fabs(o.xt[j] / o.Dv[j]));
      mx = block_max(mx, o.red);
      pinf = mx < eps_inf * nrm;
      pinf10 = approx && lhs < -10.0 * eps_inf * nrm && mx < 10.0 * eps_inf * nrm;
    }
  }
  // Ax (G rows) -> w, Px -> rhs, A'y -> xt
  ox_gs_mul<GP>(o, gd, o.x, o.w);
  ox_m_pass<false>(o.M, n, o.tmp, o.part, o.rhs);                  // (o.tmp = D x after ox_gs_mul)
  for (int j = TID; j < n; j += NT) o.rhs[j] *= cc * o.Dv[j];
  __syncthreads();
  ox_gst_mul<GP>(c, o, gd, o.y, o.dpart, o.xt);                           // (E y into the slot behind w -- at least n_c doubles, dgsqp_layout.h: the dense-dot partials, rebuilt by every product)
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
      v[6] = fmax(v[6], __builtin_fabs(o.Dv[j] * o.x[j]));
      v[7] = fmax(v[7], __builtin_fabs(o.EI[j] * o.Dv[j] * o.x[j]));
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
  if (!(pri_res <= eps_p && dua_res <= eps_d) && !pinf) {
    double nrm = 0, qdx = 0;
    for (int j = TID; j < n; j += NT) { nrm = fmax(nrm, __builtin_fabs(o.Dv[j] * o.dx[j])); qdx += cc * o.Dv[j] * o.q[j] * o.dx[j]; }
    nrm = block_max(nrm, o.red);
    qdx = block_sum(qdx, o.red);
    if (nrm > 1.0 / OSQP_INFTY && qdx < -cc * eps_inf * nrm) {
      ox_gs_mul<GP>(o, gd, o.dx, o.w);                             // w = As dx; o.tmp = D dx
      ox_m_pass<false>(o.M, n, o.tmp, o.part, o.rhs);
      double mx = 0;
      for (int j = TID; j < n; j += NT) mx = fmax(mx, __builtin_fabs(cc * o.rhs[j]));
      mx = block_max(mx, o.red);
      auto cert = [&](double e) {
        int viol = 0;
        for (int r = TID; r < nc; r += NT) {
          const double er = o.E[r], us = er * fmin(-o.g[r], OSQP_INFTY), ls = -OSQP_INFTY * er, adx = o.w[r] / er;
          const bool ok_u = us > OSQP_INFTY * OSQP_MIN_SCALING || adx < e * nrm;
          const bool ok_l = ls < -OSQP_INFTY * OSQP_MIN_SCALING || adx > -e * nrm;
          viol |= !(ok_u && ok_l);
        }
        for (int j = TID; j < n; j += NT) {
          const bool inf_b = o.EI[j] * OSQP_INFTY > OSQP_INFTY * OSQP_MIN_SCALING;
          const double adx = o.Dv[j] * o.dx[j];
          viol |= !((inf_b || adx < e * nrm) && (inf_b || adx > -e * nrm));
        }
        return !__syncthreads_or(viol);
      };
      if (mx < cc * eps_inf * nrm) dinf = cert(eps_inf);
      if (approx && qdx < -cc * 10.0 * eps_inf * nrm && mx < cc * 10.0 * eps_inf * nrm) dinf10 = cert(10.0 * eps_inf);
    }
  }
  __syncthreads();
  if (TID == 0) {
    o.scal[DG_OSQP_CHK] = pri_res; o.scal[DG_OSQP_CHK + 1] = dua_res; o.scal[DG_OSQP_CHK + 2] = eps_p; o.scal[DG_OSQP_CHK + 3] = eps_d;
    o.scal[DG_OSQP_CHK + 4] = ad_pr; o.scal[DG_OSQP_CHK + 5] = ad_dr;
    o.scal[DG_OSQP_CHK + 6] = (pinf ? 1.0 : 0.0) + (dinf ? 2.0 : 0.0) + (pinf10 ? 4.0 : 0.0) + (dinf10 ? 8.0 : 0.0);
  }
  __syncthreads();
  PROF_END(PH_O_CHECK, po5);
}

// Polish (section 4) of the ADMM point in o.z / o.y (scaled), in unscaled variables; on acceptance du / lhat (= o.x / o.y slots) hold the
// polished point.  Returns 1 accepted, -1 rejected / not attempted; *na_out = active rows.  The caller has NOT yet unscaled x, y.
template <class GP>
__device__ __noinline__ int ox_polish(const Ctx& c, GP gd, double cc, double pri_res, double dua_res, int* na_out) {
  const DgProb& D = dg_prob;
  const int n = D.n, nc = D.nc;
  const OxPtrs o = ox_ptrs(c);
  const int lane = TID & 63;
  const bool w0 = TID < 64;
  const double delta = 1e-6, cinv = 1.0 / cc;
  const int np = (n + 1) & ~1;
  // the ADMM's w is dead: five n-vectors in its place
  lptr xs = o.w, nu = xs + np, e1 = nu + np, e2 = e1 + np;       // (the slot is max(n_c, 4 n) doubles: dgsqp_layout.h)
  int na = 0;
  // ---- active rows in row order:  upper (u - z) < y,  lower (z - l) < -y
  if (w0) {
    int cnt = 0;
    for (int base = 0; base < nc; base += 64) {
      const int r = base + lane;
      bool act = false;
      if (r < nc) {
        const double er = o.E[r];
        const double us = er * fmin(-o.g[r], OSQP_INFTY), ls = -OSQP_INFTY * er;
        act = ((us - o.z[r]) < o.y[r]) || ((o.z[r] - ls) < -o.y[r]);
      }
      const unsigned long long mask = __ballot(act);
      const int pos = cnt + __popcll(mask & ((1ull << lane) - 1ull));
      if (act && pos < n) o.alist[pos] = r;
      cnt += __popcll(mask);
    }
    if (lane == 0) o.scal[1] = (double)cnt;
  }
  __syncthreads();
  na = (int)o.scal[1];
  *na_out = na;
  if (na > n) return -1;
  // delta E^-2 of the active rows: parked in the scratch (delta y is dead) -- the eliminations below take the LDS slot of the polish vectors
  for (int k = TID; k < na; k += NT) { const double er = o.E[o.alist[k]]; o.dy[k] = delta / (er * er); }
  // ---- Hu = c M + delta D^-2, factor Ju (the ADMM's factor of K is dead)
  for (int j = TID; j < n; j += NT) o.tmp[j] = delta / (o.Dv[j] * o.Dv[j]);
  __syncthreads();
  PROF_BEGIN(po6);
  // (the elimination's multipliers take the slot of w .. dpart: the polish vectors are written after both factorisations)
  const bool ok = ox_factor(o.J, n, o.Mm, o.part, o.tv, [&](int i, int k) { double a = cc * o.M[(int64_t)i * n + k]; if (i == k) a += o.tmp[i]; return a; });
  PROF_END(PH_O_PINV, po6);
  PROF_COUNT(PH_O_NACT, na);
  if (!ok) return -1;
  // ---- Y: row k = Ju' a_k  (a box / rate row: one or two rows of Ju)
  PROF_BEGIN(po7);
  const XlSplit S = xl_split(n);
  for (int k = 0; k < na; k++) {
    const int p = o.alist[k];
    const DgRow Rw = ld_row(p);
    gptr Yk = o.Y + (int64_t)k * n;
    if (Rw.dense < 0) {
      const int c1 = am_col(D, Rw.a, Rw.k, Rw.idx);
      const bool has0 = (Rw.type == DG_R_RATE_UB || Rw.type == DG_R_RATE_LB) && Rw.k > 0;
      const double sgn = (Rw.type == DG_R_IN_UB || Rw.type == DG_R_RATE_UB) ? 1.0 : -1.0;
      for (int i = TID; i < n; i += NT) {
        double dj = o.J[(int64_t)c1 * n + i];
        if (has0) dj -= o.J[(int64_t)(c1 - DGSQP_NUA) * n + i];
        Yk[i] = sgn * dj;
      }
    } else {
      __syncthreads();
      for (int col = TID; col < n; col += NT) o.rhs[col] = g_row_coef<GP>(D, gd, p, col);
      __syncthreads();
      xl_jt_mul<cgptr>(o.J, n, n, S, 0, n, o.rhs, o.xt, o.part);
      for (int i = TID; i < n; i += NT) Yk[i] = o.xt[i];
    }
  }
  XSYNC();
  // ---- S = Y Y' + delta E^-2 (lower triangle), factor Js
  bool oks = true;
  if (na > 0) {
    oks = ox_factor(o.S, na, o.Mm, o.part, o.tv, [&](int i, int k) {
      cgptr yi = o.Y + (int64_t)i * n, yk = o.Y + (int64_t)k * n;
      double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
      int t = 0;
      for (; t + 3 < n; t += 4) { s0 = __builtin_fma(yi[t], yk[t], s0); s1 = __builtin_fma(yi[t + 1], yk[t + 1], s1); s2 = __builtin_fma(yi[t + 2], yk[t + 2], s2); s3 = __builtin_fma(yi[t + 3], yk[t + 3], s3); }
      for (; t < n; t++) s0 = __builtin_fma(yi[t], yk[t], s0);
      double s = (s0 + s1) + (s2 + s3);
      if (i == k) s += o.dy[i];
      return s;
    });
  }
  PROF_END(PH_O_PROWS, po7);
  if (!oks) return -1;
  PROF_BEGIN(po8);
  const int m = na;
  lptr tq = o.rhs, dnu = o.xt;     // work vectors of the solve (n-vectors; m <= n)
  // (dx, dnu) = Kreg^-1 (r1, r2):  t = Hu^-1 r1,  dnu = S^-1 (A t - r2),  dx = t - Ju (Y' dnu).   r1 in e1, r2 in e2; dx -> o.dx, dnu -> dnu
  auto kkt_solve = [&]() {
    ox_solve(o.J, n, e1, o.tv, o.dx, o.part);                       // t
    qp_dense_dots<GP>(D, gd, o.dx, o.dpart, o.ddx);
    for (int k = TID; k < m; k += NT) tq[k] = qpw_row_dot(D, ld_row(o.alist[k]), o.dx, o.ddx) - e2[k];
    __syncthreads();
    if (m > 0) ox_solve(o.S, m, tq, o.tv, dnu, o.part);
    // tq = Y' dnu (thread = element i, rows of Y coalesced), then dx -= Ju tq
    for (int i = TID; i < n; i += NT) {
      double s0 = 0, s1 = 0;
      int k = 0;
      for (; k + 1 < m; k += 2) { s0 = __builtin_fma(o.Y[(int64_t)k * n + i], dnu[k], s0); s1 = __builtin_fma(o.Y[(int64_t)(k + 1) * n + i], dnu[k + 1], s1); }
      if (k < m) s0 = __builtin_fma(o.Y[(int64_t)k * n + i], dnu[k], s0);
      o.tmp[i] = s0 + s1;
    }
    __syncthreads();
    xl_j_mul<cgptr>(o.J, n, n, S, 0, n, o.tmp, o.tv, o.part);
    for (int i = TID; i < n; i += NT) o.dx[i] -= o.tv[i];
    __syncthreads();
  };
  // A' nu over the active rows -> out (four lanes per column)
  auto at_active = [&](clptr nuv, lptr out) {
    __syncthreads();
    for (int it = TID; it < 4 * n; it += NT) {
      const int col = it >> 2, part = it & 3;
      double s = 0;
      for (int j = part; j < m; j += 4) s = __builtin_fma(nuv[j], g_row_coef<GP>(D, gd, o.alist[j], col), s);
      s += dpp_f64<0xB1>(s);
      s += dpp_f64<0x4E>(s);
      if (part == 0) out[col] = s;
    }
    __syncthreads();
  };
  for (int j = TID; j < n; j += NT) e1[j] = -cc * o.q[j];
  for (int k = TID; k < m; k += NT) e2[k] = fmin(-o.g[o.alist[k]], OSQP_INFTY);
  __syncthreads();
  kkt_solve();
  for (int j = TID; j < n; j += NT) xs[j] = o.dx[j];
  for (int k = TID; k < m; k += NT) nu[k] = dnu[k];
  __syncthreads();
  // (rhs / xt double as tq / dnu inside kkt_solve: the residual products below finish with them before the next solve)
  lptr mx = o.EI;       // M xs: E_I is only needed by the ADMM -- dead now
  for (int rf = 0; rf < 3; rf++) {
    ox_m_pass<false>(o.M, n, xs, o.part, mx);
    at_active(nu, o.tmp);
    qp_dense_dots<GP>(D, gd, xs, o.dpart, o.ddx);
    for (int j = TID; j < n; j += NT) e1[j] = -cc * o.q[j] - (cc * mx[j] + o.tmp[j]);
    for (int k = TID; k < m; k += NT) e2[k] = fmin(-o.g[o.alist[k]], OSQP_INFTY) - qpw_row_dot(D, ld_row(o.alist[k]), xs, o.ddx);
    __syncthreads();
    kkt_solve();
    for (int j = TID; j < n; j += NT) xs[j] += o.dx[j];
    for (int k = TID; k < m; k += NT) nu[k] += dnu[k];
    __syncthreads();
  }
  ox_m_pass<false>(o.M, n, xs, o.part, mx);
  at_active(nu, o.tmp);
  qp_dense_dots<GP>(D, gd, xs, o.dpart, o.ddx);
  double pr_p = 0, dr_p = 0;
  int nonfin = 0;
  for (int r = TID; r < nc; r += NT) pr_p = fmax(pr_p, fmax(0.0, qpw_row_dot(D, ld_row(r), xs, o.ddx) - fmin(-o.g[r], OSQP_INFTY)));
  for (int j = TID; j < n; j += NT) { dr_p = fmax(dr_p, __builtin_fabs(cc * mx[j] + cc * o.q[j] + o.tmp[j])); nonfin |= !(__builtin_fabs(xs[j]) < INFINITY); }
  for (int k = TID; k < m; k += NT) nonfin |= !(__builtin_fabs(nu[k]) < INFINITY);
  pr_p = block_max(pr_p, o.red);
  dr_p = cinv * block_max(dr_p, o.red);
  nonfin = __syncthreads_or(nonfin);
  const bool better = (pr_p < pri_res && dr_p < dua_res) || (pr_p < pri_res && dua_res < 1e-10) || (dr_p < dua_res && pri_res < 1e-10);
  int polished = -1;
  if (better && !nonfin) {
    for (int j = TID; j < n; j += NT) o.x[j] = xs[j];              // (= du)
    for (int r = TID; r < nc; r += NT) o.y[r] = 0.0;               // (= lhat)
    __syncthreads();
    for (int k = TID; k < m; k += NT) o.y[o.alist[k]] = cinv * nu[k];
    polished = 1;
  }
  __syncthreads();
  PROF_END(PH_O_PSOLVE, po8);
  return polished;
}

template <class GP>
__device__ __noinline__ int dev_qp_osqp_xl_t(const Ctx& c, GP gd) {
  const DgProb& D = dg_prob;
  const int n = D.n, nc = D.nc;
  const OxPtrs o = ox_ptrs(c);
  const int max_iter = 4000, check_every = 25;
  __syncthreads();
  PROF_BEGIN(pt_qp);
  if (TID == 0) { o.scal[DG_QP_NPREV] = 0.0; o.scal[DG_XVALID] = 0.0; }
  const double cc = ox_setup<GP>(c, gd);
  if (cc != cc) {
    if (TID == 0) { o.scal[DG_OSQP_INFO] = OSQP_NAN_DATA; o.scal[DG_OSQP_INFO + 1] = 0; o.scal[DG_OSQP_INFO + 2] = 0; }
    __syncthreads();
    return 1;
  }
  const double cinv = 1.0 / cc;
  double rho = D.par.osqp_rho_carry ? o.scal[DG_OSQP_RHO] : 0.1;      // (carried from the scenario's previous call: include/dgsqp.h)
  int rho_updates = 0;
  for (int j = TID; j < n; j += NT) { o.x[j] = 0.0; o.dx[j] = 0.0; }
  for (int r = TID; r < nc; r += NT) { o.z[r] = 0.0; o.y[r] = 0.0; o.dy[r] = 0.0; o.w[r] = 0.0; }
  XSYNC();
  int status = OSQP_MAX_ITER, iters = 0, approx_flags = 0;
  double pri_res = INFINITY, dua_res = INFINITY, eps_p = 0, eps_d = 0;
  bool need_k = true;
  PROF_BEGIN(po4);
  for (int it = check_every; it <= max_iter; it += check_every) {       // one block of iterations, then a termination check
    if (need_k) {
      need_k = false;
      if (!ox_build_k(c, rho, cc)) { status = OSQP_NAN_DATA; break; }
      // (the elimination's multipliers lie over w)
      for (int r = TID; r < nc; r += NT) o.w[r] = o.E[r] * (rho * o.z[r] - o.y[r]);
      __syncthreads();
    }
    if (n <= 176) ox_iterate_block<GP, false>(c, gd, rho, cc, check_every); else ox_iterate_block<GP, true>(c, gd, rho, cc, check_every);
    iters = it;
    ox_check<GP>(c, gd, cc, it == max_iter);
    pri_res = o.scal[DG_OSQP_CHK]; dua_res = o.scal[DG_OSQP_CHK + 1]; eps_p = o.scal[DG_OSQP_CHK + 2]; eps_d = o.scal[DG_OSQP_CHK + 3];
    const double ad_pr = o.scal[DG_OSQP_CHK + 4], ad_dr = o.scal[DG_OSQP_CHK + 5];
    const int flags = (int)o.scal[DG_OSQP_CHK + 6];
    if (pri_res <= eps_p && dua_res <= eps_d) { status = OSQP_SOLVED; break; }
    if (flags & 1) { status = OSQP_PRIMAL_INFEASIBLE; break; }
    if (flags & 2) { status = OSQP_DUAL_INFEASIBLE; break; }
    approx_flags = flags;
    {
      const double rho_new = fmin(fmax(rho * sqrt(ad_pr / (ad_dr + 1e-10)), OSQP_RHO_MIN), OSQP_RHO_MAX);
      if (rho_new > rho * 5.0 || rho_new < rho / 5.0) { rho = rho_new; rho_updates++; need_k = true; }
    }
    for (int r = TID; r < nc; r += NT) o.w[r] = o.E[r] * (rho * o.z[r] - o.y[r]);
    __syncthreads();
  }
  PROF_END(PH_O_ADMM, po4);
  PROF_COUNT(PH_O_ITERS, iters);
  if (status == OSQP_MAX_ITER && iters == max_iter) {
    if (pri_res <= 10.0 * eps_p && dua_res <= 10.0 * eps_d) status = OSQP_SOLVED_INACCURATE;
    else if (approx_flags & 4) status = OSQP_PRIMAL_INFEASIBLE_INACCURATE;
    else if (approx_flags & 8) status = OSQP_DUAL_INFEASIBLE_INACCURATE;
  }
  __syncthreads();
  int polished = 0, na = 0;
  if (status == OSQP_SOLVED) polished = ox_polish<GP>(c, gd, cc, pri_res, dua_res, &na);
  __syncthreads();
  if (polished != 1) {      // the ADMM iterate, unscaled, in place (x is du's slot, y is lhat's)
    for (int j = TID; j < n; j += NT) o.x[j] = o.Dv[j] * o.x[j];
    for (int r = TID; r < nc; r += NT) o.y[r] = cinv * o.E[r] * o.y[r];
  }
  __syncthreads();
  if (TID == 0) {
    o.scal[DG_OSQP_INFO] = (double)status; o.scal[DG_OSQP_INFO + 1] = (double)iters; o.scal[DG_OSQP_INFO + 2] = (double)polished; o.scal[DG_OSQP_INFO + 3] = rho;
    atomicAdd(&dg_osqp_count[0], 1ULL); atomicAdd(&dg_osqp_count[1], (unsigned long long)iters);
    if (D.par.osqp_rho_carry) o.scal[DG_OSQP_RHO] = rho;
    o.scal[DG_OSQP_INFO + 4] = (double)rho_updates; o.scal[DG_OSQP_INFO + 5] = (double)na; o.scal[DG_OSQP_INFO + 6] = pri_res; o.scal[DG_OSQP_INFO + 7] = dua_res;
  }
  int nonfinite = 0;
  for (int j = TID; j < n; j += NT) nonfinite |= !(__builtin_fabs(o.x[j]) < INFINITY);
  for (int r = TID; r < nc; r += NT) nonfinite |= !(__builtin_fabs(o.y[r]) < INFINITY);
  nonfinite = __syncthreads_or(nonfinite);
  PROF_END(PH_QP, pt_qp);
  return (nonfinite || status == OSQP_PRIMAL_INFEASIBLE || status == OSQP_DUAL_INFEASIBLE || status == OSQP_PRIMAL_INFEASIBLE_INACCURATE ||
          status == OSQP_DUAL_INFEASIBLE_INACCURATE || status == OSQP_NAN_DATA) ? 1 : 0;
}
__device__ int dev_qp_osqp_xl(const Ctx& c) {
  if (dg_prob.gd_global) return dev_qp_osqp_xl_t<cgptr>(c, c.ws + dg_prob.ws_gd);
  return dev_qp_osqp_xl_t<clptr>(c, LP(dg_prob.L.gd));
}
