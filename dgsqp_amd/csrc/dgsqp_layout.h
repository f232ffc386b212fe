// Host+device description of one game as the HIP kernels see it: dimensions,
// the constraint-row table (reference row order, DGSQP/solvers/DGSQP.py:732-821),
// the packed storage of the distinct dense constraint gradients
//  and the LDS arena of one scenario workgroup.
#pragma once
#include <stdint.h>
#include <algorithm>

#include "../../include/dgsqp.h"

#define DG_NMAX 64        // horizon limit
#define DG_NCMAX 2048     // inequality rows limit
#define DG_NVARMAX 320     // decision variables limit (XL kernels: five 64-lane registers per column)
#define DG_NDMAX 1024     // distinct dense gradients limit (6-car merge, N = 25: 837)
#define DG_MAXEFF 8       // effective variables of one agent's dynamics (dyn bicycle: 6 states + 2 inputs)
#define DG_MAXDIR 36      // MAXEFF*(MAXEFF+1)/2 Taylor directions
#ifndef DG_BLOCK
#define DG_BLOCK 512      // threads per scenario workgroup (8 wavefronts = 2 per SIMD, to hide LDS latency)
#endif
// -DDG_BLOCK=256 (row N1, "two scenarios per CU"): 4 wavefronts = 1 per SIMD per workgroup and HALF the arena, so that two workgroups
// share a CU and one scenario's latency-bound phases overlap the other's issue-bound ones.  That build holds the LDS-resident
// explicit-inverse layout only (dgsqp_create refuses big / XL / classical-QP games: their register tilings assume 8 wavefronts).
#define DG_WG_PER_CU (512 / DG_BLOCK)
#define DG_PSD_KMAX (2 * (DG_BLOCK / 64))   // negative eigenpairs handled per batch (two per wavefront)
#define DG_NH (DG_BLOCK / 128)  // row-groups of the register-resident matrix slices (thread = column x row-group)
#define DG_LDS_LIMIT ((163840 - 512) / DG_WG_PER_CU)   // 160 KB per CU minus the kernel's static LDS (256 B per workgroup)

enum { DG_R_OBS = 0, DG_R_RATE_UB, DG_R_RATE_LB, DG_R_IN_UB, DG_R_IN_LB, DG_R_ST_UB, DG_R_ST_LB, DG_R_LANE };

struct DgRow {      // one inequality row
  int8_t type, a, b, idx, sgn;  // sgn: coefficient of the shared dense gradient (+1 / -1)
  int8_t k;
  int16_t dense;    // index of the dense gradient backing this row, or -1
};

struct DgDense {    // one distinct dense gradient: d x^a_k[idx] / du   or the obstacle gradient of pair (a,b) at stage k
  int8_t kind;      // 0 state row, 1 obstacle, 2 lane half-plane (idx = lane number; n_x dx/du + n_y dy/du)
  int8_t a, b, idx, k;
  uint8_t nt, t0lo, t0hi;  // its chunks in the dot-product task table: nt tasks from index t0lo + 256 t0hi
  int32_t off;      // offset in the packed Gd array; kind 0: 2k entries [t][j]; kind 1: 4k entries, agent a then agent b
  int16_t r_pos, r_neg;  // rows using this gradient with coefficient +1 / -1 (-1: none)
};

// Dot products with the dense gradients are cut into chunks of <= DG_CHUNK contiguous entries (one lane each, all loads
// issued together): gd[p0 .. p0+len) . v[v0 .. v0+len)
#define DG_XL_KMAX DG_NVARMAX   // XL layout: negative eigenvalues the tridiagonal _nearestPD handles (all of them; it was 64 until round 4, and the
                                 // one-sided Jacobi sweeps beyond cost ~200 Mcycles per call at n = 200: one F1 scenario with > 64 spent 27 Gcycles there)
#define DG_CHUNK 16
// Storage of the dual method's inverse Cholesky factor T (upper triangular, dgsqp_solve.h: qpt_solve): column j = 8a + b holds its
// rows 0..j followed by zeros up to row 8(a + 1) - 1, plus one unused entry; it starts at dg_tcol(j).  The zeros make groups of eight
// rows / columns of the triangle rectangular (no per-element masks in the products); the extra entry skews the column starts over
// the LDS banks.  dg_tcol(n) entries hold n columns.
__host__ __device__ inline int dg_tcol(int j) { const int a = j >> 3, b = j & 7; return 8 * (a + 1) * (4 * a + b) + j; }
struct DgTask {
  int32_t p0;
  uint16_t v0;
  uint8_t len, pad_;
};
#define DG_NTASKMAX 3072

// LDS arena, offsets in doubles
struct DgLds {
  // persistent
  int u, l, q, g, d, v, gd, yd, red, scal;
  int w_prev, w_prevlam; // active set of the previous QP (ints) and its multipliers
  int t_rows, t_dense;  // LDS copies of the row / dense-gradient tables (8 and 16 bytes per entry)
  int t_task;           // LDS copy of the dense dot-product task table (8 bytes per entry)
  int t_track;          // LDS copy of the track tables: seg_s[17], seg_curv[16], seg_ang[17], slope[16]
  int t_atan;           // atan(k) for k = 0, 1/2, 1, 3/2, inf: 5 high parts, 5 low parts
  int scr;  // start of phase scratch
  // EVAL scratch (absolute offsets)
  int e_x, e_ue, e_A[DGSQP_MAX_AGENTS], e_B[DGSQP_MAX_AGENTS], e_dJ, e_lam, e_Dxs, e_K, e_xs, e_xs2;
  // EIG scratch
  int g_Bp, g_V, g_tw, g_strip;  // packed P / packed Householder reflectors / tridiagonal workspace
  // QP scratch (P shares g_Bp)
  int p_R, p_lam, p_c, p_w, p_r, p_y, p_t, p_alist, p_rd, p_yd2, p_yslot, p_yfree, p_dpart, p_act, p_part, p_xu;
  // QP outputs that must survive trial evaluations
  int o_du, o_lhat;
  int a_z, a_y, a_E, a_dy, a_w;  // qp_method = OSQP (dgsqp_osqp.h): the ADMM's n_c-vectors z, y, E (row scaling), delta y, a work vector -- inside the
                                 // slot of the active-set factor T (p_R), which only the polish needs
  int a_tab;                     // ... followed by the per-QP index tables of the transposed product G' w (dgsqp_osqp.h: osqp_build_tables)
  int a_sw;                      // pivot-column buffers of the Gauss-Jordan sweep (4 (NH RPT + 4) doubles): the ADMM's delta-y / work vectors where they are large enough
  int a_tail;                    // ... and five n-vectors of the polish at the end of that slot (T keeps room for DgProb.osqp_namax active rows)
  // qp_method = OSQP on the XL layout (dgsqp_osqp_xl.h): z; the slot of w (>= max(n_c, 4 n): the polish's vectors) followed by the dense-dot partials
  // (>= n_c: a work vector of the checks) -- together >= 16 n + 2, the multipliers of the blocked elimination; seven n-vectors of stride ox_np;
  // partial sums of the matrix products (>= 512: the elimination's tables); the polish's active rows
  int ox_z, ox_w, ox_dpart, ox_yd2, ox_nv, ox_np, ox_part, ox_alist;
  int x_el;  // XL layout with xl_el: packed lower triangle of the QP's elimination M = L~ D L~^T (overlaps the QP outputs and c_R, dead until J is built)
  int c_R;   // classical QP: the first DgProb.c_rcap columns of the triangular factor R, packed column-major (phase-multiplexed with e_xs2)
  // LSQR scratch (s_yd2 / s_dpart: the dense-dot scratch of the dual start -- the QP's p_yd2 / p_dpart except in the
  // 'tables in constant memory' layout, where the dual start has its own behind its vectors and may overlap the QP outputs)
  int s_u, s_v, s_w, s_x, s_t, s_yd2, s_dpart;
  int total;  // doubles
};

struct DgProb {
  dgsqp_problem_t P;
  dgsqp_params_t par;
  int M, N, nq, nu, n, nc, npairs, ndense, ngd, ntask;
  int nqa[DGSQP_MAX_AGENTS], qoff[DGSQP_MAX_AGENTS], sidx[DGSQP_MAX_AGENTS], eyidx[DGSQP_MAX_AGENTS];
  double inv_track_L;
  double eig_floor;    // value given to negative eigenvalues by _nearestPD (par.eig_floor, 1e-10 when not set)
  int uniform_nqa;
  int gd_global;    // the packed constraint gradients live in the global scratch (ws_gd) instead of LDS: XL games beyond n ~ 160
  int xl_pack;      // XL layout: the symmetric matrix of the Householder tridiagonalisation lives in LDS as a packed lower triangle
                    // (n (n + 1) / 2 doubles next to the per-step vectors: n <= ~160) instead of the L2 scratch; the packed constraint
                    // gradients move to the scratch to make room (gd_global)
  int xl_el;        // XL layout: the elimination that builds J = L^-T runs on a packed lower triangle in LDS (L.x_el) and J is written to
                    // the scratch once, instead of n passes over the L2-resident J
  int xl_blk;       // XL layout: the elimination that builds J = L^-T runs 16 pivots per pass (xl_eliminate_blocked; multipliers in LDS at L.x_el)
  int xl_noblock;   // (development knob, environment DGSQP_XL_NOBLOCK: the XL warm start re-absorbs its rows one at a time)
  int tab_const;    // the row / dense-gradient / task tables are read from this constant block instead of LDS copies, the compact
                    // state-Hessian columns (e_K) live in the global scratch (ws_K) and the stage gradients share the costates'
                    // slot: games whose vectors alone nearly fill the arena (6 agents, N = 25: n = 300, 1,587 rows, 837 gradients)
  int c_rcap;       // columns of R (classical QP) that live in LDS; the rest in the global scratch
  int osqp;         // par.qp_method == DGSQP_QP_OSQP: the QP is OSQP's ADMM + polish (dgsqp_osqp.h) on the explicit-inverse layout, whatever reg
  int osqp_nacap;   // (layout search: the T slot is sized for this many active rows -- n unless the arena overflows)
  int osqp_namax;   // ... whose polish handles at most this many active rows (what the T slot holds next to the polish vectors)
  int classic_qp;   // the QP runs the classical (J = L^-T) Goldfarb-Idnani kernels of dgsqp_xl.h: XL layout, or a projected Hessian
                    // whose smallest eigenvalue (eig_floor + reg) is below 1e-8 -- the literal reg = 0 formula
  int big;          // 2: XL layout (n > 128, dgsqp_xl.h).  1: the packed inverse P and the packed Householder reflectors live in the workgroup's global scratch (L2)
                    // instead of LDS: games whose LDS-resident layout exceeds the 160 KB arena
  int lsqr_keeps_eval;  // the dual start's vectors sit ABOVE the evaluation arrays (not over them): the trajectory, A_k, B_k and the Taylor
                    // tensor of the start point survive the dual start and serve the first linearisation
  int ls_spec;      // trial step sizes of _line_search_3 rolled out concurrently (speculation width)
  int ls_spec1;     // how many of them live in the first LDS segment (e_xs); the rest in e_xs2  // every agent uses the same vehicle model (statically indexed fast paths)
  int neff[DGSQP_MAX_AGENTS], ndir[DGSQP_MAX_AGENTS];
  int effvar[DGSQP_MAX_AGENTS][DG_MAXEFF];  // effective variable -> index into z = (q_0..q_{nqa-1}, u_0, u_1)
  int t2off[DGSQP_MAX_AGENTS];              // offset (doubles) of agent block in the Taylor tensor workspace
  int t2k[DGSQP_MAX_AGENTS];                // per-stage stride of that block = nqa*ndir
  int64_t t2_doubles;
  int16_t ox_perm[DG_NVARMAX];   // ... its task order: the columns sorted by length (longest first), sixteen to a group
  int ox_gstart[DG_NVARMAX / 16 + 2];   // ... and where each group's runs start (entries)
  int ox_nya, ox_ngi;   // OSQP on the XL layout, wave-interleaved G' table (dgsqp_osqp_xl.h: ox_build_tables): entries of the per-agent multiplier lists; padded entries of the table
  int64_t wsx_gdI, wsx_tabI;   // ... its values, and its index part (group starts, per-agent list starts, the lists)
  int64_t wsx_Y, wsx_S, wsx_E, wsx_dy, wsx_tab, wsx_gdT;   // OSQP on the XL layout: the polish's Y and Schur complement (n x n each), row scaling and delta y (n_c), the transposed index table of G' w
  int64_t ws_t2, ws_q, ws_base, ws_H, ws_tang, ws_Y, ws_P, ws_V, ws_R, ws_Vp, ws_xM, ws_xJ, ws_xR, ws_bfgs, ws_gd, ws_v2, ws_K, ws_doubles; // global workspace layout (doubles): Taylor tensor, raw Q, watchdog backups,
                                                             // costate-contracted dynamics Hessians, tangent trajectories
  DgRow rows[DG_NCMAX];
  DgDense dense[DG_NDMAX];
  int16_t r_in_ub[DGSQP_MAX_AGENTS][DG_NMAX][DGSQP_NUA], r_in_lb[DGSQP_MAX_AGENTS][DG_NMAX][DGSQP_NUA];
  int16_t r_rate_ub[DGSQP_MAX_AGENTS][DG_NMAX][DGSQP_NUA], r_rate_lb[DGSQP_MAX_AGENTS][DG_NMAX][DGSQP_NUA];
  int16_t stage_row0[DG_NMAX + 2];   // first row of each stage (rows are ordered by stage)
  int16_t stage_dense0[DG_NMAX + 2]; // first dense gradient of each stage
  DgTask dtask[DG_NTASKMAX];
  DgLds L;
  double spl[9 * DGSQP_MAX_KNOTS];   // spline track (P.track_kind == DGSQP_TRACK_SPLINE): knots, x coefficients, y coefficients
};

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>

// Global workspace and LDS arena of a built problem; falls back to the big layout when the arena overflows.
static inline std::string dg_build_layout(DgProb& D) {
  // ---- global workspace (doubles)
  D.ws_t2 = 0;
  D.ws_q = D.ws_t2 + D.t2_doubles;
  D.ws_base = D.ws_q + (int64_t)D.n * D.n;
  D.ws_H = D.ws_base + 2 * D.n + 2 * D.nc + 16;
  D.ws_tang = D.ws_H + (int64_t)D.M * D.N * D.M * (DG_MAXEFF * DG_MAXEFF);
  D.ws_Y = D.ws_tang + (int64_t)(D.N + 1) * DGSQP_MAX_NQA * D.n;   // y_j = P a_j of the QP's active rows
  D.ws_P = (D.ws_Y + (int64_t)D.n * D.n + 1) & ~(int64_t)1;           // big layout: packed P, packed reflectors (16-byte aligned: the XL kernels read pairs);
  const int64_t matsz = D.big == 2 ? (int64_t)D.n * D.n : (D.big ? (int64_t)D.n * (D.n + 1) / 2 : 0);   // XL: three full matrices
  D.ws_V = D.ws_P + matsz;
  D.ws_R = D.ws_V + matsz;
  D.ws_Vp = D.ws_R + (D.big == 2 ? matsz + 4 * D.n : 0);      // XL: eigenvectors of the scenario's previous _nearestPD (Jacobi warm start)
  D.ws_doubles = D.ws_Vp + (D.big == 2 ? matsz : 0);
  // matrices of the classical QP: M / its Cholesky factor, J, R (row-major n x n).  XL: the Jacobi buffers are reused
  D.classic_qp = D.big == 2 || D.eig_floor + (D.par.reg > 0 ? D.par.reg : 0.0) < 1e-8 ||
                 (D.par.variant == DGSQP_VARIANT_V2 && D.eig_floor < 1e-8);      // v2: reg decays towards 0 during a solve (dgsqp_solve_v2.h)
  // OSQP: ADMM adds sigma = 1e-6 and the polish delta = 1e-6 to the projected Hessian, so its explicit inverses are well conditioned
  // whatever reg is; it needs the projected Hessian M itself (n x n, row-major) in the scratch; W = Gs^T Gs borrows the Y slot
  if (D.osqp && D.big == 2) {      // dgsqp_osqp_xl.h: M where dev_xl_psd leaves it (ws_R), the factor of K / Hu in ws_P, W in ws_V, plus the polish's two matrices
    D.classic_qp = 0; D.ws_xM = D.ws_R; D.ws_xJ = D.ws_P; D.ws_xR = D.ws_V;
    const int64_t ncp = (D.nc + 1) & ~1, nvp = ncp > D.n ? ncp : D.n;
    D.wsx_Y = (D.ws_doubles + 1) & ~(int64_t)1; D.wsx_S = D.wsx_Y + (int64_t)D.n * D.n; D.wsx_E = D.wsx_S + (int64_t)D.n * D.n; D.wsx_dy = D.wsx_E + nvp;
    D.wsx_tab = D.wsx_dy + nvp;          // uint32: n + 2 column starts, one entry per packed gradient element
    D.wsx_gdT = D.wsx_tab + ((int64_t)D.ngd + D.n + 6) / 2 + 2;      // the packed gradients' values in the table's (transposed) order
    D.ws_doubles = D.wsx_gdT + D.ngd + 2;
    // the same values once more in WAVE-INTERLEAVED order for the ADMM iteration's G' w (ox_iterate_block).  The columns are sorted by
    // length (ox_perm; the lengths are a property of the game) and taken sixteen to a group; task (sorted column s, quarter) = lane
    // it4 & 63 of group it4 >> 6, it4 = 4 s + quarter; entry m of that lane at ox_gstart[group] + 64 m + lane, every group padded with
    // zeros to its longest quarter column rounded up to eight entries -- a wavefront load reads ONE 512-byte run where the column-major
    // table gave it 16 cache lines, and eight of them are in flight per lane.  The multipliers come from per-agent lists (agent a: its
    // covering gradients in table order), so the hot loop reads no index at all.
    {
      int nya = 0;
      for (int d = 0; d < D.ndense; d++) nya += D.dense[d].kind == 1 ? 2 : 1;
      D.ox_nya = nya;
      const int NG = (4 * D.n + 63) / 64;
      int cnt[DG_NVARMAX];
      for (int col = 0; col < D.n; col++) {
        const int a = col / (D.N * DGSQP_NUA), t = (col % (D.N * DGSQP_NUA)) / DGSQP_NUA;
        cnt[col] = 0;
        for (int d = D.stage_dense0[t + 1]; d < D.ndense; d++) cnt[col] += (D.dense[d].a == a) || (D.dense[d].kind == 1 && D.dense[d].b == a);
        D.ox_perm[col] = (int16_t)col;
      }
      std::stable_sort(D.ox_perm, D.ox_perm + D.n, [&](int16_t x, int16_t y) { return cnt[x] > cnt[y]; });
      int tot = 0;
      for (int G = 0; G < NG; G++) {
        const int mx = 16 * G < D.n ? cnt[D.ox_perm[16 * G]] : 0;        // (sorted: the group's first column is its longest)
        D.ox_gstart[G] = tot;
        tot += 64 * ((((mx + 3) / 4) + 7) & ~7);
      }
      D.ox_gstart[NG] = tot;
      D.ox_ngi = tot;
      D.wsx_tabI = (D.ws_doubles + 1) & ~(int64_t)1;                     // uint32: M + 1 list starts, nya list entries
      D.wsx_gdI = D.wsx_tabI + (DGSQP_MAX_AGENTS + 1 + nya + 3) / 2 + 2;
      D.ws_doubles = D.wsx_gdI + tot + 2;
    }
  }
  else if (D.osqp) { D.classic_qp = 0; D.ws_xM = D.ws_doubles; D.ws_xJ = D.ws_xR = D.ws_xM; D.ws_doubles = D.ws_xM + (int64_t)D.n * D.n; }
  else if (D.big == 2) { D.ws_xM = D.ws_R; D.ws_xJ = D.ws_P; D.ws_xR = D.ws_V; }
  else if (D.classic_qp) { D.ws_xM = D.ws_doubles; D.ws_xJ = D.ws_xM + (int64_t)D.n * D.n; D.ws_xR = D.ws_xJ + (int64_t)D.n * D.n; D.ws_doubles = D.ws_xR + (int64_t)D.n * D.n; }
  // hessian_approximation = 'bfgs': u of the previous iteration and d(u_prev, l) (2 n), the Hessian used at the top of the
  // previous iteration and its projection (n x n each)
  D.ws_bfgs = D.ws_doubles;
  if (D.par.hessian_bfgs) D.ws_doubles += 2 * (int64_t)D.n + 2 * (int64_t)D.n * D.n;
  D.ws_v2 = D.ws_doubles;     // DG-SQP v2: three iteration records (checkpoint, latest, current) + (u, l) of the last m-step
  if (D.par.variant == DGSQP_VARIANT_V2) D.ws_doubles += 3 * (2 * (int64_t)D.n + 2 * D.nc + 2) + D.n + D.nc;
  D.ws_gd = D.ws_doubles;
  if (D.gd_global) D.ws_doubles += D.ngd + DG_CHUNK;   // (chunked dots read up to one chunk past the last gradient)
  D.ws_K = D.ws_doubles;
  if (D.tab_const) D.ws_doubles += (int64_t)D.M * (D.N + 1) * D.M * 5;
  D.ws_doubles = (D.ws_doubles + 31) / 32 * 32;
  // ---- LDS arena
  DgLds& L = D.L;
  const int n = D.n, nq = D.nq, nu = D.nu, N = D.N, nc = D.nc, nd = D.ndense;
  (void)nu;
  int o = 0;
  auto take = [&](int cnt) { int r = o; o += (cnt + 1) & ~1; return r; };
  L.u = take(n); L.l = take(nc); L.q = take(n); L.g = take(nc); L.d = take(n); L.v = take(n);
  L.gd = take(D.gd_global ? (nc > 2 ? nc : 2) : D.ngd);   // gd_global: only the nc-vector the trial merits keep there
  L.yd = take(nd); L.red = take(64); L.scal = take(64);
  L.w_prev = take((n + 2) / 2 + 1); L.w_prevlam = take(n + 1);   // final active set of the previous QP of this scenario (warm start)
  if (D.tab_const) { L.t_rows = L.t_dense = L.t_task = o; }
  else { L.t_rows = take(nc); L.t_dense = take(2 * nd); L.t_task = take(D.ntask); }
  L.t_track = take(4 * (DGSQP_MAX_SEGS + 1)); L.t_atan = take(10);
  L.scr = o;
  // EVAL
  o = L.scr;
  L.e_x = take((N + 1) * nq); L.e_ue = take(n);
  for (int a = 0; a < D.M; a++) { L.e_A[a] = take(N * D.nqa[a] * D.nqa[a]); L.e_B[a] = take(N * D.nqa[a] * DGSQP_NUA); }
  L.e_dJ = take((N + 1) * nq);
  L.e_lam = take(D.M * (N + 1) * nq);
  if (D.tab_const) { L.e_Dxs = L.e_lam; L.e_K = o; }      // costate recursion in place; e_K in the global scratch
  else { L.e_Dxs = take(D.M * (N + 1) * nq); L.e_K = take(D.M * (N + 1) * D.M * 5); }
  int eval_end = o;
  // EIG: packed P, packed Householder reflectors, tridiagonal workspace
  o = L.scr;
  const int npk = n * (n + 1) / 2;
  const int rpt = (n <= 32 ? 32 : (n <= 64 ? 64 : (n <= 100 ? 100 : 128))) / DG_NH;
  if (!D.big) { L.g_Bp = take(npk); L.g_V = take(npk); } else { L.g_Bp = L.g_V = -1; }
  const int xl_strips = (DG_BLOCK / 64) * 3 * n + 16, xl_packed = npk + DG_NH * n + 16;     // (the packed matrix + the partial sums of its products share the strips' slot)
  L.g_tw = D.big == 2 ? take(6 * n + DG_XL_KMAX + 16 + (D.xl_pack && xl_packed > xl_strips ? xl_packed : xl_strips))   // d, e, tau, v, w, e^2, lambda, per-wavefront strips
                      : take((5 + DG_NH) * n + 16 + DG_PSD_KMAX * n /* Z */ + 3 * (DG_NH * rpt + 4)
                             + 2 * (DG_BLOCK / 64) * (DG_NH * rpt + 4) /* per-wavefront copies of the reflector v and of w */);
  // per-wavefront strips of the twisted factorisation (3 n doubles each): the packed-P slot is free until the sweep writes
  // P at the very end of the phase; tiny problems get their own space
  const int strips = (DG_BLOCK / 64) * 3 * ((n + 1) & ~1);
  L.g_strip = (!D.big && strips <= npk) ? L.g_Bp : (D.big == 2 ? L.g_tw : take(strips));
  const int eig_end = o;
  // QP (P aliases Bp)
  o = L.scr + (D.big ? 0 : ((npk + 1) & ~1));
  {
    int tslot = D.classic_qp ? npk : dg_tcol(n);   // (the dual method's inverse factor T: padded columns, dg_tcol)
    const int ncp = (nc + 1) & ~1, np = (n + 1) & ~1;
    const bool osqp_lds = D.osqp && D.big != 2;     // (the XL layout's OSQP places its own vectors below: ox_*)
    if (osqp_lds) tslot = dg_tcol(D.osqp_nacap < n ? D.osqp_nacap : n) + (D.osqp_nacap < n ? 5 * np : 0);
    const int tabsz = 2 * n + (n + 8) / 4 + (D.ngd + 2) / 2 + 2;       // 8 int16 per column, n + 1 uint16 column starts, one uint32 per gradient entry
    if (osqp_lds && tslot < 5 * ncp + tabsz + 5 * np) tslot = 5 * ncp + tabsz + 5 * np;
    L.p_R = D.big == 2 ? -1 : take(tslot);
    if (osqp_lds) {
      L.a_z = L.p_R; L.a_y = L.a_z + ncp; L.a_E = L.a_y + ncp; L.a_dy = L.a_E + ncp; L.a_w = L.a_dy + ncp; L.a_tab = L.a_w + ncp;
      L.a_tail = L.p_R + ((tslot - 5 * np) & ~1);
      int na = 0;
      while (na < n && dg_tcol(na + 1) <= L.a_tail - L.p_R) na++;
      D.osqp_namax = na;
    }
  }
  L.p_lam = take(n + 1); L.p_c = take(n + 9); L.p_w = take(n + 9); L.p_r = take(n + 9);   // (c, w are read in groups of eight: zero tail)
  L.p_y = take(n); L.p_t = take(n); L.p_alist = take((n + 2) / 2 + 1); L.p_rd = take(n + 1);
  if (!D.tab_const && D.big != 2) {   // the dual start borrows p_yd2 / p_dpart while its own vectors (5 of length n_c) sit at L.scr: keep them apart
    const int lsqr_size = 4 * ((nc + 1) & ~1) + (((nc > n ? nc : n) + 1) & ~1);
    if (o < L.scr + lsqr_size) o = L.scr + lsqr_size;
  }
  L.p_yd2 = take(nd); L.p_yslot = take((n + 2) / 2 + 1); L.p_yfree = take((n + 2) / 2 + 1); L.p_dpart = take(D.ntask); L.p_act = take(nc / 8 + 2); L.p_part = take(DG_NH * n);
  if (D.osqp && D.big != 2) {
    // (the ADMM's delta-y and work vectors are free whenever a matrix is inverted: before the loop, after a check, in the polish)
    const int need = 4 * (DG_NH * rpt + 4);
    L.a_sw = 2 * ((nc + 1) & ~1) >= need ? L.a_dy : take(need);
  }
  L.p_xu = L.p_rd;     // unconstrained minimiser of the QP, x = x_u - Y lam at every point of the dual method (the slot the reciprocal diagonal of the Cholesky factor had)
  const int qp_end = o;
  // QP outputs live past the end of both the QP and EVAL scratch
  o = qp_end > eval_end ? qp_end : eval_end;
  L.o_du = take(n); L.o_lhat = take(nc);
  const int out_end = o;
  // LSQR: over the evaluation arrays -- or above them when the EIG / QP scratch reaches further anyway (LDS-resident layout): the
  // evaluation of the start point then survives the dual start
  {
    const int lsqr_size = 4 * ((nc + 1) & ~1) + (((nc > n ? nc : n) + 1) & ~1);
    const int reach = eig_end > out_end ? eig_end : out_end;
    D.lsqr_keeps_eval = (!D.big && !D.tab_const && eval_end + lsqr_size <= reach && eval_end + lsqr_size <= L.p_yd2 && !getenv("DGSQP_LSQR_LOW")) ? 1 : 0;
  }
  o = D.lsqr_keeps_eval ? eval_end : L.scr;
  L.s_u = take(nc); L.s_v = take(nc); L.s_w = take(nc); L.s_x = take(nc); L.s_t = take(nc > n ? nc : n);
  if (D.tab_const || D.big == 2) { L.s_yd2 = take(nd); L.s_dpart = take(D.ntask); }       // (the QP outputs are dead during the dual start)
  else { L.s_yd2 = L.p_yd2; L.s_dpart = L.p_dpart; }
  const int lsqr_end = o;
  if (L.s_yd2 < L.s_t + (nc > n ? nc : n) || L.s_dpart < L.s_t + (nc > n ? nc : n)) return "internal layout error: the dual start's dot scratch overlaps its vectors";
  {
    // give the speculative rollouts the scratch the EIG / QP phases need anyway (the arena does not grow for them)
    int tot0 = eig_end > out_end ? eig_end : out_end;
    if (lsqr_end > tot0) tot0 = lsqr_end;
    const int xsz = ((N + 1) * nq + 1) & ~1;
    bool all_dyn = true;
    for (int a = 0; a < D.M; a++) all_dyn = all_dyn && D.nqa[a] == 8;
    const int lanes_per_traj = all_dyn ? 2 * D.M : D.M;
    L.e_xs = eval_end > lsqr_end ? eval_end : lsqr_end;   // above everything a trial evaluation / merit touches
    int K = (L.o_du - L.e_xs) / xsz;                      // must end below the QP outputs (du, lhat are live during trials)
    if (K > 64 / lanes_per_traj) K = 64 / lanes_per_traj;
    if (K > 16) K = 16;
    if (K < 1) K = 0;
    // second segment above the QP outputs, up to the LDS limit (phase-multiplexed with the EIG workspace)
    int cap = 64 / lanes_per_traj; if (cap > 16) cap = 16;
    L.e_xs2 = (out_end + 1) & ~1;
    int K2 = K > 0 ? (DG_LDS_LIMIT / 8 - L.e_xs2) / xsz : 0;
    if (K2 > cap - K) K2 = cap - K;
    if (K2 < 0) K2 = 0;
    D.ls_spec1 = K;
    D.ls_spec = K + K2;
    if (L.e_xs + K * xsz > eval_end) eval_end = L.e_xs + K * xsz;
    if (K2 > 0 && L.e_xs2 + K2 * xsz > eval_end) eval_end = L.e_xs2 + K2 * xsz;
  }
  int tot = eval_end;
  if (eig_end > tot) tot = eig_end;
  if (out_end > tot) tot = out_end;
  if (lsqr_end > tot) tot = lsqr_end;
  // classical QP: as many columns of R as the arena still holds above the QP outputs (the trial trajectories of e_xs2 are dead
  // while a QP runs); a triangular solve per added row reads R column by column -- from LDS that is 10x less latency than from L2
  // XL, packed elimination: from the end of the QP scratch to the end of the arena (over the QP outputs and the R columns: all dead
  // until J = L^-T has been built)
  L.x_el = (qp_end + 1) & ~1;
  D.xl_el = D.big == 2 && D.xl_pack && L.x_el + npk + 2 <= DG_LDS_LIMIT / 8 && !getenv("DGSQP_XL_NOEL");
  if (D.xl_el && L.x_el + npk + 2 > tot) tot = L.x_el + npk + 2;       // (the triangle is part of the arena whatever R gets)
  D.xl_blk = D.big == 2 && !D.xl_el && n >= 128 && L.x_el + 16 * n + 2 <= DG_LDS_LIMIT / 8 && !getenv("DGSQP_XL_NOBLKEL");
  if (D.xl_blk && L.x_el + 16 * n + 2 > tot) tot = L.x_el + 16 * n + 2;
  L.c_R = (out_end + 1) & ~1;
  D.c_rcap = 0;
  if (D.classic_qp) {
    const int avail = DG_LDS_LIMIT / 8 - L.c_R;
    int cap = 0;
    while (cap < n && (cap + 1) * (cap + 2) / 2 <= avail) cap++;
    if (const char* e = getenv("DGSQP_RCAP")) { const int lim = atoi(e); if (lim >= 0 && lim < cap) cap = lim; }   // test hook: force the LDS / scratch split of R
    D.c_rcap = cap;
    if (L.c_R + cap * (cap + 1) / 2 > tot) tot = L.c_R + cap * (cap + 1) / 2;
  }
  if (D.osqp && D.big == 2) {
    // OSQP on the XL layout: its vectors go where the QP scratch of the active-set kernels and the (dead) evaluation arrays are -- below
    // the QP outputs (which hold x and y) -- and, what does not fit there, above them (trial trajectories / columns of R: dead or unused)
    const int ncp = (nc + 1) & ~1, np = (n + 1) & ~1;
    const int wslot = ncp > 4 * np ? ncp : 4 * np;
    const int dp_len = D.ntask > D.ox_nya ? D.ntask : D.ox_nya;       // (the dense-dot partials; between two products the slot holds the per-agent multiplier lists of G' w)
    int dslot = ((dp_len + 1) & ~1) + ((nd + 1) & ~1) + 2;
    if (dslot < ncp) dslot = ncp;
    int r1 = wslot + dslot;
    if (r1 < 16 * n + 2) r1 = 16 * n + 2;
    int a = L.scr, b = (out_end + 1) & ~1;
    const int aend = L.o_du, bend = DG_LDS_LIMIT / 8;
    bool fits = true;
    auto place = [&](int cnt) -> int {
      cnt = (cnt + 1) & ~1;
      if (a + cnt <= aend) { const int r = a; a += cnt; return r; }
      if (b + cnt <= bend) { const int r = b; b += cnt; return r; }
      fits = false;
      return L.scr;
    };
    L.ox_w = place(r1); L.ox_dpart = L.ox_w + wslot; L.ox_yd2 = L.ox_dpart + ((dp_len + 1) & ~1);
    L.ox_z = place(ncp);
    L.ox_np = np; L.ox_nv = place(7 * np);
    L.ox_part = place(DG_NH * n > 512 ? DG_NH * n : 512);
    L.ox_alist = place((n + 2) / 2 + 1);
    if (!fits) {
      if (!D.tab_const) { D.tab_const = 1; return dg_build_layout(D); }      // the tables out of LDS first
      char buf[200];
      snprintf(buf, sizeof buf, "qp_method OSQP: n=%d n_c=%d does not fit the LDS arena next to the problem's vectors (XL layout)", n, nc);
      return buf;
    }
    if (b > tot) tot = b;
  }
  L.total = tot;
  if ((long)tot * 8 > DG_LDS_LIMIT && !D.big) { D.big = 1; return dg_build_layout(D); }   // (n > 128 starts at big = 2)
  if ((long)tot * 8 > DG_LDS_LIMIT && D.big == 1 && DG_WG_PER_CU > 1 && !D.osqp && !D.gd_global) { D.gd_global = 1; return dg_build_layout(D); }   // half-arena build: the packed gradients to the scratch as well
  if ((long)tot * 8 > DG_LDS_LIMIT && D.osqp && D.big == 1 && D.osqp_nacap > 24) { D.osqp_nacap -= 8; return dg_build_layout(D); }   // OSQP: a smaller T slot (polish of fewer active rows)
  if ((long)tot * 8 > DG_LDS_LIMIT && D.big == 2 && D.xl_pack) { D.xl_pack = 0; D.gd_global = 0; return dg_build_layout(D); }   // no room for the packed matrix: the plain XL layout
  if ((long)tot * 8 > DG_LDS_LIMIT && D.big == 2 && !D.gd_global) { D.gd_global = 1; return dg_build_layout(D); }
  if ((long)tot * 8 > DG_LDS_LIMIT && D.big == 2 && !D.tab_const) { D.tab_const = 1; return dg_build_layout(D); }
  if ((long)tot * 8 > DG_LDS_LIMIT) {
    char buf[160];
    snprintf(buf, sizeof buf, "problem needs %ld B of LDS per scenario (limit %d): n=%d n_c=%d", (long)tot * 8, DG_LDS_LIMIT, n, nc);
    return buf;
  }
  if (getenv("DGSQP_LAYOUT_DEBUG"))
    fprintf(stderr, "dgsqp layout: n=%d nc=%d big=%d pack=%d scr=%d g_tw=%d eig_end=%d qp_end=%d out_end=%d x_el=%d c_R=%d rcap=%d total=%d (limit %d doubles)\n",
            n, nc, D.big, D.xl_pack, L.scr, L.g_tw, eig_end, qp_end, out_end, L.x_el, L.c_R, D.c_rcap, tot, DG_LDS_LIMIT / 8);
  return "";
}

// Build the device-side problem description. Returns empty string on success, else an error message.
static inline std::string dg_build(const dgsqp_problem_t& P, const dgsqp_params_t& par, DgProb& D) {
  memset(&D, 0, sizeof(D));
  D.P = P; D.par = par;
  D.P.spline = nullptr;          // the table is copied below; the host pointer is not part of the device description
  if (P.track_kind == DGSQP_TRACK_SPLINE) {
    if (P.n_knots < 3 || P.n_knots > DGSQP_MAX_KNOTS || !P.spline) return "bad spline track table (3 <= n_knots <= DGSQP_MAX_KNOTS)";
    memcpy(D.spl, P.spline, sizeof(double) * (size_t)(9 * P.n_knots - 8));
    for (int i = 0; i + 1 < P.n_knots; i++) if (!(P.spline[i + 1] > P.spline[i])) return "spline knots must increase";
    if (P.spline[0] != 0.0 || fabs(P.spline[P.n_knots - 1] - P.track_L) > 1e-9 * P.track_L) return "spline knots must span [0, track_L]";
  } else if (P.track_kind != DGSQP_TRACK_ARCS) return "unknown track kind";
  if (P.M < 1 || P.M > DGSQP_MAX_AGENTS) return "unsupported number of agents";
  if (P.N < 1 || P.N > DG_NMAX) return "unsupported horizon";
  if (par.merit_function == DGSQP_MERIT_SUM_OBJ_L1 && par.variant != DGSQP_VARIANT_V2) return "merit function sum_obj_l1 belongs to DG-SQP v2";
  if (P.N * P.M * DGSQP_NUA > DG_NVARMAX) return "more than 320 decision variables are not supported yet";
  if (par.qp_method != DGSQP_QP_ACTIVE_SET && par.qp_method != DGSQP_QP_OSQP) return "unknown qp_method";
  if (P.n_segs < 1 || P.n_segs > DGSQP_MAX_SEGS) return "bad track table";
  D.M = P.M; D.N = P.N; D.nq = 0; D.nu = P.M * DGSQP_NUA;
  int t2 = 0;
  D.uniform_nqa = 1;
  D.inv_track_L = 1.0 / P.track_L;
  D.eig_floor = par.eig_floor > 0.0 ? par.eig_floor : 1e-10;
  for (int a = 0; a < P.M; a++) {
    if (P.agents[a].model != P.agents[0].model) D.uniform_nqa = 0;
    const int mdl = P.agents[a].model;
    if (mdl != DGSQP_MODEL_KIN_BICYCLE && mdl != DGSQP_MODEL_DYN_BICYCLE && mdl != DGSQP_MODEL_UNICYCLE) return "unsupported vehicle model";
    D.nqa[a] = mdl == DGSQP_MODEL_DYN_BICYCLE ? 8 : (mdl == DGSQP_MODEL_UNICYCLE ? 4 : 6);
    D.qoff[a] = D.nq; D.nq += D.nqa[a];
    // Frenet states s, e_y are the last two of both bicycles.  The unicycle has none: its slots (v, psi) only ever meet
    // zero progress / competition / blocking weights, and carry the goal-tracking curvature of the last two states.
    D.sidx[a] = D.nqa[a] - 2; D.eyidx[a] = D.nqa[a] - 1;
    if (mdl == DGSQP_MODEL_UNICYCLE && (P.agents[a].w_prog != 0.0 || P.agents[a].w_comp != 0.0 || P.agents[a].w_block != 0.0))
      return "progress / competition / blocking costs need a Frenet-frame model";
    if (P.agents[a].n_lane < 0 || P.agents[a].n_lane > DGSQP_MAX_LANES) return "bad number of lane rows";
    for (int i = 2; i < D.nqa[a] - 2; i++)
      if (P.agents[a].w_goal[i] != 0.0) return "goal-tracking weights are supported on the positions and the last two states";
    // x, y (state 0,1) never enter fc: effective variables are the remaining states and the two inputs
    D.neff[a] = D.nqa[a];  // (nqa-2) states + 2 inputs
    for (int e = 0; e < D.nqa[a] - 2; e++) D.effvar[a][e] = e + 2;
    D.effvar[a][D.nqa[a] - 2] = D.nqa[a]; D.effvar[a][D.nqa[a] - 1] = D.nqa[a] + 1;
    D.ndir[a] = D.neff[a] * (D.neff[a] + 1) / 2;
    D.t2off[a] = t2; D.t2k[a] = D.nqa[a] * D.ndir[a];
    t2 += P.N * D.t2k[a];
  }
  D.n = P.N * D.nu;
  D.npairs = P.obstacle_rows ? P.M * (P.M - 1) / 2 : 0;
  for (int a = 0; a < DGSQP_MAX_AGENTS; a++)
    for (int k = 0; k < DG_NMAX; k++)
      for (int j = 0; j < DGSQP_NUA; j++) D.r_in_ub[a][k][j] = D.r_in_lb[a][k][j] = D.r_rate_ub[a][k][j] = D.r_rate_lb[a][k][j] = -1;
  // ---- rows in reference order, dense gradients in (stage, [pairs], agent, state idx) order
  int nc = 0, nd = 0, off = 0;
  auto add_row = [&](int type, int k, int a, int b, int idx, int sgn, int dense) -> bool {
    if (nc >= DG_NCMAX) return false;
    D.rows[nc] = DgRow{(int8_t)type, (int8_t)a, (int8_t)b, (int8_t)idx, (int8_t)sgn, (int8_t)k, (int16_t)dense};
    nc++;
    return true;
  };
  for (int k = 0; k <= P.N; k++) {
    D.stage_row0[k] = (int16_t)nc; D.stage_dense0[k] = (int16_t)nd;
    if (P.obstacle_rows && k >= 1)
      for (int i = 0; i < P.M; i++)
        for (int j = i + 1; j < P.M; j++) {
          if (nd >= DG_NDMAX) return "too many dense rows";
          D.dense[nd] = DgDense{1, (int8_t)i, (int8_t)j, 0, (int8_t)k, 0, 0, 0, off, (int16_t)nc, -1};
          off += 4 * k;
          if (!add_row(DG_R_OBS, k, i, j, 0, 1, nd)) return "too many rows";
          nd++;
        }
    for (int a = 0; a < P.M; a++) {
      const dgsqp_agent_t& ag = P.agents[a];
      if (k < P.N) {
        if (ag.has_rate)
          for (int j = 0; j < DGSQP_NUA; j++) {
            D.r_rate_ub[a][k][j] = (int16_t)nc; if (!add_row(DG_R_RATE_UB, k, a, -1, j, 1, -1)) return "too many rows";
            D.r_rate_lb[a][k][j] = (int16_t)nc; if (!add_row(DG_R_RATE_LB, k, a, -1, j, -1, -1)) return "too many rows";
          }
      }
      // lane half-planes at every stage, k = 0 included (a gradient of length 0: the row only depends on x_0)
      for (int j = 0; j < ag.n_lane; j++) {
        if (nd >= DG_NDMAX) return "too many dense rows";
        D.dense[nd] = DgDense{2, (int8_t)a, -1, (int8_t)j, (int8_t)k, 0, 0, 0, off, (int16_t)nc, -1};
        off += 2 * k;
        if (!add_row(DG_R_LANE, k, a, -1, j, 1, nd)) return "too many rows";
        nd++;
      }
      if (k < P.N) {
        for (int j = 0; j < DGSQP_NUA; j++)
          if (ag.in_ub[j] < INFINITY) { D.r_in_ub[a][k][j] = (int16_t)nc; if (!add_row(DG_R_IN_UB, k, a, -1, j, 1, -1)) return "too many rows"; }
        for (int j = 0; j < DGSQP_NUA; j++)
          if (ag.in_lb[j] > -INFINITY) { D.r_in_lb[a][k][j] = (int16_t)nc; if (!add_row(DG_R_IN_LB, k, a, -1, j, -1, -1)) return "too many rows"; }
      }
      if (k > 0) {
        // one dense gradient per constrained state, shared by its ub and lb rows
        int dense_of[DGSQP_MAX_NQA];
        for (int i = 0; i < D.nqa[a]; i++) {
          dense_of[i] = -1;
          if (ag.st_ub[i] < INFINITY || ag.st_lb[i] > -INFINITY) {
            if (nd >= DG_NDMAX) return "too many dense rows";
            D.dense[nd] = DgDense{0, (int8_t)a, -1, (int8_t)i, (int8_t)k, 0, 0, 0, off, -1, -1};
            off += 2 * k;
            dense_of[i] = nd++;
          }
        }
        for (int i = 0; i < D.nqa[a]; i++)
          if (ag.st_ub[i] < INFINITY) { D.dense[dense_of[i]].r_pos = (int16_t)nc; if (!add_row(DG_R_ST_UB, k, a, -1, i, 1, dense_of[i])) return "too many rows"; }
        for (int i = 0; i < D.nqa[a]; i++)
          if (ag.st_lb[i] > -INFINITY) { D.dense[dense_of[i]].r_neg = (int16_t)nc; if (!add_row(DG_R_ST_LB, k, a, -1, i, -1, dense_of[i])) return "too many rows"; }
      }
    }
  }
  D.stage_row0[P.N + 1] = (int16_t)nc; D.stage_dense0[P.N + 1] = (int16_t)nd;
  D.nc = nc; D.ndense = nd; D.ngd = off;
  {
    int nt = 0;
    for (int d = 0; d < nd; d++) {
      DgDense& dd = D.dense[d];
      const int len = DGSQP_NUA * dd.k, parts = dd.kind == 1 ? 2 : 1;
      dd.t0lo = (uint8_t)(nt & 255); dd.t0hi = (uint8_t)(nt >> 8);
      int cnt = 0;
      for (int part = 0; part < parts; part++) {
        const int vb = (part == 0 ? dd.a : dd.b) * P.N * DGSQP_NUA;
        for (int c0 = 0; c0 < len; c0 += DG_CHUNK) {
          if (nt >= DG_NTASKMAX) return "too many dense-gradient chunks";
          D.dtask[nt++] = DgTask{dd.off + part * len + c0, (uint16_t)(vb + c0), (uint8_t)(len - c0 < DG_CHUNK ? len - c0 : DG_CHUNK), 0};
          cnt++;
        }
      }
      dd.nt = (uint8_t)cnt;
    }
    D.ntask = nt;
  }
  D.t2_doubles = t2;
  D.osqp = par.qp_method == DGSQP_QP_OSQP ? 1 : 0;
  D.osqp_nacap = D.n;
  if (D.osqp && D.n <= 128 && (D.ngd >= 65536 || D.ndense >= 65536)) return "qp_method OSQP: the packed gradients exceed the 16-bit offsets of its index tables";
  if (D.osqp && D.n > 128 && D.ngd >= (1 << 22)) return "qp_method OSQP: the packed gradients exceed the 22-bit offsets of the XL layout's index table";
  D.big = D.n > 128 ? 2 : 0;   // XL layout: every matrix of the PSD / QP phases in the global scratch, generic (slow) kernels
  D.xl_noblock = getenv("DGSQP_XL_NOBLOCK") ? 1 : 0;
  if (D.big == 0 && getenv("DGSQP_FORCE_BIG")) D.big = 1;                      // (development knobs: the big layout / the gradients in the scratch for a game that would fit without)
  if (D.big == 1 && !D.osqp && getenv("DGSQP_FORCE_GD_GLOBAL")) D.gd_global = 1;
  if (D.big == 2 && D.n <= 176 && !getenv("DGSQP_XL_NOPACK")) { D.xl_pack = 1; D.gd_global = 1; }       // (tried first; dg_build_layout falls back when the arena overflows)
  return dg_build_layout(D);
}

