// XL layout (128 < n <= 320): _nearestPD + _solve_qp for games whose matrices do not fit the LDS-resident layouts
// (BASELINE configs[2]: 3 agents, N=25, n=150; scripts/DGSQP_monte_carlo_agents.py at its M=3, N=25 setting).
// Every n x n matrix lives in the workgroup's global scratch (L2); the kernels are plain block-wide loops -- a correct,
// unoptimised path with the same semantics as the fast one:
//   _nearestPD (DGSQP.py:1290-1296): eigen-decomposition of B = (Q+Q^T)/2 by one-sided (Hestenes) Jacobi rotations,
//                M = B + sum_{lambda_j < 0} (floor - lambda_j) v_j v_j^T + reg I;
//   _solve_qp  (DGSQP.py:232-266): Goldfarb-Idnani dual active set in the classical J = L^-T / R form, cold start, the same
//                pivot rules and tolerances as the test suite's CPU restatement (most violated row first, lowest index on ties).
#pragma once
#include "dgsqp_solve.h"

#define XSYNC() do { __threadfence_block(); __syncthreads(); } while (0)
#define DG_XL_BASIS 49   // scal slot: the saved eigenvector basis of the Jacobi _nearestPD is valid (XL layout)
#define XL_NV 5      // registers per lane for one column (n <= 320 = DG_NVARMAX)
#define XL_MAXP 20   // pairs of one tournament round per wavefront (160 pairs / 8 wavefronts)
static_assert(DG_BLOCK != 512 || (64 * XL_NV >= DG_NVARMAX && XL_MAXP * (DG_BLOCK / 64) * 2 >= DG_NVARMAX), "XL kernels: register tiling must cover DG_NVARMAX columns");
#ifndef XL_GRP
#define XL_GRP 4     // ... handled four at a time (their four columns each stay in registers)
#endif

// Thread (g, i) of the products with J: column / row i, the g-th part of the summation range (n <= 256: two to four parts).
struct XlSplit { int G, g, i; };
__device__ inline XlSplit xl_split(int n) {
  const int npad = (n + 31) & ~31;
  XlSplit S;
  S.G = NT / npad; if (S.G > DG_NH) S.G = DG_NH;
  S.g = TID / npad; S.i = TID - S.g * npad;
  return S;
}

// ---- rank-16 update of a matrix in the L2 scratch on the matrix cores:  C[i][j] += sum_{kk < 16} a_of(kk, i) * b_of(kk, j)  over the
// 16 x 16 tiles tile_of(t) = (ti, tj), t < ntile (tile index t covers rows / columns 16 t .. 16 t + 15; elements outside n x n are neither read nor written).
// v_mfma_f64_16x16x4_f64: lane l holds A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15] and the results C[(l >> 4) + 4 r][l & 15],
// r = 0 .. 3 -- every register of a tile is four 128-byte row segments.  The operands come from LDS one value per lane and MFMA
// (the VALU form of this update needs 16 broadcast LDS reads per ROW of 64 results: the LDS pipe, not the ALU, bounds it);
// four tiles per wavefront and pass keep 16 L2 round trips in flight.  Same fp64 FMAs as the VALU form, in the MFMA's order.
typedef double xl_v4d __attribute__((ext_vector_type(4)));
// General form: result(i, j) = c_of(i, j) + sum_{kk < 16 nk16} a_of(kk, i) b_of(kk, j), handed to s_of(i, j, value).
template <class FT, class FA, class FB, class FC, class FS>
__device__ __forceinline__ void xl_mfma_tiles(int n, int ntile, int nk16, FT tile_of, FA a_of, FB b_of, FC c_of, FS s_of) {
  const int lane = TID & 63, wave = TID >> 6, li = lane & 15, h = lane >> 4;
  constexpr int U = 4;
  for (int t0 = wave * U; t0 < ntile; t0 += (NT / 64) * U) {
    xl_v4d acc[U];
    int row0[U], col[U], rowa[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int t = t0 + u < ntile ? t0 + u : ntile - 1;      // (a short last pass repeats its last tile's loads; nothing is stored)
      int ti, tj;
      tile_of(t, ti, tj);
      row0[u] = 16 * ti + h; col[u] = 16 * tj + li; rowa[u] = 16 * ti + li;
#pragma unroll
      for (int r = 0; r < 4; r++) { const int row = row0[u] + 4 * r; acc[u][r] = (row < n && col[u] < n) ? c_of(row, col[u]) : 0.0; }
    }
    for (int kc = 0; kc < nk16; kc++) {
      double av[U][4], bv[U][4];
#pragma unroll
      for (int u = 0; u < U; u++) {
#pragma unroll
        for (int q = 0; q < 4; q++) { av[u][q] = a_of(16 * kc + 4 * q + h, rowa[u]); bv[u][q] = b_of(16 * kc + 4 * q + h, col[u]); }
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
#pragma unroll
        for (int q = 0; q < 4; q++) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u][q], bv[u][q], acc[u], 0, 0, 0);
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (t0 + u < ntile) {
#pragma unroll
        for (int r = 0; r < 4; r++) { const int row = row0[u] + 4 * r; if (row < n && col[u] < n) s_of(row, col[u], acc[u][r]); }
      }
    }
  }
}
template <class FT, class FA, class FB>
__device__ __forceinline__ void xl_mfma_rank16(gptr C, int ldc, int n, int ntile, FT tile_of, FA a_of, FB b_of) {
  xl_mfma_tiles(n, ntile, 1, tile_of, a_of, b_of, [&](int i, int j) { return C[(int64_t)i * ldc + j]; },
                [&](int i, int j, double v) { C[(int64_t)i * ldc + j] = v; });
}

// ---- _nearestPD through the tridiagonal form (the same algorithm as the fast layouts, written generically):
//   Householder tridiagonalisation of B (block-wide, B and the reflectors in the L2 scratch), Sturm-count multisection
//   for the NEGATIVE eigenvalues only, eigenvectors by twisted factorisation (one wavefront each), modified Gram-Schmidt,
//   back-transformation, M = B + sum_j (floor - lambda_j) v_j v_j^T + reg I.
// Returns false (nothing written) when there are more than XL_KMAX negative eigenvalues: the Jacobi path takes over.
#define XL_KMAX DG_XL_KMAX
#define XL_RCH 16      // Givens rotations applied to a row of J per pass
#define XL_PB 8        // columns per panel of the blocked tridiagonalisation (matrix in the L2 scratch)
#define XL_SEG 8       // 16-column segments of a row of J one thread updates per pass (loads first, stores after)
__device__ __noinline__ bool dev_xl_psd_tri(const Ctx& c, gptr Qpd) {
  const DgProb& D = dg_prob;
  const int n = D.n, lane = TID & 63, wave = TID >> 6;
  lptr lds = LP(0);
  cgptr Qg = c.ws + D.ws_q;
  gptr Bm = c.ws + D.ws_P, Vr = c.ws + D.ws_V, Mx = c.ws + D.ws_R, Z = c.ws + D.ws_Vp;   // Z: XL_KMAX x n (the Jacobi basis slot)
  lptr W = lds + D.L.g_tw;
  lptr dv = W, ev = W + n, tau = W + 2 * n, vv = W + 3 * n, pv = W + 4 * n, e2 = W + 5 * n, lamv = W + 6 * n;   // lamv: XL_KMAX
  lptr strips = W + 6 * n + XL_KMAX + 16;                                            // 3 n per wavefront
  lds_d* red = lds + D.L.red;
  lds_d* scal = lds + D.L.scal;
  lptr part = strips;                 // (the strips are free until the eigenvector phase; the QP scratch overlaps this workspace)
  const XlSplit S = xl_split(n);
  if (TID == 0) { scal[DG_XVALID] = 0.0; scal[DG_XL_BASIS] = 0.0; }   // (Z overwrites the Jacobi warm-start basis)
  __syncthreads();
  PROF_BEGIN(pt_t);
  if (D.xl_pack) {
    // ---- Householder tridiagonalisation with the matrix in LDS (packed lower triangle, row i at i (i + 1) / 2): the same reduction as
    // the loop below, whose every step pays three dependent L2 round trips (column, product, rank-2 update) -- 2.4 of the 9 Mcycles of
    // a QP iteration at n = 150.  Thread (g, i) owns row i and the g-th part of its column range, in the product and in the update.
    lptr Bp = strips, part2 = strips + n * (n + 1) / 2;
    for (int e = TID; e < n * n; e += NT) {
      const int i = e / n, k = e - i * n;
      if (k <= i) Bp[i * (i + 1) / 2 + k] = 0.5 * (Qg[(int64_t)i * n + k] + Qg[(int64_t)k * n + i]);
    }
    __syncthreads();
    for (int k = 0; k + 2 < n; k++) {
      double s2 = 0;
      for (int i = k + 1 + TID; i < n; i += NT) { const double x = Bp[i * (i + 1) / 2 + k]; vv[i] = x; s2 += x * x; }
      const double nrm2 = block_sum(s2, red);
      const double x0 = vv[k + 1];
      const double tail2 = nrm2 - x0 * x0;
      double alpha, beta;
      if (!(tail2 > 0.0)) { alpha = x0; beta = 0.0; }
      else { alpha = x0 >= 0 ? -sqrt(nrm2) : sqrt(nrm2); const double v0 = x0 - alpha; beta = 2.0 / (tail2 + v0 * v0); }
      __syncthreads();
      if (TID == 0) { dv[k] = Bp[k * (k + 1) / 2 + k]; ev[k] = alpha; tau[k] = beta; if (beta != 0.0) vv[k + 1] = x0 - alpha; }
      __syncthreads();
      for (int i = k + 1 + TID; i < n; i += NT) Vr[(int64_t)k * n + i] = vv[i];
      if (beta != 0.0) {
        const bool on = S.g < S.G && S.i > k && S.i < n;
        if (on) {      // p_i = sum_j B(i, j) v_j over the g-th part of j in (k, n): row i up to the diagonal, column i below it
          const int i = S.i, len = n - k - 1, ja = k + 1 + (S.g * len) / S.G, jb = k + 1 + ((S.g + 1) * len) / S.G;
          clptr row = Bp + i * (i + 1) / 2;
          double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
          int j = ja;
          const int jm = jb < i + 1 ? jb : i + 1;
          for (; j + 7 < jm; j += 8) {       // eight LDS reads of the matrix in flight (and eight broadcast reads of v)
            double b[8], w[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { b[u] = row[j + u]; w[u] = vv[j + u]; }
            a0 += b[0] * w[0] + b[4] * w[4]; a1 += b[1] * w[1] + b[5] * w[5]; a2 += b[2] * w[2] + b[6] * w[6]; a3 += b[3] * w[3] + b[7] * w[7];
          }
          for (; j < jm; j++) a0 += row[j] * vv[j];
          for (; j + 7 < jb; j += 8) {
            double b[8], w[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { b[u] = Bp[(j + u) * (j + u + 1) / 2 + i]; w[u] = vv[j + u]; }
            a0 += b[0] * w[0] + b[4] * w[4]; a1 += b[1] * w[1] + b[5] * w[5]; a2 += b[2] * w[2] + b[6] * w[6]; a3 += b[3] * w[3] + b[7] * w[7];
          }
          for (; j < jb; j++) a0 += Bp[j * (j + 1) / 2 + i] * vv[j];
          part2[S.g * n + i] = (a0 + a1) + (a2 + a3);
        }
        __syncthreads();
        double pvsum = 0;
        for (int i = k + 1 + TID; i < n; i += NT) {
          double a = part2[i];
          for (int g = 1; g < S.G; g++) a += part2[g * n + i];
          const double pq = beta * a;
          pv[i] = pq; pvsum += pq * vv[i];
        }
        const double K = 0.5 * beta * block_sum(pvsum, red);
        for (int i = k + 1 + TID; i < n; i += NT) pv[i] -= K * vv[i];       // w
        __syncthreads();
        if (on) {      // B(i, j) -= v_i w_j + w_i v_j on the g-th part of row i's columns (k, i]
          const int i = S.i, len = i - k, ja = k + 1 + (S.g * len) / S.G, jb = k + 1 + ((S.g + 1) * len) / S.G;
          lptr row = Bp + i * (i + 1) / 2;
          const double vi = vv[i], wi = pv[i];
          int j = ja;
          for (; j + 7 < jb; j += 8) {       // loads first, stores after
            double b[8], wj[8], vj[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { b[u] = row[j + u]; wj[u] = pv[j + u]; vj[u] = vv[j + u]; }
#pragma unroll
            for (int u = 0; u < 8; u++) row[j + u] = b[u] - (vi * wj[u] + wi * vj[u]);
          }
          for (; j < jb; j++) row[j] -= vi * pv[j] + wi * vv[j];
        }
      }
      __syncthreads();
    }
    if (TID == 0) {
      dv[n - 2] = Bp[(n - 2) * (n - 1) / 2 + n - 2]; dv[n - 1] = Bp[(n - 1) * n / 2 + n - 1];
      ev[n - 2] = Bp[(n - 1) * n / 2 + n - 2]; ev[n - 1] = 0.0; tau[n - 2] = 0.0; tau[n - 1] = 0.0;
    }
    __threadfence_block();
    __syncthreads();
  } else {
  for (int e = TID; e < n * n; e += NT) {
    const int i = e / n, k = e % n;
    Bm[e] = 0.5 * (Qg[(int64_t)i * n + k] + Qg[(int64_t)k * n + i]);
  }
  // ---- Householder tridiagonalisation, BLOCKED (the matrix lives in the L2 scratch: n = 200 .. 320).  The plain reduction reads the
  // trailing block for the product B v and reads + writes it again for the rank-2 update, every column: three L2 passes per column,
  // 14 of the 95 Mcycles of a QP iteration at n = 300.  Here the updates of XL_PB columns are held back as a panel (V, W: 2 XL_PB
  // columns in LDS; LAPACK's dlatrd scheme): column k is read with the panel's corrections applied on the fly, the product uses the
  // matrix of the panel's start plus two thin products with the panel, and the trailing block is read + written ONCE per panel
  // (B -= V W^T + W V^T) -- 1 + 2 / XL_PB passes per column.  Same arithmetic as the plain reduction up to the order of the sums.
  constexpr int PB = XL_PB;
  static_assert(NT != 512 || PB == NT / 64, "one wavefront per panel column in the thin products");
  lptr Pl = strips + 6 * n;             // panel, column-major: V_c at c n, W_c at (PB + c) n; rows <= (column's index) stay zero
  for (int e = TID; e < 2 * PB * n; e += NT) Pl[e] = 0.0;
  XSYNC();
  // product B0 v: thread (g2, ip) owns the column PAIR (2 ip, 2 ip + 1) -- 16-byte loads, n / 2 lanes per row, so that NT / (n / 2)
  // groups share the rows (3 at n = 300 where one column per thread leaves 212 threads idle and 19 dependent round trips per column)
  const bool pairs = !(n & 1) && !((uintptr_t)Bm & 15);
  const int np2 = ((n >> 1) + 31) & ~31;
  const int G2 = pairs ? (NT / np2 < 6 ? NT / np2 : 6) : S.G, g2 = pairs ? TID / np2 : S.g, ip2 = TID - g2 * np2;
  int pc = 0;                           // columns in the panel
#ifdef DG_PROF
  long long tq1 = 0, tq2 = 0, tq3 = 0, tq4 = 0, tqc = clock64();
#define TRI_T(acc) do { const long long now_ = clock64(); acc += now_ - tqc; tqc = now_; } while (0)
#else
#define TRI_T(acc) do {} while (0)
#endif
  for (int k = 0; k + 2 < n; k++) {
    // (1) column k of the UPDATED matrix below the diagonal (= row k right of it: both triangles are kept)
    double vk[PB], wk[PB];
#pragma unroll
    for (int cc = 0; cc < PB; cc++) { vk[cc] = cc < pc ? Pl[cc * n + k] : 0.0; wk[cc] = cc < pc ? Pl[(PB + cc) * n + k] : 0.0; }
    double s2 = 0;
    for (int i = k + 1 + TID; i < n; i += NT) {
      double x = Bm[(int64_t)k * n + i];
#pragma unroll
      for (int cc = 0; cc < PB; cc++) if (cc < pc) x -= Pl[cc * n + i] * wk[cc] + Pl[(PB + cc) * n + i] * vk[cc];
      vv[i] = x; s2 += x * x;
    }
    const double nrm2 = block_sum(s2, red);
    const double x0 = vv[k + 1];
    const double tail2 = nrm2 - x0 * x0;
    double alpha, beta;
    if (!(tail2 > 0.0)) { alpha = x0; beta = 0.0; }
    else { alpha = x0 >= 0 ? -sqrt(nrm2) : sqrt(nrm2); const double v0 = x0 - alpha; beta = 2.0 / (tail2 + v0 * v0); }
    __syncthreads();
    if (TID == 0) {
      double dk = Bm[(int64_t)k * n + k];
#pragma unroll
      for (int cc = 0; cc < PB; cc++) dk -= 2.0 * vk[cc] * wk[cc];
      dv[k] = dk; ev[k] = alpha; tau[k] = beta;
      if (beta != 0.0) vv[k + 1] = x0 - alpha;
    }
    __syncthreads();
    for (int i = k + 1 + TID; i < n; i += NT) Vr[(int64_t)k * n + i] = vv[i];
    TRI_T(tq1);
    if (beta != 0.0) {
      // (2) B0 v with the matrix as it stood at the panel's start (B symmetric: column i is read along rows, coalesced over i) ...
      if (pairs) {
        if (g2 < G2 && 2 * ip2 + 1 > k && 2 * ip2 < n) {
          const int len = n - k - 1, ja = k + 1 + (g2 * len) / G2, jb = k + 1 + ((g2 + 1) * len) / G2;
          const double2* pb = (const double2*)(Bm + (int64_t)ja * n + 2 * ip2);
          const int rs2 = n >> 1;
          double a0 = 0, a1 = 0, c0 = 0, c1 = 0;
          int j = ja;
          for (; j + 15 < jb; j += 16, pb += 16 * rs2) {          // sixteen L2 round trips in flight
            double2 b[16];
#pragma unroll
            for (int u = 0; u < 16; u++) b[u] = pb[u * rs2];
#pragma unroll
            for (int u = 0; u < 16; u += 2) {
              const double v0 = vv[j + u], v1 = vv[j + u + 1];
              a0 += b[u].x * v0; c0 += b[u].y * v0; a1 += b[u + 1].x * v1; c1 += b[u + 1].y * v1;
            }
          }
          for (; j + 3 < jb; j += 4, pb += 4 * rs2) {
            const double2 b0 = pb[0], b1 = pb[rs2], b2 = pb[2 * rs2], b3 = pb[3 * rs2];
            a0 += b0.x * vv[j]; c0 += b0.y * vv[j]; a1 += b1.x * vv[j + 1]; c1 += b1.y * vv[j + 1];
            a0 += b2.x * vv[j + 2]; c0 += b2.y * vv[j + 2]; a1 += b3.x * vv[j + 3]; c1 += b3.y * vv[j + 3];
          }
          for (; j < jb; j++, pb += rs2) { const double2 b0 = pb[0]; a0 += b0.x * vv[j]; c0 += b0.y * vv[j]; }
          part[g2 * n + 2 * ip2] = a0 + a1; part[g2 * n + 2 * ip2 + 1] = c0 + c1;      // (column k itself may ride along: never read)
        }
      } else if (S.g < S.G && S.i > k && S.i < n) {
        const int len = n - k - 1, ja = k + 1 + (S.g * len) / S.G, jb = k + 1 + ((S.g + 1) * len) / S.G;
        cgptr pb = Bm + (int64_t)ja * n + S.i;
        double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        int j = ja;
        for (; j + 15 < jb; j += 16, pb += 16 * n) {
          double b[16];
#pragma unroll
          for (int u = 0; u < 16; u++) b[u] = pb[u * n];
#pragma unroll
          for (int u = 0; u < 16; u += 4) { a0 += b[u] * vv[j + u]; a1 += b[u + 1] * vv[j + u + 1]; a2 += b[u + 2] * vv[j + u + 2]; a3 += b[u + 3] * vv[j + u + 3]; }
        }
        for (; j + 3 < jb; j += 4, pb += 4 * n) {
          const double b0 = pb[0], b1 = pb[n], b2 = pb[2 * n], b3 = pb[3 * n];
          a0 += b0 * vv[j]; a1 += b1 * vv[j + 1]; a2 += b2 * vv[j + 2]; a3 += b3 * vv[j + 3];
        }
        for (; j < jb; j++, pb += n) a0 += pb[0] * vv[j];
        part[S.g * n + S.i] = (a0 + a1) + (a2 + a3);
      }
      // (3) ... and the panel's thin products W^T v and V^T v, one panel column per wavefront, while the loads above are in flight
      if (wave < pc) {
        double sw = 0, sv = 0;
        for (int i = k + 1 + lane; i < n; i += 64) { const double vi = vv[i]; sw += Pl[(PB + wave) * n + i] * vi; sv += Pl[wave * n + i] * vi; }
        sw = wave_sum(sw); sv = wave_sum(sv);
        if (lane == 0) { e2[wave] = sw; e2[PB + wave] = sv; }
      }
      __syncthreads();
      TRI_T(tq2);
      // (4) p = beta (B0 v - V (W^T v) - W (V^T v)),  w = p - (beta / 2) (p . v) v;  v and w join the panel
      double pvsum = 0;
      for (int i = k + 1 + TID; i < n; i += NT) {
        double a = part[i];
        for (int g = 1; g < G2; g++) a += part[g * n + i];
#pragma unroll
        for (int cc = 0; cc < PB; cc++) if (cc < pc) a -= Pl[cc * n + i] * e2[cc] + Pl[(PB + cc) * n + i] * e2[PB + cc];
        const double pq = beta * a;
        pv[i] = pq; pvsum += pq * vv[i];
      }
      const double K = 0.5 * beta * block_sum(pvsum, red);
      for (int i = k + 1 + TID; i < n; i += NT) {
        const double w = pv[i] - K * vv[i];
        Pl[pc * n + i] = vv[i]; Pl[(PB + pc) * n + i] = w;
      }
    }
    pc++;
    __syncthreads();
    TRI_T(tq3);
    if (pc == PB || k + 3 >= n) {
      // (5) B -= V W^T + W V^T on the trailing block (rows and columns > k), a rank-16 product on the matrix cores.  Tiles start at the
      // 16-row boundary at or below k + 1: the rows and columns <= k they touch are dead (read for the last time inside this panel)
      const int t_lo = (k + 1) >> 4, t_hi = (n + 15) >> 4;
      const int TT = t_hi - t_lo;
      xl_mfma_rank16(Bm, n, n, TT * TT, [&](int t, int& ti, int& tj) { ti = t_lo + t / TT; tj = t_lo + t % TT; },
                     [&](int kk, int i) { return i < n ? -Pl[kk * n + i] : 0.0; },
                     [&](int kk, int j) { return j < n ? Pl[((kk + PB) & (2 * PB - 1)) * n + j] : 0.0; });
      XSYNC();
      for (int e = TID; e < 2 * PB * n; e += NT) Pl[e] = 0.0;
      pc = 0;
      __syncthreads();
      TRI_T(tq4);
    }
  }
  PROF_COUNT(PH_T_COL, tq1); PROF_COUNT(PH_T_MV, tq2); PROF_COUNT(PH_T_W, tq3); PROF_COUNT(PH_T_UPD, tq4);
  if (TID == 0) {
    dv[n - 2] = Bm[(int64_t)(n - 2) * n + n - 2]; dv[n - 1] = Bm[(int64_t)(n - 1) * n + n - 1];
    ev[n - 2] = Bm[(int64_t)(n - 1) * n + n - 2]; ev[n - 1] = 0.0; tau[n - 2] = 0.0; tau[n - 1] = 0.0;
  }
  __syncthreads();
  }
  PROF_END(PH_E_TRI, pt_t);
  // ---- negative eigenvalues of T
  double tn = 0;
  for (int i = TID; i < n; i += NT) { e2[i] = ev[i] * ev[i]; tn = fmax(tn, fabs(dv[i]) + fabs(ev[i]) + (i > 0 ? fabs(ev[i - 1]) : 0.0)); }
  const double tnorm = block_max(tn, red);
  const double pivmin = fmax(1e-300, 1e-290 * tnorm * tnorm);
  __syncthreads();
  const int kneg = sturm_count(dv, e2, n, 0.0, pivmin);      // every thread, same result
  PROF_COUNT(PH_E_KNEG, kneg); PROF_COUNT(PH_PD_TRY, kneg == 0 ? 1 : 0);
  if (kneg > XL_KMAX) return false;
  PROF_BEGIN(pe1);
  for (int j = wave; j < kneg; j += NT / 64) {
    double lo = -tnorm * 1.0000001 - 1e-300, hi = 0.0;
    for (int it = 0; it < 10; it++) {      // 65^10 > 2^53
      const double wdt = hi - lo;
      const double sg = lo + wdt * (double)(lane + 1) * (1.0 / 65.0);
      const int cnt = sturm_count(dv, e2, n, sg, pivmin);
      const unsigned long long above = __ballot(cnt > j);
      const int first = above ? __ffsll((long long)above) - 1 : 64;
      const double nlo = first == 0 ? lo : lane_bcast(sg, first - 1);
      const double nhi = first == 64 ? hi : lane_bcast(sg, first);
      lo = nlo; hi = nhi;
      if (hi - lo <= 4.5e-16 * fmax(fabs(lo), fabs(hi)) + 1e-300) break;
    }
    const double lam = 0.5 * (lo + hi);
    // eigenvector by twisted factorisation (Parlett & Dhillon; LAPACK dlar1v): the two pivot sequences sequentially by
    // lane 0 into the wavefront's strip, twist index and the two-term recurrences by the whole wavefront
    lptr Dp = strips + wave * 3 * n, Dm = Dp + n, zz = Dp + 2 * n;
    if (lane == 0) {
      lamv[j] = lam;
      double qf = dv[0] - lam;
      qf = (__builtin_fabs(qf) <= pivmin) ? -pivmin : qf;
      Dp[0] = qf;
      for (int i = 1; i < n; i++) { qf = (dv[i] - lam) - e2[i - 1] / qf; qf = (__builtin_fabs(qf) <= pivmin) ? -pivmin : qf; Dp[i] = qf; }
      double qb = dv[n - 1] - lam;
      qb = (__builtin_fabs(qb) <= pivmin) ? -pivmin : qb;
      Dm[n - 1] = qb;
      for (int i = n - 2; i >= 0; i--) { qb = (dv[i] - lam) - e2[i] / qb; qb = (__builtin_fabs(qb) <= pivmin) ? -pivmin : qb; Dm[i] = qb; }
    }
    __builtin_amdgcn_wave_barrier();
    double gbest = INFINITY; int r = 0;
    for (int i = lane; i < n; i += 64) { const double gm = __builtin_fabs(Dp[i] + Dm[i] - (dv[i] - lam)); if (gm < gbest) { gbest = gm; r = i; } }
    wave_argmin(gbest, r);
    if (lane == 0) {
      zz[r] = 1.0;
      for (int i = r - 1; i >= 0; i--) zz[i] = -(ev[i] / Dp[i]) * zz[i + 1];
      for (int i = r; i + 1 < n; i++) zz[i + 1] = -(ev[i] / Dm[i + 1]) * zz[i];
    }
    __builtin_amdgcn_wave_barrier();
    double mx = 0;
    for (int i = lane; i < n; i += 64) mx = fmax(mx, __builtin_fabs(zz[i]));
    mx = wave_max(mx);
    for (int i = lane; i < n; i += 64) Z[(int64_t)j * n + i] = zz[i] / mx;
  }
  XSYNC();
  PROF_END(PH_E_BIS, pe1);
  // ---- orthonormalisation (close eigenvalues give nearly parallel vectors): classical Gram-Schmidt applied twice per vector -- all the
  //      projections of vector j at once (one wavefront per earlier vector for the dot products, then one pass over z_j), instead of
  //      one block-wide step per PAIR as modified Gram-Schmidt needs: with up to 64 vectors that is 128 steps, not 2,016
  lptr dots = strips;        // (the strips are free again: XL_KMAX coefficients)
  for (int j = 0; j < kneg; j++) {
    for (int pass = 0; pass < 2 && j > 0; pass++) {
      for (int i = wave; i < j; i += NT / 64) {
        double dsum = 0;
        for (int t = lane; t < n; t += 64) dsum += Z[(int64_t)j * n + t] * Z[(int64_t)i * n + t];
        dsum = wave_sum(dsum);
        if (lane == 0) dots[i] = dsum;
      }
      __syncthreads();
      for (int t = TID; t < n; t += NT) {
        double zt = Z[(int64_t)j * n + t];
        for (int i = 0; i < j; i++) zt -= dots[i] * Z[(int64_t)i * n + t];
        Z[(int64_t)j * n + t] = zt;
      }
      XSYNC();
    }
    double nsum = 0;
    for (int t = TID; t < n; t += NT) { const double z = Z[(int64_t)j * n + t]; nsum += z * z; }
    const double nr = 1.0 / sqrt(block_sum(nsum, red));
    for (int t = TID; t < n; t += NT) Z[(int64_t)j * n + t] *= nr;
    XSYNC();
  }
  // ---- back-transformation v = H_0 ... H_{n-3} z, one wavefront per vector, z in registers
  for (int j = wave; j < kneg; j += NT / 64) {
    double z[XL_NV];
#pragma unroll
    for (int h = 0; h < XL_NV; h++) { const int i = lane + 64 * h; z[h] = i < n ? Z[(int64_t)j * n + i] : 0.0; }
    for (int k = n - 3; k >= 0; k--) {
      const double beta = tau[k];
      if (beta == 0.0) continue;
      double v[XL_NV], dt = 0;
#pragma unroll
      for (int h = 0; h < XL_NV; h++) { const int i = lane + 64 * h; v[h] = (i > k && i < n) ? Vr[(int64_t)k * n + i] : 0.0; dt += v[h] * z[h]; }
      dt = beta * wave_sum(dt);
#pragma unroll
      for (int h = 0; h < XL_NV; h++) z[h] -= dt * v[h];
    }
#pragma unroll
    for (int h = 0; h < XL_NV; h++) { const int i = lane + 64 * h; if (i < n) Z[(int64_t)j * n + i] = z[h]; }
  }
  XSYNC();
  // ---- M = B + sum_j (floor - lambda_j) v_j v_j^T + reg I
  const double reg = dev_reg();
  // (a rank-kneg product: on the matrix cores, 16 eigenvectors per pass over a tile's operands)
  {
    const int T = (n + 15) >> 4;
    const double fl = D.eig_floor;
    xl_mfma_tiles(n, T * T, (kneg + 15) >> 4, [&](int t, int& ti, int& tj) { ti = t / T; tj = t - ti * T; },
                  [&](int kk, int i) { return (kk < kneg && i < n) ? (fl - lamv[kk]) * Z[(int64_t)kk * n + i] : 0.0; },
                  [&](int kk, int j) { return (kk < kneg && j < n) ? Z[(int64_t)kk * n + j] : 0.0; },
                  [&](int i, int k) { return 0.5 * (Qg[(int64_t)i * n + k] + Qg[(int64_t)k * n + i]) + (i == k ? reg : 0.0); },
                  [&](int i, int k, double v) { Mx[(int64_t)i * n + k] = v; if (Qpd) Qpd[(int64_t)i * n + k] = v; });
  }
  XSYNC();
  PROF_END(PH_JACOBI, pt_t);
  return true;
}

// ---- _nearestPD: M (row-major, n x n) into c.ws + ws_R
__device__ __noinline__ void dev_xl_psd_jacobi(const Ctx& c, gptr Qpd) {
  const DgProb& D = dg_prob;
  const int n = D.n, lane = TID & 63, wave = TID >> 6;
  lptr lds = LP(0);
  cgptr Qg = c.ws + D.ws_q;
  gptr G = c.ws + D.ws_P, V = c.ws + D.ws_V, Mx = c.ws + D.ws_R;   // G, V column-major: column j at [j*n, j*n+n)
  lptr lamv = lds + D.L.g_tw;             // n eigenvalues
  lds_d* scal = lds + D.L.scal;
  if (TID == 0) scal[DG_XVALID] = 0.0;    // (the EIG scratch may overlap the trajectory)
  __syncthreads();
  PROF_BEGIN(pt_t);
  // Start from the eigenvectors of this scenario's previous projection when there is one (B changes little from one SQP
  // iteration to the next): G = B V_prev is then nearly orthogonal and a few sweeps suffice instead of ~11 from V = I.
  gptr Vp = c.ws + D.ws_Vp;
  const bool warm = D.par.qp_warm_start && scal[DG_XL_BASIS] != 0.0;
  for (int e = TID; e < n * n; e += NT) {
    const int col = e / n, row = e % n;
    const double b = 0.5 * (Qg[(int64_t)row * n + col] + Qg[(int64_t)col * n + row]);
    if (warm) { Mx[e] = b; V[e] = Vp[e]; }
    else { G[e] = b; V[e] = row == col ? 1.0 : 0.0; }
  }
  XSYNC();
  if (warm) {
    for (int e = TID; e < n * n; e += NT) {
      const int j = e / n, i = e % n;          // G[:, j] = B V[:, j]; B symmetric: column i of B is read along rows (coalesced over i)
      double s0 = 0, s1 = 0;
      int k = 0;
      for (; k + 1 < n; k += 2) { s0 += Mx[(int64_t)k * n + i] * Vp[(int64_t)j * n + k]; s1 += Mx[(int64_t)(k + 1) * n + i] * Vp[(int64_t)j * n + k + 1]; }
      if (k < n) s0 += Mx[(int64_t)k * n + i] * Vp[(int64_t)j * n + k];
      G[e] = s0 + s1;
    }
    XSYNC();
  }
  const int npad = n + (n & 1), rounds = npad - 1, half = npad / 2;
  // columns whose norm falls below 1e-14 |B| carry a numerically zero eigenvalue: rotating two of them against each other
  // never converges (their inner product is noise) and changes nothing that matters
  double bn = 0;
  for (int e = TID; e < n * n; e += NT) bn += G[e] * G[e];
  const double tiny = 1e-28 * block_sum(bn, lds + D.L.red);
  for (int sweep = 0; sweep < 20; sweep++) {
    if (TID == 0) scal[3] = 0.0;
    __syncthreads();
    int rotated = 0;
    for (int r = 0; r < rounds; r++) {
      // round-robin tournament: every pair meets once per sweep, the pairs of one round are disjoint.  A wavefront owns up
      // to XL_MAXP pairs of the round and handles them TOGETHER: all column loads are issued before the first reduction, so
      // the global-memory latency is paid once per round instead of once per pair.
      for (int g0 = 0; g0 < XL_MAXP; g0 += XL_GRP) {     // XL_GRP pairs at a time: their columns fit the register file
        int pp[XL_GRP], qq[XL_GRP];
        double ga[XL_GRP][XL_NV], gb[XL_GRP][XL_NV], va[XL_GRP][XL_NV], vb[XL_GRP][XL_NV];
  #pragma unroll
        for (int s = 0; s < XL_GRP; s++) {
          const int k = wave + (g0 + s) * (NT / 64);
          int p = k == 0 ? npad - 1 : (r + k) % rounds, q = k == 0 ? r : (r - k + rounds) % rounds;
          if (k >= half || p >= n || q >= n) { p = -1; q = -1; }
          else if (p > q) { const int t = p; p = q; q = t; }
          pp[s] = p; qq[s] = q;
  #pragma unroll
          for (int h = 0; h < XL_NV; h++) {
            const int i = lane + 64 * h;
            const bool ok = p >= 0 && i < n;
            ga[s][h] = ok ? G[(int64_t)p * n + i] : 0.0;
            gb[s][h] = ok ? G[(int64_t)q * n + i] : 0.0;
            va[s][h] = ok ? V[(int64_t)p * n + i] : 0.0;     // the eigenvector columns ride along: one latency per group
            vb[s][h] = ok ? V[(int64_t)q * n + i] : 0.0;
          }
        }
        double cs_[XL_GRP], sn_[XL_GRP];
        // the three inner products of every pair by wavefront reductions; the rotation angles of the XL_GRP pairs are then
        // computed side by side, pair s in lane s (fp64 sqrt / divide chains are the expensive part), and broadcast
        double mal = 0.0, mbe = 0.0, mgm = 0.0;
  #pragma unroll
        for (int s = 0; s < XL_GRP; s++) {
          double al = 0, be = 0, gm = 0;
  #pragma unroll
          for (int h = 0; h < XL_NV; h++) { al += ga[s][h] * ga[s][h]; be += gb[s][h] * gb[s][h]; gm += ga[s][h] * gb[s][h]; }
          al = wave_sum(al); be = wave_sum(be); gm = wave_sum(gm);
          if (pp[s] < 0) gm = 0.0;
          if (lane == s) { mal = al; mbe = be; mgm = gm; }
        }
        double mcs = 1.0, msn = 0.0;
        if (__builtin_fabs(mgm) > 1e-14 * sqrt(mal * mbe) && mal > tiny && mbe > tiny) {
          const double zeta = (mbe - mal) / (2.0 * mgm);
          const double t = (zeta >= 0 ? 1.0 : -1.0) / (__builtin_fabs(zeta) + sqrt(1.0 + zeta * zeta));
          mcs = 1.0 / sqrt(1.0 + t * t); msn = mcs * t;
        }
  #pragma unroll
        for (int s = 0; s < XL_GRP; s++) {
          cs_[s] = lane_bcast(mcs, s); sn_[s] = lane_bcast(msn, s);
          if (sn_[s] != 0.0) rotated = 1;
        }
  #pragma unroll
        for (int s = 0; s < XL_GRP; s++) {
          if (sn_[s] == 0.0) continue;       // wave-uniform
          const int p = pp[s], q = qq[s];
  #pragma unroll
          for (int h = 0; h < XL_NV; h++) {
            const int i = lane + 64 * h;
            if (i < n) {
              G[(int64_t)p * n + i] = cs_[s] * ga[s][h] - sn_[s] * gb[s][h];
              G[(int64_t)q * n + i] = sn_[s] * ga[s][h] + cs_[s] * gb[s][h];
              V[(int64_t)p * n + i] = cs_[s] * va[s][h] - sn_[s] * vb[s][h];
              V[(int64_t)q * n + i] = sn_[s] * va[s][h] + cs_[s] * vb[s][h];
            }
          }
        }
      }
      XSYNC();
    }
    if (rotated && lane == 0) scal[3] = 1.0;
    __syncthreads();
    const bool again = scal[3] != 0.0;
    __syncthreads();
    if (!again) break;
    PROF_COUNT(PH_C_NPREV, 1);
  }
  // eigenvalues lambda_j = v_j . (B v_j) / v_j . v_j = v_j . g_j / |v_j|^2 (the basis is carried over many projections:
  // its norms are not assumed); coefficient of the correction v_j v_j^T in lamv[n + j]
  for (int j = wave; j < n; j += NT / 64) {
    double s = 0, nv = 0;
    for (int i = lane; i < n; i += 64) { const double v = V[(int64_t)j * n + i]; s += v * G[(int64_t)j * n + i]; nv += v * v; }
    s = wave_sum(s); nv = wave_sum(nv);
    if (lane == 0) { const double lj = s / nv; lamv[j] = lj; lamv[n + j] = lj < 0.0 ? (D.eig_floor - lj) / nv : 0.0; }
  }
  __syncthreads();
  const double reg = dev_reg();
  for (int e = TID; e < n * n; e += NT) {
    const int i = e / n, k = e % n;
    double a = 0.5 * (Qg[(int64_t)i * n + k] + Qg[(int64_t)k * n + i]);
    for (int j = 0; j < n; j++) {
      const double cj = lamv[n + j];
      if (cj != 0.0) a += cj * V[(int64_t)j * n + i] * V[(int64_t)j * n + k];
    }
    if (i == k) a += reg;
    Mx[e] = a;
    if (Qpd) Qpd[e] = a;
    Vp[e] = V[e];
  }
  if (TID == 0) scal[DG_XL_BASIS] = 1.0;
  XSYNC();
  PROF_END(PH_JACOBI, pt_t);
}

// one Givens rotation of columns (ja, jb) of the row-major matrix X over rows [0, rows): [xa, xb] <- [c xa + s xb, -s xa + c xb]
__device__ inline void xl_rot_cols(gptr X, int n, int rows, int ja, int jb, double cc, double ss) {
  for (int i = TID; i < rows; i += NT) {
    const double a = X[(int64_t)i * n + ja], b = X[(int64_t)i * n + jb];
    X[(int64_t)i * n + ja] = cc * a + ss * b;
    X[(int64_t)i * n + jb] = -ss * a + cc * b;
  }
}

// ---- _solve_qp on the projected Hessian M held row-major in the scratch (ws_xM).  Also the QP of the smaller layouts when
// M is too ill-conditioned for the explicit-inverse kernels (classic_qp, dgsqp_layout.h).
// Out: du (L.o_du), lhat (L.o_lhat).  Returns 0 ok, 1 infeasible, 2 numerical failure / iteration limit.
// J (n x n, row stride js) lives in LDS -- in the slots of the packed P and R of the explicit-inverse kernels, which the classical
// method does not use -- whenever the game has the LDS-resident layout (n <= ~100), otherwise in the workgroup's L2 scratch.
// R (upper triangular) is kept column-major in the scratch: a column is what one step of a triangular solve and the append of a
// row read / write, and it is contiguous.
template <class MP> struct xl_mp { static constexpr bool lds = false; };
template <> struct xl_mp<lptr> { static constexpr bool lds = true; };
#define JSYNC() do { if constexpr (xl_mp<MP>::lds) __syncthreads(); else XSYNC(); } while (0)
// R: column k holds rows 0..k.  The first `cap` columns are packed in LDS, the others sit column-major (stride n) in the scratch.
struct XlR {
  gptr G; lptr Lp; int cap, n;
  __device__ inline double get(int i, int k) const { return k < cap ? Lp[((k * (k + 1)) >> 1) + i] : G[(int64_t)k * n + i]; }
  __device__ inline void set(int i, int k, double v) const { if (k < cap) Lp[((k * (k + 1)) >> 1) + i] = v; else G[(int64_t)k * n + i] = v; }
};

// Wavefront 0: R r = b (back substitution) and R^T y = b (forward substitution), iq x iq.  Lane l owns rows l, l + 64, ...; the
// columns of R are fetched BCH at a time (one round trip per BCH steps), the pivots travel by readlane -- no barriers.
template <int NS, int BCH>
__device__ inline void xl_wave_backsub_t(const XlR& R, int iq, clptr b, lptr out) {
  const int lane = TID & 63;
  double a[NS];
#pragma unroll
  for (int s = 0; s < NS; s++) { const int i = lane + 64 * s; a[s] = i < iq ? b[i] : 0.0; }
  for (int k0 = iq - 1; k0 >= 0; k0 -= BCH) {
    double col[BCH][NS], dg[BCH];
#pragma unroll
    for (int t = 0; t < BCH; t++) {
      const int k = k0 - t, kk = k > 0 ? k : 0;
      dg[t] = k >= 0 ? R.get(kk, kk) : 1.0;
#pragma unroll
      for (int s = 0; s < NS; s++) { const int i = lane + 64 * s; col[t][s] = (k >= 0 && i < k) ? R.get(i, kk) : 0.0; }
    }
#pragma unroll
    for (int t = 0; t < BCH; t++) {
      const int k = k0 - t;
      if (k >= 0) {       // uniform
        double piv = a[0];
#pragma unroll
        for (int s = 1; s < NS; s++) piv = (k >> 6) == s ? a[s] : piv;
        const double rk = lane_bcast(piv, k & 63) / dg[t];
        if (lane == 0) out[k] = rk;
#pragma unroll
        for (int s = 0; s < NS; s++) a[s] -= col[t][s] * rk;     // (rows >= k carry zeros)
      }
    }
  }
}
template <int NS, int BCH>
__device__ inline void xl_wave_fwdsub_t(const XlR& R, int iq, clptr b, lptr out) {
  const int lane = TID & 63;
  double y[NS];
#pragma unroll
  for (int s = 0; s < NS; s++) y[s] = 0.0;
  for (int k0 = 0; k0 < iq; k0 += BCH) {
    double col[BCH][NS], dg[BCH], bk[BCH];
#pragma unroll
    for (int t = 0; t < BCH; t++) {
      const int k = k0 + t, kk = k < iq ? k : 0;
      dg[t] = k < iq ? R.get(kk, kk) : 1.0;
      bk[t] = k < iq ? b[k] : 0.0;
#pragma unroll
      for (int s = 0; s < NS; s++) { const int i = lane + 64 * s; col[t][s] = (k < iq && i < k) ? R.get(i, kk) : 0.0; }
    }
#pragma unroll
    for (int t = 0; t < BCH; t++) {
      const int k = k0 + t;
      if (k < iq) {       // uniform
        double part = 0.0;
#pragma unroll
        for (int s = 0; s < NS; s++) part += col[t][s] * y[s];
        const double yk = (bk[t] - wave_sum(part)) / dg[t];
        if (lane == 0) out[k] = yk;
#pragma unroll
        for (int s = 0; s < NS; s++) if (lane + 64 * s == k) y[s] = yk;
      }
    }
  }
}
// (own functions: see xl_jt_mul_pairs)
__device__ __noinline__ void xl_wave_backsub(const XlR& R, int iq, clptr b, lptr out) {
  if (iq <= 64) xl_wave_backsub_t<1, 8>(R, iq, b, out);
  else if (iq <= 128) xl_wave_backsub_t<2, 8>(R, iq, b, out);
  else if (iq <= 256) xl_wave_backsub_t<4, 4>(R, iq, b, out);
  else xl_wave_backsub_t<5, 4>(R, iq, b, out);
}
__device__ __noinline__ void xl_wave_fwdsub(const XlR& R, int iq, clptr b, lptr out) {
  if (iq <= 64) xl_wave_fwdsub_t<1, 8>(R, iq, b, out);
  else if (iq <= 128) xl_wave_fwdsub_t<2, 8>(R, iq, b, out);
  else if (iq <= 256) xl_wave_fwdsub_t<4, 4>(R, iq, b, out);
  else xl_wave_fwdsub_t<5, 4>(R, iq, b, out);
}
// out[i] = sum_{k0 <= k < k1} J[k][i] v[k]   (J^T v restricted to rows k0..k1-1): consecutive threads read consecutive addresses
// J^T v for a matrix in the scratch (own function: inside the QP's body the register allocation of these loops changed with every
// unrelated edit -- 63 k or 126 k cycles per step direction).  Thread (g, ip) owns the column PAIR (2 ip, 2 ip + 1) and the g-th part
// of the rows -- 16-byte loads, sixteen L2 round trips in flight.  One column per thread and eight in flight is 38 dependent round
// trips at n = 300, where n / 2 lanes per row leave room for three row groups: 6.  Needs an even n and row stride and a 16-byte
// aligned matrix (the caller checks).
__device__ __noinline__ void xl_jt_mul_pairs(cgptr J, int js, int n, int k0, int k1, clptr v, lptr out, lptr part) {
  const int np2 = ((n >> 1) + 31) & ~31;
  const int G2 = NT / np2 < DG_NH ? NT / np2 : DG_NH, g2 = TID / np2, ip2 = TID - g2 * np2;
  if (g2 < G2 && 2 * ip2 < n) {
    const int len = k1 - k0, ka = k0 + (g2 * len) / G2, kb = k0 + ((g2 + 1) * len) / G2;
    const double2* pb = (const double2*)(J + (int64_t)ka * js + 2 * ip2);
    const int rs2 = js >> 1;
    double a0 = 0, a1 = 0, c0 = 0, c1 = 0;
    int k = ka;
    for (; k + 15 < kb; k += 16, pb += 16 * rs2) {
      double2 b[16];
#pragma unroll
      for (int u = 0; u < 16; u++) b[u] = pb[u * rs2];
#pragma unroll
      for (int u = 0; u < 16; u += 2) {
        const double v0 = v[k + u], v1 = v[k + u + 1];
        a0 += b[u].x * v0; c0 += b[u].y * v0; a1 += b[u + 1].x * v1; c1 += b[u + 1].y * v1;
      }
    }
    for (; k + 3 < kb; k += 4, pb += 4 * rs2) {
      const double2 b0 = pb[0], b1 = pb[rs2], b2 = pb[2 * rs2], b3 = pb[3 * rs2];
      a0 += b0.x * v[k]; c0 += b0.y * v[k]; a1 += b1.x * v[k + 1]; c1 += b1.y * v[k + 1];
      a0 += b2.x * v[k + 2]; c0 += b2.y * v[k + 2]; a1 += b3.x * v[k + 3]; c1 += b3.y * v[k + 3];
    }
    for (; k < kb; k++, pb += rs2) { const double2 b0 = pb[0]; a0 += b0.x * v[k]; c0 += b0.y * v[k]; }
    part[g2 * n + 2 * ip2] = a0 + a1; part[g2 * n + 2 * ip2 + 1] = c0 + c1;
  }
  __syncthreads();
  if (TID < n) { double sacc = part[TID]; for (int g = 1; g < G2; g++) sacc += part[g * n + TID]; out[TID] = sacc; }
  __syncthreads();
}
template <class MP>
__device__ inline void xl_jt_mul(MP J, int js, int n, const XlSplit& S, int k0, int k1, clptr v, lptr out, lptr part) {
  if constexpr (!xl_mp<MP>::lds) {
    if (!(n & 1) && !(js & 1) && !((uintptr_t)J & 15)) { xl_jt_mul_pairs(J, js, n, k0, k1, v, out, part); return; }
  }
  if (S.g < S.G && S.i < n) {
    const int len = k1 - k0, ka = k0 + (S.g * len) / S.G, kb = k0 + ((S.g + 1) * len) / S.G;
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0, s6 = 0, s7 = 0;
    int k = ka;
    for (; k + 7 < kb; k += 8) {
      s0 += J[k * js + S.i] * v[k]; s1 += J[(k + 1) * js + S.i] * v[k + 1];
      s2 += J[(k + 2) * js + S.i] * v[k + 2]; s3 += J[(k + 3) * js + S.i] * v[k + 3];
      s4 += J[(k + 4) * js + S.i] * v[k + 4]; s5 += J[(k + 5) * js + S.i] * v[k + 5];
      s6 += J[(k + 6) * js + S.i] * v[k + 6]; s7 += J[(k + 7) * js + S.i] * v[k + 7];
    }
    for (; k < kb; k++) s0 += J[k * js + S.i] * v[k];
    s0 += s4; s1 += s5; s2 += s6; s3 += s7;
    part[S.g * n + S.i] = (s0 + s1) + (s2 + s3);
  }
  __syncthreads();
  if (TID < n) { double s = part[TID]; for (int g = 1; g < S.G; g++) s += part[g * n + TID]; out[TID] = s; }
  __syncthreads();
}
// J v (columns k0 .. k1 - 1) for a matrix in the scratch: one wavefront per row, lanes along the row (own function, see xl_jt_mul_pairs)
__device__ __noinline__ void xl_j_mul_rows(cgptr J, int js, int n, int k0, int k1, clptr v, lptr out) {
  const int lane = TID & 63;
  // eight rows per wavefront and pass, every load of the pass (8 rows x up to 5 chunks of 64 columns, n <= 320) issued before the
  // first use: one L2 round trip per pass instead of one per chunk
  constexpr int RP = 8;
  for (int i0 = (TID >> 6) * RP; i0 < n; i0 += (NT / 64) * RP) {
    double a[RP][XL_NV], vk[XL_NV];
#pragma unroll
    for (int cidx = 0; cidx < XL_NV; cidx++) {
      const int k = k0 + lane + 64 * cidx;
      const bool on = k < k1;
      vk[cidx] = on ? v[k] : 0.0;
#pragma unroll
      for (int r = 0; r < RP; r++) a[r][cidx] = (on && i0 + r < n) ? J[(int64_t)(i0 + r) * js + k] : 0.0;
    }
#pragma unroll
    for (int r = 0; r < RP; r++) {
      const double t = wave_sum(((a[r][0] * vk[0] + a[r][1] * vk[1]) + (a[r][2] * vk[2] + a[r][3] * vk[3])) + a[r][4] * vk[4]);
      if (lane == 0 && i0 + r < n) out[i0 + r] = t;
    }
  }
}
// The same product by the wavefronts 1 .. 7 only: wavefront 0 returns at once and is free for the triangular solve r = R^-1 d1 of the same
// inner step (independent data: d1 / R against d2 / J) -- the two used to run one after the other, 20 k + 20 k cycles of the 60 k of a step
// direction at n = 300.  Same rows, same sums as xl_j_mul_rows: results are bit-identical.
__device__ __noinline__ void xl_j_mul_rows_but_wave0(cgptr J, int js, int n, int k0, int k1, clptr v, lptr out) {
  const int lane = TID & 63, wave = TID >> 6;
  if (wave == 0) return;
  constexpr int RP = 8;
  for (int i0 = (wave - 1) * RP; i0 < n; i0 += (NT / 64 - 1) * RP) {
    double a[RP][XL_NV], vk[XL_NV];
#pragma unroll
    for (int cidx = 0; cidx < XL_NV; cidx++) {
      const int k = k0 + lane + 64 * cidx;
      const bool on = k < k1;
      vk[cidx] = on ? v[k] : 0.0;
#pragma unroll
      for (int r = 0; r < RP; r++) a[r][cidx] = (on && i0 + r < n) ? J[(int64_t)(i0 + r) * js + k] : 0.0;
    }
#pragma unroll
    for (int r = 0; r < RP; r++) {
      const double t = wave_sum(((a[r][0] * vk[0] + a[r][1] * vk[1]) + (a[r][2] * vk[2] + a[r][3] * vk[3])) + a[r][4] * vk[4]);
      if (lane == 0 && i0 + r < n) out[i0 + r] = t;
    }
  }
}
// out[i] = sum_{k0 <= k < k1} J[i][k] v[k]   (J v restricted to columns k0..k1-1).  LDS: the odd row stride keeps the row-per-thread
// reads conflict-free.  Scratch: one wavefront per row, lanes along the row.
template <class MP>
__device__ inline void xl_j_mul(MP J, int js, int n, const XlSplit& S, int k0, int k1, clptr v, lptr out, lptr part) {
  if constexpr (xl_mp<MP>::lds) {
    if (S.g < S.G && S.i < n) {
      const int len = k1 - k0, ka = k0 + (S.g * len) / S.G, kb = k0 + ((S.g + 1) * len) / S.G;
      MP Ji = J + S.i * js;
      double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
      int k = ka;
      for (; k + 3 < kb; k += 4) { s0 += Ji[k] * v[k]; s1 += Ji[k + 1] * v[k + 1]; s2 += Ji[k + 2] * v[k + 2]; s3 += Ji[k + 3] * v[k + 3]; }
      for (; k < kb; k++) s0 += Ji[k] * v[k];
      part[S.g * n + S.i] = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
    if (TID < n) { double s = part[TID]; for (int g = 1; g < S.G; g++) s += part[g * n + TID]; out[TID] = s; }
    __syncthreads();
  } else {
    xl_j_mul_rows(J, js, n, k0, k1, v, out);
    __syncthreads();
  }
}


// Warm start, step (C): J <- J H_t0 ... H_{t1-1} for a matrix in the scratch, four rows per wavefront and pass (their L2 round trips
// overlap; the reflectors come from LDS: reflector t is column refc[t] of Dl, zero above entry t, scale betas[t]).
__device__ __noinline__ void xl_reflect_rows(gptr J, int js, int n, int t0, int t1, clptr betas, clptr refc, clptr Dl) {
  const int lane = TID & 63, wave = TID >> 6;
  constexpr int RP = 4;
  for (int i0 = wave * RP; i0 < n; i0 += (NT / 64) * RP) {
    double r[RP][XL_NV];
#pragma unroll
    for (int rr = 0; rr < RP; rr++) {
#pragma unroll
      for (int h = 0; h < XL_NV; h++) { const int k = lane + 64 * h; r[rr][h] = (k < n && i0 + rr < n) ? J[(int64_t)(i0 + rr) * js + k] : 0.0; }
    }
    for (int t = t0; t < t1; t++) {
      const double bt = betas[t];
      if (bt == 0.0) continue;
      clptr v = Dl + (int)refc[t] * n;
      double vk[XL_NV], ds[RP];
#pragma unroll
      for (int rr = 0; rr < RP; rr++) ds[rr] = 0.0;
#pragma unroll
      for (int h = 0; h < XL_NV; h++) {
        const int k = lane + 64 * h;
        vk[h] = (k >= t && k < n) ? v[k] : 0.0;
#pragma unroll
        for (int rr = 0; rr < RP; rr++) ds[rr] += r[rr][h] * vk[h];
      }
#pragma unroll
      for (int rr = 0; rr < RP; rr++) ds[rr] = bt * wave_sum(ds[rr]);
#pragma unroll
      for (int rr = 0; rr < RP; rr++) {
#pragma unroll
        for (int h = 0; h < XL_NV; h++) r[rr][h] -= ds[rr] * vk[h];
      }
    }
#pragma unroll
    for (int rr = 0; rr < RP; rr++) {
#pragma unroll
      for (int h = 0; h < XL_NV; h++) { const int k = lane + 64 * h; if (k < n && i0 + rr < n) J[(int64_t)(i0 + rr) * js + k] = r[rr][h]; }
    }
  }
}

// ---- J = L^-T, first part: the elimination M = L~ D L~^T with X = L~^-1 accumulated in place (lower triangle of J, d on the diagonal),
// BLOCKED for the matrix in the L2 scratch.  The column-by-column form below reads and writes the whole active region once per pivot
// (15 of the 56 Mcycles of a QP at n = 300); here 16 pivots form a panel:
//   (a) the 16 x 16 diagonal block is factored by one wavefront (rows on lanes, pivots exchanged by v_readlane);
//   (b) every row below the panel computes its 16 multipliers from its own 16 entries (no barrier: the pivot rows are the block's);
//   (c) the panel's own rows complete their part of X by a forward substitution per column (16 values in registers);
//   (d) ONE pass over the rows below the panel applies all 16 pivots, a rank-16 product on the matrix cores:
//       X[i][k] -= sum_c m[i][c] X[j0 + c][k]  (k < j0),   S[i][k] -= sum_c m[i][c] (m[k][c] d_c)  (j0 + 16 <= k <= i),
//       and X[i][j0 + c] = -(m_i Xpp)[c] with Xpp the inverse of the block's unit factor.
// Same eliminations as the column-by-column form (the sums over a panel's pivots are taken in the MFMA's order).
// LDS: the multipliers (16 columns of n) at `Mm`, two 16 x 16 tables at `tab`, 16 pivots + reciprocals + a flag at `sm`.
__device__ __noinline__ bool xl_eliminate_blocked(gptr J, const int js, const int n, lptr Mm, lptr tab, lptr sm) {
  const int lane = TID & 63, wave = TID >> 6;
  lptr Lm = tab, Xp = tab + 256, dd = sm, dinv = sm + 16, flag = sm + 32;
  const int T = (n + 15) >> 4;
  if (TID == 0) flag[0] = 0.0;
  for (int j0 = 0; j0 < n; j0 += 16) {
    const int pb = n - j0 < 16 ? n - j0 : 16, jt = j0 >> 4;
    __syncthreads();
    // (a) diagonal block
    if (wave == 0) {
      const int r = lane & 15;
      double a[16];
#pragma unroll
      for (int cc = 0; cc < 16; cc++) a[cc] = (r < pb && cc <= r) ? J[(int64_t)(j0 + r) * js + j0 + cc] : 0.0;
      bool okp = true;
#pragma unroll
      for (int cc = 0; cc < 16; cc++) {
        const double dcc = lane_bcast(a[cc], cc);
        const bool live = cc < pb;
        if (live && !(dcc > 0.0)) okp = false;
        const double inv = (live && dcc > 0.0) ? 1.0 / dcc : 0.0;
        const double m = r > cc ? a[cc] * inv : 0.0;
#pragma unroll
        for (int c2 = cc + 1; c2 < 16; c2++) {
          const double piv = lane_bcast(a[cc], c2);        // (row c2, column cc) = L[c2][cc] d_cc
          if (r >= c2) a[c2] -= m * piv;
        }
        if (lane < 16) {
          Lm[r * 16 + cc] = m;
          if (r == cc) { dd[cc] = live ? dcc : 0.0; dinv[cc] = inv; if (live) J[(int64_t)(j0 + r) * js + j0 + cc] = dcc; }
        }
      }
      if (!okp && lane == 0) flag[0] = 1.0;
    }
    __syncthreads();
    if (flag[0] != 0.0) return true;
    // (b) multipliers of the rows below the panel (thread = row)
    const int i = j0 + 16 + TID;
    const bool below = i < n;
    double m[16];
    if (below) {
      double a[16];
#pragma unroll
      for (int cc = 0; cc < 16; cc++) a[cc] = J[(int64_t)i * js + j0 + cc];
#pragma unroll
      for (int cc = 0; cc < 16; cc++) {
        m[cc] = a[cc] * dinv[cc];
#pragma unroll
        for (int c2 = cc + 1; c2 < 16; c2++) a[c2] -= a[cc] * Lm[c2 * 16 + cc];       // a[cc] = m d_cc: the unnormalised entry
        Mm[cc * n + i] = m[cc];
      }
    }
    // (c) X of the panel's rows: columns k < j0 (forward substitution on what previous panels left) and the block itself (on the identity)
    if (TID < j0 + pb) {
      const int k = TID, kc = k - j0;
      double x[16];
#pragma unroll
      for (int r = 0; r < 16; r++) x[r] = kc < 0 ? (r < pb ? J[(int64_t)(j0 + r) * js + k] : 0.0) : (r == kc ? 1.0 : 0.0);
#pragma unroll
      for (int r = 1; r < 16; r++) {
        double sacc = x[r];
#pragma unroll
        for (int cc = 0; cc < r; cc++) sacc -= Lm[r * 16 + cc] * x[cc];
        x[r] = sacc;
      }
#pragma unroll
      for (int r = 0; r < 16; r++) {
        if (r < pb && r > kc) J[(int64_t)(j0 + r) * js + k] = x[r];
        if (kc >= 0) Xp[r * 16 + kc] = x[r];
      }
    } else if (TID < j0 + 16) {           // (a short last panel: the columns it does not have)
      const int kc = TID - j0;
#pragma unroll
      for (int r = 0; r < 16; r++) Xp[r * 16 + kc] = 0.0;
    }
    XSYNC();
    if (pb < 16 || j0 + 16 >= n) break;       // nothing below the last panel
    // (d) the panel's own columns of the rows below:  X[i][j0 + c] = -(m[c] + sum_{c' > c} m[c'] Xpp[c'][c])
    if (below) {
#pragma unroll
      for (int cc = 0; cc < 16; cc++) {
        double sacc = m[cc];
#pragma unroll
        for (int c2 = cc + 1; c2 < 16; c2++) sacc += m[c2] * Xp[c2 * 16 + cc];
        J[(int64_t)i * js + j0 + cc] = -sacc;
      }
    }
    // ... and everything else of those rows on the matrix cores: tile row ti > jt has the tile columns [0, jt) and (jt, ti]
    // (the tile loop as a function of its own and the multipliers read back from LDS were tried: 2.05 -> 2.5 Mcycles at n = 300)
    {
      const int base = (jt + 1) * jt / 2, ntile = T * (T - 1) / 2 - base;
      xl_mfma_rank16(J, js, n, ntile,
                     [&](int t, int& ti, int& tj) {
                       const int g = t + base;
                       int rr = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)g)) * 0.5f);
                       while (rr * (rr - 1) / 2 > g) rr--;
                       while ((rr + 1) * rr / 2 <= g) rr++;
                       const int cidx = g - rr * (rr - 1) / 2;
                       ti = rr; tj = cidx < jt ? cidx : cidx + 1;
                     },
                     [&](int kk, int ii) { return ii < n ? -Mm[kk * n + ii] : 0.0; },
                     [&](int kk, int jj) { return jj < j0 ? J[(int64_t)(j0 + kk) * js + jj] : (jj < n ? Mm[kk * n + jj] * dd[kk] : 0.0); });
    }
    XSYNC();
  }
  return false;
}

template <class MP>
__device__ __noinline__ int dev_xl_qp_t(const Ctx& c, MP J, const int js) {
  const DgProb& D = dg_prob;
  const DgLds& L = D.L;
  lptr lds = LP(0);
  const int n = D.n, nc = D.nc;
  const QpPtrs q = qp_ptrs(c);
  cgptr Mx = c.ws + D.ws_xM;
  const XlR R{c.ws + D.ws_xR, lds + L.c_R, D.c_rcap, n};
  lptr lhat = lds + L.o_lhat, x = q.xv, np = q.yv, dv = q.cvec, zv = q.wv, rv = q.rv, uu = q.lam, tv = q.tv, acc = q.rd;
  lptr part = lds + L.p_part;
  lds_d* scal = lds + L.scal;
  lds_d* red = lds + L.red;
  const int NONE = 0x7fffffff;
  const double TOL = 1e-10;
  const XlSplit S = xl_split(n);
  const int tr = TID >> 4, tc = TID & 15;       // 32 row groups x 16 consecutive columns for the element-wise updates of J
  PROF_BEGIN(pt_qp);
  __syncthreads();
  for (int r = TID; r < nc; r += NT) q.act[r] = 0;
  bool bad = false;
  if constexpr (!xl_mp<MP>::lds) {
  if (D.xl_el) {
    // ---- the same elimination on a packed lower triangle in LDS (row i at i (i + 1) / 2; the region overlaps the QP outputs and the
    // LDS columns of R, all dead until J exists): two LDS-latency barriers per column instead of two L2 round trips, then ONE pass
    // writes J = L^-T (upper triangular at this point) to the scratch.
    PROF_BEGIN(pxe);
    lptr X = lds + L.x_el;
    for (int e = TID; e < n * n; e += NT) { const int i = e / n, k = e - i * n; if (k <= i) X[i * (i + 1) / 2 + k] = Mx[(int64_t)i * n + k]; }
    __syncthreads();
    for (int j = 0; j < n; j++) {
      const double djj = X[j * (j + 1) / 2 + j];
      if (!(djj > 0.0)) { bad = true; break; }
      if (j == n - 1) break;
      const double inv = 1.0 / djj;
      for (int k = TID; k < n; k += NT) {
        if (k < j) tv[k] = X[j * (j + 1) / 2 + k];
        else if (k > j) { const double akj = X[k * (k + 1) / 2 + j]; tv[k] = akj; np[k] = akj * inv; }
      }
      __syncthreads();
      for (int i = j + 1 + TID; i < n; i += NT) X[i * (i + 1) / 2 + j] = -np[i];
      if (S.g < S.G && S.i < n && S.i != j) {       // thread (g, k): column k of the rows i >= max(j + 1, k), i = j + 1 + g (mod G)
        const int k = S.i, G = S.G;
        const double pk = tv[k];
        int i = j + 1 + S.g;
        if (i < k) i += ((k - i + G - 1) / G) * G;
        for (; i + 3 * G < n; i += 4 * G) {
          const int i1 = i + G, i2 = i + 2 * G, i3 = i + 3 * G;
          lptr p0 = X + i * (i + 1) / 2 + k, p1 = X + i1 * (i1 + 1) / 2 + k, p2 = X + i2 * (i2 + 1) / 2 + k, p3 = X + i3 * (i3 + 1) / 2 + k;
          const double a0 = *p0, a1 = *p1, a2 = *p2, a3 = *p3, m0 = np[i], m1 = np[i1], m2 = np[i2], m3 = np[i3];
          *p0 = a0 - m0 * pk; *p1 = a1 - m1 * pk; *p2 = a2 - m2 * pk; *p3 = a3 - m3 * pk;
        }
        for (; i < n; i += G) X[i * (i + 1) / 2 + k] -= np[i] * pk;
      }
      __syncthreads();
    }
    if (!bad) {
      for (int j = TID; j < n; j += NT) tv[j] = 1.0 / sqrt(X[j * (j + 1) / 2 + j]);
      __syncthreads();
      for (int e = TID; e < n * n; e += NT) {       // J[a][b] = X[b][a] / sqrt(d_b) above the diagonal, 1 / sqrt(d_a) on it, 0 below
        const int a = e / n, b2 = e - a * n;
        J[e] = a < b2 ? X[b2 * (b2 + 1) / 2 + a] * tv[b2] : (a == b2 ? tv[a] : 0.0);
      }
    }
    XSYNC();
    PROF_END(PH_Q_WARM, pxe);
  }
  }
  for (int r = TID; r < nc; r += NT) lhat[r] = 0.0;          // (after the packed elimination: its triangle overlaps the QP outputs)
  const bool el_done = !xl_mp<MP>::lds && D.xl_el;
  if (!el_done) {
  // ---- J = L^-T with M = L L^T, inside J's storage: elimination M = L~ D L~^T by columns with the inverse of the unit factor
  //      accumulated in place.  At step j the pivot vector holds row j left of the diagonal (= row j of X = L~^-1, final) and column j
  //      below it (a_kj, the multipliers m_i = a_ij / d_j); row i > j becomes row_i - m_i * pivot vector on its whole prefix [0, i]:
  //      left of j that is the forward substitution on the identity, right of j the Schur update, and (i, j) itself turns into -m_i.
  //      Every element is independent: two barriers per column, all threads busy.  Then J[c][i] = X[i][c] / sqrt(d_i).
  PROF_BEGIN(px1);
  for (int e = TID; e < n * n; e += NT) { const int i = e / n, k = e - i * n; if (k <= i) J[i * js + k] = Mx[(int64_t)i * n + k]; }
  JSYNC();
#ifdef DG_PROF
  long long pa_ = 0, pb_ = 0, pc_ = 0;
#endif
  bool blocked_el = false;
  if constexpr (!xl_mp<MP>::lds) {
    if (D.xl_blk) {        // 16 pivots per pass over the matrix, the pass on the matrix cores (xl_eliminate_blocked)
#ifdef DG_PROF
      const long long p0_ = clock64();
#endif
      bad = xl_eliminate_blocked(J, js, n, lds + L.x_el, part, tv);
      blocked_el = true;
      for (int r = TID; r < nc; r += NT) lhat[r] = 0.0;      // (the multipliers' slot overlaps the QP outputs)
      __syncthreads();
#ifdef DG_PROF
      pb_ += clock64() - p0_;
#endif
    }
  }
  if (!blocked_el)
  for (int j = 0; j < n; j++) {
#ifdef DG_PROF
    long long p0_ = clock64();
#endif
    const double djj = J[j * js + j];
    if (!(djj > 0.0)) { bad = true; break; }       // (every thread reads the same pivot)
    if (j == n - 1) break;
    const double inv = 1.0 / djj;
    for (int k = TID; k < n; k += NT) {
      if (k < j) tv[k] = J[j * js + k];
      else if (k > j) { const double akj = J[k * js + j]; tv[k] = akj; np[k] = akj * inv; }      // pivot vector / multipliers m_k
    }
    __syncthreads();
#ifdef DG_PROF
    { const long long t_ = clock64(); pa_ += t_ - p0_; p0_ = t_; }
#endif
    for (int i = j + 1 + TID; i < n; i += NT) J[i * js + j] = -np[i];
    // thread (g, k): column k != j of the rows i >= max(j + 1, k), i = j + 1 + g (mod G) -- its pivot-vector entry is loaded once, the
    // multiplier of a row is a broadcast read; four rows per pass, loads first, stores after; no guards inside the loop
    if (S.g < S.G && S.i < n && S.i != j) {
      const int k = S.i, G = S.G;
      const double pk = tv[k];
      int i = j + 1 + S.g;
      if (i < k) i += ((k - i + G - 1) / G) * G;
      MP pj = J + (int64_t)i * js + k;
      clptr pm = np + i;
      const int rs = G * js;
      if constexpr (!xl_mp<MP>::lds) {
        for (; i + 7 * G < n; i += 8 * G, pj += 8 * rs, pm += 8 * G) {      // scratch: eight L2 round trips in flight
          double a[8];
#pragma unroll
          for (int u = 0; u < 8; u++) a[u] = pj[u * rs];
#pragma unroll
          for (int u = 0; u < 8; u++) pj[u * rs] = a[u] - pm[u * G] * pk;
        }
      }
      for (; i + 3 * G < n; i += 4 * G, pj += 4 * rs, pm += 4 * G) {
        const double a0 = pj[0], a1 = pj[rs], a2 = pj[2 * rs], a3 = pj[3 * rs];
        const double m0 = pm[0], m1 = pm[G], m2 = pm[2 * G], m3 = pm[3 * G];
        pj[0] = a0 - m0 * pk; pj[rs] = a1 - m1 * pk; pj[2 * rs] = a2 - m2 * pk; pj[3 * rs] = a3 - m3 * pk;
      }
      for (; i < n; i += G, pj += rs, pm += G) pj[0] = pj[0] - pm[0] * pk;
    }
#ifdef DG_PROF
    { const long long t_ = clock64(); pb_ += t_ - p0_; p0_ = t_; }
#endif
    JSYNC();
#ifdef DG_PROF
    { const long long t_ = clock64(); pc_ += t_ - p0_; p0_ = t_; }
#endif
  }
  PROF_COUNT(PH_W_BUILD, pa_); PROF_COUNT(PH_W_MULT, pb_); PROF_COUNT(PH_W_X, pc_);
  if (!bad) {
  PROF_END(PH_Q_WARM, px1);
  PROF_BEGIN(px2);
  for (int j = TID; j < n; j += NT) tv[j] = 1.0 / sqrt(J[j * js + j]);
  __syncthreads();
  for (int i = tr; i < n; i += 32) {
    const double ri = tv[i];
    MP Ji = J + (int64_t)i * js;
    for (int k = tc; k < i; k += 16) { J[k * js + i] = Ji[k] * ri; Ji[k] = 0.0; }
  }
  for (int j = TID; j < n; j += NT) J[j * js + j] = tv[j];
  JSYNC();
  PROF_END(PH_Q_Y, px2);
  }
  }
  if (bad) { if (TID == 0) scal[DG_QP_NPREV] = 0.0; __syncthreads(); PROF_END(PH_QP, pt_qp); return 2; }
  // ---- x = -M^-1 q = -J (J^T q)
  xl_jt_mul<MP>(J, js, n, S, 0, n, lds + L.q, dv, part);
  xl_j_mul<MP>(J, js, n, S, 0, n, dv, x, part);
  for (int i = TID; i < n; i += NT) x[i] = -x[i];
  __syncthreads();
  int iq = 0, ret = 2;
  auto row_slack = [&](int p) -> double {       // -(g_p + a_p . x), block-uniform; tv must hold a_p
    double s = 0;
    for (int i = TID; i < n; i += NT) s += tv[i] * x[i];
    return -(q.g[p] + block_sum(s, red));
  };
  // a_p into tv, n_p = -a_p into np, d = J^T n_p into dv (an input-bound / rate row has one or two unit coefficients: d is a row of J)
  auto row_products = [&](int p) {
    const DgRow Rw = ld_row(p);
    __syncthreads();
    if (Rw.dense < 0) {
      const int c1 = am_col(D, Rw.a, Rw.k, Rw.idx);
      const bool has0 = (Rw.type == DG_R_RATE_UB || Rw.type == DG_R_RATE_LB) && Rw.k > 0;
      const double sgn = (Rw.type == DG_R_IN_UB || Rw.type == DG_R_RATE_UB) ? 1.0 : -1.0;
      for (int i = TID; i < n; i += NT) {
        double av = i == c1 ? sgn : 0.0;
        if (has0 && i == c1 - DGSQP_NUA) av = -sgn;
        tv[i] = av; np[i] = -av;
        double dj = J[c1 * js + i];
        if (has0) dj -= J[(c1 - DGSQP_NUA) * js + i];
        dv[i] = -sgn * dj;
      }
      __syncthreads();
    } else {
      for (int col = TID; col < n; col += NT) { const double a = q.gdG ? g_row_coef<cgptr>(D, q.gdG, p, col) : g_row_coef<clptr>(D, q.gd, p, col); tv[col] = a; np[col] = -a; }
      __syncthreads();
      xl_jt_mul<MP>(J, js, n, S, 0, n, np, dv, part);
    }
  };
  // `rot` (optional): a vector c = J^T w held in LDS; the rotations that act on the columns of J act on its entries l .. iq the same
  // way, so it follows J without another pass over it (one pass over J per dropped row in the warm start's clean-up and in the main loop)
  auto drop = [&](int l, lptr rot = nullptr, lptr rot2 = nullptr) {   // remove the active constraint at position l, restore R upper triangular (rotations also on J)
    if (TID == 0) q.act[q.alist[l]] = 0;
    __syncthreads();
    // R loses column l: new column j = old column j + 1 (rows 0..j+1, upper Hessenberg from column l on).  Rows above l are copied
    // (thread = row, columns in ascending order: a thread only ever touches its own row)
    for (int i = TID; i < l; i += NT)
      for (int k = l; k < iq - 1; k++) R.set(i, k, R.get(i, k + 1));
    if (TID == 0) {
      for (int k = l; k < iq - 1; k++) { q.alist[k] = q.alist[k + 1]; uu[k] = uu[k + 1]; }
      q.alist[iq - 1] = q.alist[iq]; uu[iq - 1] = uu[iq]; uu[iq] = 0.0; q.alist[iq] = -1;
    }
    XSYNC();
    iq--;
    // Givens rotations of rows (k, k+1), k = l .. iq-1, restore the triangle.  Thread j builds NEW column j from OLD column j + 1 and
    // carries the current value of row k in a register; the coefficients of rotation k come from the thread of column k through LDS.
    // The rows it needs are prefetched XL_RCH at a time -- always before the step (and its barrier) in which the thread of the
    // neighbouring column overwrites them.
    lptr gc = acc, gs = part;
    {
      const int j2 = TID;
      const bool mine = j2 >= l && j2 < iq;
      double ra = mine ? R.get(l, j2 + 1) : 0.0;
      for (int k0 = l; k0 < iq; k0 += XL_RCH) {
        double rbv[XL_RCH];
#pragma unroll
        for (int t = 0; t < XL_RCH; t++) { const int k = k0 + t; rbv[t] = (mine && k < iq && j2 >= k) ? R.get(k + 1, j2 + 1) : 0.0; }
#pragma unroll
        for (int t = 0; t < XL_RCH; t++) {
          const int k = k0 + t;
          if (k < iq) {       // uniform
            if (j2 == k) {
              const double h = sqrt(ra * ra + rbv[t] * rbv[t]);
              gc[k] = h != 0.0 ? ra / h : 1.0; gs[k] = h != 0.0 ? rbv[t] / h : 0.0;
            }
            __syncthreads();
            if (mine && j2 >= k) {
              const double cc = gc[k], s2 = gs[k];
              R.set(k, j2, cc * ra + s2 * rbv[t]);
              ra = -s2 * ra + cc * rbv[t];
            }
          }
        }
      }
    }
    __syncthreads();
    if ((rot && TID == NT - 1) || (rot2 && TID == NT - 2)) {      // (threads that own no row of J for n < NT - 1)
      lptr rv_ = TID == NT - 1 ? rot : rot2;
      double carry = rv_[l];
      for (int k = l; k < iq; k++) {
        const double cc = gc[k], s2 = gs[k], jb = rv_[k + 1];
        rv_[k] = cc * carry + s2 * jb;
        carry = -s2 * carry + cc * jb;
      }
      rv_[iq] = carry;
    }
    // the same rotations on the columns of J: every thread carries its own row through the whole sequence
    for (int i = TID; i < n; i += NT) {
      MP Ji = J + (int64_t)i * js;
      double carry = Ji[l];
      for (int k0 = l; k0 < iq; k0 += XL_RCH) {
        const int cnt = iq - k0 < XL_RCH ? iq - k0 : XL_RCH;
        double jb[XL_RCH], out[XL_RCH];
#pragma unroll
        for (int t = 0; t < XL_RCH; t++) jb[t] = t < cnt ? Ji[k0 + t + 1] : 0.0;
#pragma unroll
        for (int t = 0; t < XL_RCH; t++) {
          const double cc = t < cnt ? gc[k0 + t] : 1.0, s2 = t < cnt ? gs[k0 + t] : 0.0;
          out[t] = cc * carry + s2 * jb[t];
          carry = t < cnt ? -s2 * carry + cc * jb[t] : carry;
        }
#pragma unroll
        for (int t = 0; t < XL_RCH; t++) if (t < cnt) Ji[k0 + t] = out[t];
      }
      Ji[iq] = carry;
    }
    XSYNC();
  };
  // Append row p (d = J^T n_p in dv, sig2 = |d[iq..n-1]|^2) to the factorisation: ONE Householder reflection H maps d2 = d[iq..] onto
  // delta e_1; J2 <- J2 H = J2 - (beta J2 v) v^T with v = d2 - delta e_1, and J2 v = J2 d2 - delta J[:, iq] costs nothing when the
  // step direction z = J2 d2 is at hand (have_z).  Every element of J2 is updated independently -- the rotation-by-rotation form of
  // the textbook method is a serial chain per row.  R gets the column (d1, delta); p becomes active.
  auto absorb = [&](int ip, bool have_z, double sig2) {
    const double x0 = dv[iq];
    const double tail2 = sig2 - x0 * x0;
    double delta = x0, beta = 0.0, v0 = 0.0;
    if (iq + 1 < n && tail2 > 0.0) { delta = x0 >= 0.0 ? -sqrt(sig2) : sqrt(sig2); v0 = x0 - delta; beta = -1.0 / (delta * v0); }
    __syncthreads();        // (x0 is read)
    if (beta != 0.0) {      // uniform
      if (!have_z) xl_j_mul<MP>(J, js, n, S, iq, n, dv, zv, part);
      for (int i = TID; i < n; i += NT) zv[i] = beta * (zv[i] - delta * J[i * js + iq]);
      if (TID == 0) dv[iq] = v0;
      __syncthreads();
      bool paired = false;
      if constexpr (!xl_mp<MP>::lds) {
        // scratch: thread (g, ip) owns the column PAIR (kb + 2 ip, + 1), kb = iq rounded down to even, of the rows g, g + G2, ... --
        // 16-byte loads and stores, n / 2 lanes per row and so three row groups at n = 300, where one column per thread (below)
        // walks all 300 rows alone: 38 dependent L2 round trips per added row against 13.  (A pair that straddles iq rewrites
        // column iq - 1 with its own value: v = 0 there.)
        if (!(n & 1) && !(js & 1) && !((uintptr_t)J & 15)) {
          paired = true;
          const int np2 = ((n >> 1) + 31) & ~31;
          const int G2 = NT / np2, g2 = TID / np2, ip2 = TID - g2 * np2, kb = iq & ~1, k = kb + 2 * ip2;
          if (g2 < G2 && k < n) {
            const double vk0 = k >= iq ? dv[k] : 0.0, vk1 = dv[k + 1];
            int i = g2;
            double2* pj = (double2*)(J + (int64_t)i * js + k);
            const int rs2 = G2 * (js >> 1);
            clptr pw = zv + i;
            for (; i + 7 * G2 < n; i += 8 * G2, pj += 8 * rs2, pw += 8 * G2) {
              double2 a[8];
#pragma unroll
              for (int u = 0; u < 8; u++) a[u] = pj[u * rs2];
#pragma unroll
              for (int u = 0; u < 8; u++) { const double w = pw[u * G2]; a[u].x -= w * vk0; a[u].y -= w * vk1; pj[u * rs2] = a[u]; }
            }
            for (; i < n; i += G2, pj += rs2, pw += G2) { double2 a0 = pj[0]; const double w = pw[0]; a0.x -= w * vk0; a0.y -= w * vk1; pj[0] = a0; }
          }
        }
      }
      if (!paired && S.g < S.G && S.i + iq < n) {     // thread (g, k): column k >= iq of the rows g, g + G, ...
        const int k = S.i + iq, G = S.G, rs = G * js;
        const double vk = dv[k];
        int i = S.g;
        MP pj = J + (int64_t)i * js + k;
        clptr pw = zv + i;
        if constexpr (!xl_mp<MP>::lds) {
          for (; i + 7 * G < n; i += 8 * G, pj += 8 * rs, pw += 8 * G) {
            double a[8];
#pragma unroll
            for (int u = 0; u < 8; u++) a[u] = pj[u * rs];
#pragma unroll
            for (int u = 0; u < 8; u++) pj[u * rs] = a[u] - pw[u * G] * vk;
          }
        }
        for (; i + 3 * G < n; i += 4 * G, pj += 4 * rs, pw += 4 * G) {
          const double a0 = pj[0], a1 = pj[rs], a2 = pj[2 * rs], a3 = pj[3 * rs];
          const double w0 = pw[0], w1 = pw[G], w2 = pw[2 * G], w3 = pw[3 * G];
          pj[0] = a0 - w0 * vk; pj[rs] = a1 - w1 * vk; pj[2 * rs] = a2 - w2 * vk; pj[3 * rs] = a3 - w3 * vk;
        }
        for (; i < n; i += G, pj += rs, pw += G) pj[0] = pj[0] - pw[0] * vk;
      }
    }
    for (int i = TID; i < iq; i += NT) R.set(i, iq, dv[i]);
    if (TID == 0) { R.set(iq, iq, delta); q.act[ip] = 1; }
    XSYNC();
    iq++;
  };
  // ---- warm start (par.qp_warm_start): consecutive QPs of a scenario end on nearly the same active set.  Rebuild the factorisation
  // for the rows the previous QP ended with (skipping rows that have become dependent), take the minimiser on that set,
  //   y1 = R^-T g_W,  u = R^-1 (y1 + J1^T q),  x = J1 y1 - J2 J2^T q      (J^T N = [R; 0] with N = -A_W^T: rows in the form n'x - g >= 0),
  // and drop rows with negative multipliers until the pair (x, W) is dual feasible -- a valid state of the dual method, from
  // which the main loop adds whatever is still violated.  The minimiser is unique: the start only shortens the path.
  // Used by the XL layout only.  The smaller layouts come here for the literal reg = 0 projection (condition 1e12), where the
  // order in which rows enter decides the +-1e-15 residual an active input bound is left with and hence _get_mu's switch
  // (DESIGN.md section 2): there the cold start, which follows the oracle's path, is kept (measured: merge N = 20 32/32 scenarios
  // identical to the oracle cold, 28/32 warm, for a 10 % gain).
  const int nprev = (D.par.qp_warm_start && D.big == 2) ? (int)scal[DG_QP_NPREV] : 0;
  if (nprev > 0) {
    PROF_BEGIN(pxw);
    int jp0 = 0;
    if constexpr (!xl_mp<MP>::lds) {
      // Blocked form (J in the scratch).  Re-absorbing the guessed rows one at a time costs three L2 passes over J per row (J^T n_p, the
      // step direction, the reflection): 84 k cycles per row at n = 300, 10 of the 55 Mcycles of a QP iteration there.  The reflections
      // only depend on D = J^T N_W, so, mcap rows at a time (as many columns of D as fit between the end of the QP scratch and the LDS
      // columns of R: the QP outputs in between are dead until the start has its x): (A) D for the block's rows into LDS (a row of J
      // for a box / rate row, one product for a dense row), (B) Householder QR of D there, column by column with the dependence test
      // of the sequential path -- R and the reflectors --, (C) ONE pass over J applying the block's reflectors to every row
      // (xl_reflect_rows).  Same arithmetic as the loop below up to the order of the sums; a remainder of fewer than four rows goes
      // through that loop.  (Until round 4 only the first block was taken: 5 rows at n = 300.)
      const int mcap = (L.c_R - L.x_el) / n;
      while (mcap >= 4 && nprev - jp0 >= 4 && iq < n && !D.xl_noblock) {
        const int mb = nprev - jp0 < mcap ? nprev - jp0 : mcap, iq0 = iq;
        lptr Dl = lds + L.x_el, betas = zv, refc = rv, npn = acc;
        const int lane = TID & 63, wave = TID >> 6;
        for (int j = 0; j < mb; j++) {                    // (A) dense rows, one product each
          const int p = q.prev[jp0 + j];
          if (ld_row(p).dense < 0) continue;              // uniform
          row_products(p);
          double sn = 0;
          for (int k = TID; k < n; k += NT) { sn += np[k] * np[k]; Dl[j * n + k] = dv[k]; }
          sn = block_sum(sn, red);
          if (TID == 0) npn[j] = sn;
        }
        __syncthreads();
        for (int e = TID; e < mb * n; e += NT) {          // (A) box / rate rows: d = -+ (row c1 of J - row c1 - 2)
          const int j = e / n, i = e - j * n;
          const DgRow Rw = ld_row(q.prev[jp0 + j]);
          if (Rw.dense >= 0) continue;
          const int c1 = am_col(D, Rw.a, Rw.k, Rw.idx);
          const bool has0 = (Rw.type == DG_R_RATE_UB || Rw.type == DG_R_RATE_LB) && Rw.k > 0;
          const double sgn = (Rw.type == DG_R_IN_UB || Rw.type == DG_R_RATE_UB) ? 1.0 : -1.0;
          double dj = J[(int64_t)c1 * js + i];
          if (has0) dj -= J[(int64_t)(c1 - DGSQP_NUA) * js + i];
          Dl[e] = -sgn * dj;
          if (i == 0) npn[j] = has0 ? 2.0 : 1.0;
        }
        __syncthreads();
        for (int j = 0; j < mb && iq < n; j++) {          // (B) QR of D in LDS
          lptr d = Dl + j * n;
          double s2 = 0;
          for (int k = iq + TID; k < n; k += NT) s2 += d[k] * d[k];
          const double s_d2 = block_sum(s2, red);
          if (!(s_d2 > 1e-12 * npn[j])) continue;         // (numerically) dependent on the rows taken so far
          const double x0 = d[iq];
          const double tail2 = s_d2 - x0 * x0;
          double delta = x0, beta = 0.0, v0 = 0.0;
          if (iq + 1 < n && tail2 > 0.0) { delta = x0 >= 0.0 ? -sqrt(s_d2) : sqrt(s_d2); v0 = x0 - delta; beta = -1.0 / (delta * v0); }
          __syncthreads();
          for (int i = TID; i < iq; i += NT) R.set(i, iq, d[i]);
          if (TID == 0) {
            const int p = q.prev[jp0 + j];
            R.set(iq, iq, delta); q.act[p] = 1; q.alist[iq] = p; uu[iq] = 0.0;
            if (beta != 0.0) d[iq] = v0;
            betas[iq] = beta; refc[iq] = (double)j;
          }
          __syncthreads();
          if (beta != 0.0)
            for (int cc = j + 1 + wave; cc < mb; cc += NT / 64) {      // the reflection on the later columns, one wavefront each
              lptr dc = Dl + cc * n;
              double t = 0;
              for (int k = iq + lane; k < n; k += 64) t += d[k] * dc[k];
              t = beta * wave_sum(t);
              for (int k = iq + lane; k < n; k += 64) dc[k] -= t * d[k];
            }
          iq++;
          __syncthreads();
        }
        xl_reflect_rows(J, js, n, iq0, iq, betas, refc, Dl);          // (C) J <- J H_iq0 ... H_{iq-1}
        XSYNC();
        jp0 += mb;
      }
    }
    PROF_END(PH_QW_BLK, pxw);
    PROF_COUNT(PH_QW_NPREV, nprev);
    PROF_BEGIN(pxs);
    for (int jp = jp0; jp < nprev && iq < n; jp++) {
      const int p = q.prev[jp];
      row_products(p);
      double s_d2 = 0, s_np = 0;
      for (int k = TID; k < n; k += NT) { if (k >= iq) s_d2 += dv[k] * dv[k]; s_np += np[k] * np[k]; }
      s_d2 = block_sum(s_d2, red); s_np = block_sum(s_np, red);
      if (!(s_d2 > 1e-12 * s_np)) continue;               // (numerically) dependent on the rows taken so far
      if (TID == 0) { q.alist[iq] = p; uu[iq] = 0.0; }
      __syncthreads();
      absorb(p, false, s_d2);
    }
    PROF_END(PH_QW_SEQ, pxs);
    PROF_BEGIN(pxf);
    int ndrop_ = 0;
    // c = J^T q  (dv)
    xl_jt_mul<MP>(J, js, n, S, 0, n, lds + L.q, dv, part);
    // forward substitution R^T y1 = g_W once: a dropped row's rotations G turn R into [R'; 0] and y1 into the leading part of G y1
    // (R'^T (G y1) = g_W' column by column), so y1 follows the rotations like c = J^T q does -- one triangular solve per drop, not two
    for (int k = TID; k < n; k += NT) { acc[k] = k < iq ? q.g[q.alist[k]] : 0.0; zv[k] = 0.0; }
    __syncthreads();
    if (iq > 0 && TID < 64) xl_wave_fwdsub(R, iq, acc, zv);
    __syncthreads();
    for (int guard = 0; guard <= n && iq > 0; guard++) {
      // back substitution R u = y1 + c1 (wavefront 0)
      for (int k = TID; k < iq; k += NT) acc[k] = zv[k] + dv[k];
      __syncthreads();
      if (TID < 64) xl_wave_backsub(R, iq, acc, uu);
      __syncthreads();
      double umin = INFINITY, umax = 0.0; int kmin = NONE;
      for (int k = TID; k < iq; k += NT) { if (uu[k] < umin) { umin = uu[k]; kmin = k; } umax = fmax(umax, fabs(uu[k])); }
      { double bv; int bi; block_argmin(umin, kmin, red, bv, bi); umin = bv; kmin = bi; }
      umax = block_max(umax, red);
      if (!(umin < -1e-10 * (1.0 + umax))) break;
      // drop the row with the most negative multiplier; J's rotations act on c = J^T q as well
      drop(kmin, dv, zv); ndrop_++;
    }
    PROF_END(PH_QW_FIN, pxf);
    PROF_COUNT(PH_QW_DROPS, ndrop_);
    // x = J1 y1 - J2 c2 ; multipliers clipped at 0 (rounding)
    for (int k = TID; k < n; k += NT) np[k] = k < iq ? zv[k] : -dv[k];
    __syncthreads();
    xl_j_mul<MP>(J, js, n, S, 0, n, np, x, part);
    for (int k = TID; k < iq; k += NT) uu[k] = fmax(uu[k], 0.0);
    for (int r = TID; r < nc; r += NT) lhat[r] = 0.0;          // (the blocks of D reach over the QP outputs)
    __syncthreads();
    PROF_END(PH_Q_WARM, pxw);
  }
  const int max_outer = 20 * (n + nc);
  for (int iter = 0; iter < max_outer; iter++) {
    PROF_BEGIN(px3);
    const int ip = qp_scan(q, TOL);
    PROF_END(PH_Q_SCAN, px3);
    if (ip == NONE) { ret = 0; break; }
    PROF_BEGIN(px4a);
    row_products(ip);
    if (TID == 0) { uu[iq] = 0.0; q.alist[iq] = ip; }
    double npnp;
    { double s = 0; for (int i = TID; i < n; i += NT) s += np[i] * np[i]; npnp = block_sum(s, red); }
    double sp = row_slack(ip);
    PROF_END(PH_Q_DIR, px4a);
    int st = -1;          // -1 running, 0 constraint added, 1 infeasible, 2 iteration limit
    bool have_d = true;   // dv = J^T n_p is current (row_products); a drop rotates it along with J
    bool have_z = false;  // zv = J2 d2 of the previous inner step is still valid up to the one column a drop moved into J2
    for (int inner = 0; inner < 10 * (n + nc) && st < 0; inner++) {
      // step 2a: d = J^T np ; z = J2 d2 ; r = R^-1 d1
      PROF_BEGIN(px4);
      if (!have_d) xl_jt_mul<MP>(J, js, n, S, 0, n, np, dv, part);
      have_d = false;
      // z = J2 d2: one pass over J -- except right after a drop, where J2 has only gained the column iq and d2 the entry d[iq]
      // (the rotations of the drop touch the columns l .. iq, none of the old J2): z += d[iq] J[:, iq]
      if (have_z) {
        for (int i = TID; i < n; i += NT) zv[i] += dv[iq] * J[(int64_t)i * js + iq];
        __syncthreads();
        if (TID < 64) xl_wave_backsub(R, iq, dv, rv);
      } else if constexpr (!xl_mp<MP>::lds) {
        // matrix in the scratch: wavefront 0 solves r = R^-1 d1 while the other seven form z = J2 d2
        if (TID < 64) xl_wave_backsub(R, iq, dv, rv);
        xl_j_mul_rows_but_wave0(J, js, n, iq, n, dv, zv);
      } else {
        xl_j_mul<MP>(J, js, n, S, iq, n, dv, zv, part);
        if (TID < 64) xl_wave_backsub(R, iq, dv, rv);
      }
      have_z = false;
      __syncthreads();
      PROF_END(PH_Q_DIR, px4);
      // step 2b: step lengths
      double t1 = INFINITY; int lidx = NONE;
      for (int k = TID; k < iq; k += NT) if (rv[k] > 0.0) { const double tt = uu[k] / rv[k]; if (tt < t1) { t1 = tt; lidx = k; } }
      { double bv; int bi; block_argmin(t1, lidx, red, bv, bi); t1 = bv; lidx = bi; }
      double znp;
      { double s = 0; for (int k = iq + TID; k < n; k += NT) s += dv[k] * dv[k]; znp = block_sum(s, red); }
      const double t2 = (znp > 1e-18 * npnp && iq < n) ? -sp / znp : INFINITY;
      const double t = fmin(t1, t2);
      if (!(t < INFINITY)) { st = 1; break; }
      if (!(t2 < INFINITY)) {   // dual step only
        for (int k = TID; k < iq; k += NT) uu[k] -= t * rv[k];
        if (TID == 0) uu[iq] += t;
        __syncthreads();
        drop(lidx, dv);         // (d = J^T n_p follows the rotations)
        have_d = true; have_z = true;
        continue;
      }
      for (int i = TID; i < n; i += NT) x[i] += t * zv[i];
      for (int k = TID; k < iq; k += NT) uu[k] -= t * rv[k];
      if (TID == 0) uu[iq] += t;
      __syncthreads();
      PROF_BEGIN(px5);
      if (t == t2) {   // full step: add constraint ip
        absorb(ip, true, znp);
        st = 0;
      } else {          // partial step: drop the blocking constraint, recompute the slack of ip
        drop(lidx, dv);
        have_d = true; have_z = true;
        sp = row_slack(ip);
      }
      PROF_END(PH_Q_UPD, px5);
    }
    if (st < 0) st = 2;
    if (st != 0) { ret = st; break; }
  }
  // ---- polish: the KKT point of the final active set W.  The dual method reaches the optimal active set, but its iterate is an
  // accumulation that starts at x = -M^-1 q -- of size 1e10 |q| along the directions _nearestPD floored at 1e-10 -- and cancels
  // back to O(1): on the literal reg = 0 projection the point it ends on is off by up to 0.5 |x| (tools/reg0_qp_study.py), while the
  // minimiser itself is well determined (the active rows pin those directions; the KKT matrix has condition ~1e7).  The reference
  // returns OSQP's POLISHED point, the solution of the KKT system of the active set (DGSQP.py:186, osqp polish); the oracle solves
  // that system by dense LU.  Here: two steps of iterative refinement with fp64 residuals, the factorisation at hand (J^T M J = I to
  // ~1e-4, J^T N = [R; 0] to ~1e-10) as the approximate inverse -- each step gains four or more digits:
  //     r1 = -(q + M x + A_W^T u),  r2 = -(g_W + A_W x);   R^T y1 = -r2,  y2 = J2^T r1;   dx = J [y1; y2],  du = R^-1 (y1 - J1^T r1).
  // The polished point is kept when it is primal and dual feasible to 1e-9 (else the iterate stands: OSQP's "polish unsuccessful").
  if (ret == 0 && iq > 0) {
    PROF_BEGIN(pxp);
    lptr xb = tv, ub = acc, r1 = np, rhs2 = zv, y1 = rv;
    __syncthreads();
    for (int i = TID; i < n; i += NT) xb[i] = x[i];
    for (int k = TID; k < iq; k += NT) ub[k] = uu[k];
    for (int pass = 0; pass < 2; pass++) {
      __syncthreads();
      for (int k = TID; k < iq; k += NT) lhat[q.alist[k]] = uu[k];        // (lhat is zero elsewhere)
      __syncthreads();
      gt_mul(c, lhat, r1);                                                  // A_W^T u
      xl_jt_mul<cgptr>(Mx, n, n, S, 0, n, x, dv, part);                     // M x (M symmetric)
      if (q.gdG) qp_dense_dots<cgptr>(D, q.gdG, x, q.dpart, q.ddx); else qp_dense_dots<clptr>(D, q.gd, x, q.dpart, q.ddx);
      for (int i = TID; i < n; i += NT) r1[i] = -(lds[L.q + i] + dv[i] + r1[i]);
      for (int k = TID; k < iq; k += NT) { const int r = q.alist[k]; rhs2[k] = q.g[r] + qpw_row_dot(D, ld_row(r), x, q.ddx); }      // = -r2
      __syncthreads();
      xl_jt_mul<MP>(J, js, n, S, 0, n, r1, dv, part);                       // [J1^T r1; J2^T r1]
      if (TID < 64) xl_wave_fwdsub(R, iq, rhs2, y1);
      __syncthreads();
      for (int k = TID; k < iq; k += NT) { rhs2[k] = y1[k] - dv[k]; dv[k] = y1[k]; }
      __syncthreads();
      xl_j_mul<MP>(J, js, n, S, 0, n, dv, r1, part);                        // dx
      if (TID < 64) xl_wave_backsub(R, iq, rhs2, y1);                       // du
      __syncthreads();
      for (int i = TID; i < n; i += NT) x[i] += r1[i];
      for (int k = TID; k < iq; k += NT) uu[k] += y1[k];
    }
    __syncthreads();
    double lmin = 0.0, lmax = 0.0;
    for (int k = TID; k < iq; k += NT) { lmin = fmin(lmin, uu[k]); lmax = fmax(lmax, fabs(uu[k])); }
    lmin = -block_max(-lmin, red); lmax = block_max(lmax, red);
    bool keep = lmin >= -1e-9 * (1.0 + lmax);
    for (int i = TID; i < n; i += NT) keep = keep && (x[i] == x[i]);
    keep = !__syncthreads_or(!keep) && qp_scan(q, 1e-9) == NONE;
    if (!keep) {
      for (int i = TID; i < n; i += NT) x[i] = xb[i];
      for (int k = TID; k < iq; k += NT) uu[k] = ub[k];
    } else {
      for (int k = TID; k < iq; k += NT) uu[k] = fmax(uu[k], 0.0);
    }
    PROF_COUNT(PH_C_MWARM, keep ? 1 : 0);
    __syncthreads();
    for (int k = TID; k < iq; k += NT) lhat[q.alist[k]] = 0.0;
    PROF_END(PH_Q_REFINE, pxp);
  }
  if (ret == 0) {
    __syncthreads();
    for (int k = TID; k < iq; k += NT) { lhat[q.alist[k]] = uu[k]; q.prev[k] = q.alist[k]; }
    __syncthreads();
    // par.snap_active_bounds (default 0 = literal): put du exactly on its active input bounds (same knob in dev_qp and the oracle)
    for (int k = TID; D.par.snap_active_bounds && k < iq; k += NT) {
      const int r = q.alist[k];
      const DgRow Rw = ld_row(r);
      if (uu[k] > 0.0 && Rw.type == DG_R_IN_UB) x[am_col(D, Rw.a, Rw.k, Rw.idx)] = -q.g[r];
      else if (uu[k] > 0.0 && Rw.type == DG_R_IN_LB) x[am_col(D, Rw.a, Rw.k, Rw.idx)] = q.g[r];
    }
  }
  __syncthreads();
  if (TID == 0) scal[DG_QP_NPREV] = ret == 0 ? (double)iq : 0.0;      // final active set kept in q.prev for the next QP of this scenario
  __syncthreads();
  PROF_END(PH_QP, pt_qp);
  return ret;
}

__device__ inline int dev_xl_qp(const Ctx& c) {
  const DgProb& D = dg_prob;
  if (D.big == 0) return dev_xl_qp_t<lptr>(c, LP(D.L.g_Bp), D.n + 1);     // P + R slots: n (n + 1) doubles
  return dev_xl_qp_t<gptr>(c, c.ws + D.ws_xJ, D.n);
}

__device__ void dev_xl_psd(const Ctx& c, gptr Qpd) {
  if (!dev_xl_psd_tri(c, Qpd)) dev_xl_psd_jacobi(c, Qpd);     // (more than XL_KMAX negative eigenvalues)
}
