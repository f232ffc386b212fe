// PSD projection (_nearestPD), the QP sub-problem (_solve_qp), the dual initialisation (LSQR),
// the merit / line-search / watchdog logic and the SQP outer loop of DGSQP.solve() on device.
#pragma once
#include "dgsqp_eval.h"

__device__ inline int tri(int i, int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }

// ------------------------------------------------------------------------------------------------
// Fast path of _nearestPD + reg + inverse.  The symmetrised game Hessian has only a handful of negative
// eigenvalues (1-8 in the racing games), so instead of a full eigendecomposition:
//   1. Householder tridiagonalisation  B = Q T Q^T  in LDS (full storage, reflectors kept in place),
//   2. Sturm count at 0 -> number of negative eigenvalues; each one by 64-way multisection (one wavefront
//      evaluates 64 Sturm sequences per step),
//   3. eigenvectors of T by inverse iteration (tridiagonal LU with partial pivoting, one lane each),
//      modified Gram-Schmidt, back-transformation through the reflectors (one wavefront per vector),
//   4. M = B + sum_j (floor - lambda_j) v_j v_j^T + reg I   (== U diag(s') U^T + reg I of DGSQP.py:1290-1296, floor = 1e-10 there),
//   5. P = M^-1 by the symmetric Gauss-Jordan sweep (SPD => no pivoting), packed into L.g_Bp.
// Returns false (nothing written) when there are more than PSD_KMAX negative eigenvalues: caller falls back
// to the Jacobi route.
// ------------------------------------------------------------------------------------------------
#include <type_traits>
#define PSD_KMAX DG_PSD_KMAX
// Number of eigenvalues of the symmetric tridiagonal (d, e) below sigma = number of sign changes of the
// Sturm sequence p_0 = 1, p_1 = d_0 - s, p_{i+1} = (d_i - s) p_i - e_{i-1}^2 p_{i-1}  (division-free, rescaled).
__device__ inline int sturm_count(clptr d, clptr e2, int n, double sigma, double pivmin) {
  (void)pivmin;
  double pp = 1.0, pc = d[0] - sigma;
  int cnt = pc < 0.0 || pc == 0.0;  // a zero is counted as a sign change (same convention as q = -pivmin)
  if (pc == 0.0) pc = -1e-300;
  for (int i = 1; i < n; i++) {
    double pn = (d[i] - sigma) * pc - e2[i - 1] * pp;
    if (pn == 0.0) pn = pc > 0.0 ? -1e-300 * fabs(pc) - 1e-320 : 1e-300 * fabs(pc) + 1e-320;
    cnt += (pn < 0.0) != (pc < 0.0);
    pp = pc; pc = pn;
    const double a = fabs(pc);
    if (a > 1e120) { pc *= 1e-120; pp *= 1e-120; }
    else if (a < 1e-120) { pc *= 1e120; pp *= 1e120; }
  }
  return cnt;
}
// Sturm count (number of eigenvalues of the tridiagonal below sigma) with d and e^2 DISTRIBUTED OVER THE LANES of the
// calling wavefront (xA: rows 0..63, xB: rows 64..127) and fetched with v_readlane: the loop index is wave-uniform, so
// the operands arrive as scalars and the only latency left is the recurrence itself.  Product form of the Sturm sequence
//   p_0 = 1, p_1 = d_0 - s, p_{i+1} = (d_i - s) p_i - e_{i-1}^2 p_{i-1}      (count = sign changes, a zero counts as one)
// whose dependent chain is one multiply and one fma per row; both carried terms are rescaled by a power of two every
// four rows (frexp / ldexp), which keeps them far inside the fp64 range for |T| < 1e30.  sigma may differ per lane.
__device__ inline int sturm_count_reg(double dA, double dB, double e2A, double e2B, int n, double sigma, double pivmin) {
  (void)pivmin;
  double pp = 1.0, pc = lane_bcast(dA, 0) - sigma;
  bool neg = !(pc > 0.0);           // "not positive": an exact zero counts as negative; then p_{i+1} = -e^2 p_{i-1}
  int cnt = neg;                    // has the opposite sign of p_{i-1}, so no special value is needed
  // one row; HD / HE select the register half of d_r / e^2_{r-1} at compile time
#define STURM_ROW(r, HD, HE)                                                                     \
  {                                                                                              \
    const double di = lane_bcast(HD ? dB : dA, (r) & 63), ei = lane_bcast(HE ? e2B : e2A, ((r) - 1) & 63); \
    const double pn = __builtin_fma(di - sigma, pc, -ei * pp);                                   \
    const bool nn = !(pn > 0.0);                                                                 \
    cnt += nn != neg;                                                                            \
    neg = nn; pp = pc; pc = pn;                                                                  \
  }
#define STURM_RESCALE()                                                                          \
  {                                                                                              \
    const int ex = __builtin_amdgcn_frexp_exp(pc);                                               \
    const int sc = (ex > 200 || ex < -200) ? -ex : 0;                                            \
    pc = __builtin_ldexp(pc, sc); pp = __builtin_ldexp(pp, sc);                                  \
  }
  const int nA = n < 64 ? n : 64;   // rows 1 .. nA-1: d and e^2 both from the A half
  int i = 1;
  for (; i + 3 < nA; i += 4) { STURM_ROW(i, 0, 0) STURM_ROW(i + 1, 0, 0) STURM_ROW(i + 2, 0, 0) STURM_ROW(i + 3, 0, 0) STURM_RESCALE() }
  for (; i < nA; i++) STURM_ROW(i, 0, 0)
  if (n > 64) {
    STURM_ROW(64, 1, 0)             // d_64 from the B half, e^2_63 from the A half
    STURM_RESCALE()
    i = 65;
    for (; i + 3 < n; i += 4) { STURM_ROW(i, 1, 1) STURM_ROW(i + 1, 1, 1) STURM_ROW(i + 2, 1, 1) STURM_ROW(i + 3, 1, 1) STURM_RESCALE() }
    for (; i < n; i++) STURM_ROW(i, 1, 1)
  }
#undef STURM_ROW
#undef STURM_RESCALE
  return cnt;
}
// inclusive prefix / suffix products over the 64 lanes of a wavefront
__device__ inline double wave_prefix_prod(double v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const double t = __shfl_up(v, d, 64); v = lane >= d ? v * t : v; }
  return v;
}
__device__ inline double wave_suffix_prod(double v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const double t = __shfl_down(v, d, 64); v = lane + d < 64 ? v * t : v; }
  return v;
}
// Eigenvector of the tridiagonal for the (accurately known) eigenvalue lam by the twisted factorisation of T - lam I
// (Parlett & Dhillon; LAPACK dlar1v): stationary qd from the top, progressive qd from the bottom, twist where
// |gamma_r| = |D+_r + D-_r - (d_r - lam)| is smallest, then z_r = 1 and the two two-term recurrences
//   z_i = -(e_i / D+_i) z_{i+1}  (i < r),   z_{i+1} = -(e_i / D-_{i+1}) z_i  (i >= r),
// which are prefix / suffix products and are evaluated as wavefront scans.  One wavefront per eigenvalue; d, e come from
// lane-distributed registers (lane i <-> rows i, i+64), the two pivot sequences go through the wavefront's LDS strip
// `strip` (3 n doubles).  z is returned max-normalised.
__device__ inline void twisted_eigvec(double dA, double dB, double eA, double eB, int n, double lam, double pivmin, int lane,
                                      lptr strip, double& zA, double& zB) {
  const double e2A = eA * eA, e2B = eB * eB;
  const double tA = dA - lam, tB = dB - lam;
  const int npad = (n + 1) & ~1;
  lptr Dp = strip, Dm = strip + npad, Ds = strip + 2 * npad;     // D+_i, D-_i, D-_{i+1}
  const int nl = n - 1;
  double qf = lane_bcast(tA, 0);
  qf = (__builtin_fabs(qf) <= pivmin) ? -pivmin : qf;
  double qb = lane_bcast(nl < 64 ? tB * 0.0 + tA : tB, nl & 63);
  qb = (__builtin_fabs(qb) <= pivmin) ? -pivmin : qb;
  if (lane == 0) { Dp[0] = qf; Dm[nl] = qb; if (nl >= 1) Ds[nl - 1] = qb; Ds[nl] = 1.0; }
  for (int s = 1; s < n; s++) {
    const int i = s, ib = nl - s;
    const double ti = lane_bcast(i < 64 ? tA : tB, i & 63), ei = lane_bcast(i - 1 < 64 ? e2A : e2B, (i - 1) & 63);
    const double tb = lane_bcast(ib < 64 ? tA : tB, ib & 63), eb = lane_bcast(ib < 64 ? e2A : e2B, ib & 63);
    qf = ti - ei * fast_rcp(qf);
    qb = tb - eb * fast_rcp(qb);
    qf = (__builtin_fabs(qf) <= pivmin) ? -pivmin : qf;
    qb = (__builtin_fabs(qb) <= pivmin) ? -pivmin : qb;
    if (lane == 0) { Dp[i] = qf; Dm[ib] = qb; if (ib >= 1) Ds[ib - 1] = qb; }
  }
  // twist index
  const double pA = lane < n ? Dp[lane] : 1.0, pB = lane + 64 < n ? Dp[lane + 64] : 1.0;
  double gA = lane < n ? __builtin_fabs(pA + Dm[lane] - tA) : INFINITY;
  double gB = lane + 64 < n ? __builtin_fabs(pB + Dm[lane + 64] - tB) : INFINITY;
  int r = lane;
  if (gB < gA) { gA = gB; r = lane + 64; }
  wave_argmin(gA, r);
  // multipliers, set to 1 outside their range:  L_k (k < r) below the twist, U_k (r <= k < n-1) above it
  const int ka = lane, kb = lane + 64;
  const double LA = ka < r ? -eA * fast_rcp(pA) : 1.0, LB = kb < r ? -eB * fast_rcp(pB) : 1.0;
  const double UA = (ka >= r && ka < nl) ? -eA * fast_rcp(Ds[ka]) : 1.0, UB = (kb >= r && kb < nl) ? -eB * fast_rcp(Ds[kb]) : 1.0;
  // z_i = prod_{k=i}^{r-1} L_k  (suffix product)  *  prod_{k=r}^{i-1} U_k  (exclusive prefix product)
  const double sB = wave_suffix_prod(LB, lane);
  const double sA = wave_suffix_prod(LA, lane) * lane_bcast(sB, 0);
  const double pfA = wave_prefix_prod(UA, lane);
  const double pfB = wave_prefix_prod(UB, lane) * lane_bcast(pfA, 63);
  // exclusive: shift by one lane
  double exA = __shfl_up(pfA, 1, 64), exB = __shfl_up(pfB, 1, 64);
  exA = lane == 0 ? 1.0 : exA;
  exB = lane == 0 ? lane_bcast(pfA, 63) : exB;
  zA = lane < n ? sA * exA : 0.0;
  zB = lane + 64 < n ? sB * exB : 0.0;
  const double nr = 1.0 / wave_max(fmax(__builtin_fabs(zA), __builtin_fabs(zB)));
  zA *= nr; zB *= nr;
}
// Symmetric Gauss-Jordan sweep of an SPD matrix held in registers (thread (jc = TID & 127, hf = TID >> 7) owns column jc, rows
// hf + NH r): after all n pivots the slice holds -A^-1 (SPD => no pivoting).  TWO pivots per barrier: with the columns k and k + 1 of
// the current matrix published, the sweep of pivot k followed by the sweep of pivot k + 1 is
//   d1 = a_kk,  r1_j = a_kj / d1,  c'_i = a_i,k+1 - a_ik r1_{k+1}  (column k + 1 after the first sweep),  d2 = c'_{k+1},  r2_j = c'_j / d2,
//   a_ij <- a_ij - a_ik r1_j - c'_i r2_j                                                  (i, j outside {k, k + 1})
//   column k: a_ik / d1 - c'_i r2_k,   column k + 1: c'_i / d2,   row k: r1_j - r1_{k+1} r2_j,   row k + 1: r2_j,
//   [k][k] = -1/d1 - r1_{k+1} r2_k,   [k][k+1] = [k+1][k] = r2_k = r1_{k+1} / d2,   [k+1][k+1] = -1/d2
// -- every thread derives the scalars itself from the two published columns, so a step is one round of LDS reads, two multiply-adds per
// entry and one barrier (the one-pivot version paid the read latency and the barrier per pivot: 2.5 kcycles each, 100 of them).
// tws: 4 (NH RPT + 4) doubles of LDS (two column pairs, double buffered).
// CHECK: returns false (block-uniform; the slice is left half swept) at the first non-positive pivot -- the inertia test of _nearestPD's
// shortcut: a symmetric matrix whose elimination pivots are all positive has no eigenvalue <= 0.
template <int RPT, bool CHECK = false>
__device__ __forceinline__ bool spd_sweep_regs(double (&Br)[RPT], lptr tws, int n) {
  constexpr int NH = DG_NH, CS = NH * RPT + 4;
  const int jc = TID & 127, hf = TID >> 7;
  const bool colok = jc < n;
  for (int i = TID; i < 4 * CS; i += NT) tws[i] = 0.0;    // padding rows stay zero
  __syncthreads();
  if (jc < 2) {
#pragma unroll
    for (int r = 0; r < RPT; r++) tws[jc * CS + hf + NH * r] = Br[r];
  }
  __syncthreads();
  int k = 0;
  for (; k + 1 < n; k += 2) {
    clptr c1 = tws + ((k >> 1) & 1) * 2 * CS, c2 = c1 + CS;
    lptr nx = tws + (((k >> 1) + 1) & 1) * 2 * CS;
    const double d1 = c1[k], a12 = c1[k + 1], a22 = c2[k + 1];
    const double ck = colok ? c1[jc] : 0.0, ck1 = colok ? c2[jc] : 0.0;
    const double id1 = fast_rcp(d1);
    const double r1k1 = a12 * id1;
    const double d2 = __builtin_fma(-a12, r1k1, a22);
    if (CHECK && !(d1 > 0.0 && d2 > 0.0)) return false;      // (every thread reads the same two published columns)
    const double id2 = fast_rcp(d2);
    const double r2k = r1k1 * id2;
    const double r1 = ck * id1;
    const double r2 = __builtin_fma(-ck, r1k1, ck1) * id2;
    // the update in two halves of the rows: half as many column entries live at a time (the slice itself fills a third of the registers)
    constexpr int RH = (RPT + 1) / 2;
#pragma unroll
    for (int h0 = 0; h0 < RPT; h0 += RH) {
      double ci1[RH], ci2[RH];
#pragma unroll
      for (int r = 0; r < RH; r++) { const int rr = h0 + r < RPT ? h0 + r : RPT - 1; ci1[r] = c1[hf + NH * rr]; ci2[r] = c2[hf + NH * rr]; }
#pragma unroll
      for (int r = 0; r < RH; r++) ci2[r] = __builtin_fma(-ci1[r], r1k1, ci2[r]);        // c'_i (rows k, k + 1 are overwritten below)
      if (jc != k && jc != k + 1) {
#pragma unroll
        for (int r = 0; r < RH; r++) if (h0 + r < RPT) Br[h0 + r] = __builtin_fma(-ci2[r], r2, __builtin_fma(-ci1[r], r1, Br[h0 + r]));
      } else if (jc == k) {
#pragma unroll
        for (int r = 0; r < RH; r++) if (h0 + r < RPT) Br[h0 + r] = __builtin_fma(-ci2[r], r2k, ci1[r] * id1);
      } else {
#pragma unroll
        for (int r = 0; r < RH; r++) if (h0 + r < RPT) Br[h0 + r] = ci2[r] * id2;
      }
    }
    if (hf == (k & (NH - 1))) {            // row k: one register (index k / NH, wave-uniform) of these threads
      const int rs = k / NH;
      const double v = jc == k ? __builtin_fma(-r1k1, r2k, -id1) : (jc == k + 1 ? r2k : __builtin_fma(-r1k1, r2, r1));
#pragma unroll
      for (int r = 0; r < RPT; r++) Br[r] = (r == rs) ? v : Br[r];
    }
    if (hf == ((k + 1) & (NH - 1))) {      // row k + 1
      const int rs = (k + 1) / NH;
      const double v = jc == k ? r2k : (jc == k + 1 ? -id2 : r2);
#pragma unroll
      for (int r = 0; r < RPT; r++) Br[r] = (r == rs) ? v : Br[r];
    }
    if (jc == k + 2 || jc == k + 3) {      // the next pair of pivot columns is final as soon as this update is done
      lptr dst = nx + (jc - k - 2) * CS;
#pragma unroll
      for (int r = 0; r < RPT; r++) dst[hf + NH * r] = Br[r];
    }
    __syncthreads();
  }
  if (k < n) {                             // odd n: the last pivot alone (its column is the first of the current pair)
    clptr colk = tws + ((k >> 1) & 1) * 2 * CS;
    if (CHECK && !(colk[k] > 0.0)) return false;
    const double dinv = fast_rcp(colk[k]);
    const double rj = colok ? colk[jc] * dinv : 0.0;
    const bool pc = jc == k;
    double ci[RPT];
#pragma unroll
    for (int r = 0; r < RPT; r++) ci[r] = colk[hf + NH * r];
    if (!pc) {
#pragma unroll
      for (int r = 0; r < RPT; r++) Br[r] = fma(-ci[r], rj, Br[r]);
    } else {
#pragma unroll
      for (int r = 0; r < RPT; r++) Br[r] = ci[r] * dinv;
    }
    if (hf == (k & (NH - 1))) {
      const int rs = k / NH;
      const double rowv = pc ? -dinv : rj;
#pragma unroll
      for (int r = 0; r < RPT; r++) Br[r] = (r == rs) ? rowv : Br[r];
    }
    __syncthreads();
  }
  return true;
}
#define DG_PSD_PD 61     // scal slot: the scenario's previous _nearestPD found no negative eigenvalue (the next one tries the shortcut)
// Register-resident layout: thread (jc = TID & 127, hf = TID >> 7) owns column jc, rows hf, hf+2, hf+4, ...
// (RPT of them) of the symmetric matrix for the whole Householder reduction AND the Gauss-Jordan sweep; LDS
// only carries the broadcast vectors (reflector v, w, pivot column) and the stored reflectors.
template <int RPT>
__device__ __noinline__ bool dev_psd_inverse_tridiag(const Ctx& c, gptr Qpd, bool want_inverse) {
  const DgProb& D = dg_prob;
  const DgLds& L = D.L;
  lptr lds = LP(0);
  const int n = D.n;
  if (n > DG_NH * RPT || n > 128 || n < 4) return false;
  lds_d* Rf = lds + (D.big ? 0 : L.g_V);   // Householder reflectors, strict lower triangle packed by columns
  gptr RfG = c.ws + D.ws_V;                // ... in the workgroup's global scratch for games beyond the LDS layout
  const bool big = D.big != 0;
#define RFOFF(k) ((k) * (2 * n - (k) - 1) / 2)
  lds_d* Wk = lds + L.g_tw;  // workspace (aliases the packed slot when LDS is tight: P is written there last)
  constexpr int NH = DG_NH;
  lds_d *dv = Wk, *ev = Wk + n, *tau = Wk + 2 * n, *vv = Wk + 3 * n, *pp = Wk + 4 * n /* NH*n */;
  lptr lamv = Wk + (4 + NH) * n;       // PSD_KMAX (+ pad)
  lptr Z = Wk + (4 + NH) * n + 16;     // PSD_KMAX x n eigenvectors
  lptr tws = Z + PSD_KMAX * n;        // published-column / w buffers of the Householder and sweep loops
  lds_d* red = lds + L.red;
  lds_d* scal = lds + L.scal;
  cgptr Qg = c.ws + D.ws_q;
  const int lane = TID & 63, wave = TID >> 6;
  const int jc = TID & 127, hf = TID >> 7;
  const bool colok = jc < n;
  __syncthreads();
  PROF_BEGIN(pt_t);
  // ---- 1. B = (Q + Q^T)/2, this thread's slice
  double Br[RPT];
#pragma unroll
  for (int r = 0; r < RPT; r++) {
    const int i = hf + NH * r;
    Br[r] = (colok && i < n) ? 0.5 * (Qg[(int64_t)i * n + jc] + Qg[(int64_t)jc * n + i]) : 0.0;
  }
  // ---- 1b. Shortcut (round 4).  _nearestPD only changes B when it has negative eigenvalues, and 41 % of configs[1]'s calls find
  //          none -- after 0.5 Mcycles of tridiagonalisation.  When the scenario's PREVIOUS call found none either, the inertia is
  //          tested first: the pivots of a symmetric elimination of B (the Gauss-Jordan sweep of step 5, stopped at the first non-positive
  //          pivot) are all positive exactly when no eigenvalue is <= 0.  Then M = B + reg I as step 4a forms it from the same
  //          numbers -- identical results, without steps 2 and 3.  A failed test costs one sweep (0.2 Mcycles), two when the certified
  //          sweep of M passes its pivots but misses the norm bound and the inertia sweep of B then fails.
  bool pd_fast = false, have_P = false, m_pivot_failed = false;
  const double reg0 = dev_reg();
  if (scal[DG_PSD_PD] == 1.0 && want_inverse && reg0 > 0.0 && !Qpd) {
    // ... and cheaper still when P = M^-1 is wanted anyway: sweep M = B + reg I itself (pivots checked: M is positive definite) and
    // bound its inverse, lambda_max(M^-1) <= ||M^-1||_inf.  If that is below 1 / reg, lambda_min(M) > reg, i.e. B has no eigenvalue
    // <= 0: the sweep's result IS the P of the full path (same M, same sweep) -- one sweep instead of a test sweep plus that one.
    // On configs[1] every call the shortcut takes is certified this way (||M^-1||_inf ~ 1.2 against 1 / reg = 1,000).
#pragma unroll
    for (int r = 0; r < RPT; r++) if (colok && hf + NH * r == jc) Br[r] += reg0;
    bool okm = spd_sweep_regs<RPT, true>(Br, tws, n);
    __syncthreads();
    m_pivot_failed = !okm;        // M = B + reg I is not positive definite => neither is B: the inertia sweep below would fail too
    if (okm) {        // block-uniform
      double cs = 0;
#pragma unroll
      for (int r = 0; r < RPT; r++) { const int i = hf + NH * r; if (colok && i < n) cs += fabs(Br[r]); }
      Z[hf * 128 + jc] = cs;           // (the eigenvector area is free: NH x 128 partial column sums of the symmetric inverse)
      __syncthreads();
      double tot = 0.0;
      if (TID < 128) {
#pragma unroll
        for (int h = 0; h < NH; h++) tot += Z[h * 128 + TID];
        if (TID >= n) tot = 0.0;
      }
      const double nrm = block_max(tot, red);
      okm = nrm < 0.999 / reg0;
    }
    if (okm) { pd_fast = true; have_P = true; }
    else {
#pragma unroll
      for (int r = 0; r < RPT; r++) {
        const int i = hf + NH * r;
        Br[r] = (colok && i < n) ? 0.5 * (Qg[(int64_t)i * n + jc] + Qg[(int64_t)jc * n + i]) : 0.0;
      }
    }
    PROF_COUNT(PH_T_COL, okm ? 1 : 0);
  }
  if (!have_P && !m_pivot_failed && scal[DG_PSD_PD] == 1.0) {       // block-uniform (only the NORM bound can have failed above: B may still be positive definite)
    pd_fast = spd_sweep_regs<RPT, true>(Br, tws, n);
    __syncthreads();
    if (!pd_fast) {
#pragma unroll
      for (int r = 0; r < RPT; r++) {
        const int i = hf + NH * r;
        Br[r] = (colok && i < n) ? 0.5 * (Qg[(int64_t)i * n + jc] + Qg[(int64_t)jc * n + i]) : 0.0;
      }
    }
    PROF_COUNT(PH_PD_TRY, pd_fast ? 1 : 0);
  }
  double tnorm = 0.0, pivmin = 0.0;
  int kneg = 0;
  if (!pd_fast) {
  // ---- 2. Householder tridiagonalisation, three barriers per step.  Every wavefront derives alpha / beta / K
  //         redundantly from the published column (no serial wave-0 sections); v is the published column masked to
  //         rows > k (with v_{k+1} = x_{k+1} - alpha), w is shared through wf at FULL row index (zero for rows <= k
  //         and the padding rows), so the per-thread loops are branch-free and their LDS reads are batched.
  lptr wf = tws;                        // NH*RPT + 4
  lptr cb0 = tws + (NH * RPT + 4);      // published column, double buffered
  lptr cb1 = tws + 2 * (NH * RPT + 4);
  for (int i = TID; i < NH * RPT + 4; i += NT) { wf[i] = 0.0; cb0[i] = 0.0; cb1[i] = 0.0; }
  __syncthreads();
  if (jc == 0) {
#pragma unroll
    for (int r = 0; r < RPT; r++) cb0[hf + NH * r] = Br[r];
  }
  __syncthreads();
  // Two barriers per step: every wavefront keeps its OWN copy of w (in the eigenvector area, unused until later), so
  // w never has to be published.  Rows that are already reduced (i <= k for every thread) are skipped in quarter-steps
  // of the register tile (R0), which removes about a third of the sweep work.
  lptr wfw = tws + 3 * (NH * RPT + 4) + wave * (NH * RPT + 4);
  lptr vfw = tws + (3 + NT / 64) * (NH * RPT + 4) + wave * (NH * RPT + 4);   // masked reflector: zero for rows <= k
  for (int i = lane; i < NH * RPT + 4; i += 64) { wfw[i] = 0.0; vfw[i] = 0.0; }
  auto hh_step = [&](int k, auto r0tag) {
    constexpr int R0 = decltype(r0tag)::value;
    const int m = n - k - 1;
    clptr cb = (k & 1) ? cb1 : cb0;
    lptr cbn = (k & 1) ? cb0 : cb1;
    // -- every wave: x = cb[k+1..n-1]; alpha = -sign(x0)|x|; v = x - alpha e1; beta = 2 / v^T v
    const double xa = lane < m ? cb[k + 1 + lane] : 0.0, xb = lane + 64 < m ? cb[k + 1 + lane + 64] : 0.0;
    const double x0 = lane_bcast(xa, 0);
    const double nrm2 = wave_sum(xa * xa + xb * xb);
    const double tail2 = nrm2 - x0 * x0;
    double alpha, beta;
    if (!(tail2 > 0.0)) { alpha = x0; beta = 0.0; }
    else {
      alpha = x0 >= 0 ? -sqrt(nrm2) : sqrt(nrm2);
      const double v0 = x0 - alpha;
      beta = 2.0 / (tail2 + v0 * v0);
    }
    const double dshift = beta != 0.0 ? alpha : 0.0;   // v_{k+1} = x_{k+1} - dshift
    {
      const double va = lane == 0 ? xa - dshift : xa;
      if (lane < m) vfw[k + 1 + lane] = va;
      if (lane + 64 < m) vfw[k + 1 + lane + 64] = xb;
      if (lane == 0) vfw[k] = 0.0;
      if (wave == 0) {
        if (big) {
          if (lane < m) RfG[RFOFF(k) + lane] = va;
          if (lane + 64 < m) RfG[RFOFF(k) + lane + 64] = xb;
        } else {
          if (lane < m) Rf[RFOFF(k) + lane] = va;
          if (lane + 64 < m) Rf[RFOFF(k) + lane + 64] = xb;
        }
        if (lane == 0) { dv[k] = cb[k]; ev[k] = alpha; tau[k] = beta; }
      }
    }
    if (beta != 0.0) {   // wave-uniform (all waves computed the same beta)
      // p_j = sum_i B[i][j] v_i over this thread's rows (B symmetric => column sums give the matvec)
      double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
      double vr[RPT];          // v on this thread's rows: read once, used by the product and by the rank-2 update below
#pragma unroll
      for (int r = R0; r < RPT; r++) vr[r] = vfw[hf + NH * r];
#pragma unroll
      for (int r = R0; r < RPT; r++) {
        const double vi = vr[r];
        if ((r & 3) == 0) s0 += Br[r] * vi; else if ((r & 3) == 1) s1 += Br[r] * vi; else if ((r & 3) == 2) s2 += Br[r] * vi; else s3 += Br[r] * vi;
      }
      if (colok) pp[hf * n + jc] = (s0 + s1) + (s2 + s3);
      __syncthreads();                                                            // barrier 1: partial sums
      // -- every wave: p, K = beta/2 p^T v, w = p - K v into its private copy (full row index, zero for rows <= k)
      double pa = 0.0, pb = 0.0;
#pragma unroll
      for (int h = 0; h < NH; h++) {
        if (lane < m) pa += pp[h * n + k + 1 + lane];
        if (lane + 64 < m) pb += pp[h * n + k + 1 + lane + 64];
      }
      pa *= beta; pb *= beta;
      const double va = lane == 0 ? xa - dshift : xa, vb = xb;
      const double K = 0.5 * beta * wave_sum(pa * va + pb * vb);
      if (lane < m) wfw[k + 1 + lane] = pa - K * va;
      if (lane + 64 < m) wfw[k + 1 + lane + 64] = pb - K * vb;
      if (lane == 0) wfw[k] = 0.0;
      {
        const double vj = colok ? vfw[jc] : 0.0, wj = colok ? wfw[jc] : 0.0;   // both zero for columns <= k
#pragma unroll
        for (int r = R0; r < RPT; r++) {
          const int i = hf + NH * r;
          Br[r] -= vr[r] * wj + wfw[i] * vj;
        }
      }
    }
    if (jc == k + 1) {   // owners of the next column publish their (updated) slice into the other buffer
#pragma unroll
      for (int r = 0; r < RPT; r++) cbn[hf + NH * r] = Br[r];
    }
    __syncthreads();                                                              // barrier 2: next column
  };
  {
    constexpr int Q1 = RPT / 4, Q2 = RPT / 2, Q3 = (3 * RPT) / 4;
    int k = 0;
    for (; k < n - 2 && k < NH * Q1 - 1; k++) hh_step(k, std::integral_constant<int, 0>());
    for (; k < n - 2 && k < NH * Q2 - 1; k++) hh_step(k, std::integral_constant<int, Q1>());
    for (; k < n - 2 && k < NH * Q3 - 1; k++) hh_step(k, std::integral_constant<int, Q2>());
    for (; k < n - 2; k++) hh_step(k, std::integral_constant<int, Q3>());
  }
#pragma unroll
  for (int r = 0; r < RPT; r++) {
    const int i = hf + NH * r;
    if (jc == n - 2 && i == n - 2) dv[n - 2] = Br[r];
    if (jc == n - 1 && i == n - 1) { dv[n - 1] = Br[r]; ev[n - 1] = 0.0; }
    if (jc == n - 2 && i == n - 1) ev[n - 2] = Br[r];
  }
  if (big) __threadfence_block();
  __syncthreads();
  PROF_END(PH_E_TRI, pt_t);
  // ---- 3. negative eigenvalues of T by Sturm counts.  pp <- e^2
  double tn = 0;
  for (int i = TID; i < n; i += NT) { pp[i] = ev[i] * ev[i]; tn = fmax(tn, fabs(dv[i]) + fabs(ev[i]) + (i > 0 ? fabs(ev[i - 1]) : 0.0)); }
  tnorm = block_max(tn, red);
  pivmin = fmax(1e-300, 1e-290 * tnorm * tnorm);   // e^2 / pivmin stays finite
  kneg = sturm_count_reg(lane < n ? dv[lane] : 0.0, lane + 64 < n ? dv[lane + 64] : 0.0, lane < n ? pp[lane] : 0.0,
                         lane + 64 < n ? pp[lane + 64] : 0.0, n, 0.0, pivmin);
#ifdef DG_PROF
  if (TID == 0) { atomicAdd(&dg_prof[2 * PH_E_KNEG], (unsigned long long)kneg); atomicAdd(&dg_prof[2 * PH_E_KNEG + 1], 1ULL); }
#endif
  }
  __syncthreads();
  if (TID == 0) scal[DG_PSD_PD] = kneg == 0 ? 1.0 : 0.0;
  // ---- 4a. M = B + reg I (this thread's slice, back into Br); the negative part is corrected batch by batch
  const double reg = reg0;
  if (!have_P) {
#pragma unroll
    for (int r = 0; r < RPT; r++) {
      const int i = hf + NH * r;
      double a = 0.0;
      if (colok && i < n) {
        a = 0.5 * (Qg[(int64_t)i * n + jc] + Qg[(int64_t)jc * n + i]);
        if (i == jc) a += reg;
      }
      Br[r] = a;
    }
  }
  for (int j0 = 0; j0 < kneg; j0 += PSD_KMAX) {
    const int kb = kneg - j0 < PSD_KMAX ? kneg - j0 : PSD_KMAX;
    __syncthreads();
    PROF_BEGIN(pe1);
    {
      // this wavefront's copy of the tridiagonal, one row (two for n > 64) per lane
      const double dA = lane < n ? dv[lane] : 0.0, dB = lane + 64 < n ? dv[lane + 64] : 0.0;
      const double eA = lane < n ? ev[lane] : 0.0, eB = lane + 64 < n ? ev[lane + 64] : 0.0;
      const double e2A = eA * eA, e2B = eB * eB;
      for (int jj = wave; jj < kb; jj += NT / 64) {
        const int j = j0 + jj;
        // eigenvalue j (ascending) lies in [lo, hi) with count(lo) <= j < count(hi): 64-way multisection
        double lo = -tnorm * 1.0000001 - 1e-300, hi = 0.0;
        PROF_BEGIN(pms);
        for (int it = 0; it < 10; it++) {   // 65^10 > 2^53: ten 64-way multisection steps always reach fp64 resolution
          const double wdt = hi - lo;
          const double sg = lo + wdt * (double)(lane + 1) * (1.0 / 65.0);
          const int cnt = sturm_count_reg(dA, dB, e2A, e2B, n, sg, pivmin);
          const unsigned long long above = __ballot(cnt > j);
          const int first = above ? __ffsll((long long)above) - 1 : 64;
          const double nlo = first == 0 ? lo : lane_bcast(sg, first - 1);
          const double nhi = first == 64 ? hi : lane_bcast(sg, first);
          lo = nlo; hi = nhi;
          if (hi - lo <= 4.5e-16 * fmax(fabs(lo), fabs(hi)) + 1e-300) break;
        }
        const double lam = 0.5 * (lo + hi);
        PROF_END(PH_E_VEC, pms);
        if (lane == 0) lamv[jj] = lam;
        // ---- 3b. eigenvector of T by twisted factorisation
        double zA, zB;
        twisted_eigvec(dA, dB, eA, eB, n, lam, pivmin, lane, lds + L.g_strip + wave * 3 * ((n + 1) & ~1), zA, zB);
        if (lane < n) Z[jj * n + lane] = zA;
        if (lane + 64 < n) Z[jj * n + lane + 64] = zB;
      }
    }
    __syncthreads();
    PROF_END(PH_E_BIS, pe1);
    PROF_BEGIN(pe3);
    // ---- 3c. modified Gram-Schmidt (wavefront 0), then back-transformation v = H_0 ... H_{n-3} z (one wavefront per vector)
    if (wave == 0) {
      for (int j = 0; j < kb; j++) {
        double za = lane < n ? Z[j * n + lane] : 0.0, zb = lane + 64 < n ? Z[j * n + lane + 64] : 0.0;
        for (int i = 0; i < j; i++) {
          const double ya = lane < n ? Z[i * n + lane] : 0.0, yb = lane + 64 < n ? Z[i * n + lane + 64] : 0.0;
          double dt = wave_sum(za * ya + zb * yb);
          za -= dt * ya; zb -= dt * yb;
        }
        double nr = wave_sum(za * za + zb * zb);
        nr = 1.0 / sqrt(nr);
        za *= nr; zb *= nr;
        if (lane < n) Z[j * n + lane] = za;
        if (lane + 64 < n) Z[j * n + lane + 64] = zb;
      }
    }
    __syncthreads();
    for (int j = wave; j < kb; j += NT / 64) {
      double za = lane < n ? Z[j * n + lane] : 0.0, zb = lane + 64 < n ? Z[j * n + lane + 64] : 0.0;
      for (int k = n - 3; k >= 0; k--) {
        const double beta = tau[k];
        if (beta == 0.0) continue;
        double va = 0.0, vb = 0.0;
        if (big) {
          if (lane > k && lane < n) va = RfG[RFOFF(k) + lane - k - 1];
          if (lane + 64 > k && lane + 64 < n) vb = RfG[RFOFF(k) + lane + 64 - k - 1];
        } else {
          if (lane > k && lane < n) va = Rf[RFOFF(k) + lane - k - 1];
          if (lane + 64 > k && lane + 64 < n) vb = Rf[RFOFF(k) + lane + 64 - k - 1];
        }
        double dt = wave_sum(va * za + vb * zb);
        dt *= beta;
        za -= dt * va; zb -= dt * vb;
      }
      if (lane < n) Z[j * n + lane] = za;
      if (lane + 64 < n) Z[j * n + lane + 64] = zb;
    }
    __syncthreads();
    PROF_END(PH_E_BACK, pe3);
    // ---- 4b. M += sum_j (floor - lam_j) v_j v_j^T   (== U diag(s') U^T of DGSQP.py:1290-1296 on the negative part)
#pragma unroll
    for (int r = 0; r < RPT; r++) {
      const int i = hf + NH * r;
      if (colok && i < n) {
        double a = Br[r];
        for (int j = 0; j < kb; j++) a += (D.eig_floor - lamv[j]) * Z[j * n + i] * Z[j * n + jc];
        Br[r] = a;
      }
    }
  }
  if (Qpd) {
#pragma unroll
    for (int r = 0; r < RPT; r++) {
      const int i = hf + NH * r;
      if (colok && i < n) Qpd[i * n + jc] = Br[r];
    }
    __threadfence_block();   // (read back by other wavefronts when the classical QP takes over)
  }
  __syncthreads();
  PROF_END(PH_JACOBI, pt_t);
  if (!want_inverse) return true;      // the classical QP works on M itself (written to Qpd above): no explicit inverse
  PROF_BEGIN(pt_s);
  // ---- 5. symmetric Gauss-Jordan sweep in registers: after all pivots the slice holds -M^-1 (the certified shortcut already has it)
  if (!have_P) spd_sweep_regs<RPT>(Br, tws, n);
  if (big) {
    gptr Pp = c.ws + D.ws_P;
#pragma unroll
    for (int r = 0; r < RPT; r++) {
      const int i = hf + NH * r;
      if (colok && i < n && i >= jc) Pp[i * (i + 1) / 2 + jc] = -Br[r];
    }
    __threadfence_block();
  } else {
    lds_d* Pp = lds + L.g_Bp;
#pragma unroll
    for (int r = 0; r < RPT; r++) {
      const int i = hf + NH * r;
      if (colok && i < n && i >= jc) Pp[i * (i + 1) / 2 + jc] = -Br[r];
    }
  }
  __syncthreads();
  PROF_END(PH_PFORM, pt_s);
  return true;
}

__device__ inline void dev_psd_inverse(const Ctx& c, gptr Qpd, bool want_inverse = true) {
  if (TID == 0) LP(dg_prob.L.scal)[DG_XVALID] = 0.0;   // the EIG scratch overwrites the trajectory
  const int n = dg_prob.n;
  bool ok;
  if (n <= 32) ok = dev_psd_inverse_tridiag<32 / DG_NH>(c, Qpd, want_inverse);
  else if (n <= 64) ok = dev_psd_inverse_tridiag<64 / DG_NH>(c, Qpd, want_inverse);
  else if (n <= 100) ok = dev_psd_inverse_tridiag<100 / DG_NH>(c, Qpd, want_inverse);
  else ok = dev_psd_inverse_tridiag<128 / DG_NH>(c, Qpd, want_inverse);
  (void)ok;  // n <= 128 is enforced by dgsqp_create
}

// out = scale * P t  (P packed symmetric in LDS).  Every row is split over NSEG threads, each with four independent
// accumulators so that the LDS reads of a segment are in flight together; the partial sums meet in LDS.
template <class PT>
__device__ inline void dev_p_mul_t(PT Pp, clptr t, lptr out, double scale) {
  const DgProb& D = dg_prob;
  lptr part = LP(D.L.p_part);  // NSEG x n partial sums
  const int n = D.n;
  constexpr int NSEG = NT / 128;
  const int i = TID & 127, sg = TID >> 7;
  __syncthreads();
  if (i < n) {
    const int len = (n + NSEG - 1) / NSEG;
    const int j0 = sg * len, j1 = (j0 + len < n) ? j0 + len : n;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    // columns j <= i come from row i of the packed lower triangle (contiguous), j > i from column i (stride j)
    const int split = (i + 1 < j1) ? ((i + 1 > j0) ? i + 1 : j0) : j1;
    const PT row = Pp + i * (i + 1) / 2;
    int j = j0;
    for (; j + 3 < split; j += 4) {
      a0 += row[j] * t[j]; a1 += row[j + 1] * t[j + 1]; a2 += row[j + 2] * t[j + 2]; a3 += row[j + 3] * t[j + 3];
    }
    for (; j < split; j++) a0 += row[j] * t[j];
    for (; j + 3 < j1; j += 4) {
      a0 += Pp[j * (j + 1) / 2 + i] * t[j];
      a1 += Pp[(j + 1) * (j + 2) / 2 + i] * t[j + 1];
      a2 += Pp[(j + 2) * (j + 3) / 2 + i] * t[j + 2];
      a3 += Pp[(j + 3) * (j + 4) / 2 + i] * t[j + 3];
    }
    for (; j < j1; j++) a0 += Pp[j * (j + 1) / 2 + i] * t[j];
    part[sg * n + i] = (a0 + a1) + (a2 + a3);
  }
  __syncthreads();
  if (TID < n) {
    double s = 0;
#pragma unroll
    for (int g = 0; g < NSEG; g++) s += part[g * n + TID];
    out[TID] = scale * s;
  }
  __syncthreads();
}

__device__ inline void dev_p_mul(const Ctx& c, clptr t, lptr out, double scale) {
  if (dg_prob.big) dev_p_mul_t<cgptr>(c.ws + dg_prob.ws_P, t, out, scale);
  else dev_p_mul_t<clptr>(LP(dg_prob.L.g_Bp), t, out, scale);
}

// coefficient of constraint row r at column col
template <class GP>
__device__ inline double g_row_coef(const DgProb& D, GP gd, int r, int col) {
  const DgRow R = ld_row(r);
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
      const DgDense dd = ld_dense(R.dense);
      if (t >= dd.k) return 0.0;
      if (a == dd.a) return R.sgn * gd[dd.off + t * DGSQP_NUA + j];
      if (dd.kind == 1 && a == dd.b) return R.sgn * gd[dd.off + 2 * dd.k + t * DGSQP_NUA + j];
      return 0.0;
    }
  }
}

// wavefront-0 helpers of the dual active-set QP (dgsqp_qp.h).  The Schur complement S = A_W P A_W^T of the active rows is held
// through the INVERSE of its Cholesky factor, T = R^-1 (S = R^T R; upper triangular), so that  w = R^-T c = T^T c  and
// r = R^-1 w = T w  are two lane-parallel products (column sums, row sums: m independent multiply-adds per lane) instead of two
// triangular solves of m dependent steps each -- the active-set loop of a thrashing scenario runs tens of thousands of them.
// Adding a row borders R by (w; rho): T gets the column (-r / rho; 1 / rho), r = T w -- both at hand.  Removing row jd: the Givens
// rotations that restore the triangle of R after its column jd is deleted act on the COLUMNS of T; they are the rotations that sweep
// row jd of T into the last column, so their angles follow from that row alone (running norms), every row of T is then transformed
// independently (lane = row), and T' is what remains without row jd and the last column.
// Storage: columns padded with zeros to a multiple of eight rows (dg_tcol, dgsqp_layout.h) -- a group of eight rows of a column, or
// of eight columns at a row, is then either entirely inside (entries or zeros) or entirely outside the lane's part of the triangle:
// one execution mask per group, no masks or address clamps per element, neighbouring entries read in pairs.
typedef __attribute__((address_space(3))) int lds_i_t;
typedef __attribute__((address_space(3))) unsigned char lds_b_t;
#define QPT_U 8
// Writes w = T^T c to wv, r = T w to rv (lane j holds entries j and j + 64) and returns |w|^2 to every lane.  cvec / wv must be
// readable (finite values or zeros) up to the next multiple of eight beyond m: both are zero-filled here.
__device__ inline double qpt_solve(clptr T, int m_, int lane, lptr cvec, lptr wv, lptr rv, double& r0_out, double& r1_out) {
  const int m = __builtin_amdgcn_readfirstlane(m_);       // (uniform by construction; says so to the compiler: scalar loops and branches)
  const bool two = m > 64;
  const int m1 = two ? 64 : m;                 // rows / columns of the first half
  const int m8 = (m + 7) & ~7;
  const int la = lane < m ? lane : 0, lb = lane + 64 < m ? lane + 64 : 64;
  const int colA = dg_tcol(la), colB = dg_tcol(lb);
  if (lane >= m && lane < m8) cvec[lane] = 0.0;
  if (lane + 64 >= m && lane + 64 < m8) cvec[lane + 64] = 0.0;
  double w0, w1;
  {
    double a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
    // rows i < m1 in groups of eight: column `lane` takes part while i <= lane (its zeros cover the rest of its last group),
    // column lane + 64 in every group
    for (int i = 0; i < m1; i += QPT_U) {
      double cc[QPT_U];
#pragma unroll
      for (int k = 0; k < QPT_U; k++) cc[k] = cvec[i + k];
      if (la >= i) {
        double ta[QPT_U];
#pragma unroll
        for (int k = 0; k < QPT_U; k++) ta[k] = T[colA + i + k];
#pragma unroll
        for (int k = 0; k < QPT_U; k++) a[k & 3] = __builtin_fma(ta[k], cc[k], a[k & 3]);
      }
      if (two) {
        double tb[QPT_U];
#pragma unroll
        for (int k = 0; k < QPT_U; k++) tb[k] = T[colB + i + k];
#pragma unroll
        for (int k = 0; k < QPT_U; k++) b[k & 3] = __builtin_fma(tb[k], cc[k], b[k & 3]);
      }
    }
    // rows i >= 64: only columns lane + 64 >= i
    if (two)
      for (int i = 64; i < m; i += QPT_U) {
        if (lb >= i) {
          double tb[QPT_U], cc[QPT_U];
#pragma unroll
          for (int k = 0; k < QPT_U; k++) { cc[k] = cvec[i + k]; tb[k] = T[colB + i + k]; }
#pragma unroll
          for (int k = 0; k < QPT_U; k++) b[k & 3] = __builtin_fma(tb[k], cc[k], b[k & 3]);
        }
      }
    w0 = lane < m ? (a[0] + a[1]) + (a[2] + a[3]) : 0.0;
    w1 = lane + 64 < m ? (b[0] + b[1]) + (b[2] + b[3]) : 0.0;
  }
  if (lane < m8) wv[lane] = w0;
  if (lane + 64 < m8) wv[lane + 64] = w1;
  const double ww = wave_sum(w0 * w0 + w1 * w1);
  double r0, r1;
  {
    double a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
    // columns j in groups of eight (beyond m: w = 0 against zeros or entries of T's slot -- every column < m8 <= 8 ceil(n / 8) is inside
    // it): row `lane` takes part while lane < j + 8 (the columns' zeros cover the rows below their diagonal inside the group)
    for (int j = 0; j < m; j += QPT_U) {
      const int base = dg_tcol(j), stride = 8 * ((j >> 3) + 1) + 1;
      const int last = m - 1 - j;      // columns beyond m - 1 (last group only) hold whatever was there: read column m - 1 instead, w = 0 there
      double wj[QPT_U];
#pragma unroll
      for (int k = 0; k < QPT_U; k++) wj[k] = wv[j + k];
      if (la < j + QPT_U) {
        double ta[QPT_U];
#pragma unroll
        for (int k = 0; k < QPT_U; k++) ta[k] = T[base + (k < last ? k : last) * stride + la];
#pragma unroll
        for (int k = 0; k < QPT_U; k++) a[k & 3] = __builtin_fma(ta[k], wj[k], a[k & 3]);
      }
      if (two && j >= 64 && lb < j + QPT_U) {
        double tb[QPT_U];
#pragma unroll
        for (int k = 0; k < QPT_U; k++) tb[k] = T[base + (k < last ? k : last) * stride + lb];
#pragma unroll
        for (int k = 0; k < QPT_U; k++) b[k & 3] = __builtin_fma(tb[k], wj[k], b[k & 3]);
      }
    }
    r0 = lane < m ? (a[0] + a[1]) + (a[2] + a[3]) : 0.0;
    r1 = lane + 64 < m ? (b[0] + b[1]) + (b[2] + b[3]) : 0.0;
  }
  if (lane < m) rv[lane] = r0;
  if (lane + 64 < m) rv[lane + 64] = r1;
  r0_out = r0; r1_out = r1;
  return ww;
}
// column m of T (m rows before): the entries v_i (lane i and i + 64 hold v0, v1), the diagonal d, zeros down to the end of the group
__device__ inline void qpt_put_column(lptr T, int m, int lane, double v0, double v1, double d) {
  const int base = dg_tcol(m), pend = 8 * ((m >> 3) + 1);
  if (lane < pend) T[base + lane] = lane < m ? v0 : (lane == m ? d : 0.0);
  if (lane + 64 < pend) T[base + lane + 64] = lane + 64 < m ? v1 : (lane + 64 == m ? d : 0.0);
}
// inclusive prefix sum over the 64 lanes of a wavefront
__device__ inline double wave_prefix_sum(double v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const double t = __shfl_up(v, d); if (lane >= d) v += t; }
  return v;
}
// remove active constraint jd (m rows before): bookkeeping arrays shift down, T is rotated (see above).  cs / sn: two scratch vectors
// of m doubles in LDS.
__device__ inline void qpt_drop(lptr T, lds_i_t* alist, lds_i_t* yslot, lptr lam, lds_b_t* act, int m_, int jd_, int lane, lptr cs, lptr sn) {
  const int m = __builtin_amdgcn_readfirstlane(m_), jd = __builtin_amdgcn_readfirstlane(jd_);
  const int mn = m - 1;
  const int ca = lane, cb = lane + 64;
  const bool sa = ca >= jd && ca < mn, sb = cb >= jd && cb < mn;
  if (lane == 0) act[alist[jd]] = 0;
  const int ala = sa ? alist[ca + 1] : 0, alb = sb ? alist[cb + 1] : 0;
  const int ysa = sa ? yslot[ca + 1] : 0, ysb = sb ? yslot[cb + 1] : 0;
  const double lma = sa ? lam[ca + 1] : 0.0, lmb = sb ? lam[cb + 1] : 0.0;
  // rotation k (jd <= k < mn) mixes the running column (row-jd entry a_k) with old column k + 1 (row-jd entry e_{k+1}):
  // a_jd = e_jd, a_{k+1} = sqrt(a_k^2 + e_{k+1}^2) = sqrt(sum_{i=jd..k+1} e_i^2);  c_k = e_{k+1} / a_{k+1},  s_k = a_k / a_{k+1}
  const double ea = (ca >= jd && ca < m) ? T[dg_tcol(ca) + jd] : 0.0, eb = (cb >= jd && cb < m) ? T[dg_tcol(cb) + jd] : 0.0;
  const double pa = wave_prefix_sum(ea * ea, lane);
  const double pb = wave_prefix_sum(eb * eb, lane) + lane_bcast(pa, 63);
  // a_k for k = lane / lane + 64 (the first one keeps its sign)
  const double aa = ca == jd ? ea : sqrt(pa), ab = cb == jd ? eb : sqrt(pb);
  // rotation k needs e_{k+1} and a_{k+1}: lane k + 1 publishes c_k, s_k needs a_k from lane k -- go through LDS
  if (ca > jd && ca < m) cs[ca - 1] = ea / aa;          // c_{ca-1} = e_ca / a_ca
  if (cb > jd && cb < m) cs[cb - 1] = eb / ab;
  if (ca >= jd && ca < mn) sn[ca] = aa;                  // a_k, divided by a_{k+1} below
  if (cb >= jd && cb < mn) sn[cb] = ab;
  if (ca > jd && ca < m) sn[ca - 1] = sn[ca - 1] / aa;   // (same lane order within the wavefront: the store above is visible)
  if (cb > jd && cb < m) sn[cb - 1] = sn[cb - 1] / ab;
  // rows: lane handles old rows ca and cb (row jd itself disappears)
  const bool ra = ca < m && ca != jd, rb = cb < m && cb != jd;
  const int na = ca < jd ? ca : ca - 1, nb = cb < jd ? cb : cb - 1;      // their new indices
  const int cjd = dg_tcol(jd);
  double runa = (ra && ca <= jd) ? T[cjd + ca] : 0.0, runb = (rb && cb <= jd) ? T[cjd + cb] : 0.0;
  for (int k = jd; k < mn; k++) {
    const double c_ = cs[k], s_ = sn[k];
    const int base = dg_tcol(k + 1), dst = dg_tcol(k);
    const bool ia = ra && ca <= k + 1, ib = rb && cb <= k + 1;
    const double ba = ia ? T[base + ca] : 0.0, bb = ib ? T[base + cb] : 0.0;
    // new column k = s col_{k+1} - c run  (the sign that keeps the diagonal positive);  run <- s run + c col_{k+1}
    if (ia) T[dst + na] = s_ * ba - c_ * runa;
    if (ib) T[dst + nb] = s_ * bb - c_ * runb;
    runa = s_ * runa + c_ * ba;
    runb = s_ * runb + c_ * bb;
  }
  // (the zeros below the diagonals stay: new column k has k + 1 entries like the old one)
  if (sa) { alist[ca] = ala; lam[ca] = lma; yslot[ca] = ysa; }
  if (sb) { alist[cb] = alb; lam[cb] = lmb; yslot[cb] = ysb; }
}

#include "dgsqp_qp.h"
#include "dgsqp_osqp.h"
__device__ void dev_xl_psd(const Ctx& c, gptr Qpd);   // XL layout (n > 128), dgsqp_xl.h
__device__ int dev_xl_qp(const Ctx& c);
__device__ int dev_qp_osqp_xl(const Ctx& c);          // ... with OSQP's arithmetic, dgsqp_osqp_xl.h

// ------------------------------------------------------------------------------------------------
// dual initialisation  l = max(0, -lsqr(G G^T, G q))   (DGSQP.py:320-327).
// LSQR restated from scipy 1.15.3 scipy/sparse/linalg/_isolve/lsqr.py (Paige & Saunders 1982) with
// its defaults damp=0, atol=btol=1e-6, conlim=1e8, iter_lim=2*n_c; the operator G G^T is applied as
// G (G^T v) through the packed constraint gradients instead of being assembled.
// ------------------------------------------------------------------------------------------------
__device__ inline void dev_ggt_mul(const Ctx& c, clptr vin, lptr vout, lptr tmpn) {
  const DgProb& D = dg_prob;
  gt_mul(c, vin, tmpn);
  lptr dd2 = LP(D.L.s_yd2);     // QP scratch, idle during the dual start (its own copy in the tab_const layout)
  if (D.gd_global) qp_dense_dots<cgptr>(D, dev_gd_global(c), tmpn, LP(D.L.s_dpart), dd2);
  else qp_dense_dots<clptr>(D, LP(D.L.gd), tmpn, LP(D.L.s_dpart), dd2);
  for (int r = TID; r < D.nc; r += NT) vout[r] = qpw_row_dot(D, ld_row(r), tmpn, dd2);
  __syncthreads();
}
__device__ inline void dev_sym_ortho(double a, double b, double& cs, double& sn, double& r) {
  if (b == 0) { cs = (a > 0) - (a < 0); sn = 0; r = fabs(a); }
  else if (a == 0) { cs = 0; sn = (b > 0) - (b < 0); r = fabs(b); }
  else if (fabs(b) > fabs(a)) { const double tau = a / b; sn = ((b > 0) - (b < 0)) / sqrt(1 + tau * tau); cs = sn * tau; r = b / sn; }
  else { const double tau = b / a; cs = ((a > 0) - (a < 0)) / sqrt(1 + tau * tau); sn = cs * tau; r = a / cs; }
}
__device__ __noinline__ void dev_dual_init(const Ctx& c) {
  if (TID == 0 && !dg_prob.lsqr_keeps_eval) LP(dg_prob.L.scal)[DG_XVALID] = 0.0;   // the LSQR vectors overwrite the trajectory
  const DgProb& D = dg_prob;
  const DgLds& L = D.L;
  lptr lds = LP(0);
  const int nc = D.nc;
  lds_d *u = lds + L.s_u, *v = lds + L.s_v, *w = lds + L.s_w, *x = lds + L.s_x, *tmp = lds + L.s_t;
  lds_d* tn = lds + L.d;  // n-vector scratch (d is recomputed afterwards)
  lds_d* red = lds + L.red;
  lds_d* l = lds + L.l;
  PROF_BEGIN(pt_l);
  const double eps = 2.220446049250313e-16;
  const double atol = D.par.lsqr_atol, btol = D.par.lsqr_btol, ctol = 1e-8;
  const int iter_lim = D.par.lsqr_iter_lim > 0 ? D.par.lsqr_iter_lim : 2 * nc;
  // b = G q
  __syncthreads();
  double p = 0;
  for (int r = TID; r < nc; r += NT) { const double b = g_row_dot_any(c, r, lds + L.q); u[r] = b; x[r] = 0.0; p += b * b; }
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
#define DG_V2_OBJ 48       // scal slot, DG-SQP v2 merit 'sum_obj_l1': sum of the agents' costs at the current linearisation point
struct LinScal {
  double phi, dphi;   // merit and its directional derivative at the base point (with the caller's mu)
  double S0, S1;      // sum(s), sum(ds) with s = min(0,g), ds = g + G du - s   (DGSQP.py:414-415)
  double dstat, vio;  // f_dstat_norm and sum(g - s)
};

// d = q + G^T l  (also the stationarity vector of the convergence test, DGSQP.py:368)
__device__ inline void dev_stat_vector(const Ctx& c, clptr lvec, lptr dout) {
  const DgProb& D = dg_prob;
  gt_mul(c, lvec, dout);
  for (int i = TID; i < D.n; i += NT) dout[i] += LP(0)[D.L.q + i];
  __syncthreads();
}
// v = Qraw^T d
__device__ inline void dev_qt_mul(const Ctx& c) {
  const DgProb& D = dg_prob;
  cgptr Qg = c.ws + D.ws_q;
  const lds_d* d = LP(0) + D.L.d;
  __syncthreads();
  PROF_BEGIN(pt_q);
  for (int j = TID; j < D.n; j += NT) {
    double s = 0;
    for (int i = 0; i < D.n; i++) s += Qg[(int64_t)i * D.n + j] * d[i];
    LP(0)[D.L.v + j] = s;
  }
  __syncthreads();
  PROF_END(PH_QTMUL, pt_q);
}
// after a QP solve at the current linearisation: everything phi / dphi / mu need
__device__ __noinline__ void dev_step_scalars(const Ctx& c, LinScal& S) {
  const DgProb& D = dg_prob;
  const DgLds& L = D.L;
  lptr lds = LP(0);
  const lds_d *q = lds + L.q, *g = lds + L.g, *l = lds + L.l, *d = lds + L.d, *v = lds + L.v;
  const lds_d *du = lds + L.o_du, *lhat = lds + L.o_lhat;
  lds_d* t = lds + L.p_t;  // G^T lhat
  lds_d* red = lds + L.red;
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
    ssum += gr + g_row_dot_any(c, r, du);  // s + ds
  }
  a1 = block_sum(a1, red); a2 = block_sum(a2, red); lGdu = block_sum(lGdu, red); dd = block_sum(dd, red);
  lg = block_sum(lg, red); lhg = block_sum(lhg, red); vio = block_sum(vio, red); s0 = block_sum(s0, red); ssum = block_sum(ssum, red);
  S.dstat = a1 + a2 + lg * (lGdu + (lhg - lg));
  if (D.par.variant == DGSQP_VARIANT_V2) S.dstat = a1 + a2;     // d/d(u,l) of 1/2 |d|^2 along (du, dl)  (DGSQP_v2.py:1143-1145)
  const bool sum_obj = D.par.variant == DGSQP_VARIANT_V2 && D.par.merit_function == DGSQP_MERIT_SUM_OBJ_L1;
  if (sum_obj) S.dstat = a1;                                    // L.v holds Du(sum_a J^a) here: dobj = v . du  (DGSQP_v2.py:1152)
  S.vio = vio;
  S.S0 = s0;
  S.S1 = ssum - s0;
  S.phi = 0.5 * (dd + lg * lg);  // + mu*vio added by the caller
  if (D.par.variant == DGSQP_VARIANT_V2) S.phi = 0.5 * dd;
  if (sum_obj) S.phi = lds[L.scal + DG_V2_OBJ];
  S.dphi = S.dstat;
  PROF_END(PH_MERIT, pt_m);
}
// DG-SQP v2's merit (DGSQP_v2.py:1141-1160, 'stat_l1'):  1/2 |d|^2 + mu sum(max(0, g))  -- no complementarity term, slacks
// s = max(0, g) of the trial point itself.  |d|^2 and the violation are left in two scalar slots for the caller, which also
// needs the value with mu = 1 (DGSQP_v2.py:754).
#define DG_V2_DD 50
#define DG_V2_VIO 51
// DG-SQP v2, merit 'sum_obj_l1' (DGSQP_v2.py:1151-1152, 1161-1164): obj = sum_a J^a along the trajectory in the EVAL scratch (inputs
// e_ue).  With `grad` the gradient of obj w.r.t. ALL inputs goes to L.v -- the slot of Q^T d, which this merit does not use --: stage
// gradients of the summed cost, one costate sweep per agent block (the joint dynamics are block diagonal), B_t^T lam_{t+1} + the
// direct input-cost terms.  Needs x, A_k, B_k of the point in the EVAL scratch, i.e. runs before the QP.
__device__ __noinline__ double dev_v2_sum_obj(const Ctx& c, bool grad) {
  const DgProb& D = dg_prob;
  const DgLds& L = D.L;
  lptr lds = LP(0);
  const int nq = D.nq, N = D.N, M = D.M, n = D.n;
  clptr x = lds + L.e_x, ue = lds + L.e_ue;
  lptr Dxs = lds + L.e_Dxs;       // [k][nq]: d/dx_k of the summed stage cost (first agent slice of the costate scratch)
  lptr lam = lds + L.e_lam;       // [k][nq]: costates of the summed cost
  __syncthreads();
  double part = 0;
  for (int k = TID; k <= N; k += NT) {
    double Dx[DGSQP_MAX_AGENTS * DGSQP_MAX_NQA];
    for (int i = 0; i < nq; i++) Dx[i] = 0.0;
    for (int a = 0; a < M; a++) part += dev_state_cost(D, a, x + k * nq, k == N, grad ? Dx : (double*)nullptr, (double*)nullptr);
    if (grad) for (int i = 0; i < nq; i++) Dxs[k * nq + i] = Dx[i];
  }
  for (int i = TID; i < n; i += NT) {
    const int a = i / (N * DGSQP_NUA), rem = i % (N * DGSQP_NUA), t = rem / DGSQP_NUA, j = rem % DGSQP_NUA;
    const dgsqp_agent_t& ag = D.P.agents[a];
    const double uk = ue[i], um = t > 0 ? ue[i - DGSQP_NUA] : 0.0;
    part += 0.5 * ag.w_in[j] * uk * uk + 0.5 * ag.w_rate[j] * (uk - um) * (uk - um);
  }
  const double obj = block_sum(part, lds + L.red);
  if (!grad) return obj;
  __syncthreads();
  if (TID < M) {
    const int a = TID, nqa = D.nqa[a], qo = D.qoff[a];
    double lk[DGSQP_MAX_NQA], nx[DGSQP_MAX_NQA];
    for (int m = 0; m < nqa; m++) lk[m] = Dxs[N * nq + qo + m];
    for (int k = N - 1; k >= 0; k--) {
      for (int m = 0; m < nqa; m++) lam[(k + 1) * nq + qo + m] = lk[m];
      clptr A = lds + L.e_A[a] + k * nqa * nqa;
      for (int z = 0; z < nqa; z++) { double sacc = Dxs[k * nq + qo + z]; for (int o = 0; o < nqa; o++) sacc += A[o * nqa + z] * lk[o]; nx[z] = sacc; }
      for (int m = 0; m < nqa; m++) lk[m] = nx[m];
    }
  }
  __syncthreads();
  for (int i = TID; i < n; i += NT) {
    const int a = i / (N * DGSQP_NUA), rem = i % (N * DGSQP_NUA), t = rem / DGSQP_NUA, j = rem % DGSQP_NUA;
    const dgsqp_agent_t& ag = D.P.agents[a];
    const double uk = ue[i], um = t > 0 ? ue[i - DGSQP_NUA] : 0.0;
    double sacc = ag.w_in[j] * uk + ag.w_rate[j] * (uk - um);
    if (t + 1 < N) sacc -= ag.w_rate[j] * (ue[i + DGSQP_NUA] - uk);
    const int nqa = D.nqa[a];
    clptr B = lds + L.e_B[a] + t * nqa * DGSQP_NUA;
    clptr lkp = lam + (t + 1) * nq + D.qoff[a];
    for (int m = 0; m < nqa; m++) sacc += B[m * DGSQP_NUA + j] * lkp[m];
    lds[L.v + i] = sacc;
  }
  if (TID == 0) lds[L.scal + DG_V2_OBJ] = obj;
  __syncthreads();
  return obj;
}
__device__ inline double dev_v2_trial_phi(const Ctx& c, double dd, double mu) {
  const DgProb& D = dg_prob;
  lptr lds = LP(0);
  if (D.par.merit_function == DGSQP_MERIT_SUM_OBJ_L1) dd = 2.0 * dev_v2_sum_obj(c, false);      // (callers keep 1/2 dd: the objective part of the merit)
  double v = 0;
  for (int r = TID; r < D.nc; r += NT) v += fmax(lds[D.L.g + r], 0.0);
  v = block_sum(v, lds + D.L.red);
  if (TID == 0) { lds[D.L.scal + DG_V2_DD] = dd; lds[D.L.scal + DG_V2_VIO] = v; }
  __syncthreads();
  return 0.5 * dd + mu * v;
}
// merit of a trial point through q and the packed G (fallback when the packed-G area is too small to hold the trial
// multipliers: games with hardly any state / obstacle rows): current (q, g, G) in LDS belong to the trial u
__device__ __noinline__ double dev_phi_trial_dense(const Ctx& c, double alpha, double sum_s, double mu) {
  const DgProb& D = dg_prob;
  const DgLds& L = D.L;
  lptr lds = LP(0);
  lds_d* lt = lds + L.s_x;
  lds_d* dt = lds + L.s_t;
  lds_d* red = lds + L.red;
  const lds_d *l = lds + L.l, *lhat = lds + L.o_lhat, *g = lds + L.g;
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
  if (D.par.variant == DGSQP_VARIANT_V2) phi = dev_v2_trial_phi(c, dd, mu);
  PROF_END(PH_MERIT, pt_m);
  return phi;
}

// Merit of a trial point u + alpha du with multipliers l + alpha (lhat - l):  1/2 |d|^2 + 1/2 (l'g)^2 + mu sum(g - s),
// d = q + G' l = stacked gradients of the agents' Lagrangians w.r.t. their own inputs.  d comes from one costate sweep
// per agent at the trial trajectory (x, A_k, B_k and g are current; q and the packed G are not needed):
//   d_(a,t,j) = dJ^a/du direct + box / rate multipliers + B_t^T lam^a_{t+1}
// The trial multipliers and d live in the packed-G area, which is dead until the next full linearisation.
__device__ __noinline__ double dev_phi_trial_adjoint(const Ctx& c, double alpha, double sum_s, double mu) {
  const DgProb& D = dg_prob;
  const DgLds& L = D.L;
  lptr lds = LP(0);
  lds_d* lt = lds + L.gd;
  lds_d* red = lds + L.red;
  const lds_d *l = lds + L.l, *lhat = lds + L.o_lhat, *g = lds + L.g;
  clptr ue = lds + L.e_ue;
  __syncthreads();
  PROF_BEGIN(pt_m);
  double lg = 0, sg = 0;
  for (int r = TID; r < D.nc; r += NT) {
    const double v = l[r] + alpha * (lhat[r] - l[r]);
    lt[r] = v;
    lg += v * g[r];
    sg += g[r];
  }
  __syncthreads();
  dev_costates(c, lt);
  clptr lam = lds + L.e_lam;
  double dd = 0;
  for (int i = TID; i < D.n; i += NT) {
    const int a = i / (D.N * DGSQP_NUA), rem = i % (D.N * DGSQP_NUA), t = rem / DGSQP_NUA, j = rem % DGSQP_NUA;
    const dgsqp_agent_t& ag = D.P.agents[a];
    const double uk = ue[i], um = t > 0 ? ue[i - DGSQP_NUA] : 0.0;
    double s = ag.w_in[j] * uk + ag.w_rate[j] * (uk - um);
    if (t + 1 < D.N) s -= ag.w_rate[j] * (ue[i + DGSQP_NUA] - uk);
    int r;
    if ((r = D.r_in_ub[a][t][j]) >= 0) s += lt[r];
    if ((r = D.r_in_lb[a][t][j]) >= 0) s -= lt[r];
    if ((r = D.r_rate_ub[a][t][j]) >= 0) s += lt[r];
    if ((r = D.r_rate_lb[a][t][j]) >= 0) s -= lt[r];
    if (t + 1 < D.N) {
      if ((r = D.r_rate_ub[a][t + 1][j]) >= 0) s -= lt[r];
      if ((r = D.r_rate_lb[a][t + 1][j]) >= 0) s += lt[r];
    }
    const int nqa = D.nqa[a];
    clptr B = lds + L.e_B[a] + t * nqa * DGSQP_NUA;
    clptr lk = lam + (a * (D.N + 1) + t + 1) * D.nq + D.qoff[a];
    for (int m = 0; m < nqa; m++) s += B[m * DGSQP_NUA + j] * lk[m];
    dd += s * s;
  }
  dd = block_sum(dd, red); lg = block_sum(lg, red); sg = block_sum(sg, red);
  double phi = 0.5 * (dd + lg * lg);
  if (D.par.merit_function == DGSQP_MERIT_STAT_L1) phi += mu * (sg - sum_s);
  if (D.par.variant == DGSQP_VARIANT_V2) phi = dev_v2_trial_phi(c, dd, mu);
  PROF_END(PH_MERIT, pt_m);
  return phi;
}

// derivatives + merit of the trial point prepared by dev_evaluate_point
__device__ inline double dev_trial_merit(const Ctx& c, double alpha, double sum_s, double mu) {
  if (dg_prob.ngd >= dg_prob.nc) {
    dev_evaluate_trial_derivs(c);
    return dev_phi_trial_adjoint(c, alpha, sum_s, mu);
  }
  dev_evaluate_derivs(c, false);
  return dev_phi_trial_dense(c, alpha, sum_s, mu);
}

// ---- cooperative line search (dgsqp_device.h: DgCoop) ----------------------------------------------------------------------
#define DG_COOP_FLIP 57      // scal slot: which of the workgroup's two job slots its next line search uses
#define DG_BCAST 58          // scal slot: one 64-bit word handed from thread 0 to the workgroup
#define AT_LOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
// one value read by thread 0 and handed to the whole workgroup (control flow must stay uniform)
__device__ inline unsigned long long dev_bcast_u64(unsigned long long v) {
  lptr sc = LP(dg_prob.L.scal);
  __syncthreads();
  if (TID == 0) ((__attribute__((address_space(3))) unsigned long long*)sc)[DG_BCAST] = v;
  __syncthreads();
  return ((__attribute__((address_space(3))) unsigned long long*)sc)[DG_BCAST];
}
// The derivative-free part of the merit exceeds the Armijo bound: the trial is rejected from its constraint values alone.
__device__ inline bool dev_trial_pruned(const Ctx& c, double alpha, double mu, double phi, double dphi, double S0, double S1) {
  const DgProb& D = dg_prob;
  const DgLds& L = D.L;
  lptr lds = LP(0);
  double lg = 0, sg = 0;
  for (int r = TID; r < D.nc; r += NT) {
    const double gr = lds[L.g + r];
    lg += (lds[L.l + r] + alpha * (lds[L.o_lhat + r] - lds[L.l + r])) * gr;
    sg += gr;
  }
  lg = block_sum(lg, lds + L.red); sg = block_sum(sg, lds + L.red);
  double lb = 0.5 * lg * lg;
  if (D.par.merit_function == DGSQP_MERIT_STAT_L1) lb += mu * (sg - (S0 + alpha * S1));
  const double bound = phi + D.par.beta * alpha * dphi;
  PROF_COUNT(PH_C_TRIALS, lb > bound + 1e-9 * (fabs(bound) + fabs(lb)) ? 1 : 0);
  return lb > bound + 1e-9 * (fabs(bound) + fabs(lb));
}
// Helper side: evaluate trial j of `job` exactly as its owner would (own rollout of that one trajectory).  The base point is in
// this workgroup's LDS (loaded by dev_coop_help).
__device__ __noinline__ void dev_coop_trial(const Ctx& c, DgCoopJob* job, int j) {
  const DgProb& D = dg_prob;
  const DgLds& L = D.L;
  lptr lds = LP(0);
  const int xsz = ((D.N + 1) * D.nq + 1) & ~1;
  double alpha = 1.0;
  for (int t = 0; t < j; t++) alpha *= D.par.tau;            // the owner's alpha *= tau, j times
  const double mu = AT_LOAD(&job->mu), phi = AT_LOAD(&job->phi), dphi = AT_LOAD(&job->dphi), S0 = AT_LOAD(&job->S0), S1 = AT_LOAD(&job->S1);
  const int iters = __hip_atomic_load(&job->iters, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  dev_rollout_multi(c, lds + L.u, lds + L.o_du, alpha, D.par.tau, 1, lds + L.e_xs, xsz, D.ls_spec1, lds + L.e_xs2);
  dev_evaluate_point(c, lds + L.u, alpha, lds + L.o_du, lds + L.e_xs);
  bool pruned = false;
  if (j + 1 < iters) pruned = dev_trial_pruned(c, alpha, mu, phi, dphi, S0, S1);
  double phit = 0.0;
  if (!pruned) phit = dev_trial_merit(c, alpha, S0 + alpha * S1, mu);
  __syncthreads();
  if (TID == 0) {
    job->phi_out[j] = phit;
    if (pruned) __hip_atomic_fetch_or(&job->pruned, 1ull << j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    __hip_atomic_fetch_or(&job->ready, 1ull << j, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&c.coop->helped, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
}
// A workgroup that found the ticket queue empty: help the line searches of the workgroups still solving until every scenario of
// the launch is done.
// Returns 1 when a deferred scenario is waiting to be resumed (the caller takes it), 0 when the launch is over.
__device__ __noinline__ int dev_coop_help(Ctx& c) {
  const DgProb& D = dg_prob;
  const DgLds& L = D.L;
  lptr lds = LP(0);
  DgCoop* co = c.coop;
  const int njobs = 2 * (int)gridDim.x;
  const int NONE = 0x7fffffff;
  // The first `coop_helpers` workgroups to run out of scenarios help; the others only wait for the launch to end: a failing line search
  // has at most 48 trials on offer, and 250 workgroups evaluating trials nobody will need keep the whole chip under load -- its clock
  // sags and the scenarios still solving, serial chains all of them, run slower than in a plain launch's quiet tail.
  // `active_helpers` counts the workgroups evaluating trials right now: a helper that leaves to resume a deferred scenario gives its place
  // back, one that returns takes a place only if there is one (`idle` counts every workgroup in here, active or asleep)
  const unsigned int rank = (unsigned int)dev_bcast_u64(TID == 0 ? (unsigned long long)__hip_atomic_fetch_add(&co->active_helpers, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull);
  const bool passive = (int)rank >= c.coop_helpers;
  if (TID == 0) {
    if (passive) __hip_atomic_fetch_sub(&co->active_helpers, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&co->idle, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&co->helper_regs, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  int start = (2 * (int)blockIdx.x + 2) % njobs;
  unsigned long long spins = 0;
  int ret = 0;
  while (true) {
    if (passive) {
      const unsigned long long w = dev_bcast_u64(TID == 0 ? ((unsigned long long)(AT_LOAD(&co->finished) >= c.coop_total) << 32) | (unsigned long long)AT_LOAD(&co->park_avail) : 0ull);
      if (w & 0xffffffffull) { ret = 1; break; }
      if (w >> 32) break;
      for (int t = 0; t < 32; t++) __builtin_amdgcn_s_sleep(127);
      if (++spins > (1ull << 22)) break;
      continue;
    }
    {
      // two words to poll while nothing is on offer (hundreds of workgroups may be idle: they must not hammer the memory system)
      if (dev_bcast_u64(TID == 0 ? (unsigned long long)AT_LOAD(&co->park_avail) : 0ull)) { ret = 1; break; }
      const unsigned long long w = dev_bcast_u64(TID == 0 ? ((unsigned long long)(AT_LOAD(&co->finished) >= c.coop_total) << 32) | (unsigned long long)AT_LOAD(&co->open) : 0ull);
      if (w >> 32) break;
      if ((w & 0xffffffffull) == 0ull) {
        for (int t = 0; t < 8; t++) __builtin_amdgcn_s_sleep(127);
        if (++spins > (1ull << 24)) break;
        continue;
      }
    }
    // every thread looks at some job slots: open, with a trial nobody has taken?  Helpers take [lo, iters) upwards, then
    // lo-1 .. 1 downwards (trial 0 is always the owner's).  The nearest such slot after `start` wins.
    double key = 1e300; int slot = NONE;
    for (int sl = TID; sl < njobs; sl += NT) {
      DgCoopJob* jb = &co->jobs[sl];
      const unsigned int sq = __hip_atomic_load(&jb->seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
      if (!(sq & 1u)) continue;
      const int lo = __hip_atomic_load(&jb->lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int iters = __hip_atomic_load(&jb->iters, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      { const int lim = __hip_atomic_load(&jb->pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + __hip_atomic_load(&jb->window, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (lim < iters) iters = lim; }
      if (iters <= lo) continue;
      const unsigned long long cl = AT_LOAD(&jb->claimed);
      const unsigned long long all = iters >= 64 ? ~0ull : ((1ull << iters) - 1ull);
      if ((cl & all) == all || lo < 1 || iters > DG_COOP_PHI) continue;
      if ((cl >> lo) == (all >> lo)) continue;            // nothing left in [lo, iters): the trials below lo are the owner's
      const double k2 = (double)((sl - start + njobs) % njobs);
      if (k2 < key) { key = k2; slot = sl; }
    }
    double kb; int sb;
    block_argmin(key, slot, lds + L.red, kb, sb);
    unsigned long long pick = ~0ull;        // (slot << 8) | trial, ~0: nothing
    if (sb != NONE && TID == 0) {
      DgCoopJob* jb = &co->jobs[sb];
      const unsigned int sq = __hip_atomic_load(&jb->seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
      if (sq & 1u) {
        __hip_atomic_fetch_add(&jb->active, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        bool ok = __hip_atomic_load(&jb->seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == sq;     // still the job we looked at: its fields are stable while `active` is held
        int cand = -1;
        if (ok) {
          const int lo = __hip_atomic_load(&jb->lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          int iters = __hip_atomic_load(&jb->iters, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          { const int lim = __hip_atomic_load(&jb->pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + __hip_atomic_load(&jb->window, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (lim < iters) iters = lim; }
          for (int tries = 0; tries < 64 && cand < 0; tries++) {
            const unsigned long long cl = AT_LOAD(&jb->claimed);
            int j2 = -1;
            for (int j = lo; j < iters; j++) if (!((cl >> j) & 1ull)) { j2 = j; break; }
            if (j2 < 0) break;
            if (!((__hip_atomic_fetch_or(&jb->claimed, 1ull << j2, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) >> j2) & 1ull)) cand = j2;
          }
        }
        if (cand >= 0) pick = ((unsigned long long)sb << 8) | (unsigned long long)cand;
        else __hip_atomic_fetch_sub(&jb->active, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    pick = dev_bcast_u64(pick);
    if (pick == ~0ull) { __builtin_amdgcn_s_sleep(127); if (++spins > (1ull << 24)) break; continue; }
    DgCoopJob* job = &co->jobs[pick >> 8];
    const int j = (int)(pick & 255ull);
    start = (int)(pick >> 8);
    // base point of the job into this workgroup's LDS (its owner does not touch the payload while `active` is held)
    __threadfence();
    {
      const double* pl = (const double*)__hip_atomic_load((unsigned long long*)&job->payload, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int i = TID; i < D.n; i += NT) { lds[L.u + i] = pl[i]; lds[L.o_du + i] = pl[D.n + i]; }
      for (int r = TID; r < D.nc; r += NT) { lds[L.l + r] = pl[2 * D.n + r]; lds[L.o_lhat + r] = pl[2 * D.n + D.nc + r]; }
      if (TID == 0) lds[L.scal + DG_XVALID] = 0.0;
      c.x0 = (cgptr)(const double*)__hip_atomic_load((unsigned long long*)&job->x0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    dev_coop_trial(c, job, j);
    if (TID == 0) __hip_atomic_fetch_sub(&job->active, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (TID == 0) {
    __hip_atomic_fetch_sub(&co->idle, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!passive) __hip_atomic_fetch_sub(&co->active_helpers, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  return ret;
}

// _line_search_3 (DGSQP.py:1057-1081) from the base (u, du, l, lhat) held in LDS.  On return u and l hold
// the LAST trial point; returns its merit.
__device__ inline double dev_line_search(const Ctx& c, double mu, double phi, double dphi, double S0, double S1) {
  const DgProb& D = dg_prob;
  const DgLds& L = D.L;
  lptr lds = LP(0);
  double alpha = 1.0, phit = 0.0;
  const int K = D.ls_spec, xsz = ((D.N + 1) * D.nq + 1) & ~1;
  const int iters = D.par.line_search_iters;
  // ---- cooperative mode: once `start` trials have been rejected the remaining ones are offered to idle workgroups, if there are
  // any (never while an event trace is recorded: it lists every trial).  Short searches -- the common case -- stay private.
  DgCoopJob* job = nullptr;
  const bool coop_ok = c.coop && !c.trace && K > 1 && iters > K && iters <= DG_COOP_PHI && D.ngd >= D.nc;
  const int C = 4;                       // the owner claims its own next trials in chunks of C, the helpers work upwards from `lo`
  int lo = iters;
  bool verify = false;
  unsigned long long mine = ~0ull;      // trials this workgroup evaluates itself (bit mask); without a job: all of them
  int rolled = -1;                       // block of K trials whose trajectories are in the speculation slots
  for (int i = 0; i < iters; i++) {
    if (coop_ok && !job && i > 0) {
      if (i == c.coop_start && i + C < iters && dev_bcast_u64(TID == 0 ? (unsigned long long)AT_LOAD(&c.coop->idle) : 0ull) > 0ull) {
        const int flip = (int)lds[L.scal + DG_COOP_FLIP];
        DgCoopJob* jb = &c.coop->jobs[2 * (int)blockIdx.x + flip];
        // the slot was closed two line searches ago; a helper may still be finishing a trial of that job (rare): then go alone
        if (dev_bcast_u64(TID == 0 ? (unsigned long long)__hip_atomic_load(&jb->active, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) : 0ull) == 0ull) {
          job = jb;
          lo = i + C;
          verify = c.coop_verify != 0;
          double* pl = c.coop_payload + (size_t)flip * (2 * D.n + 2 * D.nc);
          for (int t = TID; t < D.n; t += NT) { pl[t] = lds[L.u + t]; pl[D.n + t] = lds[L.o_du + t]; }
          for (int r = TID; r < D.nc; r += NT) { pl[2 * D.n + r] = lds[L.l + r]; pl[2 * D.n + D.nc + r] = lds[L.o_lhat + r]; }
          __threadfence();
          __syncthreads();
          if (TID == 0) {
            lds[L.scal + DG_COOP_FLIP] = (double)(1 - flip);
            job->claimed = (1ull << lo) - 1ull; job->ready = 0ull; job->pruned = 0ull;        // trials below lo are the owner's
            job->lo = lo; job->iters = iters; job->pos = i; job->window = c.coop_window;
            job->mu = mu; job->phi = phi; job->dphi = dphi; job->S0 = S0; job->S1 = S1;
            job->x0 = (const double*)c.x0; job->payload = pl;
            __threadfence();
            __hip_atomic_store(&job->seq, job->seq + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);     // even -> odd: open
            __hip_atomic_fetch_add(&c.coop->open, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
          }
          __syncthreads();
        }
      }
    }
    if (job && TID == 0) __hip_atomic_store(&job->pos, i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool have_h = false, h_pruned = false;     // a helper's value for this trial
    double h_phi = 0.0;
    if (job && i >= lo) {
      // claim the next chunk of trials the helpers have not taken yet; the others arrive through the job slot
      if ((i - lo) % C == 0) {
        unsigned long long want = 0ull;
        for (int j = i; j < i + C && j < iters; j++) want |= 1ull << j;
        const unsigned long long prev = dev_bcast_u64(TID == 0 ? __hip_atomic_fetch_or(&job->claimed, want, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) : 0ull);
        mine = (mine & ~want) | (want & ~prev);
      }
      if (!((mine >> i) & 1ull)) {
        // A helper has taken this trial.  Its value is used if it is there; the owner never WAITS for one: a helper needs a rollout of
        // its own (1.8 Mcycles) for what costs the owner 0.3 M on top of the block's rollout, so waiting loses whenever the search ends
        // within a few more trials (measured: twelve bench batches 596 -> 624 ms with waiting).  A search that fails runs into the
        // helpers' values after ~7 trials of its own and takes the other 40 from them.
        const unsigned long long rdy = dev_bcast_u64(TID == 0 ? __hip_atomic_load(&job->ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) : 0ull);
        if ((rdy >> i) & 1ull) {
          __threadfence();
          h_pruned = (dev_bcast_u64(TID == 0 ? AT_LOAD(&job->pruned) : 0ull) >> i) & 1ull;
          h_phi = __longlong_as_double((long long)dev_bcast_u64(TID == 0 ? (unsigned long long)__double_as_longlong(AT_LOAD(&job->phi_out[i])) : 0ull));
          have_h = true;
          if (TID == 0) __hip_atomic_fetch_add(&c.coop->used, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (!verify) {
            if (h_pruned) { alpha *= D.par.tau; continue; }
            phit = h_phi;
            if (phit <= phi + D.par.beta * alpha * dphi) break;
            if (i + 1 < iters) alpha *= D.par.tau;
            continue;
          }
        } else mine |= 1ull << i;         // not there yet: evaluate it here -- same value
      }
    }
    if (K > 1) {
      // trial step sizes are known in advance: roll the K of a block out concurrently (one instruction stream), test them in order
      if (rolled != i / K) {
        PROF_BEGIN(pt_);
        const int b0 = (i / K) * K, left = iters - b0;
        double ab = 1.0;
        for (int t = 0; t < b0; t++) ab *= D.par.tau;       // alpha of trial b0: the same products as the running alpha *= tau
        dev_rollout_multi(c, lds + L.u, lds + L.o_du, ab, D.par.tau, left < K ? left : K, lds + L.e_xs, xsz, D.ls_spec1, lds + L.e_xs2);
        rolled = i / K;
        PROF_END(PH_ROLLOUT, pt_);
      }
      const int jt = i % K;
      dev_evaluate_point(c, lds + L.u, alpha, lds + L.o_du, jt < D.ls_spec1 ? lds + L.e_xs + jt * xsz : lds + L.e_xs2 + (jt - D.ls_spec1) * xsz);
    } else dev_evaluate_point(c, lds + L.u, alpha, lds + L.o_du);
    // The merit is  1/2 |q + G'l|^2 + 1/2 (l'g)^2 + mu sum(g - s):  the first term is >= 0 and is the only one that needs
    // derivatives.  When the other two alone exceed the Armijo bound the trial is rejected from the constraint values (the
    // decision cannot differ; a relative margin covers rounding).  Never for the last allowed trial (its merit is returned)
    // and not while an event trace is recorded (the trace lists every trial's merit).
    if (!c.trace && i + 1 < iters) {
      if (dev_trial_pruned(c, alpha, mu, phi, dphi, S0, S1)) {
        if (have_h && !h_pruned && TID == 0) __hip_atomic_fetch_add(&c.coop->mismatches, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        alpha *= D.par.tau; continue;
      }
    }
    phit = dev_trial_merit(c, alpha, S0 + alpha * S1, mu);
    if (have_h && (h_pruned || __double_as_longlong(h_phi) != __double_as_longlong(phit)) && TID == 0)
      __hip_atomic_fetch_add(&c.coop->mismatches, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    dev_tr(c, 30, alpha); dev_tr(c, 31, phit);
    if (phit <= phi + D.par.beta * alpha * dphi) break;
    if (i + 1 < iters) alpha *= D.par.tau;
  }
  __syncthreads();
  if (job && TID == 0) {
    __hip_atomic_store(&job->seq, job->seq + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);     // odd -> even: closed
    __hip_atomic_fetch_sub(&c.coop->open, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
  for (int i = TID; i < D.n; i += NT) lds[L.u + i] = step_u(lds[L.u + i], alpha, lds[L.o_du + i]);
  for (int r = TID; r < D.nc; r += NT) lds[L.l + r] += alpha * (lds[L.o_lhat + r] - lds[L.l + r]);
  __syncthreads();
  return phit;
}

// full linearisation + QP at the current (u, l): _evaluate(hessian=True) followed by _solve_qp.
// Returns the QP flag (0 ok).  Leaves d, v, du, lhat and P in LDS.
__device__ inline int dev_linearize_and_qp(const Ctx& c, bool do_qp, double* cond3, gptr Qpd) {
  const DgProb& D = dg_prob;
  const DgLds& L = D.L;
  lptr lds = LP(0);
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
  if (dg_prob.osqp && dg_prob.big == 2) { dev_xl_psd(c, Qpd); return dev_qp_osqp_xl(c); }   // n > 128: dgsqp_osqp_xl.h
  if (dg_prob.osqp) {          // OSQP's arithmetic (dgsqp_osqp.h) on the projected Hessian M itself
    dev_psd_inverse(c, c.ws + dg_prob.ws_xM, false);
    if (Qpd) { for (int e = TID; e < dg_prob.n * dg_prob.n; e += NT) Qpd[e] = (c.ws + dg_prob.ws_xM)[e]; }
    return dev_qp_osqp(c);
  }
  if (dg_prob.big == 2) { dev_xl_psd(c, Qpd); return dev_xl_qp(c); }   // n > 128: dgsqp_xl.h
  if (dg_prob.classic_qp) {    // literal reg = 0 projection: condition ~1e12, classical active-set kernels on M itself
    dev_psd_inverse(c, c.ws + dg_prob.ws_xM, false);
    if (Qpd) { for (int e = TID; e < dg_prob.n * dg_prob.n; e += NT) Qpd[e] = (c.ws + dg_prob.ws_xM)[e]; }
    return dev_xl_qp(c);
  }
  dev_psd_inverse(c, Qpd);
  return dev_qp(c);
}

__device__ inline void dev_save_base(const Ctx& c) {
  const DgProb& D = dg_prob;
  const DgLds& L = D.L;
  gptr b = c.ws + D.ws_base;
  for (int i = TID; i < D.n; i += NT) { b[i] = LP(0)[L.u + i]; b[D.n + i] = LP(0)[L.o_du + i]; }
  for (int r = TID; r < D.nc; r += NT) { b[2 * D.n + r] = LP(0)[L.l + r]; b[2 * D.n + D.nc + r] = LP(0)[L.o_lhat + r]; }
  __syncthreads();
}
__device__ inline void dev_restore_base(const Ctx& c) {
  const DgProb& D = dg_prob;
  const DgLds& L = D.L;
  cgptr b = c.ws + D.ws_base;
  __syncthreads();
  for (int i = TID; i < D.n; i += NT) { LP(0)[L.u + i] = b[i]; LP(0)[L.o_du + i] = b[D.n + i]; }
  for (int r = TID; r < D.nc; r += NT) { LP(0)[L.l + r] = b[2 * D.n + r]; LP(0)[L.o_lhat + r] = b[2 * D.n + D.nc + r]; }
  __syncthreads();
}
__device__ inline void dev_take_full_step(const Ctx& c) {  // u += du ; l = lhat
  const DgProb& D = dg_prob;
  const DgLds& L = D.L;
  __syncthreads();
  for (int i = TID; i < D.n; i += NT) LP(0)[L.u + i] = step_u(LP(0)[L.u + i], 1.0, LP(0)[L.o_du + i]);
  for (int r = TID; r < D.nc; r += NT) LP(0)[L.l + r] = LP(0)[L.o_lhat + r];
  __syncthreads();
}

// _watchdog_line_search_4 (DGSQP.py:1174-1288; branch order of SURVEY.md A.7).  Base point and step
// (u_k, du_k, l_k, lhat_k) are in LDS on entry; returns the number of extra QP solves (negated when, with qp_method OSQP, one of them
// was reported infeasible: the solve ends there).
// One reading of the 100 MHz constant-rate counter for the whole workgroup (thread 0 reads, everybody gets the same value),
// so that wall-clock decisions (time_limit, DGSQP.py:470 and :1243-1247) are uniform across the 8 wavefronts of a scenario.
#define DG_CLOCK 56
__device__ inline double dev_block_clock() {
  lptr sc = LP(dg_prob.L.scal);
  __syncthreads();
  if (TID == 0) sc[DG_CLOCK] = (double)wall_clock64();
  __syncthreads();
  return sc[DG_CLOCK];
}
// iterate log (solve(): iter_data u_sol / l_sol); the record counter lives in an LDS scalar slot (uniform for the workgroup)
#define DG_ITREC 55
__device__ inline void dev_log_iterate(const Ctx& c) {
  if (!c.itlog) return;
  const DgProb& D = dg_prob;
  lptr sc = LP(D.L.scal);
  __syncthreads();
  const int rec = (int)sc[DG_ITREC];
  if (rec < c.itlog_cap) {
    gptr o = c.itlog + 1 + (int64_t)rec * (D.n + D.nc);
    for (int i = TID; i < D.n; i += NT) o[i] = LP(D.L.u)[i];
    for (int r = TID; r < D.nc; r += NT) o[D.n + r] = LP(D.L.l)[r];
  }
  __syncthreads();
  if (TID == 0) { sc[DG_ITREC] = (double)(rec + 1); c.itlog[0] = (double)(rec + 1); }
  __syncthreads();
}

__device__ inline int dev_watchdog(const Ctx& c, double mu, const LinScal& Sk) {
  const DgProb& D = dg_prob;
  const DgLds& L = D.L;
  lptr lds = LP(0);
  const double beta = D.par.beta;
  const double phi_k = Sk.phi, dphi_k = Sk.dphi;
  int nqp = 0;
  dev_save_base(c);
  // relaxed (full) step; most of them are accepted and re-linearised: roll out fused with the derivative pass
  dev_evaluate_point(c, lds + L.u, 1.0, lds + L.o_du, nullptr, true);
  const double phi1 = dev_trial_merit(c, 1.0, Sk.S0 + Sk.S1, mu);
  dev_tr(c, 20, phi1);
  if (phi1 <= phi_k + beta * dphi_k) { dev_take_full_step(c); return 0; }
  dev_take_full_step(c);  // (u_t, l_t) = (u_k + du_k, l_k + dl_k)
  bool fail = false;
  LinScal S;
  double phi_n = 0;
  const bool timed = D.par.time_limit >= 0.0;
  const double wd_start = timed ? dev_block_clock() : 0.0;     // start_time (DGSQP.py:1205)
  for (int t = 0; t < 5; t++) {
    const int flag = dev_linearize_and_qp(c, true, nullptr, nullptr);
    nqp++;
    if (flag != 0 && D.osqp) return -nqp;      // OSQP reported infeasibility: the reference's NaN step ends the solve (DGSQP.py:566-585)
    if (flag != 0) { fail = true; break; }
    dev_step_scalars(c, S);
    dev_evaluate_point(c, lds + L.u, 1.0, lds + L.o_du, nullptr, true);
    phi_n = dev_trial_merit(c, 1.0, S.S0 + S.S1, mu);
    dev_tr(c, 21, phi_n);
    if (phi_n > 1e6) break;                                   // merit_max; (u_t, l_t) not advanced
    if (phi_n <= phi_k + beta * dphi_k) { dev_take_full_step(c); return nqp; }
    dev_take_full_step(c);
    if (timed && (dev_block_clock() - wd_start) * 1e-8 > D.par.time_limit) { fail = true; break; }   // DGSQP.py:1243-1247
  }
  // insist on merit decrease
  {
    const int flag = dev_linearize_and_qp(c, true, nullptr, nullptr);
    nqp++;
    if (flag != 0 && D.osqp) return -nqp;
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
      if (flag != 0 && D.osqp) return -(nqp + 1);
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
// hessian_approximation = 'bfgs' (DGSQP.py:357-364, :535-557): damped BFGS update (Nocedal & Wright, Procedure 18.2) of the
// PROJECTED Hessian of the previous iteration,
//   s = u - u_prev,  y = d(u, l) - d(u_prev, l)  with  d = q + G^T l  at the CURRENT multipliers,
//   B = _nearestPD(Q_prev),  theta = 1 if s'y >= 0.2 s'Bs else 0.8 s'Bs / (s'Bs - s'y),  r = theta y + (1 - theta) B s,
//   Q = B - (B s)(B s)'/(s'Bs) + r r'/(s'r).
// Leaves q, g, packed G at u (no exact Hessian is evaluated) and Q in the raw-Hessian slot of the workspace.
__device__ __noinline__ void dev_bfgs_hessian(const Ctx& c) {
  const DgProb& D = dg_prob;
  const DgLds& L = D.L;
  lptr lds = LP(0);
  const int n = D.n;
  gptr up = c.ws + D.ws_bfgs, dm = up + n, Qk = c.ws + D.ws_bfgs + 2 * n, Bm = Qk + (int64_t)n * n, Qg = c.ws + D.ws_q;
  lptr uprev = lds + L.v;                 // persistent n-vector, rewritten by dev_qt_mul later in the iteration
  lptr yv = lds + L.p_y, Bs = lds + L.p_t, rv = lds + L.p_c;    // QP scratch (idle here)
  __syncthreads();
  for (int i = TID; i < n; i += NT) uprev[i] = up[i];
  __syncthreads();
  dev_evaluate(c, uprev, 0.0, nullptr, false);
  dev_stat_vector(c, lds + L.l, lds + L.d);
  for (int i = TID; i < n; i += NT) dm[i] = lds[L.d + i];
  // B = _nearestPD(Q_prev) (no reg, :364): through the projection kernel, whose output includes reg on the diagonal
  for (int e = TID; e < n * n; e += NT) Qg[e] = Qk[e];
  __threadfence_block();
  __syncthreads();
  if (D.big == 2) dev_xl_psd(c, Bm); else dev_psd_inverse(c, Bm, false);
  __threadfence_block();
  __syncthreads();
  dev_evaluate(c, lds + L.u, 0.0, nullptr, false);
  dev_stat_vector(c, lds + L.l, lds + L.d);
  const double reg = dev_reg();
  for (int i = TID; i < n; i += NT) { uprev[i] = lds[L.u + i] - up[i]; yv[i] = lds[L.d + i] - dm[i]; }   // uprev <- s
  __syncthreads();
  double a = 0, b = 0;
  for (int i = TID; i < n; i += NT) {
    double t = 0;
    for (int j = 0; j < n; j++) t += (Bm[(int64_t)j * n + i] - (i == j ? reg : 0.0)) * uprev[j];     // B symmetric: coalesced over i
    Bs[i] = t; a += uprev[i] * t; b += uprev[i] * yv[i];
  }
  const double sBs = block_sum(a, lds + L.red), sy = block_sum(b, lds + L.red);
  const double th = sy >= 0.2 * sBs ? 1.0 : 0.8 * sBs / (sBs - sy);
  a = 0;
  for (int i = TID; i < n; i += NT) { const double r = th * yv[i] + (1.0 - th) * Bs[i]; rv[i] = r; a += uprev[i] * r; }
  const double sr = block_sum(a, lds + L.red);
  for (int e = TID; e < n * n; e += NT) {
    const int i = e / n, j = e % n;
    Qg[e] = (Bm[e] - (i == j ? reg : 0.0)) - Bs[i] * Bs[j] / sBs + rv[i] * rv[j] / sr;
  }
  __threadfence_block();
  __syncthreads();
}

// DGSQP.solve() for one scenario (DGSQP.py:302-507)
// ------------------------------------------------------------------------------------------------
struct SolveOutPtrs {
  double *u, *l, *x, *cond, *cost;
  int32_t *status, *iters, *qp_solves;
};
// ---- deferral of long scenarios (DgPark, dgsqp_device.h) ----
// A slot for this scenario, or -1: not yet long enough, no fresh ticket left (nothing to make room for), or no slot free.  Block-uniform.
__device__ inline long long dev_park_reserve(const Ctx& c, int sqp_it, unsigned long long ticks0) {
  if (!c.park.entries || sqp_it < c.park.min_it) return -1;
  long long r = -1;
  if (TID == 0) {
    DgCoop* co = c.coop;
    const unsigned long long fin = AT_LOAD(&co->finished), di = AT_LOAD(&co->done_iters);
    // (nothing is deferred before 32 scenarios have finished: what "long" means for this game is not known yet)
    bool long_enough = fin >= 32ull && (unsigned long long)sqp_it * 16ull * fin >= (unsigned long long)c.park.factor_x16 * di;
    if (c.park.time_mode) {       // by time spent instead of by iterations
      const unsigned long long fresh = AT_LOAD(&co->done_fresh), dt = AT_LOAD(&co->done_ticks);
      long_enough = fresh >= 32ull && (wall_clock64() - ticks0) * 16ull * fresh >= (unsigned long long)c.park.factor_x16 * dt;
    }
    // ... and only while at least two more rounds of fresh scenarios wait: setting a scenario aside just before the queue runs empty
    // only delays it (a single 1,024-scenario batch of the 3-car merge lost 14 % that way)
    if (long_enough && AT_LOAD(c.ticket) + 2ull * gridDim.x < c.coop_total) {
      const unsigned int idx = __hip_atomic_fetch_add(&co->park_pushed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (idx < c.park.cap) r = (long long)idx;
    }
  }
  return (long long)dev_bcast_u64((unsigned long long)r);
}
// the scenario's state -- LDS arena, workgroup scratch, loop variables -- into slot idx
__device__ __noinline__ void dev_park_store(const Ctx& c, unsigned int idx, int sqp_it, int rel_tol_its, int total_qp, long long ticket, unsigned long long key, const double* cond,
                                            const double* xd = nullptr, const int* xi = nullptr) {
  const DgProb& D = dg_prob;
  double* slot = c.park.store + (size_t)idx * c.park.slot_doubles;
  __syncthreads();
  for (int i = TID; i < D.L.total; i += NT) slot[i] = dg_lds[i];
  {
    cgptr w = c.ws;
    double* sw = slot + D.L.total;
    const int64_t nw = D.ws_doubles;
    int64_t i = TID;
    for (; i + 7 * NT < nw; i += 8 * NT) {
      double t[8];
#pragma unroll
      for (int k = 0; k < 8; k++) t[k] = w[i + k * NT];
#pragma unroll
      for (int k = 0; k < 8; k++) sw[i + k * NT] = t[k];
    }
    for (; i < nw; i += NT) sw[i] = w[i];
  }
  __threadfence();
  __syncthreads();
  if (TID == 0) {
    DgParkEntry* e = &c.park.entries[idx];
    e->sqp_it = sqp_it; e->rel_tol_its = rel_tol_its; e->total_qp = total_qp; e->ticket = ticket; e->key = key;
    e->t_park = wall_clock64() - AT_LOAD(&c.coop->t_first);
    for (int i = 0; i < 3; i++) e->cond[i] = cond[i];
    if (xd) for (int i = 0; i < 6; i++) e->xd[i] = xd[i];
    if (xi) for (int i = 0; i < 6; i++) e->xi[i] = xi[i];
    __threadfence();
    __hip_atomic_store(&e->state, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&c.coop->park_avail, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
}
// ... and back (the workgroup's own line-search slot toggle survives: it belongs to the workgroup, not to the scenario)
__device__ __noinline__ void dev_park_load(const Ctx& c, unsigned int idx) {
  const DgProb& D = dg_prob;
  const double* slot = c.park.store + (size_t)idx * c.park.slot_doubles;
  __threadfence();
  __syncthreads();
  const double flip = dg_lds[D.L.scal + DG_COOP_FLIP];
  __syncthreads();
  for (int i = TID; i < D.L.total; i += NT) dg_lds[i] = slot[i];
  {
    gptr w = c.ws;
    const double* sw = slot + D.L.total;
    const int64_t nw = D.ws_doubles;
    int64_t i = TID;
    for (; i + 7 * NT < nw; i += 8 * NT) {
      double t[8];
#pragma unroll
      for (int k = 0; k < 8; k++) t[k] = sw[i + k * NT];
#pragma unroll
      for (int k = 0; k < 8; k++) w[i + k * NT] = t[k];
    }
    for (; i < nw; i += NT) w[i] = sw[i];
  }
  __threadfence_block();
  __syncthreads();
  if (TID == 0) dg_lds[D.L.scal + DG_COOP_FLIP] = flip;
  __syncthreads();
}
// The deferred scenario that has cost the most so far, or -1 when none is waiting.  Block-uniform.
__device__ __noinline__ long long dev_park_pop(const Ctx& c) {
  if (!c.park.entries) return -1;
  const DgProb& D = dg_prob;
  lptr lds = LP(0);
  DgCoop* co = c.coop;
  const int NONE = 0x7fffffff;
  for (int tries = 0; tries < 64; tries++) {
    const unsigned long long w = dev_bcast_u64(TID == 0 ? ((unsigned long long)AT_LOAD(&co->park_avail) << 32) | (unsigned long long)AT_LOAD(&co->park_pushed) : 0ull);
    if ((w >> 32) == 0ull) return -1;
    unsigned int np = (unsigned int)(w & 0xffffffffull);
    if (np > c.park.cap) np = c.park.cap;
    double key = 1e300; int slot = NONE;
    for (unsigned int sl = TID; sl < np; sl += NT) {
      DgParkEntry* e = &c.park.entries[sl];
      if (__hip_atomic_load(&e->state, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 1u) continue;
      const double k2 = -(double)AT_LOAD(&e->key);
      if (k2 < key) { key = k2; slot = (int)sl; }
    }
    double kb; int sb;
    block_argmin(key, slot, lds + D.L.red, kb, sb);
    if (sb == NONE) continue;            // (an entry counted in park_avail whose state is not visible yet, or somebody else took the last one)
    unsigned long long got = 0ull;
    if (TID == 0) {
      unsigned int expect = 1u;
      if (__hip_atomic_compare_exchange_strong(&c.park.entries[sb].state, &expect, 2u, __ATOMIC_ACQ_REL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        __hip_atomic_fetch_sub(&co->park_avail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&co->park_resumed, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        got = 1ull;
      }
    }
    if (dev_bcast_u64(got)) return (long long)sb;
  }
  return -1;
}

// The trajectory of the final u for the outputs: it is still in the evaluation scratch when u is, bit for bit, the point evaluated
// last (an exit at the convergence test, or right after a line search / full step whose last trial is the new iterate) and nothing
// has overwritten it since (tag in scal[DG_XVALID]); otherwise one more rollout.
__device__ inline void dev_final_rollout(const Ctx& c) {
  const DgProb& D = dg_prob;
  const DgLds& L = D.L;
  lptr lds = LP(0);
  lptr ue = lds + L.e_ue;
  const bool tagged = lds[L.scal + DG_XVALID] != 0.0;
  int differs = 0;
  if (tagged)
    for (int i = TID; i < D.n; i += NT) differs |= (__double_as_longlong(lds[L.u + i]) != __double_as_longlong(ue[i]));
  if (!tagged || __syncthreads_or(differs)) {
    __syncthreads();
    for (int i = TID; i < D.n; i += NT) ue[i] = lds[L.u + i];
    dev_rollout(c, ue, lds + L.e_x);
  }
  __syncthreads();
}

// Returns true when the scenario was DEFERRED (its state is in a slot; nothing was written to the outputs), false when it is done.
// resume: the entry of a deferred scenario whose state dev_park_load has just put back.  iters_out: SQP iterations of a finished solve.
__device__ inline bool dev_solve(const Ctx& c, cgptr u_ws, int64_t b, const SolveOutPtrs& O, const DgParkEntry* resume = nullptr,
                                 long long ticket = 0, unsigned long long ticks0 = 0ull, int* iters_out = nullptr) {
  const DgProb& D = dg_prob;
  const DgLds& L = D.L;
  lptr lds = LP(0);
  const int n = D.n, nc = D.nc;
  __syncthreads();
  const bool timed = D.par.time_limit >= 0.0;              // < 0: no limit (DGSQPParams.time_limit = None -> inf, DGSQP.py:64-67)
  double t_start = 0.0;
  int sqp_it = 0, rel_tol_its = 0, status = DGSQP_MAX_IT, total_qp = 0;
  if (!resume) {
    if (TID == 0) { lds[L.scal + DG_XVALID] = 0.0; lds[L.scal + DG_QP_NPREV] = 0.0; lds[L.scal + DG_REG] = D.par.reg; lds[L.scal + DG_PSD_PD] = 0.0; lds[L.scal + DG_OSQP_RHO] = 0.1; }
    for (int i = TID; i < n; i += NT) lds[L.u + i] = u_ws[i];
    for (int r = TID; r < nc; r += NT) lds[L.l + r] = 0.0;
    __syncthreads();
    t_start = timed ? dev_block_clock() : 0.0;             // solve_start (DGSQP.py:304): before the dual start
    // dual warm start.  Where the evaluation survives it (DgProb.lsqr_keeps_eval) the rollout is fused with the second-order derivative
    // pass the first linearisation needs anyway: one pass instead of rollout + first derivatives here and a fused pass there.
    dev_evaluate_point(c, lds + L.u, 0.0, nullptr, nullptr, D.lsqr_keeps_eval != 0);
    dev_evaluate_derivs(c, false);
    dev_dual_init(c);
    if (TID == 0) lds[L.scal + DG_ITREC] = 0.0;
    dev_log_iterate(c);                                    // record 0: (u_ws, dual start) = solve_info['init']
  } else {
    sqp_it = resume->sqp_it; rel_tol_its = resume->rel_tol_its; total_qp = resume->total_qp;
  }
  double cond[3] = {0, 0, 0};
  const bool l1 = D.par.merit_function == DGSQP_MERIT_STAT_L1;
  while (true) {
    if (sqp_it == 0 || !D.par.hessian_bfgs) dev_evaluate(c, lds + L.u, 0.0, nullptr, true);   // exact Hessian (DGSQP.py:353-355)
    else dev_bfgs_hessian(c);
    if (D.par.hessian_bfgs) {    // what the next iteration's update starts from: this iteration's Hessian and point
      gptr up = c.ws + D.ws_bfgs, Qk = c.ws + D.ws_bfgs + 2 * n;
      cgptr Qg = c.ws + D.ws_q;
      for (int i = TID; i < n; i += NT) up[i] = lds[L.u + i];
      for (int e = TID; e < n * n; e += NT) Qk[e] = Qg[e];
      __threadfence_block();
      __syncthreads();
    }
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
    const int qp_before = total_qp;      // trace code 40: QP solves of this iteration, at the reference's iter_data records (:386-451)
    if (cond[2] > 1e5) { dev_tr(c, 40, 0.0); dev_log_iterate(c); status = DGSQP_DIVERGED; break; }
    if (cond[0] < D.par.p_tol && cond[1] < D.par.d_tol && cond[2] < D.par.d_tol) { dev_tr(c, 40, 0.0); dev_log_iterate(c); status = DGSQP_CONV_ABS_TOL; break; }
    dev_qt_mul(c);
    int flag;
    if (D.osqp && D.big == 2) { dev_xl_psd(c, nullptr); flag = dev_qp_osqp_xl(c); }
    else if (D.osqp) { dev_psd_inverse(c, c.ws + D.ws_xM, false); flag = dev_qp_osqp(c); }
    else if (D.big == 2) { dev_xl_psd(c, nullptr); flag = dev_xl_qp(c); }   // n > 128: dgsqp_xl.h
    else if (D.classic_qp) { dev_psd_inverse(c, c.ws + D.ws_xM, false); flag = dev_xl_qp(c); }
    else { dev_psd_inverse(c, nullptr); flag = dev_qp(c); }
    total_qp++;
    if (flag != 0) { dev_tr(c, 40, 1.0); dev_log_iterate(c); status = DGSQP_QP_FAIL; break; }
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
    if (D.par.nonmono_ls) {
      const int wq = dev_watchdog(c, mu, S);
      if (wq < 0) {     // (qp_method OSQP only) a QP inside the watchdog was infeasible
        total_qp += -wq;
        dev_tr(c, 40, (double)(total_qp - qp_before)); dev_log_iterate(c); status = DGSQP_QP_FAIL; break;
      }
      total_qp += wq;
    } else dev_line_search(c, mu, S.phi, S.dphi, S.S0, S.S1);
    // relative-tolerance exit (DGSQP.py:454-462)
    double du2 = 0, dl2 = 0;
    cgptr bk = c.ws + D.ws_base;
    for (int i = TID; i < n; i += NT) { const double t = lds[L.u + i] - bk[i]; du2 += t * t; }
    for (int r = TID; r < nc; r += NT) { const double t = lds[L.l + r] - bk[2 * n + r]; dl2 += t * t; }
    du2 = block_sum(du2, lds + L.red);
    dl2 = block_sum(dl2, lds + L.red);
    dev_tr(c, 40, (double)(total_qp - qp_before));
    dev_log_iterate(c);
    if (sqrt(du2) < D.par.p_tol / 2 && sqrt(dl2) < D.par.d_tol / 2) {
      rel_tol_its++;
      if (rel_tol_its >= D.par.rel_tol_req && cond[0] < D.par.p_tol) { status = DGSQP_CONV_REL_TOL; break; }
    } else rel_tol_its = 0;
    sqp_it++;
    if (sqp_it >= D.par.sqp_iters) { status = DGSQP_MAX_IT; break; }
    if (timed && (dev_block_clock() - t_start) * 1e-8 > D.par.time_limit) { status = DGSQP_TIME_LIMIT; break; }   // block-uniform
    if (!resume && !timed) {      // (a wall-clock limit counts the time spent waiting: such solves are never deferred)
      const long long slot = dev_park_reserve(c, sqp_it, ticks0);
      if (slot >= 0) {
        const unsigned long long now = dev_bcast_u64(TID == 0 ? wall_clock64() : 0ull);
        // resume order: cost so far, weighted up for a scenario that is far from stationarity -- the ones that run to the iteration
        // limit are (measured on 3,243 deferred configs[1] scenarios: the resume phase takes 0.90 s ordered by cost alone, 0.73 s with
        // this weight, 0.66 s with the order an oracle would choose)
        const double sm = cond[2] == cond[2] ? fmin(cond[2], 1e30) : 1e30;
        const unsigned long long key = (unsigned long long)((double)(now - ticks0) * (1.0 + 2.0 * log10(1.0 + sm)));
        dev_park_store(c, (unsigned int)slot, sqp_it, rel_tol_its, total_qp, ticket, key, cond);
        return true;
      }
    }
  }
  if (iters_out) *iters_out = sqp_it;
  // outputs: q_pred = evaluate_dynamics(u, x0) (DGSQP.py:476), cost = f_J (:492)
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
