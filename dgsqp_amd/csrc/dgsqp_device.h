// Device side of the batched DG-SQP solver for gfx950 (MI355X).
//
// One 256-thread workgroup (4 wavefronts) owns one Monte-Carlo scenario for
// its whole solve (reference DGSQP.solve(), DGSQP/solvers/DGSQP.py:302-507):
// iterates, multipliers, the packed dense constraint gradients, the packed
// symmetric KKT Hessian, its eigenvectors and the active-set factor all live
// in the workgroup's 160 KB of LDS; only the Taylor tensor of the dynamics, the
// raw (unsymmetric) game Hessian and the watchdog's base-point backup go to a
// per-workgroup HBM/L2 scratch.  All functions below are block-cooperative:
// every thread of the workgroup calls them with identical arguments.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>

#include "dgsqp_layout.h"

#define NT DG_BLOCK
#define TID ((int)threadIdx.x)

// Explicit address spaces: every LDS access must compile to ds_read/ds_write and every workspace access to
// global_load/global_store.  (With plain `double*` the LDS base travels through a struct and hipcc falls back to
// flat_load/flat_store for all of them.)
typedef __attribute__((address_space(3))) double lds_d;
typedef lds_d* lptr;
typedef const lds_d* clptr;
typedef __attribute__((address_space(1))) double glb_d;
typedef glb_d* gptr;
typedef const glb_d* cgptr;
extern __shared__ double dg_lds[];
#define LP(off) ((lptr)dg_lds + (off))
// The game description lives in constant memory: wave-uniform reads become scalar loads in every function.
__constant__ DgProb dg_prob;
// row / dense-gradient tables are copied to LDS once per workgroup (dev_load_tables): they are indexed per lane
// inside sequential loops, where a constant-memory vector load would cost a global-memory round trip each time.
typedef __attribute__((address_space(3))) unsigned long long lds_u64;
static_assert(sizeof(DgRow) == 8 && sizeof(DgDense) == 16, "table entries are moved as 64-bit words");
// (DgProb.tab_const: games whose vectors nearly fill the arena read the tables from the constant block itself)
__device__ inline DgRow ld_row(int r) {
  if (dg_prob.tab_const) return dg_prob.rows[r];
  const unsigned long long w = ((const lds_u64*)LP(dg_prob.L.t_rows))[r];
  DgRow R;
  __builtin_memcpy(&R, &w, 8);
  return R;
}
__device__ inline DgDense ld_dense(int d) {
  if (dg_prob.tab_const) return dg_prob.dense[d];
  const lds_u64* p = (const lds_u64*)LP(dg_prob.L.t_dense) + 2 * d;
  const unsigned long long w[2] = {p[0], p[1]};
  DgDense R;
  __builtin_memcpy(&R, w, 16);
  return R;
}
__device__ inline DgTask ld_task(int t) {
  if (dg_prob.tab_const) return dg_prob.dtask[t];
  const unsigned long long w = ((const lds_u64*)LP(dg_prob.L.t_task))[t];
  DgTask R;
  __builtin_memcpy(&R, &w, 8);
  return R;
}
__device__ inline void dev_load_tables() {
  const DgProb& D = dg_prob;
  lds_u64* tr = (lds_u64*)LP(D.L.t_rows);
  lds_u64* td = (lds_u64*)LP(D.L.t_dense);
  const unsigned long long* sr = (const unsigned long long*)D.rows;
  const unsigned long long* sd = (const unsigned long long*)D.dense;
  if (!D.tab_const) for (int r = threadIdx.x; r < D.nc; r += DG_BLOCK) tr[r] = sr[r];
  if (!D.tab_const) for (int d = threadIdx.x; d < 2 * D.ndense; d += DG_BLOCK) td[d] = sd[d];
  if (!D.tab_const) {
    lds_u64* tk = (lds_u64*)LP(D.L.t_task);
    const unsigned long long* sk = (const unsigned long long*)D.dtask;
    for (int t = threadIdx.x; t < D.ntask; t += DG_BLOCK) tk[t] = sk[t];
  }
  if (threadIdx.x == 0) {
    lptr ta = LP(D.L.t_atan);
    ta[0] = 0.0; ta[1] = 4.63647609000806093515e-01; ta[2] = 7.85398163397448278999e-01; ta[3] = 9.82793723247329054082e-01; ta[4] = 1.57079632679489655800e+00;
    ta[5] = 0.0; ta[6] = 2.26987774529616870924e-17; ta[7] = 3.06161699786838301793e-17; ta[8] = 1.39033110312309984516e-17; ta[9] = 6.12323399573676603587e-17;
  }
  lptr tt = LP(D.L.t_track);
  constexpr int S1 = DGSQP_MAX_SEGS + 1;
  for (int i = threadIdx.x; i < S1; i += DG_BLOCK) {
    tt[i] = i <= D.P.n_segs ? D.P.seg_s[i] : 1e300;
    tt[S1 + i] = i < D.P.n_segs ? D.P.seg_curv[i] : 0.0;
    tt[2 * S1 + i] = i <= D.P.n_segs ? D.P.seg_ang[i] : 0.0;
    tt[3 * S1 + i] = i < D.P.n_segs ? (D.P.seg_ang[i + 1] - D.P.seg_ang[i]) / (D.P.seg_s[i + 1] - D.P.seg_s[i]) : 0.0;
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// Cooperative line search (dgsqp_solve.h: dev_line_search / dev_coop_help).  A launch ends with its slowest scenario -- 50
// iterations of failing 50-trial line searches, 0.7 s alone while the balanced share of a 1,024-scenario batch is 0.09 s.  The
// trial points of one line search are independent evaluations, so workgroups that find the ticket queue empty stay as HELPERS:
// they poll the job slots of the workgroups still solving, claim trial indices, evaluate them with the very same device
// functions (bit-identical merits) and hand the values back; the owner replays the sequential accept / reject logic on them.
// ------------------------------------------------------------------------------------------------
#define DG_COOP_PHI 64
struct DgCoopJob {
  unsigned int seq;               // odd: open (published with release semantics), even: closed
  unsigned int active;            // helpers currently working on this job (the slot is reused only when 0)
  unsigned long long claimed;     // bit j: trial j is taken (owner or helper), atomic OR
  unsigned long long ready;       // bit j: its result is in phi_out / pruned
  unsigned long long pruned;      // bit j: rejected by the derivative-free bound (no merit value)
  int lo, iters;                  // helpers take trials [lo, iters), lowest first ...
  int pos, window;                // ... but none beyond pos + window, pos = the trial its owner is at (bounds the work wasted when the search ends early)
  double mu, phi, dphi, S0, S1;   // the Armijo test's constants
  const double* x0;               // the scenario's initial state
  double* payload;                // u[n], du[n], l[nc], lhat[nc] of the base point
  double phi_out[DG_COOP_PHI];
};
struct DgCoop {
  unsigned int idle;              // helper workgroups currently polling
  unsigned int helper_regs;       // times a workgroup entered the helper loop (diagnostic)
  unsigned long long finished;    // scenarios completed by this launch
  unsigned int open;              // job slots currently open: the one word idle helpers poll
  unsigned int active_helpers;    // workgroups evaluating trials right now (at most Ctx.coop_helpers; the others sleep)
  unsigned long long helped;      // trials evaluated by helpers (diagnostic)
  unsigned int pad1_;
  unsigned int mismatches;        // verify mode: helper values whose bits differ from the owner's own evaluation (must stay 0)
  unsigned long long used;        // helper values the owners consumed (diagnostic)
  // deferral of long scenarios (DgPark below)
  unsigned int park_pushed;       // slots handed out
  unsigned int park_avail;        // deferred scenarios nobody has resumed yet: polled by idle workgroups next to `open`
  unsigned long long done_iters;  // SQP iterations of the scenarios finished so far (their mean sets the deferral threshold)
  unsigned long long park_resumed;   // (diagnostic)
  unsigned long long t_first;     // 100 MHz counter when the launch's first workgroup started (diagnostic time base)
  unsigned long long done_ticks;  // 100 MHz ticks the scenarios finished WITHOUT a deferral took, and how many those are: their mean is the
  unsigned long long done_fresh;  // deferral threshold in time mode (DgPark.time_mode: qp_method OSQP, where a scenario's cost is its ADMM iterations)
  DgCoopJob jobs[1];              // 2 per workgroup of the grid (double buffered)
};
// Deferral of long scenarios (scheduling only; cooperative launches).  A launch cannot end before its slowest scenario, and nothing in a
// scenario's inputs tells how long it will run; what does is its own history.  A scenario still iterating after ~twice the mean
// iteration count of the launch's finished scenarios is DEFERRED while fresh tickets remain: its workgroup stores the scenario's whole
// state -- the LDS arena and the workgroup's scratch -- in a slot and takes the next ticket.  Once the queue is empty the deferred
// scenarios are resumed, the ones that have cost the most so far first (longest processing time first), by whichever workgroup is
// free, from the stored image: the same instructions on the same data as an uninterrupted solve, bit for bit.  The long scenarios of
// the launch's LAST batches thus start their long tails while the chip still has other work, not after it.
struct DgParkEntry {
  unsigned int state;             // 0 empty, 1 stored (published with release semantics), 2 taken
  int sqp_it, rel_tol_its, total_qp;     // dev_solve's loop variables
  long long ticket;               // the scenario (ticket of the launch)
  unsigned long long key;         // 100 MHz ticks spent on it so far x (1 + 2 log10(1 + stationarity)): resumed in descending order
  unsigned long long t_park, t_resume, t_done;   // diagnostic (dgsqp_deferral_log): 100 MHz ticks since the launch's first ticket
  int final_its, final_qps;
  double cond[3];                 // convergence measures of its last iteration before it was set aside (diagnostic)
  double xd[6];                   // DG-SQP v2's further loop variables: reg, delta, both at the checkpoint, 100 MHz ticks spent solving so far
  int xi[6];                      // ... m-step count, checkpoint counter / index, merit-memory ring (entries, head)
};
struct DgPark {
  DgParkEntry* entries;           // null: no deferral
  double* store;                  // cap slots of slot_doubles (LDS image, then scratch image)
  unsigned int cap;
  int min_it;                     // never defer before this many iterations ...
  int factor_x16;                 // ... nor before factor x mean iterations of the finished scenarios (fixed point, 1/16)
  int time_mode;                  // 1: ... x mean TIME of the scenarios finished without a deferral instead (100 MHz ticks)
  unsigned long long slot_doubles;
};

struct Ctx {
  DgCoop* coop;    // cooperative line search of this launch (null: off)
  double* coop_payload;   // this workgroup's two payload buffers (2 x (2 n + 2 n_c) doubles)
  unsigned long long coop_total;   // scenarios of this launch (helpers leave when that many are finished)
  int coop_start;  // a line search is offered to helpers once this many of its trials have been rejected (short ones stay private)
  int coop_window; // helpers run at most this many trials ahead of the owner
  int coop_helpers; // at most this many idle workgroups evaluate trials; the others sleep until the launch ends
  int coop_verify; // diagnostic: the owner evaluates every trial itself as well and counts helper values that differ in their bits
  DgPark park;     // deferral of long scenarios (entries == null: off)
  const unsigned long long* ticket;   // the launch's ticket counter (deferral only while fresh tickets remain)
  gptr ws;      // this workgroup's global workspace
  cgptr x0;
  gptr trace;   // optional per-scenario event log: [0] = number of (code, value) pairs, then the pairs
  int trace_cap;   // capacity in pairs
  gptr itlog;   // optional per-scenario iterate log: [0] = number of records, then records of (n + n_c) doubles (u, l):
  int itlog_cap;   // record 0 = (u_ws, dual start), record i = iterates after SQP iteration i (iter_data u_sol / l_sol, DGSQP.py:386-451)
};
// Regularisation added to the projected Hessian (DGSQP.py:238-239).  Constant in DG-SQP v1; v2 decays it from iteration to
// iteration (DGSQP_v2.py:563,592), so it lives in an LDS scalar slot that dev_solve / dev_solve_v2 set per scenario.
#define DG_REG 52
#define DG_OSQP_F32 62   // scal slot: the current K^-1 of the XL ADMM iteration is stored in fp32 (dgsqp_params_t.mixed_precision and K well conditioned)
#define DG_OSQP_RHO 47   // scal slot: rho the scenario's previous OSQP call ended with (dgsqp_params_t.osqp_rho_carry); 0.1 at the start of a solve
__device__ inline double dev_reg() { const double r = LP(dg_prob.L.scal)[DG_REG]; return r > 0.0 ? r : 0.0; }
// event log compared event-by-event with the oracle's (tests/test_gpu.py::test_event_trace_parity)
__device__ inline void dev_tr(const Ctx& c, int code, double v) {
  if (c.trace && threadIdx.x == 0) {
    const int p = (int)c.trace[0];
    if (p < c.trace_cap) { c.trace[1 + 2 * p] = (double)code; c.trace[2 + 2 * p] = v; }
    c.trace[0] = (double)(p + 1);        // keeps counting past the capacity: the host sees count > capacity = truncated log
  }
}

// qp_method OSQP: QP calls and ADMM iterations since the last reset (dgsqp_osqp_counters): the measured mean iteration count behind
// bench.py's flop model (always on: two atomics per QP)
__device__ unsigned long long dg_osqp_count[2];

// Diagnostic build only (-DDG_PROF): per-phase cycle counters accumulated by thread 0 of every workgroup
// into a global array; the production library compiles these to nothing.
enum { PH_ROLLOUT = 0, PH_DERIV1, PH_DERIV2, PH_CHAINS, PH_DP, PH_JACOBI, PH_PFORM, PH_QP, PH_MERIT, PH_LSQR, PH_QTMUL, PH_SWEEP, PH_WGTOTAL, PH_WGMAX, PH_Q_SCAN, PH_Q_Y, PH_Q_DIR, PH_Q_STEP, PH_Q_UPD, PH_Q_REFINE, PH_Q_WARM, PH_W_BUILD, PH_W_MULT, PH_W_X, PH_E_TRI, PH_E_BIS, PH_E_VEC, PH_E_BACK, PH_E_KNEG, PH_C_NPREV, PH_C_MBUILD, PH_C_MWARM, PH_C_MFINAL, PH_C_TRIALS, PH_H_INJ, PH_H_COST, PH_H_CONTR, PH_H_ROWS, PH_O_SCALE, PH_O_W, PH_O_KINV, PH_O_ADMM, PH_O_ITERS, PH_O_CHECK, PH_O_PINV, PH_O_PROWS, PH_O_PSOLVE, PH_O_NACT, PH_O_GT, PH_O_PMUL, PH_O_GS, PH_O_UPD, PH_T_COL, PH_T_MV, PH_T_W, PH_T_UPD, PH_QW_BLK, PH_QW_SEQ, PH_QW_FIN, PH_QW_NPREV, PH_QW_DROPS, PH_PD_TRY, PH_COUNT };
#ifdef DG_PROF
__device__ unsigned long long dg_prof[PH_COUNT * 2];
__device__ unsigned long long dg_prof_scn[16384];
#define PROF_BEGIN(v) const long long v = clock64()
#define PROF_END(ph, v) do { if (threadIdx.x == 0) { atomicAdd(&dg_prof[2 * (ph)], (unsigned long long)(clock64() - v)); atomicAdd(&dg_prof[2 * (ph) + 1], 1ULL); } } while (0)
#define PROF_COUNT(ph, val) do { if (threadIdx.x == 0) { atomicAdd(&dg_prof[2 * (ph)], (unsigned long long)(val)); atomicAdd(&dg_prof[2 * (ph) + 1], 1ULL); } } while (0)
#else
#define PROF_BEGIN(v) do {} while (0)
#define PROF_END(ph, v) do {} while (0)
#define PROF_COUNT(ph, val) do {} while (0)
#endif

// ------------------------------------------------------------------------------------------------
// reductions (fixed tree => bitwise reproducible)
// ------------------------------------------------------------------------------------------------
// Cross-lane data movement uses DPP (row-local, a few cycles) and v_readlane (scalar broadcast) instead of
// ds_bpermute-based __shfl, whose LDS round trip dominated the many small reductions of this solver.
template <int CTRL>
__device__ inline double dpp_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ inline int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }
__device__ inline double lane_bcast(double v, int src_lane) {  // src_lane must be wave-uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}
// sum over the 64 lanes, returned to every lane (wave-uniform)
__device__ inline double wave_sum(double v) {
  v += dpp_f64<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_f64<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_f64<0x141>(v);  // row_half_mirror
  v += dpp_f64<0x140>(v);  // row_mirror: every lane now holds the sum of its 16-lane row
  return (lane_bcast(v, 0) + lane_bcast(v, 16)) + (lane_bcast(v, 32) + lane_bcast(v, 48));
}
__device__ inline double wave_max(double v) {
  v = fmax(v, dpp_f64<0xB1>(v));
  v = fmax(v, dpp_f64<0x4E>(v));
  v = fmax(v, dpp_f64<0x141>(v));
  v = fmax(v, dpp_f64<0x140>(v));
  return fmax(fmax(lane_bcast(v, 0), lane_bcast(v, 16)), fmax(lane_bcast(v, 32), lane_bcast(v, 48)));
}
__device__ inline double block_sum(double v, lptr red) {
  v = wave_sum(v);
  __syncthreads();
  if ((TID & 63) == 0) red[TID >> 6] = v;
  __syncthreads();
  double t = 0;
#pragma unroll
  for (int w = 0; w < NT / 64; w++) t += red[w];
  return t;
}
__device__ inline double block_max(double v, lptr red) {
  v = wave_max(v);
  __syncthreads();
  if ((TID & 63) == 0) red[TID >> 6] = v;
  __syncthreads();
  double t = red[0];
#pragma unroll
  for (int w = 1; w < NT / 64; w++) t = fmax(t, red[w]);
  return t;
}
// minimum with lowest index on ties
__device__ inline void argmin_pick(double& v, int& idx, double v2, int i2) {
  if (v2 < v || (v2 == v && i2 < idx)) { v = v2; idx = i2; }
}
// (value, index) minimum over the wave, lowest index on ties, result wave-uniform
__device__ inline void wave_argmin(double& v, int& idx) {
  argmin_pick(v, idx, dpp_f64<0xB1>(v), dpp_i32<0xB1>(idx));
  argmin_pick(v, idx, dpp_f64<0x4E>(v), dpp_i32<0x4E>(idx));
  argmin_pick(v, idx, dpp_f64<0x141>(v), dpp_i32<0x141>(idx));
  argmin_pick(v, idx, dpp_f64<0x140>(v), dpp_i32<0x140>(idx));
  double vr = lane_bcast(v, 0); int ir = __builtin_amdgcn_readlane(idx, 0);
  argmin_pick(vr, ir, lane_bcast(v, 16), __builtin_amdgcn_readlane(idx, 16));
  argmin_pick(vr, ir, lane_bcast(v, 32), __builtin_amdgcn_readlane(idx, 32));
  argmin_pick(vr, ir, lane_bcast(v, 48), __builtin_amdgcn_readlane(idx, 48));
  v = vr; idx = ir;
}
__device__ inline void block_argmin(double v, int idx, lptr red, double& vout, int& iout) {
  wave_argmin(v, idx);
  __syncthreads();
  if ((TID & 63) == 0) { red[TID >> 6] = v; red[32 + (TID >> 6)] = (double)idx; }
  __syncthreads();
  vout = red[0]; iout = (int)red[32];
  for (int w = 1; w < NT / 64; w++) {
    double v2 = red[w]; int i2 = (int)red[32 + w];
    if (v2 < vout || (v2 == vout && i2 < iout)) { vout = v2; iout = i2; }
  }
}

// ------------------------------------------------------------------------------------------------
// Lean fp64 elementary functions (<= 1 ulp on the arguments that occur here: angles and slip ratios, |x| < 2^20).
// The OCML versions carry large-argument reduction and full IEEE division (sincos ~200, atan2 ~125, tan ~220 ISA
// instructions); f_c is evaluated ~10^3 times per rollout on a handful of lanes, so the instruction count of these
// is the latency of the whole rollout.  Kernels are the classical fdlibm minimax polynomials.
// ------------------------------------------------------------------------------------------------
__device__ inline double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);          // v_rcp_f64: ~2^-23 relative
  double e = __builtin_fma(-x, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-x, r, 1.0);
  return __builtin_fma(r, e, r);
}
// p * z + C with the coefficient as a scalar operand of a three-operand v_fma_f64.  Left to itself the compiler keeps the
// coefficients in VGPRs and emits v_mov_b64 + v_fmac_f64 (destructive) for every Horner step.
__device__ inline double horner(double p, double z, double C) {
  double r;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(p), "v"(z), "s"(C));
  return r;
}
// fdlibm kernels on the reduced argument: sin r = r + r^3 S(r^2), cos r = 1 - r^2/2 + r^4 C(r^2)
__device__ inline double sin_poly(double z) {
  double ps = horner(1.58969099521155010221e-10, z, -2.50507602534068634195e-08);
  ps = horner(ps, z, 2.75573137070700676789e-06);
  ps = horner(ps, z, -1.98412698298579493134e-04);
  ps = horner(ps, z, 8.33333333332248946124e-03);
  return horner(ps, z, -1.66666666666666324348e-01);
}
__device__ inline double cos_poly(double z) {
  double pc = horner(-1.13596475577881948265e-11, z, 2.08757232129817482790e-09);
  pc = horner(pc, z, -2.75573143513906633035e-07);
  pc = horner(pc, z, 2.48015872894767294178e-05);
  pc = horner(pc, z, -1.38888888888741095749e-03);
  return horner(pc, z, 4.16666666666666019037e-02);
}
__device__ inline void dev_sincos(double x, double& so, double& co) {
  const double k = __builtin_rint(x * 0.63661977236758138);
  double r = __builtin_fma(-k, 1.5707963267948966, x);
  r = __builtin_fma(-k, 6.123233995736766e-17, r);
  const double z = r * r;
  const double ps = sin_poly(z), pc = cos_poly(z);
  const double sn = __builtin_fma(r * z, ps, r);
  const double cs = __builtin_fma(z * z, pc, __builtin_fma(-0.5, z, 1.0));
  const int q = (int)k;
  const double s1 = (q & 1) ? cs : sn, c1 = (q & 1) ? sn : cs;
  // quadrant signs by flipping the sign bit: sin negative for q = 2, 3 ; cos negative for q = 1, 2
  so = __hiloint2double(__double2hiint(s1) ^ ((q & 2) << 30), __double2loint(s1));
  co = __hiloint2double(__double2hiint(c1) ^ (((q + 1) & 2) << 30), __double2loint(c1));
}
// atan polynomial on the reduced argument t (|t| <= 7/16): t - t^3 P(t^2)
__device__ inline double atan_poly(double z) {
  double p = horner(1.62858201153657823623e-02, z, -3.65315727442169155270e-02);
  p = horner(p, z, 4.97687799461593236017e-02);
  p = horner(p, z, -5.83357013379057348645e-02);
  p = horner(p, z, 6.66107313738753120669e-02);
  p = horner(p, z, -7.69187620504482999495e-02);
  p = horner(p, z, 9.09088713343650656196e-02);
  p = horner(p, z, -1.11111104054623557880e-01);
  p = horner(p, z, 1.42857142725034663711e-01);
  p = horner(p, z, -1.99999999998764832476e-01);
  return horner(p, z, 3.33333333333329318027e-01);
}
__device__ inline double atan_poly_tail(double t, double hi, double lo) {
  const double z = t * t;
  const double p = atan_poly(z);
  return hi + ((lo - t * z * p) + t);
}
// Argument reduction atan(t) = atan(k) + atan((t - k)/(1 + k t)), k in {0, 1/2, 1, 3/2, inf}, folded into ONE division.  The
// range index is computed arithmetically (rid = number of thresholds 7/16, 11/16, 19/16, 39/16 <= t, no divergent blocks)
// and atan(k) comes from a 10-entry LDS table (dev_load_tables).
__device__ inline double dev_atan2(double y, double x) {
  const double ay = __builtin_fabs(y), ax = __builtin_fabs(x);
  const double y16 = 16.0 * ay;
  const int rid = (int)(y16 >= 7.0 * ax) + (int)(y16 >= 11.0 * ax) + (int)(y16 >= 19.0 * ax) + (int)(y16 >= 39.0 * ax);
  clptr ta = LP(dg_prob.L.t_atan);
  const double hi = ta[rid], lo = ta[5 + rid];
  const bool big = rid == 4;
  const double kk = 0.5 * (double)(big ? 0 : rid);
  const double num = big ? -ax : __builtin_fma(-kk, ax, ay);
  double den = big ? ay : __builtin_fma(kk, ay, ax);
  den = den == 0.0 ? 1.0 : den;                 // atan2(0, 0) = 0
  double r = atan_poly_tail(num * fast_rcp(den), hi, lo);
  r = x < 0.0 ? (3.141592653589793 - r) + 1.2246467991473532e-16 : r;
  return y < 0.0 ? -r : r;
}
__device__ inline double dev_atan(double x) {
  const double ay = __builtin_fabs(x);
  const int rid = (int)(ay >= 0.4375) + (int)(ay >= 0.6875) + (int)(ay >= 1.1875) + (int)(ay >= 2.4375);
  clptr ta = LP(dg_prob.L.t_atan);
  const double hi = ta[rid], lo = ta[5 + rid];
  const bool big = rid == 4;
  const double kk = 0.5 * (double)(big ? 0 : rid);
  const double num = big ? -1.0 : ay - kk;
  const double den = big ? ay : __builtin_fma(kk, ay, 1.0);      // >= 1
  const double r = atan_poly_tail(num * fast_rcp(den), hi, lo);
  return x < 0.0 ? -r : r;
}
__device__ inline double dev_tan(double x) { double s, c; dev_sincos(x, s, c); return s * fast_rcp(c); }

// ------------------------------------------------------------------------------------------------
// truncated univariate Taylor arithmetic, f(t) = c0 + c1 t + c2 t^2.
// Second derivatives of the discrete dynamics are recovered from the t^2 coefficient along the
// directions e_i and e_i+e_j (one direction per lane), first derivatives from the t coefficient.
// Kink conventions follow CasADi (if_else differentiates the taken branch, comparisons are constant).
// ------------------------------------------------------------------------------------------------
template <int DEG>
struct Ty {
  double c[DEG + 1];
};
template <int DEG> __device__ inline Ty<DEG> ty_const(double v) { Ty<DEG> r; r.c[0] = v; for (int i = 1; i <= DEG; i++) r.c[i] = 0.0; return r; }
template <int DEG> __device__ inline Ty<DEG> operator+(const Ty<DEG>& a, const Ty<DEG>& b) { Ty<DEG> r; for (int i = 0; i <= DEG; i++) r.c[i] = a.c[i] + b.c[i]; return r; }
template <int DEG> __device__ inline Ty<DEG> operator-(const Ty<DEG>& a, const Ty<DEG>& b) { Ty<DEG> r; for (int i = 0; i <= DEG; i++) r.c[i] = a.c[i] - b.c[i]; return r; }
template <int DEG> __device__ inline Ty<DEG> operator-(const Ty<DEG>& a) { Ty<DEG> r; for (int i = 0; i <= DEG; i++) r.c[i] = -a.c[i]; return r; }
template <int DEG> __device__ inline Ty<DEG> operator+(const Ty<DEG>& a, double b) { Ty<DEG> r = a; r.c[0] += b; return r; }
template <int DEG> __device__ inline Ty<DEG> operator+(double b, const Ty<DEG>& a) { return a + b; }
template <int DEG> __device__ inline Ty<DEG> operator-(const Ty<DEG>& a, double b) { Ty<DEG> r = a; r.c[0] -= b; return r; }
template <int DEG> __device__ inline Ty<DEG> operator-(double b, const Ty<DEG>& a) { return (-a) + b; }
template <int DEG> __device__ inline Ty<DEG> operator*(const Ty<DEG>& a, double b) { Ty<DEG> r; for (int i = 0; i <= DEG; i++) r.c[i] = a.c[i] * b; return r; }
template <int DEG> __device__ inline Ty<DEG> operator*(double b, const Ty<DEG>& a) { return a * b; }
template <int DEG> __device__ inline Ty<DEG> operator/(const Ty<DEG>& a, double b) { return a * (1.0 / b); }
template <int DEG> __device__ inline Ty<DEG> operator*(const Ty<DEG>& a, const Ty<DEG>& b) {
  Ty<DEG> r;
  r.c[0] = a.c[0] * b.c[0];
  if constexpr (DEG >= 1) r.c[1] = a.c[0] * b.c[1] + a.c[1] * b.c[0];
  if constexpr (DEG >= 2) r.c[2] = a.c[0] * b.c[2] + a.c[1] * b.c[1] + a.c[2] * b.c[0];
  return r;
}
template <int DEG> __device__ inline Ty<DEG> ty_recip(const Ty<DEG>& a) {
  Ty<DEG> r;
  r.c[0] = fast_rcp(a.c[0]);
  if constexpr (DEG >= 1) r.c[1] = -a.c[1] * r.c[0] * r.c[0];
  if constexpr (DEG >= 2) r.c[2] = -(a.c[2] * r.c[0] + a.c[1] * r.c[1]) * r.c[0];
  return r;
}
template <int DEG> __device__ inline Ty<DEG> operator/(const Ty<DEG>& a, const Ty<DEG>& b) { return a * ty_recip(b); }
template <int DEG> __device__ inline Ty<DEG> operator/(double a, const Ty<DEG>& b) { return ty_recip(b) * a; }
template <int DEG> __device__ inline void ty_sincos(const Ty<DEG>& a, Ty<DEG>& s, Ty<DEG>& c) {
  double s0, c0;
  dev_sincos(a.c[0], s0, c0);
  s.c[0] = s0; c.c[0] = c0;
  if constexpr (DEG >= 1) { s.c[1] = c0 * a.c[1]; c.c[1] = -s0 * a.c[1]; }
  if constexpr (DEG >= 2) { s.c[2] = 0.5 * a.c[1] * c.c[1] + a.c[2] * c0; c.c[2] = -0.5 * a.c[1] * s.c[1] - a.c[2] * s0; }
}
template <int DEG> __device__ inline Ty<DEG> ty_tan(const Ty<DEG>& a) {
  Ty<DEG> r;
  r.c[0] = dev_tan(a.c[0]);
  const double w0 = 1.0 + r.c[0] * r.c[0];
  if constexpr (DEG >= 1) r.c[1] = w0 * a.c[1];
  if constexpr (DEG >= 2) r.c[2] = w0 * a.c[2] + r.c[0] * r.c[1] * a.c[1];
  return r;
}
template <int DEG> __device__ inline Ty<DEG> ty_atan(const Ty<DEG>& a) {
  Ty<DEG> r;
  r.c[0] = dev_atan(a.c[0]);
  const double iw0 = fast_rcp(1.0 + a.c[0] * a.c[0]);
  if constexpr (DEG >= 1) r.c[1] = a.c[1] * iw0;
  if constexpr (DEG >= 2) r.c[2] = (a.c[2] - a.c[0] * a.c[1] * r.c[1]) * iw0;
  return r;
}
template <int DEG> __device__ inline Ty<DEG> ty_atan2(const Ty<DEG>& y, const Ty<DEG>& x) {
  Ty<DEG> r;
  r.c[0] = dev_atan2(y.c[0], x.c[0]);
  const double w0 = x.c[0] * x.c[0] + y.c[0] * y.c[0];
  const double iw0 = DEG >= 1 ? fast_rcp(w0) : 0.0;
  if constexpr (DEG >= 1) r.c[1] = (x.c[0] * y.c[1] - y.c[0] * x.c[1]) * iw0;
  if constexpr (DEG >= 2) {
    const double w1 = 2.0 * (x.c[0] * x.c[1] + y.c[0] * y.c[1]);
    const double n1 = 2.0 * (x.c[0] * y.c[2] - y.c[0] * x.c[2]);
    r.c[2] = (n1 - r.c[1] * w1) * (0.5 * iw0);
  }
  return r;
}
template <int DEG> __device__ inline Ty<DEG> ty_sqrt(const Ty<DEG>& a) {
  Ty<DEG> r;
  r.c[0] = sqrt(a.c[0]);
  const double ih = DEG >= 1 ? 0.5 * fast_rcp(r.c[0]) : 0.0;
  if constexpr (DEG >= 1) r.c[1] = a.c[1] * ih;
  if constexpr (DEG >= 2) r.c[2] = (a.c[2] - r.c[1] * r.c[1]) * ih;
  return r;
}
template <int DEG> __device__ inline Ty<DEG> ty_pow(const Ty<DEG>& a, double p) {
  Ty<DEG> r;
  r.c[0] = pow(a.c[0], p);
  const double ia = DEG >= 1 ? fast_rcp(a.c[0]) : 0.0;
  if constexpr (DEG >= 1) r.c[1] = p * r.c[0] * a.c[1] * ia;
  if constexpr (DEG >= 2) r.c[2] = (p * (2.0 * r.c[0] * a.c[2] + r.c[1] * a.c[1]) - a.c[1] * r.c[1]) * (0.5 * ia);
  return r;
}
template <int DEG> __device__ inline Ty<DEG> ty_abs(const Ty<DEG>& a) { return a.c[0] > 0 ? a : -a; }  // ca_abs, dynamics_models.py:228-234

// ------------------------------------------------------------------------------------------------
// track tables (radius_arclength_track.py:199-225): curvature piecewise constant, tangent piecewise linear
// ------------------------------------------------------------------------------------------------
// sbar = fmod(fmod(s, L) + L, L) for |s| < 2^40 L: one exact remainder step per fmod (s - L*trunc(s/L) is exact in fp64
// when the quotient is small), cheaper than the generic OCML loop.
__device__ inline double wrap_s(double s, double L, double invL) {
  double r = __builtin_fma(-L, __builtin_floor(s * invL), s);   // s mod L; the quotient may be one ulp off at the seam
  if (r < 0.0) r += L;
  if (r >= L) r -= L;
  return r;
}
template <int DEG>
__device__ inline void dev_track(const dgsqp_problem_t& P, const Ty<DEG>& s, double& curv, Ty<DEG>& psi) {
  constexpr int S1 = DGSQP_MAX_SEGS + 1;
  clptr tt = LP(dg_prob.L.t_track);
  const double sbar = wrap_s(s.c[0], P.track_L, dg_prob.inv_track_L);
  int seg = 0;
  for (int i = 1; i < P.n_segs; i++) seg += (sbar >= tt[i]) ? 1 : 0;
  curv = tt[S1 + seg];
  psi = (s + (sbar - s.c[0] - tt[seg])) * tt[3 * S1 + seg] + tt[2 * S1 + seg];
}

// Cubic-spline centre line (CasadiBSplineTrack, casadi_bspline_track.py:122-149; BASELINE configs[3]'s F1 track): curvature
// (x'y'' - y'x'') / (x'^2 + y'^2)^1.5 and tangent atan2(y', x') of the piecewise cubics x(s), y(s), evaluated in Taylor
// arithmetic -- the derivatives of the curvature that fAd / fEd need (the fourth derivative of a cubic piece is 0) come out
// of the propagation, no separate derivative tables.  The table (dg_prob.spl: knots, x coefficients, y coefficients) sits in
// the per-device constant block; waypoints are close to equispaced, so the interval is found from a proportional guess.
template <int DEG>
__device__ inline void dev_track_spline(const dgsqp_problem_t& P, const Ty<DEG>& s, Ty<DEG>& curv, Ty<DEG>& psi) {
  typedef Ty<DEG> T;
  const int nk = P.n_knots;
  const double* kn = dg_prob.spl;
  const double sbar = wrap_s(s.c[0], P.track_L, dg_prob.inv_track_L);
  int i = (int)(sbar * dg_prob.inv_track_L * (double)(nk - 1));
  i = i < 0 ? 0 : (i > nk - 2 ? nk - 2 : i);
  while (i > 0 && sbar < kn[i]) i--;
  while (i < nk - 2 && sbar >= kn[i + 1]) i++;
  const double* cx = dg_prob.spl + nk + 4 * i;
  const double* cy = dg_prob.spl + nk + 4 * (nk - 1) + 4 * i;
  const double x1 = cx[1], x2 = cx[2], x3 = cx[3], y1 = cy[1], y2 = cy[2], y3 = cy[3];
  const T t = s + (sbar - s.c[0] - kn[i]);            // d sbar / d s = 1 (fmod)
  const T dx = (t * (3.0 * x3) + 2.0 * x2) * t + x1, dy = (t * (3.0 * y3) + 2.0 * y2) * t + y1;
  const T ddx = t * (6.0 * x3) + 2.0 * x2, ddy = t * (6.0 * y3) + 2.0 * y2;
  const T n2 = dx * dx + dy * dy;
  curv = (dx * ddy - dy * ddx) * ty_recip(n2 * ty_sqrt(n2));
  psi = ty_atan2(dy, dx);
}

// kinematic bicycle in the Frenet frame (dynamics_models.py:1046-1070); q = [x,y,v,e_psi,s,e_y], u = [a, delta]
// terms of f_c that depend on the (zero-order-hold) input only: evaluated once per stage, not once per rk sub-stage
template <int DEG>
struct FcPre { Ty<DEG> a0, s0, c0; double im, iz, ilr; };   // kin: beta, sin beta, cos beta ; dyn: -, sin delta, cos delta
template <int DEG>
__device__ inline void dev_fc_pre_kin(const dgsqp_agent_t& ag, const Ty<DEG>* u, FcPre<DEG>& pre) {
  pre.im = 1.0 / ag.mass; pre.ilr = 1.0 / ag.L_r; pre.iz = 0.0;
  pre.a0 = ty_atan2(ty_tan(u[1]) * ag.L_r, ty_const<DEG>(ag.L_f + ag.L_r));
  ty_sincos(pre.a0, pre.s0, pre.c0);
}
template <int DEG>
__device__ inline void dev_fc_pre_dyn(const dgsqp_agent_t& ag, const Ty<DEG>* u, FcPre<DEG>& pre) {
  pre.im = 1.0 / ag.mass; pre.iz = 1.0 / ag.I_z; pre.ilr = 0.0;
  pre.a0 = u[1];
  ty_sincos(u[1], pre.s0, pre.c0);
}
// SPL: the track is a cubic spline (curvature is a Taylor value); a template parameter so that the arc-track instantiations --
// the hot ones -- carry nothing of the spline path in their register allocation
template <int DEG, bool SPL = false>
__device__ inline void dev_fc_kin(const dgsqp_problem_t& P, const dgsqp_agent_t& ag, const Ty<DEG>* q, const Ty<DEG>* u, const FcPre<DEG>& pre, Ty<DEG>* dq) {
  typedef Ty<DEG> T;
  const T &beta = pre.a0, &sb = pre.s0, &cb = pre.c0;
  const T psidot = q[2] * sb * pre.ilr;
  T F = q[2] * (-ag.c_da) - q[2] * ty_abs(q[2]) * ag.c_dr - psidot * psidot * ag.c_s;
  if (ag.c_r != 0.0) F = F - ty_pow(ty_abs(q[2]), ag.p_r) * (q[2] / ty_sqrt(q[2] * q[2] + 1e-6)) * ag.c_r;
  typename std::conditional<SPL, T, double>::type c;
  T psit;
  constexpr bool spl = SPL;
  if constexpr (spl) dev_track_spline<DEG>(P, q[4], c, psit);
  else dev_track(P, q[4], c, psit);
  T s1, c1, s2, c2;
  ty_sincos(beta + psit + q[3], s1, c1);
  ty_sincos(beta + q[3], s2, c2);
  const T inv = ty_recip(1.0 - q[5] * c);
  const T vlon = q[2] * c2 * inv;
  dq[0] = q[2] * c1;
  dq[1] = q[2] * s1;
  dq[2] = u[0] + F * pre.im;
  dq[3] = psidot - vlon * c;
  dq[4] = vlon;
  dq[5] = q[2] * s2;
}

// dynamic bicycle, Pacejka / linear tyres (dynamics_models.py:2008-2062); q = [x,y,vx,vy,w,e_psi,s,e_y]
template <int DEG, bool SPL = false>
__device__ inline void dev_fc_dyn(const dgsqp_problem_t& P, const dgsqp_agent_t& ag, const Ty<DEG>* q, const Ty<DEG>* u, const FcPre<DEG>& pre, Ty<DEG>* dq) {
  typedef Ty<DEG> T;
  const T &vx = q[2], &vy = q[3], &w = q[4];
  typename std::conditional<SPL, T, double>::type c;
  T psit;
  constexpr bool spl = SPL;
  if constexpr (spl) dev_track_spline<DEG>(P, q[6], c, psit);
  else dev_track(P, q[6], c, psit);
  const T &sd = pre.s0, &cd = pre.c0;
  const T vyf = vy + w * ag.L_f;
  T af;
  if (ag.simple_slip) af = u[1] - ty_atan2(vyf, vx);
  else af = -ty_atan2(vyf * cd - vx * sd, vx * cd + vyf * sd);
  const T ar = -ty_atan2(vy - w * ag.L_r, vx);
  T fyf, fyr;
  if (ag.tire_model == 0) {
    T s_, c_;
    ty_sincos(ty_atan(af * ag.pac_Bf) * ag.pac_Cf, s_, c_);
    fyf = s_ * ag.pac_Df;
    ty_sincos(ty_atan(ar * ag.pac_Br) * ag.pac_Cr, s_, c_);
    fyr = s_ * ag.pac_Dr;
  } else {
    fyf = af * (ag.lin_Bf * ag.mass * ag.gravity * ag.L_r / (ag.L_f + ag.L_r));
    fyr = ar * (ag.lin_Br * ag.mass * ag.gravity * ag.L_f / (ag.L_f + ag.L_r));
  }
  T F = vx * (-ag.c_da) - vx * ty_abs(vx) * ag.c_dr;
  if (ag.c_r != 0.0) F = F - ty_pow(ty_abs(vx), ag.p_r) * (vx / ty_sqrt(vx * vx + 1e-6)) * ag.c_r;
  T a_r, a_f;
  if (ag.drive_wheels == 0) { a_r = u[0] * 0.5; a_f = u[0] * 0.5; } else { a_r = u[0]; a_f = ty_const<DEG>(0.0); }
  const T ax = a_r + a_f * cd + (F - fyf * sd) * pre.im;
  const T ay = a_f * sd + (fyf * cd + fyr) * pre.im;
  T se, ce, st, ct;
  ty_sincos(q[5], se, ce);
  ty_sincos(q[5] + psit, st, ct);
  const T vlon = (vx * ce - vy * se) * ty_recip(1.0 - q[7] * c);
  dq[0] = vx * ct - vy * st;
  dq[1] = vy * ct + vx * st;
  dq[2] = ax + w * vy;
  dq[3] = ay - w * vx;
  dq[4] = (fyf * cd * ag.L_f - fyr * ag.L_r) * pre.iz;
  dq[5] = w - vlon * c;
  dq[6] = vlon;
  dq[7] = vx * se + vy * ce;
}

// kinematic unicycle in the global frame (dynamics_models.py:331-339); q = [x, y, v, psi], u = [F, omega]
template <int DEG>
__device__ inline void dev_fc_uni(const Ty<DEG>* q, const Ty<DEG>* u, const FcPre<DEG>& pre, Ty<DEG>* dq) {
  Ty<DEG> s, c;
  ty_sincos(q[3], s, c);
  dq[0] = q[2] * c;
  dq[1] = q[2] * s;
  dq[2] = u[0] * pre.im;
  dq[3] = u[1];
}

template <int DEG, int NQA, bool SPL = false>
__device__ inline void dev_fc(const dgsqp_problem_t& P, const dgsqp_agent_t& ag, const Ty<DEG>* q, const Ty<DEG>* u, const FcPre<DEG>& pre, Ty<DEG>* dq) {
  if constexpr (NQA == 8) dev_fc_dyn<DEG, SPL>(P, ag, q, u, pre, dq);
  else if constexpr (NQA == 4) dev_fc_uni<DEG>(q, u, pre, dq);
  else dev_fc_kin<DEG, SPL>(P, ag, q, u, pre, dq);
}

// one discrete step of the joint model's integrator (dynamics_models.py:88-99, :188-219).  The integrator is a template
// parameter so that every instantiation keeps only the stage arrays it needs in registers (rk4: x, k-accumulator, k, t).
template <int DEG, int NQA, int INTEG, bool SPL = false>
__device__ inline void dev_fd_t(const dgsqp_problem_t& P, const dgsqp_agent_t& ag, const Ty<DEG>* q, const Ty<DEG>* u, Ty<DEG>* qn) {
  typedef Ty<DEG> T;
  T x[NQA], k1[NQA], k2[NQA], t[NQA];
  FcPre<DEG> pre;
  if constexpr (NQA == 8) dev_fc_pre_dyn<DEG>(ag, u, pre);
  else if constexpr (NQA == 4) { pre.im = 1.0 / ag.mass; pre.iz = 0.0; pre.ilr = 0.0; }
  else dev_fc_pre_kin<DEG>(ag, u, pre);
#pragma unroll
  for (int i = 0; i < NQA; i++) x[i] = q[i];
  if constexpr (INTEG == DGSQP_INT_EULER) {
    dev_fc<DEG, NQA, SPL>(P, ag, x, u, pre, k1);
#pragma unroll
    for (int i = 0; i < NQA; i++) qn[i] = x[i] + k1[i] * P.dt;
    return;
  }
  const double h = P.dt / P.substeps;
  for (int m = 0; m < P.substeps; m++) {
    if constexpr (INTEG == DGSQP_INT_RK4) {
      dev_fc<DEG, NQA, SPL>(P, ag, x, u, pre, k1);
#pragma unroll
      for (int i = 0; i < NQA; i++) t[i] = x[i] + k1[i] * (h / 2);
      dev_fc<DEG, NQA, SPL>(P, ag, t, u, pre, k2);
#pragma unroll
      for (int i = 0; i < NQA; i++) { t[i] = x[i] + k2[i] * (h / 2); k1[i] = k1[i] + k2[i] * 2.0; }
      dev_fc<DEG, NQA, SPL>(P, ag, t, u, pre, k2);
#pragma unroll
      for (int i = 0; i < NQA; i++) { t[i] = x[i] + k2[i] * h; k1[i] = k1[i] + k2[i] * 2.0; }
      dev_fc<DEG, NQA, SPL>(P, ag, t, u, pre, k2);
#pragma unroll
      for (int i = 0; i < NQA; i++) x[i] = x[i] + (k1[i] + k2[i]) * (h / 6.0);
    } else if constexpr (INTEG == DGSQP_INT_RK3) {
      T k3[NQA];
      dev_fc<DEG, NQA, SPL>(P, ag, x, u, pre, k1);
#pragma unroll
      for (int i = 0; i < NQA; i++) { k1[i] = k1[i] * h; t[i] = x[i] + k1[i] * 0.5; }
      dev_fc<DEG, NQA, SPL>(P, ag, t, u, pre, k2);
#pragma unroll
      for (int i = 0; i < NQA; i++) { k2[i] = k2[i] * h; t[i] = x[i] - k1[i] + k2[i] * 2.0; }
      dev_fc<DEG, NQA, SPL>(P, ag, t, u, pre, k3);
#pragma unroll
      for (int i = 0; i < NQA; i++) x[i] = x[i] + (k1[i] + k2[i] * 4.0 + k3[i] * h) / 6.0;
    } else {
      dev_fc<DEG, NQA, SPL>(P, ag, x, u, pre, k1);
#pragma unroll
      for (int i = 0; i < NQA; i++) t[i] = x[i] + k1[i] * h;
      dev_fc<DEG, NQA, SPL>(P, ag, t, u, pre, k2);
#pragma unroll
      for (int i = 0; i < NQA; i++) x[i] = x[i] + (k1[i] + k2[i]) * (h / 2);
    }
  }
#pragma unroll
  for (int i = 0; i < NQA; i++) qn[i] = x[i];
}
template <int DEG, int NQA, bool SPL = false>
__device__ inline void dev_fd(const dgsqp_problem_t& P, const dgsqp_agent_t& ag, const Ty<DEG>* q, const Ty<DEG>* u, Ty<DEG>* qn) {
  switch (P.integrator) {   // uniform
    case DGSQP_INT_EULER: dev_fd_t<DEG, NQA, DGSQP_INT_EULER, SPL>(P, ag, q, u, qn); break;
    case DGSQP_INT_RK4: dev_fd_t<DEG, NQA, DGSQP_INT_RK4, SPL>(P, ag, q, u, qn); break;
    case DGSQP_INT_RK3: dev_fd_t<DEG, NQA, DGSQP_INT_RK3, SPL>(P, ag, q, u, qn); break;
    default: dev_fd_t<DEG, NQA, DGSQP_INT_RK2, SPL>(P, ag, q, u, qn); break;
  }
}

__device__ inline int am_col(const DgProb& D, int a, int k, int j) { return a * D.N * DGSQP_NUA + k * DGSQP_NUA + j; }

// ------------------------------------------------------------------------------------------------
// structured products with the constraint Jacobian G (n_c x n), never formed densely
// ------------------------------------------------------------------------------------------------
// y[r] = (G x)[r] for one row.  GP: where the packed gradients live -- LDS (clptr) or, for games whose gradients exceed the
// arena (XL layout, n > ~160), the workgroup's global scratch (cgptr)
template <class GP>
__device__ inline double g_row_dot(const DgProb& D, GP gd, int r, clptr x) {
  const DgRow R = ld_row(r);
  switch (R.type) {
    case DG_R_IN_UB: return x[am_col(D, R.a, R.k, R.idx)];
    case DG_R_IN_LB: return -x[am_col(D, R.a, R.k, R.idx)];
    case DG_R_RATE_UB:
    case DG_R_RATE_LB: {
      double t = x[am_col(D, R.a, R.k, R.idx)];
      if (R.k > 0) t -= x[am_col(D, R.a, R.k - 1, R.idx)];
      return R.type == DG_R_RATE_UB ? t : -t;
    }
    default: {
      const DgDense dd = ld_dense(R.dense);
      const GP p = gd + dd.off;
      const int len = 2 * dd.k;
      double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
      clptr xa = x + dd.a * D.N * DGSQP_NUA;
      int i = 0;
      for (; i + 3 < len; i += 4) { s0 += p[i] * xa[i]; s1 += p[i + 1] * xa[i + 1]; s2 += p[i + 2] * xa[i + 2]; s3 += p[i + 3] * xa[i + 3]; }
      for (; i < len; i++) s0 += p[i] * xa[i];
      if (dd.kind == 1) {
        clptr xb = x + dd.b * D.N * DGSQP_NUA;
        const GP pb = p + len;
        for (i = 0; i + 3 < len; i += 4) { s0 += pb[i] * xb[i]; s1 += pb[i + 1] * xb[i + 1]; s2 += pb[i + 2] * xb[i + 2]; s3 += pb[i + 3] * xb[i + 3]; }
        for (; i < len; i++) s0 += pb[i] * xb[i];
      }
      return R.sgn * ((s0 + s1) + (s2 + s3));
    }
  }
}
// out[n] = G^T y.  yd is an LDS scratch of ndense doubles.  Contains barriers.
template <class GP>
__device__ __noinline__ void gt_mul_t(const Ctx& c, GP gd, clptr y, lptr out) {
  const DgProb& D = dg_prob;
  lptr yd = LP(D.L.yd);
  __syncthreads();
  for (int d = TID; d < D.ndense; d += NT) {
    const DgDense dd = ld_dense(d);
    yd[d] = (dd.r_pos >= 0 ? y[dd.r_pos] : 0.0) - (dd.r_neg >= 0 ? y[dd.r_neg] : 0.0);
  }
  __syncthreads();
  // four lanes per column: lane part 0 adds the box / rate rows, all four share the dense gradients (d = part, part+4, ..)
  for (int it = TID; it < 4 * D.n; it += NT) {
    const int col = it >> 2, part = it & 3;
    const int a = col / (D.N * DGSQP_NUA), rem = col % (D.N * DGSQP_NUA), t = rem / DGSQP_NUA, j = rem % DGSQP_NUA;
    double s = 0;
    if (part == 0) {
      int r;
      if ((r = D.r_in_ub[a][t][j]) >= 0) s += y[r];
      if ((r = D.r_in_lb[a][t][j]) >= 0) s -= y[r];
      if ((r = D.r_rate_ub[a][t][j]) >= 0) s += y[r];
      if ((r = D.r_rate_lb[a][t][j]) >= 0) s -= y[r];
      if (t + 1 < D.N) {
        if ((r = D.r_rate_ub[a][t + 1][j]) >= 0) s -= y[r];
        if ((r = D.r_rate_lb[a][t + 1][j]) >= 0) s += y[r];
      }
    }
    for (int d = D.stage_dense0[t + 1] + part; d < D.ndense; d += 4) {   // only gradients of later stages reach column (a,t,j)
      const DgDense dd = ld_dense(d);
      if (dd.a == a) s += yd[d] * gd[dd.off + t * DGSQP_NUA + j];
      else if (dd.kind == 1 && dd.b == a) s += yd[d] * gd[dd.off + 2 * dd.k + t * DGSQP_NUA + j];
    }
    s += dpp_f64<0xB1>(s);
    s += dpp_f64<0x4E>(s);
    if (part == 0) out[col] = s;
  }
  __syncthreads();
}
// the packed gradients of this workgroup: LDS, or the global scratch when they do not fit (DgProb.gd_global)
__device__ inline cgptr dev_gd_global(const Ctx& c) { return c.ws + dg_prob.ws_gd; }
__device__ inline void gt_mul(const Ctx& c, clptr y, lptr out) {
  if (dg_prob.gd_global) gt_mul_t<cgptr>(c, dev_gd_global(c), y, out);
  else gt_mul_t<clptr>(c, LP(dg_prob.L.gd), y, out);
}
__device__ inline double g_row_dot_any(const Ctx& c, int r, clptr x) {
  return dg_prob.gd_global ? g_row_dot<cgptr>(dg_prob, dev_gd_global(c), r, x) : g_row_dot<clptr>(dg_prob, LP(dg_prob.L.gd), r, x);
}
