/*
 * dgsqp.h -- C-ABI of the MI355X batched Dynamic-Game-SQP solver.
 *
 * This is the drop-in boundary for ONE hot path of zhu-edward/DGSQP: the
 * Monte-Carlo SQP inner loop `DGSQP.solve()` (reference
 * DGSQP/solvers/DGSQP.py:302-507) together with everything it calls per
 * sample (`_evaluate` :509-533, `_solve_qp` :232-266, `_nearestPD`
 * :1290-1296, `_get_mu` :559-585, `_line_search_3` :1057-1081,
 * `_watchdog_line_search_4` :1174-1288).
 *
 * The reference is pure Python; its "FFI" for this path is the set of CasADi
 * `Function.__call__`s and the `ca.conic` call made from `solve()`.  A
 * maintainer binds this library with `ctypes.CDLL`.
 * Only plain pointers, fixed-width integers and doubles cross the boundary.
 *
 * Because CasADi Function objects cannot cross a C boundary, the symbolic
 * game (dynamics / costs / constraints built by
 * scripts/DGSQP_*_monte_carlo_*.py) is passed as a declarative POD
 * description, `dgsqp_problem_t`.
 */
#ifndef DGSQP_H
#define DGSQP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DGSQP_MAX_AGENTS 6
#define DGSQP_MAX_SEGS 16
#define DGSQP_MAX_NQA 8 /* largest per-agent state (dynamic bicycle) */
#define DGSQP_NUA 2     /* every vehicle model of the path has 2 inputs */

/* dynamics model ids (reference: DGSQP/dynamics/dynamics_models.py) */
enum {
  DGSQP_MODEL_KIN_BICYCLE = 0, /* CasadiKinematicBicycleCombined :997  */
  DGSQP_MODEL_DYN_BICYCLE = 1, /* CasadiDynamicBicycleCombined   :1945 */
  DGSQP_MODEL_UNICYCLE = 2,    /* CasadiKinematicUnicycle        :306 : state [x, y, v, psi], input [F, omega] */
};
#define DGSQP_MAX_LANES 2 /* lane half-plane rows per agent and stage */
/* discretisation (dynamics_models.py:88-125, :188-219) */
enum { DGSQP_INT_EULER = 0, DGSQP_INT_RK4 = 1, DGSQP_INT_RK3 = 2, DGSQP_INT_RK2 = 3 };
/* terminal competition cost shape */
enum { DGSQP_COMP_ATAN = 0, DGSQP_COMP_LINEAR = 1 };
/* merit function (DGSQP.py:971-978) */
enum { DGSQP_MERIT_STAT_L1 = 0, DGSQP_MERIT_STAT = 1, DGSQP_MERIT_SUM_OBJ_L1 = 2 /* DG-SQP v2 only (DGSQP_v2.py:1161-1164) */ };

/* per-scenario exit codes == reference `msg` strings (DGSQP.py:388,396,408,458,466,471) */
enum {
  DGSQP_CONV_ABS_TOL = 0,
  DGSQP_CONV_REL_TOL = 1,
  DGSQP_MAX_IT = 2,
  DGSQP_DIVERGED = 3,
  DGSQP_QP_FAIL = 4,
  DGSQP_TIME_LIMIT = 5
};

/* library error codes */
enum {
  DGSQP_OK = 0,
  DGSQP_E_ARG = -1,      /* bad argument / unsupported problem */
  DGSQP_E_DEVICE = -2,   /* HIP error, no GPU, or kernel image missing */
  DGSQP_E_NOMEM = -3,
  DGSQP_E_TOO_LARGE = -4 /* problem exceeds the compiled LDS/workspace limits */
};

/*
 * One agent: vehicle model (model_types.py:34-105), cost weights
 * (scripts/DGSQP_ALGAMES_monte_carlo_chicane.py:111-122,223-277) and
 * constraint data (:80-109,282-293).
 */
typedef struct {
  int32_t model;        /* DGSQP_MODEL_* */
  int32_t tire_model;   /* 0 pacejka, 1 linear   (dynamics_models.py:2022-2029) */
  int32_t drive_wheels; /* 0 all, 1 rear         (dynamics_models.py:2036-2041) */
  int32_t simple_slip;  /* dynamics_models.py:2015-2020 */
  double L_f, L_r, mass, I_z, gravity;
  double c_dr, c_da, c_s, c_r, p_r; /* drag, damping, slip, rolling resistance (+exponent) */
  double pac_Bf, pac_Br, pac_Cf, pac_Cr, pac_Df, pac_Dr;
  double lin_Bf, lin_Br;

  /* stage cost 1/2 sum w_in u^2 + 1/2 sum w_rate (u-u_prev)^2 */
  double w_in[DGSQP_NUA];
  double w_rate[DGSQP_NUA];
  /* terminal cost  -w_prog*s_a + w_comp * sum_{b!=a} comp(s_b - s_a) */
  double w_prog, w_comp;
  int32_t comp_type; /* DGSQP_COMP_* */
  int32_t _pad0;
  /* stage+terminal state costs (ablation script :229-230):
     1/2 w_block sum_b (ey_a-ey_b)^2 + 1/2 w_obs sum_b max(0, d_ab-|p_a-p_b|)^2 */
  double w_block, w_obs, obs_cost_r;

  /* constraints */
  int32_t has_rate; /* 4 rate rows per stage, order [u0 ub, u0 lb, u1 ub, u1 lb] */
  int32_t _pad1;
  double rate_ub[DGSQP_NUA], rate_lb[DGSQP_NUA]; /* per second; multiplied by dt inside */
  double in_ub[DGSQP_NUA], in_lb[DGSQP_NUA];     /* +-inf = absent (DGSQP.py:145-148) */
  double st_ub[DGSQP_MAX_NQA], st_lb[DGSQP_MAX_NQA];
  double radius; /* obstacle row for pair (i,j): (r_i+r_j)^2 - |p_i-p_j|^2 <= 0 */

  /* goal-tracking state cost (scripts/DGSQP_merge_monte_carlo.py:253-261): stage 1/2 sum_i w_goal[i] (q_i - goal[i])^2 at
     k = 0..N-1, goal_term_mult times that at k = N.  Weights may sit on the positions and the last two states. */
  double w_goal[DGSQP_MAX_NQA], goal[DGSQP_MAX_NQA];
  double goal_term_mult;
  /* lane half-planes (merge.py:66-74, :316-318), rows at every stage k = 0..N placed where the rate rows would be:
       n(p_x)^T (p - anchor) + r n^T n <= 0,   n(p_x) = n_lo + (n_hi - n_lo) [p_x >= brk]   (CasADi pw_const)        */
  int32_t n_lane;
  int32_t _pad2;
  struct {
    double brk;          /* +inf: constant normal n_lo */
    double n_lo[2], n_hi[2];
    double anchor[2];
    double r;
  } lane[DGSQP_MAX_LANES];
} dgsqp_agent_t;

/*
 * The game.  Track tables follow
 * DGSQP/tracks/radius_arclength_track.py:199-225 exactly:
 *   curvature(s) = pw_const(sbar, seg_s[1..n_segs-1], seg_curv[0..n_segs-1])
 *   tangent(s)   = pw_lin  (sbar, seg_s[0..n_segs],   seg_ang[0..n_segs])
 *   sbar = fmod(fmod(s, L) + L, L)
 */
typedef struct {
  int32_t M;          /* agents */
  int32_t N;          /* horizon */
  int32_t integrator; /* DGSQP_INT_* of the JOINT model (dynamics_models.py:2482-2530) */
  int32_t substeps;   /* RK sub-steps per dt ("M" in DynamicsConfig) */
  double dt;
  int32_t n_segs;
  int32_t obstacle_rows; /* 1: shared obstacle rows for every pair at k=1..N */
  double track_L;
  double seg_s[DGSQP_MAX_SEGS + 1];
  double seg_curv[DGSQP_MAX_SEGS];
  double seg_ang[DGSQP_MAX_SEGS + 1];
  dgsqp_agent_t agents[DGSQP_MAX_AGENTS];
  /* centre line of the Frenet-frame models.  DGSQP_TRACK_ARCS: the segment tables above (RadiusArclengthTrack: curvature
     piecewise constant, tangent piecewise linear, radius_arclength_track.py:199-225).  DGSQP_TRACK_SPLINE: cubic-spline
     interpolants x(s), y(s) (CasadiBSplineTrack, casadi_bspline_track.py:56-71): curvature (x'y'' - y'x'')/(x'^2 + y'^2)^1.5
     (:122-135), tangent atan2(y', x') (:137-149), s wrapped into [0, track_L).  spline = [n_knots] knots (first one 0), then
     [n_knots-1][4] coefficients of x and [n_knots-1][4] of y in ascending powers of (s - knot_i); caller-owned, copied by
     dgsqp_create (n_knots <= DGSQP_MAX_KNOTS). */
  int32_t track_kind;
  int32_t n_knots;
  const double* spline;
} dgsqp_problem_t;

#define DGSQP_TRACK_ARCS 0
#define DGSQP_TRACK_SPLINE 1
#define DGSQP_MAX_KNOTS 1152
#define DGSQP_VARIANT_V1 0
#define DGSQP_VARIANT_V2 1
#define DGSQP_QP_ACTIVE_SET 0
#define DGSQP_QP_OSQP 1
#define DGSQP_DECREASE_ARMIJO 0
#define DGSQP_DECREASE_MAX 1
/* DGSQPParams (solver_types.py:91-127), numeric subset used by solve() */
typedef struct {
  double beta, tau, p_tol, d_tol, reg;
  int32_t line_search_iters;
  int32_t nonmono_ls;
  int32_t sqp_iters;
  int32_t merit_function; /* DGSQP_MERIT_* */
  int32_t rel_tol_req;    /* 3 (DGSQP.py:56) */
  int32_t lsqr_iter_lim;  /* 0 => 2*n_c (scipy default, DGSQP.py:324) */
  double lsqr_atol, lsqr_btol; /* scipy 1.15 defaults 1e-6 */
  int32_t qp_warm_start;  /* implementation knob (not in DGSQPParams): 1 = start each QP's active-set search from the
                             previous QP's final active set (same minimiser, shorter path); 0 = cold start */
  int32_t hessian_bfgs;   /* DGSQPParams.hessian_approximation: 0 'none' (exact game Hessian every iteration), 1 'bfgs' (damped BFGS
                             update of the projected Hessian after the first iteration, DGSQP.py:353-364, :535-557) */
  double eig_floor;       /* value _nearestPD gives to the negative eigenvalues; the reference uses 1e-10 (DGSQP.py:1293).
                             <= 0 selects 1e-10, which is also what the host mirror passes by default.  With reg = 0 that
                             leaves a Hessian with condition 1e12 (solved by the classical J = L^-T active-set kernels);
                             a larger floor (e.g. 1e-6) is an explicit opt-in that keeps such games on the faster
                             explicit-inverse kernels. */
  double time_limit;      /* DGSQPParams.time_limit: wall-clock seconds per solve() after which the scenario ends with DGSQP_TIME_LIMIT
                             (checked at the end of every SQP iteration, DGSQP.py:470, and inside the watchdog's relaxed steps,
                             :1243-1247); < 0: none (the reference's None -> inf, DGSQP.py:64-67); 0 ends the solve after its
                             first iteration, as in the reference.  Counted from the moment a workgroup starts the scenario
                             (before the dual start, DGSQP.py:304), not from the launch of the batch. */
  int32_t snap_active_bounds; /* implementation knob, default 0 = literal.  1: after each QP put du exactly on the input bounds
                             whose multiplier is positive.  An exact QP solver leaves them at +-1 ulp, OSQP's polish at ~1e-12
                             (sign random); the rows are linear, the next iterate inherits that residual and _get_mu switches
                             on the sign of sum(g - s) with threshold 0 (DGSQP.py:559-585) -- a coin flip in the reference
                             itself; 1 makes it deterministic (mu = 0 when nothing else is violated). */
  int32_t variant;        /* DGSQP_VARIANT_V1: DGSQP/solvers/DGSQP.py (the fields above).  DGSQP_VARIANT_V2: DGSQP/solvers/DGSQP_v2.py
                             :322-720 -- d-step / m-step non-monotone strategy with checkpoints, decaying regularisation, merit
                             memory; p_tol/d_tol, reg (= reg_init), tau, line_search_iters, sqp_iters (counts m-steps), rel_tol_req
                             (10), time_limit and the LSQR / QP knobs above keep their meaning; beta and nonmono_ls are unused */
  /* DGSQPV2Params (solver_types.py:130-175) */
  int32_t nms;                       /* non-monotone strategy on (d-steps / m-steps) */
  int32_t nms_frequency;             /* m-step at the latest after this many d-steps (nms_mstep_frequency) */
  int32_t nms_memory_size;           /* length of the merit memory (<= 16) */
  int32_t merit_decrease_condition;  /* DGSQP_DECREASE_ARMIJO / DGSQP_DECREASE_MAX (DGSQP_v2.py:731-737) */
  int32_t qp_method;                 /* how _solve_qp (DGSQP.py:232-266) is computed.  DGSQP_QP_ACTIVE_SET (0, default): the exact KKT point of
                                        the strictly convex QP -- dual active-set method + polish -- i.e. what OSQP(polish=True) returns when its
                                        polish succeeds.  DGSQP_QP_OSQP (1): OSQP's own arithmetic as the reference runs it through CasADi's
                                        conic plugin (DGSQP.py:183-201, :246-249): Ruiz equilibration, ADMM to eps 1e-3, adaptive rho, polish
                                        accepted on residuals alone -- the iterate the reference's loop actually continues from (1e-3 .. 1e-6
                                        off the exact KKT point, occasionally negative multipliers); every size up to 320 unknowns.  A primal /
                                        dual infeasible QP ends the solve with DGSQP_QP_FAIL wherever it occurs (the reference's NaN step raises
                                        in _get_mu one iteration later, DGSQP.py:566-585) */
  int32_t osqp_rho_carry;            /* qp_method = DGSQP_QP_OSQP only.  0 (default): every OSQP call starts at rho = 0.1, as the CPU restatements
                                        the parity tests compare with do.  1: a call starts from the rho the previous call of the same solve()
                                        ended with -- inside CasADi's conic plugin the OSQP workspace persists and keeps its adapted rho
                                        (SURVEY.md parity hazard 7); the first call of a solve starts at 0.1 */
  int32_t mixed_precision;           /* 0 (default): fp64 storage and arithmetic throughout.  1: qp_method = DGSQP_QP_OSQP on the XL layout
                                        (n > 128: BASELINE configs[2], [3], [4], which name fp32) keeps the explicit K^-1 of the ADMM iteration in
                                        fp32 -- fp64 accumulation; the iteration is memory-bound and K^-1 is 60 % of the bytes it streams --;
                                        factorisations, residual checks, the polish and every other kernel stay fp64 -- and so does K^-1 when
                                        reg < 1e-4 (measured: at reg = 0 the flat directions of the projected Hessian drown in the rounding and
                                        the six-car merge of configs[4] loses half its converged solves, so the switch leaves that game in fp64).
                                        At reg = 1e-3 a QP whose polish succeeds in both returns the fp64 kernel's point to 1e-6 (the polish is
                                        fp64); ADMM iteration counts are no longer bit-comparable with the fp64 restatements, and along a whole
                                        solve only the solvable three-car game keeps its converged set (20 vs 21 of 24) -- on the chaotic F1 game
                                        single paths differ and only the statistics agree (256 scenarios: 44.9 % vs 47.3 % converged) */
  int32_t reserved_;
  double reg_decay;                  /* reg <- reg * reg_decay after every m-step / line-search step */
  double delta_decay;                /* gamma: d-step radius decay */
  double merit_decrease;             /* sigma */
  double merit_parameter;            /* mu; < 0: adaptive (DGSQPV2Params.merit_parameter = None, _get_mu DGSQP_v2.py:665-690) */
} dgsqp_params_t;

/* PID lane follower used for the Monte-Carlo warm start (DGSQP/solvers/PID.py through
   scripts/DGSQP_ALGAMES_monte_carlo_chicane.py:411-447) */
typedef struct {
  double kp_v;        /* speed P gain (1) */
  double kp_s, ki_s;  /* steering PI gains (1, 0.005) on  ey_gain (e_y - e_y0) + e_psi */
  double ey_gain;     /* 5 */
  double ei_max;      /* integrator clamp (100) */
  double u_max[2];    /* |u_a|, |u_steer| saturation */
  double du_max[2];   /* per-step change saturation, applied before the magnitude saturation */
  int32_t substeps;   /* rk4 sub-steps of the plant per dt (10) */
  int32_t reserved_;
} dgsqp_pid_t;

typedef struct {
  int32_t M, N, n_q, n_u, n, n_c; /* n = N*n_u decision vars, n_c inequality rows */
  int32_t n_dense;                /* distinct dense constraint gradients */
  int32_t lds_bytes;              /* dynamic LDS per scenario workgroup */
  int64_t workspace_bytes;        /* HBM scratch per resident workgroup */
  int32_t layout;                 /* 0 LDS-resident, 1 big (P and reflectors in the L2 scratch), 2 XL (n > 128, generic kernels) */
  int32_t reserved_;
} dgsqp_dims_t;

typedef struct {
  double h2d_ms, kernel_ms, d2h_ms, total_ms; /* HIP-event times of the last call */
  int32_t grid, block;
} dgsqp_timing_t;

/* Fixed 88-byte per-scenario record of the ONE collective of a sharded Monte-Carlo batch: what the
   convergence statistics need (process_data_curve.py:44-53).  cost holds every agent's cost (DGSQP_MAX_AGENTS slots). */
typedef struct {
  int32_t status, iters, qp_solves, rank;
  double p_feas, comp, stat;
  double cost[DGSQP_MAX_AGENTS];   /* f_J of every agent (DGSQP.py:492); zeros beyond M */
} dgsqp_stat_record_t;

typedef struct dgsqp_solver* dgsqp_handle_t;

/* Build a solver for one game on one HIP device (one process per GPU).
   Replaces DGSQP.__init__/_build_solver (DGSQP.py:26-230, :587-1030). */
int dgsqp_create(const dgsqp_problem_t* prob, const dgsqp_params_t* par, int device,
                 dgsqp_handle_t* out);
void dgsqp_destroy(dgsqp_handle_t h);
int dgsqp_dims(dgsqp_handle_t h, dgsqp_dims_t* out);
/* What dgsqp_create would build for this game -- dimensions, LDS / scratch need and the layout chosen -- without touching a
   device (host only).  Returns DGSQP_E_TOO_LARGE / DGSQP_E_ARG with the reason in msg when the game is not supported. */
int dgsqp_plan(const dgsqp_problem_t* prob, const dgsqp_params_t* par, dgsqp_dims_t* out, char* msg, int msglen);
const char* dgsqp_last_error(dgsqp_handle_t h); /* h may be NULL: last create() error */
int dgsqp_backend_info(char* buf, int buflen);  /* device name / arch / CU count */

/*
 * Solve B independent scenarios == B calls of DGSQP.solve() (DGSQP.py:302-507).
 * All arrays are caller-owned, C-contiguous host memory.
 *   x0     [B][n_q]          joint initial state (state2q, dynamics_models.py:2554-2561)
 *   u_ws   [B][n]            agent-major warm start (set_warm_start, DGSQP.py:271-281)
 *   u_out  [B][n]            agent-major solution
 *   l_out  [B][n_c]          multipliers (l_pred)
 *   x_out  [B][(N+1)*n_q]    q_pred
 *   status [B] DGSQP_* code, iters [B] num_iters, qp_solves [B]
 *   cond   [B][3]            p_feas, comp, stat of the last iteration (DGSQP.py:376-379)
 *   cost   [B][M]            f_J at the solution (DGSQP.py:492)
 * Any output pointer may be NULL.
 */
int dgsqp_solve_batch(dgsqp_handle_t h, int64_t B, const double* x0, const double* u_ws,
                      double* u_out, double* l_out, double* x_out, int32_t* status,
                      int32_t* iters, int32_t* qp_solves, double* cond, double* cost,
                      dgsqp_timing_t* timing);

/* The same call with single-precision arrays at the boundary ("double / float selected by dtype"): x0, u_ws
   are widened to fp64 on the device, the solve runs in fp64 -- there are no fp32 arithmetic kernels -- and u, l, x, cond,
   cost are rounded to fp32 on the way out.  Halves the HBM / PCIe bytes of the batch, changes nothing else: on inputs that are
   exactly representable in fp32 the integer outputs equal those of dgsqp_solve_batch. */
int dgsqp_solve_batch_f32(dgsqp_handle_t h, int64_t B, const float* x0, const float* u_ws,
                          float* u_out, float* l_out, float* x_out, int32_t* status,
                          int32_t* iters, int32_t* qp_solves, float* cond, float* cost,
                          dgsqp_timing_t* timing);

/*
 * Device-resident variant used by bench.py: inputs are staged once with
 * dgsqp_stage_inputs(); dgsqp_solve_staged() runs only the solve kernel on
 * the handle's stream; dgsqp_fetch_results() copies results back.
 */
int dgsqp_stage_inputs(dgsqp_handle_t h, int64_t B, const double* x0, const double* u_ws);
int dgsqp_solve_staged(dgsqp_handle_t h, dgsqp_timing_t* timing);
/* Asynchronous halves of dgsqp_solve_staged(): enqueue the solve on the handle's own stream / wait for it.  Independent
   batches held by different handles of the SAME game (same dgsqp_problem_t and dgsqp_params_t) can be in flight together:
   the workgroups of the later launch take over the compute units as the earlier launch drains its slowest scenarios.
   Handles of DIFFERENT games are safe too, but serialised: such a launch first waits for the launches in flight on the
   device (the kernels read the game from one per-device constant block). */
int dgsqp_launch_staged(dgsqp_handle_t h);
int dgsqp_wait(dgsqp_handle_t h, dgsqp_timing_t* timing);
/* ONE launch over the staged batches of `count` handles (same game, same batch size, same device; at most 64): the scenarios of all
   of them share one ticket queue, every batch keeps its own input and output buffers, results are bit-identical to separate
   launches.  A launch ends with its slowest scenario, so few long launches keep the compute units busier than many short ones; the
   number of launches in flight is bounded by the hardware queues (16 here).  hs[0] leads: its stream, workspace and events are
   used; every handle of the group is in flight until ITS dgsqp_wait / dgsqp_fetch_results (which wait for the group's kernel).
   Event and iterate logs are not available in grouped launches. */
int dgsqp_launch_staged_group(const dgsqp_handle_t* hs, int count);
/* Cooperative line search.  A launch ends with its slowest scenario (dozens of failing 50-trial line searches) while most
   workgroups have long run out of scenarios.  In a cooperative launch those workgroups stay and evaluate line-search trial points
   for the ones still solving (the very same device code: results are bit-identical to a non-cooperative launch); they keep their
   compute units until the launch's last scenario is done.  mode 0: never; 1 (default): in the synchronous calls only
   (dgsqp_solve_batch, dgsqp_solve_staged -- nothing else is waiting for the compute units); 2: every launch of this handle, also
   the asynchronous ones -- for the LAST launch of a pipeline.  The leader's setting governs a grouped launch. */
int dgsqp_set_cooperative(dgsqp_handle_t h, int mode);
/* Diagnostic: counters of the handle's last cooperative launch -- out6 = {trials evaluated by helpers, times a workgroup entered
   the helper loop (at most coop_helpers of them evaluate trials at any moment, the others sleep), scenarios finished, helpers still registered (0 after the launch), helper values the owners consumed, helper values
   whose bits differed from the owner's own evaluation (verify mode, environment DGSQP_COOP_VERIFY=1; must be 0)}. */
int dgsqp_coop_stats(dgsqp_handle_t h, uint64_t* out6);
/* qp_method = DGSQP_QP_OSQP: {QP calls, ADMM iterations} of every solve on h's DEVICE since the last reset (waits for h's launches; callers
   that want a clean count let the device's other handles finish first).  out2 may be NULL (reset only).  Nothing in the reference: OSQP
   reports `info.iter` per call; this is the sum bench.py divides to price the ADMM work it measures (mean iterations per QP). */
int dgsqp_osqp_counters(dgsqp_handle_t h, uint64_t* out2, int reset);
/* Deferral of long scenarios (cooperative launches; scheduling only, results are bit-identical).  Nothing in a scenario's inputs
   tells how long its solve will run; its own history does.  A scenario still iterating after max(min_iters, factor x the mean
   iteration count of the launch's finished scenarios) SQP iterations is set aside while fresh scenarios remain in the queue -- its
   workgroup stores the LDS arena and its scratch in a slot and takes the next ticket -- and resumed from that image once the queue
   is empty, the scenarios that have cost the most so far first.  The long solves of a launch's last batches thereby run while the
   chip still has other work.  Defaults: min_iters 8, factor 2.0; min_iters 0 switches it off.  Never applied while logs are recorded or in
   non-cooperative launches; DG-SQP v1 solves with a wall-clock limit (dgsqp_params_t.time_limit >= 0) are never deferred either.
   DG-SQP v2 solves can be deferred too, but only after an explicit dgsqp_set_deferral (measured: it does not pay for v2's length
   distribution); its study always sets a limit (600 s): the time a scenario spends set aside does not count towards it.  The slots
   (LDS image + scratch image per scenario: 0.65 MB for the 2-agent N = 25 games, up to 4 MB for the XL layouts) are one pool per
   device: sized for what the launch may defer (a quarter of its scenarios; 257 slots = 0.17 GB for a 1,024-scenario batch, 3.4 GB for
   a group of 20), grown geometrically by later, larger launches, never beyond DGSQP_DEFER_POOL_BYTES (environment; default 16 GiB,
   0 = no deferral), kept until the process's last handle is destroyed; a cooperative launch that finds the pool in use by another
   launch in flight simply does not defer.  Nothing is
   deferred before 32 scenarios of the launch have finished nor when fewer than two rounds of fresh scenarios remain. */
int dgsqp_set_deferral(dgsqp_handle_t h, int32_t min_iters, double factor);
/* Sizes the device's deferral pool now for a cooperative launch of `scenarios` scenarios of h's game (what that launch would do itself
   on entry: a hipFree + hipMalloc of up to several GB when the pool has to grow -- 1 to 100 ms that a caller may not want inside a
   timed or latency-critical region).  Nothing in the reference. */
int dgsqp_reserve_deferral(dgsqp_handle_t h, int64_t scenarios);
/* Diagnostic: out2 = {scenarios deferred, scenarios resumed} of the handle's last cooperative launch (equal after the launch). */
int dgsqp_deferral_stats(dgsqp_handle_t h, uint64_t* out2);
/* Diagnostic: one row of 11 values per deferred scenario of the handle's last cooperative launch, at most cap_rows rows --
   {ticket, SQP iterations and QP solves when it was set aside, 100 MHz ticks it had cost by then (the resume order's key), ticks since
   the launch's start when it was set aside / resumed / finished, (QP solves << 32) | iterations at the end, and -- as the bits of three doubles -- the convergence measures (constraint violation,
   complementarity, stationarity) of its last iteration before it was set aside}.  Returns the number of
   rows written, negative on error. */
int dgsqp_deferral_log(dgsqp_handle_t h, uint64_t* out, int64_t cap_rows);
/* 1 once the handle's last launch has handed out its last scenario (it only drains from then on, compute units are
   becoming free) or when nothing is in flight; 0 while scenarios are still queued.  Polled by bench.py to start the next
   independent batch on another handle at exactly that moment. */
int dgsqp_draining(dgsqp_handle_t h);
/* 1 once the launch of dgsqp_launch_staged has completed (or nothing is in flight), 0 while it runs; never blocks.  A batch ends
   with its slowest scenario: a caller that keeps several launches in flight retires whichever has finished, not the oldest. */
int dgsqp_finished(dgsqp_handle_t h);
int dgsqp_fetch_results(dgsqp_handle_t h, double* u_out, double* l_out, double* x_out,
                        int32_t* status, int32_t* iters, int32_t* qp_solves, double* cond,
                        double* cost);

/*
 * Test hook == one DGSQP._evaluate(u, l, x0, up=0, hessian=True) per scenario
 * (DGSQP.py:509-533) plus the dual initialisation of DGSQP.py:320-327.
 *   l   [B][n_c] multipliers used for Q (may be NULL => zeros)
 *   q   [B][n], g [B][n_c], G [B][n_c][n] dense row-major, Q [B][n][n] (raw,
 *   unsymmetrised), x [B][(N+1)*n_q], l0 [B][n_c] = max(0,-lsqr(GG^T, Gq)).
 */
int dgsqp_evaluate_batch(dgsqp_handle_t h, int64_t B, const double* x0, const double* u,
                         const double* l, double* q, double* g, double* G, double* Q,
                         double* x, double* l0);

/*
 * Test hook == DGSQP._solve_qp(Q, q, G, g) (DGSQP.py:232-266) on the game's
 * own constraint structure: evaluates at (u, l), PSD-projects Q (+reg) and
 * solves the QP.  du [B][n], lhat [B][n_c], Qpd [B][n][n] (projected +
 * regularised Hessian), flag [B] (0 ok, 1 infeasible / failed).
 */
int dgsqp_qp_batch(dgsqp_handle_t h, int64_t B, const double* x0, const double* u,
                   const double* l, double* du, double* lhat, double* Qpd, int32_t* flag);
/* The same hook with OSQP's diagnostics when the handle runs qp_method = DGSQP_QP_OSQP (zeros otherwise): info8 [B][8] = {OSQP status
   (1 solved, 2 solved inaccurate, -2 iteration limit, -3 primal infeasible, -4 dual infeasible, -10 non-finite data), ADMM iterations,
   polish (1 accepted, -1 rejected, 0 not attempted), final rho, rho updates, active rows handed to the polish, primal and dual residual
   of the ADMM iterate} -- what tests compare with the CPU restatements of OSQP (oracle/osqp.hpp, oracle/osqp_restate.py). */
int dgsqp_qp_batch_info(dgsqp_handle_t h, int64_t B, const double* x0, const double* u,
                        const double* l, double* du, double* lhat, double* Qpd, int32_t* flag, double* info8);

/* Warm start of a Monte-Carlo batch (row (f) of the hot-path scope): for every scenario and agent roll the PID lane follower
   out from q0[B][n_q] over the horizon with the agent's own continuous model (rk4).  u_ws[B][n] agent-major (what
   dgsqp_solve_batch takes); q_ws[B][(N+1) n_q] and collide[B] (1 = some pair closer than r_i + r_j at some stage,
   check_collision chicane.py:38-43) are optional.  Replaces the per-sample loop chicane.py:411-447 / curve.py:440-467. */
int dgsqp_pid_warm_start_batch(dgsqp_handle_t h, int64_t B, const double* q0, const dgsqp_pid_t* pid, double* u_ws,
                               double* q_ws, int32_t* collide);

/*
 * Rejection samplers of the Monte-Carlo scripts on the device (row (f) of the hot-path scope): random placement, PID warm start
 * (zero inputs for the merge), collision check along the warm start, accepted candidates kept in candidate order.  The random
 * numbers are counter-based (Philox4x32-10): uniform k of candidate c depends on (seed, c, k) only, so the host mirror
 * (dgsqp_amd/sampler.py) reproduces the stream bit for bit whatever the round sizes.
 *   DGSQP_SAMPLER_FIRST_SEGMENT  scripts/DGSQP_ALGAMES_monte_carlo_chicane.py:384-404, _curve.py (two cars)
 *   DGSQP_SAMPLER_INDEPENDENT    scripts/DGSQP_monte_carlo_agents.py:262-308 (M cars)
 *   DGSQP_SAMPLER_CIRCUIT        scripts/DGSQP_comp_monte_carlo.py:365-382 (M cars on a closed track)
 *   DGSQP_SAMPLER_MERGE          scripts/DGSQP_merge_monte_carlo.py:429-473 (unicycles, zero warm start)
 * key_pts: the arc track's key points (x, y, psi, cumulative length, segment length, signed curvature of the segment ENDING
 * there), n_key = segments + 1 (radius_arclength_track.py:361-408).
 */
enum { DGSQP_SAMPLER_FIRST_SEGMENT = 0, DGSQP_SAMPLER_INDEPENDENT = 1, DGSQP_SAMPLER_CIRCUIT = 2, DGSQP_SAMPLER_MERGE = 3 };
typedef struct {
  int32_t kind;
  int32_t n_key;
  uint64_t seed;
  double half_width;   /* placement range of e_y */
  double obs_d;        /* obstacle distance of the relative placements */
  double seg0_len;     /* length of the first track segment */
  double x_nom[DGSQP_MAX_AGENTS];   /* merge: nominal x of every car (merge.py:430,443,456) */
  double key_pts[DGSQP_MAX_SEGS + 1][6];
} dgsqp_sampler_t;
/* Draw until B scenarios are accepted: x0_out [B][n_q], u_ws_out [B][n] agent-major (host, either may be NULL).  stage != 0: the
   batch also becomes the handle's staged input (as after dgsqp_stage_inputs) -- sampling, solve and statistics never leave the
   device.  *candidates (may be NULL) = candidates consumed, i.e. the index after the last accepted one. */
int dgsqp_sample_batch(dgsqp_handle_t h, int64_t B, const dgsqp_sampler_t* spec, const dgsqp_pid_t* pid, double* x0_out,
                       double* u_ws_out, int64_t* candidates, int stage);

/*
 * Test hook: event log of the SQP state machine (convergence measures, merit values, step lengths,
 * watchdog branches) of every scenario of the next solve calls; compared event-by-event with the
 * oracle's log.  Layout per scenario: [count, (code, value) x pairs_per_scenario].  0 disables.
 */
int dgsqp_set_trace(dgsqp_handle_t h, int pairs_per_scenario);
/* out: [B_launch][1 + 2 * pairs_per_scenario] of the last launch; capacity_doubles = what out can hold (DGSQP_E_ARG when it
   is too small).  A count above pairs_per_scenario means that scenario's log was truncated. */
int dgsqp_fetch_trace(dgsqp_handle_t h, double* out, int64_t capacity_doubles);

/* Iterate log for solve()'s iter_data / init (DGSQP.py:328, :386-451): per scenario [count, records x (n + n_c)], record 0 =
   (u_ws, dual start l), record i = (u, l) at the end of SQP iteration i (or at the exit test that ended the solve).
   0 disables (default: Monte-Carlo batches only keep the final iterates). */
int dgsqp_set_iterate_log(dgsqp_handle_t h, int records_per_scenario);
int dgsqp_fetch_iterate_log(dgsqp_handle_t h, double* out, int64_t capacity_doubles);

/* Wait for everything enqueued on the handle's stream (what a caller without a HIP runtime of its own uses as a fence). */
int dgsqp_synchronize(dgsqp_handle_t h);


/*
 * Multi-GPU (one process per GPU, scenarios sharded, no data-path collective): the library owns the RCCL communicator.
 *   dgsqp_comm_unique_id   rank 0: 128-byte ncclUniqueId to hand to the other ranks (file, socket, environment ...)
 *   dgsqp_comm_init        every rank: ncclCommInitRank on the handle's device (collective call)
 *   dgsqp_gather_stats     the ONE collective per batch: ncclAllGather over xGMI of the 88-byte records of the handle's last solve.
 *                          B_pad = the largest shard (shards may differ by one scenario); out[world * B_pad] in rank order, padding
 *                          rows carry status -1.  Every rank receives all records.
 *   dgsqp_comm_barrier, dgsqp_comm_allreduce_max   fences / max-over-ranks of small host vectors for benchmark timing
 * world = 1 works without any peer (and is what the single-GPU test exercises).
 */
int dgsqp_comm_unique_id(char* out128);
int dgsqp_comm_init(dgsqp_handle_t h, const char* id128, int rank, int world);
int dgsqp_comm_destroy(dgsqp_handle_t h);
int dgsqp_gather_stats(dgsqp_handle_t h, int64_t B_pad, dgsqp_stat_record_t* out);
int dgsqp_comm_barrier(dgsqp_handle_t h);
int dgsqp_comm_allreduce_max(dgsqp_handle_t h, double* values, int count);

#ifdef __cplusplus
}
#endif
#endif /* DGSQP_H */
