"""Host-side mirror of the reference solver class for the Monte-Carlo hot path.

``DGSQP`` keeps the reference surface (DGSQP/solvers/DGSQP.py:25-34 constructor,
:268 ``initialize``, :271-281 ``set_warm_start``, :283-297 ``step``, :299
``get_prediction``, :302-507 ``solve`` and its ``solve_info`` keys) and adds
``solve_batch`` -- B independent ``solve()`` calls executed by the HIP kernels
through the C-ABI of include/dgsqp.h.  All numerics run on the GPU; this file
only lowers the game description to PODs and reshapes results.
"""
from __future__ import annotations

import copy
import ctypes as C
import time
import warnings
from typing import Dict, List, Optional

import numpy as np

from . import _ffi
from .dynamics import CasadiDecoupledMultiAgentDynamicsModel, INTEGRATORS
from .game import CollisionAvoidance, GoalTrackingCost, InputRateLimits, LaneBoundaries, RacingCost
from .solver_types import DGSQPParams, DGSQPV2Params
from .types import VehiclePrediction, VehicleState


class AbstractSolver:
    """Reference DGSQP/solvers/abstract_solver.py:9-48 (runtime interface only)."""
    needs_env_state = False

    def initialize(self):
        pass

    def solve(self):
        raise NotImplementedError

    def step(self, estimated_state, env_state=None):
        raise NotImplementedError

    def get_prediction(self):
        return VehiclePrediction()


# ---------------------------------------------------------------------------------------------
# lowering of the game to the C PODs (pure host logic, no GPU needed)
# ---------------------------------------------------------------------------------------------
def build_problem(joint_dynamics: CasadiDecoupledMultiAgentDynamicsModel,
                  costs: List[RacingCost],
                  agent_constraints: List[Optional[InputRateLimits]],
                  shared_constraints: Optional[CollisionAvoidance],
                  bounds: Dict[str, List[VehicleState]],
                  params: DGSQPParams) -> _ffi.ProblemT:
    M = joint_dynamics.n_a
    if len(costs) != M:
        raise ValueError('Number of agents: %i, but %i cost functions were provided' % (M, len(costs)))
    if M > _ffi.MAX_AGENTS:
        raise ValueError(f'at most {_ffi.MAX_AGENTS} agents are supported')
    P = _ffi.ProblemT()
    P.M, P.N = M, int(params.N)
    cfg = joint_dynamics.model_config
    P.integrator = INTEGRATORS[cfg.discretization_method]
    P.substeps = int(cfg.M) if cfg.discretization_method != 'euler' else 1
    P.dt = float(cfg.dt)
    track = joint_dynamics.track
    if track is None:                       # global-frame models (unicycle): a one-segment placeholder table
        L, seg_s, seg_curv, seg_ang = 1.0, [0.0, 1.0], [0.0], [0.0, 0.0]
    elif getattr(track, 'kind', 'arcs') == 'spline':      # CasadiBSplineTrack (F1): cubic-spline table, see include/dgsqp.h
        table = track.spline_table()
        if len(track.knots) > _ffi.MAX_KNOTS:
            raise ValueError(f'spline track has {len(track.knots)} knots, limit is {_ffi.MAX_KNOTS}')
        P.track_kind, P.n_knots = 1, len(track.knots)
        P._spline_keepalive = table                       # the POD only borrows the host buffer
        P.spline = table.ctypes.data
        L, seg_s, seg_curv, seg_ang = track.track_length, [0.0, track.track_length], [0.0], [0.0, 0.0]
    else:
        L, seg_s, seg_curv, seg_ang = track.tables()
    n_segs = len(seg_curv)
    if n_segs > _ffi.MAX_SEGS:
        raise ValueError(f'track has {n_segs} segments, limit is {_ffi.MAX_SEGS}')
    P.n_segs, P.track_L = n_segs, float(L)
    for i in range(n_segs + 1):
        P.seg_s[i] = float(seg_s[i])
        P.seg_ang[i] = float(seg_ang[i])
    for i in range(n_segs):
        P.seg_curv[i] = float(seg_curv[i])
    P.obstacle_rows = 1 if shared_constraints is not None else 0
    if shared_constraints is not None and len(shared_constraints.radii) != M:
        raise ValueError('CollisionAvoidance.radii must have one entry per agent')
    for a, mdl in enumerate(joint_dynamics.dynamics_models):
        A = P.agents[a]
        c = mdl.model_config
        A.model = mdl.model_id
        if mdl.model_id != 2:
            A.L_f, A.L_r, A.mass = c.wheel_dist_front, c.wheel_dist_rear, c.mass
            A.c_dr, A.c_da = c.drag_coefficient, c.damping_coefficient
            A.c_r, A.p_r = c.rolling_resistance, c.rolling_resistance_exponent
        if mdl.model_id == 2:
            A.L_f = A.L_r = 0.13
            A.mass, A.I_z, A.gravity = float(mdl.m), 1.0, 9.81
            A.c_dr = A.c_da = A.c_r = 0.0
            A.p_r = 0.5
        elif mdl.model_id == 0:
            A.c_s = c.slip_coefficient
            A.I_z, A.gravity = 1.0, 9.81
        else:
            A.I_z, A.gravity = c.yaw_inertia, c.gravity
            A.tire_model = 0 if c.tire_model == 'pacejka' else 1
            A.drive_wheels = 0 if c.drive_wheels == 'all' else 1
            A.simple_slip = int(bool(c.simple_slip))
            A.pac_Bf, A.pac_Br, A.pac_Cf, A.pac_Cr = c.pacejka_b_front, c.pacejka_b_rear, c.pacejka_c_front, c.pacejka_c_rear
            A.pac_Df, A.pac_Dr, A.lin_Bf, A.lin_Br = c.pacejka_d_front, c.pacejka_d_rear, c.linear_bf, c.linear_br
        cost = costs[a]
        for j in range(2):
            A.w_in[j] = float(cost.input_weight[j])
            A.w_rate[j] = float(cost.input_rate_weight[j])
        if isinstance(cost, GoalTrackingCost):
            if len(cost.state_weight) != mdl.n_q or len(cost.goal) != mdl.n_q:
                raise ValueError('GoalTrackingCost needs one weight and one goal entry per state')
            for i in range(mdl.n_q):
                if cost.state_weight[i] != 0 and i not in (0, 1, mdl.n_q - 2, mdl.n_q - 1):
                    raise NotImplementedError('goal-tracking weights are supported on the positions and the last two states')
                A.w_goal[i], A.goal[i] = float(cost.state_weight[i]), float(cost.goal[i])
            A.goal_term_mult = float(cost.terminal_multiplier)
        else:
            A.w_prog, A.w_comp = float(cost.comp_weights[0]), float(cost.comp_weights[1])
            A.comp_type = {'atan': 0, 'linear': 1}[cost.comp_type]
            A.w_block, A.w_obs, A.obs_cost_r = float(cost.blocking_weight), float(cost.obs_weight), float(cost.obs_r)
        rate = agent_constraints[a] if agent_constraints is not None else None
        if isinstance(rate, LaneBoundaries):
            if len(rate.lanes) > _ffi.MAX_LANES:
                raise ValueError(f'at most {_ffi.MAX_LANES} lane rows per agent')
            A.n_lane = len(rate.lanes)
            for j, ln in enumerate(rate.lanes):
                hi = ln.n_lo if ln.n_hi is None else ln.n_hi
                A.lane[j].brk, A.lane[j].r = float(ln.brk), float(ln.r)
                for i in range(2):
                    A.lane[j].n_lo[i], A.lane[j].n_hi[i], A.lane[j].anchor[i] = float(ln.n_lo[i]), float(hi[i]), float(ln.anchor[i])
            rate = None
        A.has_rate = 0 if rate is None else 1
        for j in range(2):
            A.rate_ub[j] = 0.0 if rate is None else float(rate.rate_max[j])
            A.rate_lb[j] = 0.0 if rate is None else float(rate.rate_min[j])
        # box bounds -> index sets of finite entries (DGSQP.py:135-148)
        su, iu = mdl.state2qu(bounds['ub'][a])
        sl, il = mdl.state2qu(bounds['lb'][a])
        for j in range(2):
            A.in_ub[j], A.in_lb[j] = float(iu[j]), float(il[j])
        for i in range(_ffi.MAX_NQA):
            A.st_ub[i] = float(su[i]) if i < mdl.n_q else np.inf
            A.st_lb[i] = float(sl[i]) if i < mdl.n_q else -np.inf
        A.radius = float(shared_constraints.radii[a]) if shared_constraints is not None else 0.0
    return P


QP_METHODS = {'active_set': 0, 'osqp': 1}


_WARNED_OSQP_DEFAULT = False


def resolve_qp_method(params, qp_method: Optional[str]) -> int:
    """How ``_solve_qp`` (DGSQP.py:232-266) is computed on the device -- ``dgsqp_params_t.qp_method``.

    The reference hands the QP to ``ca.conic('qp', params.qp_solver, ...)`` (DGSQP.py:183-201): 'osqp' (its default, ADMM + polish:
    an approximate KKT point), or one of the exact solvers 'qrqp' / 'qpoases' / 'cplex' (active-set / simplex: THE KKT point).
    Here ``'active_set'`` is the exact dual active-set method, ``'osqp'`` the restatement of OSQP's own arithmetic
    (csrc/dgsqp_osqp.h).  ``qp_method=None`` picks by ``params.qp_solver``: the exact solvers map to 'active_set'; 'osqp' maps to
    'active_set' as well -- the KKT point OSQP's polish aims at, the faster kernels -- unless the caller opts into
    ``qp_method='osqp'`` to follow the reference's own iterates."""
    if params.qp_solver not in ('osqp', 'qrqp', 'qpoases', 'cplex'):         # whatever qp_method says: the reference would run another algorithm
        raise ValueError(f'Unsupported QP solver {params.qp_solver}')           # (superscs: a conic splitting solver, not restated)
    if qp_method is None:
        global _WARNED_OSQP_DEFAULT
        if params.qp_solver == 'osqp' and not _WARNED_OSQP_DEFAULT:
            _WARNED_OSQP_DEFAULT = True
            warnings.warn("DGSQPParams.qp_solver='osqp' is solved with the exact active-set QP (the KKT point OSQP's polish aims at); its iterates "
                          "differ from OSQP's by 1e-3..1e-6 per QP.  Pass qp_method='osqp' to DGSQP(...) for OSQP's own ADMM + polish arithmetic, "
                          "or qp_method='active_set' to silence this note.", stacklevel=3)
        return QP_METHODS['active_set']
    if qp_method not in QP_METHODS:
        raise ValueError(f'qp_method must be one of {sorted(QP_METHODS)}')
    return QP_METHODS[qp_method]


def build_params(params: DGSQPParams, eig_floor: Optional[float] = None, snap_active_bounds: bool = False,
                 lsqr_tol: Optional[float] = None, qp_warm_start: bool = True, qp_method: Optional[str] = None,
                 osqp_rho_carry: bool = False, mixed_precision: bool = False) -> _ffi.ParamsT:
    """``eig_floor``: value ``_nearestPD`` gives to negative eigenvalues; ``None`` = the reference's literal 1e-10
    (DGSQP.py:1293).  At ``reg = 0`` (curve.py, comp.py, merge.py) that leaves a QP of condition 1e12, which the device solves
    with its classical (J = L^-T) active-set kernels; passing a larger floor (1e-6) is an explicit opt-in that keeps such games
    on the faster explicit-inverse kernels.  ``snap_active_bounds``: see include/dgsqp.h (default literal).
    ``lsqr_tol``: atol = btol of the LSQR dual start; ``None`` = scipy's defaults (1e-6), what ``DGSQP.py:324`` runs with.
    ``osqp_rho_carry`` (``qp_method='osqp'`` only): an OSQP call starts from the rho the previous call of the same solve ended with, as
    inside CasADi's persistent conic plugin, instead of 0.1 (include/dgsqp.h; default off = the committed restatements).
    ``mixed_precision`` (``qp_method='osqp'`` on the XL layout only): the ADMM iteration's explicit K^-1 in fp32, everything else fp64."""
    if isinstance(params, DGSQPV2Params):
        p2 = _build_params_v2(params, eig_floor, snap_active_bounds, lsqr_tol, qp_warm_start, qp_method)
        p2.osqp_rho_carry = int(bool(osqp_rho_carry))
        p2.mixed_precision = int(bool(mixed_precision))
        return p2
    if not params.conv_approx:
        raise NotImplementedError('conv_approx=False (IPOPT Newton step, DGSQP.py:204-228) is not on the Monte-Carlo path')
    if params.hessian_approximation not in ('none', 'bfgs'):
        raise ValueError(f'Hessian approximation method {params.hessian_approximation} not implmented')   # (DGSQP.py:550)
    if params.merit_function not in ('stat_l1', 'stat'):
        raise ValueError(f'Merit function option {params.merit_function} not recognized')
    p = _ffi.ParamsT()
    p.beta, p.tau, p.p_tol, p.d_tol, p.reg = params.beta, params.tau, params.p_tol, params.d_tol, params.reg
    p.line_search_iters, p.nonmono_ls, p.sqp_iters = params.line_search_iters, int(params.nonmono_ls), params.sqp_iters
    p.merit_function = 0 if params.merit_function == 'stat_l1' else 1
    p.rel_tol_req = 3                      # DGSQP.py:56
    p.lsqr_iter_lim = 0                    # scipy default 2*n_c
    p.lsqr_atol = p.lsqr_btol = 1e-6 if lsqr_tol is None else float(lsqr_tol)      # scipy >= 1.12 defaults of sparse.linalg.lsqr
    p.qp_warm_start = int(bool(qp_warm_start))      # start each QP's active-set search from the previous QP's active set (same minimiser)
    p.hessian_bfgs = 1 if params.hessian_approximation == 'bfgs' else 0
    p.time_limit = -1.0 if params.time_limit is None else float(params.time_limit)     # None -> no limit (DGSQP.py:64-67)
    p.eig_floor = 1e-10 if eig_floor is None else float(eig_floor)
    p.snap_active_bounds = int(bool(snap_active_bounds))
    p.qp_method = resolve_qp_method(params, qp_method)
    p.osqp_rho_carry = int(bool(osqp_rho_carry))
    p.mixed_precision = int(bool(mixed_precision))
    return p


def _build_params_v2(params: DGSQPV2Params, eig_floor, snap_active_bounds, lsqr_tol, qp_warm_start, qp_method=None) -> _ffi.ParamsT:
    """DGSQPV2Params -> dgsqp_params_t for DG-SQP v2 (reference DGSQP/solvers/DGSQP_v2.py:66-222)."""
    if params.merit_function not in ('stat_l1', 'sum_obj_l1'):
        raise ValueError(f'Merit function option {params.merit_function} not recognized')      # (DGSQP_v2.py:1165-1166)
    if params.merit_decrease_condition not in ('armijo', 'max'):
        raise ValueError(f'merit_decrease_condition {params.merit_decrease_condition!r} not recognized')
    if params.hessian_approximation != 'none':
        raise ValueError(f'Hessian approximation method {params.hessian_approximation} not implmented')
    if not 1 <= int(params.nms_memory_size) <= 16:
        raise ValueError('nms_memory_size must be in 1..16')
    p = _ffi.ParamsT()
    p.variant = 1
    p.tau, p.p_tol, p.d_tol, p.reg = params.tau, params.p_tol, params.d_tol, params.reg
    p.beta = params.beta                   # (unused by v2's line search, which takes merit_decrease)
    p.line_search_iters, p.sqp_iters, p.nonmono_ls = params.line_search_iters, params.sqp_iters, 0
    p.merit_function = 0 if params.merit_function == 'stat_l1' else 2      # DGSQP_MERIT_STAT_L1 / DGSQP_MERIT_SUM_OBJ_L1
    p.rel_tol_req = 10                     # DGSQP_v2.py:84
    p.lsqr_iter_lim = 0
    p.lsqr_atol = p.lsqr_btol = 1e-6 if lsqr_tol is None else float(lsqr_tol)
    p.qp_warm_start = int(bool(qp_warm_start))
    p.hessian_bfgs = 0
    p.time_limit = -1.0 if params.time_limit is None else float(params.time_limit)
    p.eig_floor = 1e-10 if eig_floor is None else float(eig_floor)
    p.snap_active_bounds = int(bool(snap_active_bounds))
    p.nms, p.nms_frequency, p.nms_memory_size = int(bool(params.nms)), int(params.nms_frequency), int(params.nms_memory_size)
    p.merit_decrease_condition = 0 if params.merit_decrease_condition == 'armijo' else 1
    p.reg_decay, p.delta_decay, p.merit_decrease = float(params.reg_decay), float(params.delta_decay), float(params.merit_decrease)
    p.merit_parameter = -1.0 if params.merit_parameter is None else float(params.merit_parameter)
    p.qp_method = resolve_qp_method(params, qp_method)
    return p


def solve_batches(solvers, batches) -> list:
    """Several Monte-Carlo batches of the SAME game and size in ONE launch (``dgsqp_launch_staged_group``): ``solvers`` are DGSQP
    objects of that game (one per batch: every batch keeps its own device buffers), ``batches`` the matching ``(x0, u_ws)`` pairs as
    ``solve_batch`` takes them -- or the batch size alone for a batch that is staged on the device already (``sample_batch(stage=True)``:
    sampling, solve and results never pass through host arrays in between).  A launch ends with its slowest scenario; grouping lets the workgroups that are done with one batch
    go on with the next instead of idling behind that tail.  Returns one ``solve_batch``-style dictionary per batch, bit-identical
    to separate ``solve_batch`` calls."""
    if len(solvers) != len(batches) or not solvers:
        raise ValueError('one solver per batch')
    lib = solvers[0]._lib
    staged = []
    for s, bt in zip(solvers, batches):
        if isinstance(bt, (int, np.integer)):          # already staged on the device (DGSQP.sample_batch(..., stage=True)): just its size
            staged.append(int(bt))
            continue
        x0, u_ws = bt
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        u_ws = np.asarray(u_ws, dtype=np.float64)
        if u_ws.ndim == 3:
            u_ws = s._to_agent_major(u_ws)
        u_ws = np.ascontiguousarray(u_ws)
        B = x0.shape[0]
        if x0.shape != (B, s.n_q) or u_ws.shape != (B, s.n):
            raise RuntimeError(f'bad batch shapes x0 {x0.shape} u_ws {u_ws.shape}')
        if lib.dgsqp_stage_inputs(s._h, B, _ffi.dptr(x0), _ffi.dptr(u_ws)) != 0:
            raise RuntimeError('dgsqp_stage_inputs failed: ' + lib.dgsqp_last_error(s._h).decode())
        staged.append(B)
    t0 = time.time()
    arr = (C.c_void_p * len(solvers))(*[s._h for s in solvers])
    # the caller waits for this launch, so a leader in the default mode 1 ("synchronous calls only") runs it cooperatively: idle
    # workgroups help with its line searches.  An explicit set_cooperative(0) / (2) of the leader is honoured and left as it is.
    mode = getattr(solvers[0], '_coop_mode', 1)
    if mode == 1:
        lib.dgsqp_set_cooperative(solvers[0]._h, 2)
    rc = lib.dgsqp_launch_staged_group(arr, len(solvers))
    if mode == 1:
        lib.dgsqp_set_cooperative(solvers[0]._h, 1)
    if rc != 0:
        raise RuntimeError('dgsqp_launch_staged_group failed: ' + lib.dgsqp_last_error(solvers[0]._h).decode())
    tm = _ffi.TimingT()
    outs = []
    for s, B in zip(solvers, staged):
        if lib.dgsqp_wait(s._h, C.byref(tm)) != 0:
            raise RuntimeError('dgsqp_wait failed: ' + lib.dgsqp_last_error(s._h).decode())
        out = dict(u=np.empty((B, s.n)), l=np.empty((B, s.n_c_total)), x=np.empty((B, s.N + 1, s.n_q)), status=np.empty(B, np.int32),
                   num_iters=np.empty(B, np.int32), qp_solves=np.empty(B, np.int32), cond=np.empty((B, 3)), cost=np.empty((B, s.M)))
        rc = lib.dgsqp_fetch_results(s._h, _ffi.dptr(out['u']), _ffi.dptr(out['l']), _ffi.dptr(out['x']), _ffi.iptr(out['status']),
                                     _ffi.iptr(out['num_iters']), _ffi.iptr(out['qp_solves']), _ffi.dptr(out['cond']), _ffi.dptr(out['cost']))
        if rc != 0:
            raise RuntimeError('dgsqp_fetch_results failed: ' + lib.dgsqp_last_error(s._h).decode())
        out['time'] = time.time() - t0
        out['kernel_ms'] = tm.kernel_ms
        out['msg'] = [_ffi.STATUS_MSG[v] for v in out['status']]
        out['converged'] = out['status'] <= 1
        out['u_pred'] = s._to_time_major(out['u'])
        outs.append(out)
    return outs


def plan(P: _ffi.ProblemT, par: _ffi.ParamsT) -> dict:
    """What ``dgsqp_create`` would build for this game (host only, no GPU): dimensions, LDS bytes, scratch bytes and the
    layout (0 LDS-resident, 1 big, 2 XL); raises ``ValueError`` with the library's reason for unsupported games."""
    lib = _ffi.load_library()
    d = _ffi.DimsT()
    msg = C.create_string_buffer(256)
    rc = lib.dgsqp_plan(C.byref(P), C.byref(par), C.byref(d), msg, 256)
    if rc != 0:
        raise ValueError(f'unsupported game ({rc}): {msg.value.decode()}')
    return {k: getattr(d, k) for k, _ in _ffi.DimsT._fields_ if k != 'reserved_'}


def problem_dims(P: _ffi.ProblemT):
    """(n_q, n_u, n, n_c) from the constraint-assembly rules DGSQP.py:732-821."""
    M, N = P.M, P.N
    nqa = [{0: 6, 1: 8, 2: 4}[P.agents[a].model] for a in range(M)]
    n_q, n_u = sum(nqa), 2 * M
    n_c = 0
    pairs = M * (M - 1) // 2 if P.obstacle_rows else 0
    for k in range(N + 1):
        if k >= 1:
            n_c += pairs
        for a in range(M):
            A = P.agents[a]
            n_c += A.n_lane                      # lane rows sit at every stage, k = 0 and k = N included
            if k < N:
                n_c += (4 if A.has_rate else 0) + sum(A.in_ub[j] < np.inf for j in range(2)) + sum(A.in_lb[j] > -np.inf for j in range(2))
            if k > 0:
                n_c += sum(A.st_ub[i] < np.inf for i in range(nqa[a])) + sum(A.st_lb[i] > -np.inf for i in range(nqa[a]))
    return n_q, n_u, N * n_u, int(n_c)


# ---------------------------------------------------------------------------------------------
# solver
# ---------------------------------------------------------------------------------------------
class DGSQP(AbstractSolver):
    def __init__(self, joint_dynamics: CasadiDecoupledMultiAgentDynamicsModel,
                 costs: List[RacingCost],
                 agent_constraints: List[Optional[InputRateLimits]],
                 shared_constraints: Optional[CollisionAvoidance],
                 bounds: Dict[str, List[VehicleState]],
                 params: DGSQPParams = DGSQPParams(),
                 print_method=print,
                 xy_plot=None,
                 use_mx: bool = False,
                 device: int = 0,
                 eig_floor: Optional[float] = None,
                 snap_active_bounds: bool = False,
                 lsqr_tol: Optional[float] = None,
                 qp_warm_start: bool = True,
                 qp_method: Optional[str] = None,
                 osqp_rho_carry: bool = False,
                 mixed_precision: bool = False,
                 workgroups_per_cu: int = 1):
        """``eig_floor``, ``snap_active_bounds``: implementation knobs, see ``build_params``; the defaults are the reference's
        literal formulas (``_nearestPD`` floor 1e-10, DGSQP.py:1293; no adjustment of the QP step).  ``qp_method``: 'active_set'
        (exact KKT point) or 'osqp' (OSQP's own ADMM + polish arithmetic), see ``resolve_qp_method``.  ``workgroups_per_cu=2``: this solver
        runs on ``libdgsqp_hip_b256.so`` (256-thread workgroups, half the LDS arena, two per CU): 1.2-1.5 x the throughput on batches >> 512 of
        n <= 64 games, identical results; games it cannot hold raise (DGSQP_E_TOO_LARGE).  Solvers of different builds do not share launches."""
        self.joint_dynamics = joint_dynamics
        self.M = joint_dynamics.n_a
        self.print_method = (lambda s: None) if print_method is None else print_method
        self.params = params
        self.N = params.N
        self.solver_name = params.solver_name
        self.verbose = params.verbose
        self.save_iter_data = params.save_iter_data
        self.n_u, self.n_q = joint_dynamics.n_u, joint_dynamics.n_q
        self.num_qa_d = [int(m.n_q) for m in joint_dynamics.dynamics_models]
        self.num_ua_d = [int(m.n_u) for m in joint_dynamics.dynamics_models]
        self.num_ua_el = [int(self.N * m.n_u) for m in joint_dynamics.dynamics_models]

        self._problem = build_problem(joint_dynamics, costs, agent_constraints, shared_constraints, bounds, params)
        self._cparams = build_params(params, eig_floor=eig_floor, snap_active_bounds=snap_active_bounds, lsqr_tol=lsqr_tol,
                                     qp_warm_start=qp_warm_start, qp_method=qp_method, osqp_rho_carry=osqp_rho_carry,
                                     mixed_precision=mixed_precision)
        _, _, self.n, n_c = problem_dims(self._problem)
        self.n_c_total = n_c

        self._lib = _ffi.load_library(workgroups_per_cu)           # raises if the HIP library is missing
        self._h = C.c_void_p()
        rc = self._lib.dgsqp_create(C.byref(self._problem), C.byref(self._cparams), int(device), C.byref(self._h))
        if rc != 0:
            msg = self._lib.dgsqp_last_error(None)
            raise RuntimeError(f'dgsqp_create failed ({rc}): {msg.decode() if msg else ""}')
        d = _ffi.DimsT()
        self._lib.dgsqp_dims(self._h, C.byref(d))
        assert (d.n, d.n_c, d.n_q) == (self.n, n_c, self.n_q), 'host/device layout mismatch'
        self.dims = d

        self.state_input_predictions = [VehiclePrediction() for _ in range(self.M)]
        self.q_pred = np.zeros((self.N + 1, self.n_q))
        self.u_pred = np.zeros((self.N, self.n_u))
        self.l_pred = np.zeros(n_c)
        self.u_prev = np.zeros(self.n_u)
        self.u_ws = np.zeros(self.N * self.n_u)
        self.l_ws = None
        self.initialized = True

    def __del__(self):
        h = getattr(self, '_h', None)
        if h is not None and h.value:
            self._lib.dgsqp_destroy(h)
            self._h = C.c_void_p()

    # ---- layout helpers (DGSQP.py:271-281, :477-482) ------------------------------------------
    def _to_agent_major(self, u_tm: np.ndarray) -> np.ndarray:
        """[..., N, n_u] time-major joint inputs -> [..., n] agent-major decision vector."""
        parts, si = [], 0
        for nu in self.num_ua_d:
            parts.append(u_tm[..., :, si:si + nu].reshape(*u_tm.shape[:-2], self.N * nu))
            si += nu
        return np.concatenate(parts, axis=-1)

    def _to_time_major(self, u_am: np.ndarray) -> np.ndarray:
        parts, si = [], 0
        for nu, nel in zip(self.num_ua_d, self.num_ua_el):
            parts.append(u_am[..., si:si + nel].reshape(*u_am.shape[:-1], self.N, nu))
            si += nel
        return np.concatenate(parts, axis=-1)

    def initialize(self):
        pass

    def set_warm_start(self, u_ws: np.ndarray, l_ws: np.ndarray = None):
        if u_ws.shape[0] != self.N or u_ws.shape[1] != self.n_u:
            raise RuntimeError('Warm start state sequence of shape (%i,%i) is incompatible with required shape (%i,%i)'
                               % (u_ws.shape[0], u_ws.shape[1], self.N, self.n_u))
        self.u_ws = self._to_agent_major(np.asarray(u_ws, dtype=float))
        self.l_ws = l_ws

    # ---- batched entry point -------------------------------------------------------------------
    def solve_batch(self, x0: np.ndarray, u_ws: np.ndarray, dtype=np.float64) -> dict:
        """B independent ``solve()`` calls.  ``x0`` [B, n_q]; ``u_ws`` [B, N, n_u] (time-major,
        as ``set_warm_start``) or [B, n] (agent-major).  ``dtype=np.float32``: single-precision arrays at the boundary
        (``dgsqp_solve_batch_f32``: widened on the device, fp64 solve, results rounded to fp32)."""
        if np.dtype(dtype) == np.float32:
            return self._solve_batch_f32(x0, u_ws)
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        B = x0.shape[0]
        u_ws = np.asarray(u_ws, dtype=np.float64)
        if u_ws.ndim == 3:
            if u_ws.shape[1:] != (self.N, self.n_u):
                raise RuntimeError('Warm start state sequence of shape (%i,%i) is incompatible with required shape (%i,%i)'
                                   % (u_ws.shape[1], u_ws.shape[2], self.N, self.n_u))
            u_ws = self._to_agent_major(u_ws)
        u_ws = np.ascontiguousarray(u_ws)
        if x0.shape != (B, self.n_q) or u_ws.shape != (B, self.n):
            raise RuntimeError(f'bad batch shapes x0 {x0.shape} u_ws {u_ws.shape}')
        out = dict(u=np.empty((B, self.n)), l=np.empty((B, self.n_c_total)), x=np.empty((B, self.N + 1, self.n_q)),
                   status=np.empty(B, np.int32), num_iters=np.empty(B, np.int32), qp_solves=np.empty(B, np.int32),
                   cond=np.empty((B, 3)), cost=np.empty((B, self.M)))
        tm = _ffi.TimingT()
        t0 = time.time()
        rc = self._lib.dgsqp_solve_batch(self._h, B, _ffi.dptr(x0), _ffi.dptr(u_ws), _ffi.dptr(out['u']), _ffi.dptr(out['l']),
                                         _ffi.dptr(out['x']), _ffi.iptr(out['status']), _ffi.iptr(out['num_iters']),
                                         _ffi.iptr(out['qp_solves']), _ffi.dptr(out['cond']), _ffi.dptr(out['cost']), C.byref(tm))
        if rc != 0:
            raise RuntimeError(f'dgsqp_solve_batch failed ({rc}): {self._lib.dgsqp_last_error(self._h).decode()}')
        out['time'] = time.time() - t0
        out['kernel_ms'] = tm.kernel_ms
        out['msg'] = [_ffi.STATUS_MSG[s] for s in out['status']]
        out['converged'] = out['status'] <= 1
        out['u_pred'] = self._to_time_major(out['u'])
        return out

    def _solve_batch_f32(self, x0, u_ws) -> dict:
        x0 = np.ascontiguousarray(x0, dtype=np.float32)
        B = x0.shape[0]
        u_ws = np.asarray(u_ws, dtype=np.float32)
        if u_ws.ndim == 3:
            u_ws = self._to_agent_major(u_ws)
        u_ws = np.ascontiguousarray(u_ws, dtype=np.float32)
        if x0.shape != (B, self.n_q) or u_ws.shape != (B, self.n):
            raise RuntimeError(f'bad batch shapes x0 {x0.shape} u_ws {u_ws.shape}')
        f32 = np.float32
        out = dict(u=np.empty((B, self.n), f32), l=np.empty((B, self.n_c_total), f32), x=np.empty((B, self.N + 1, self.n_q), f32),
                   status=np.empty(B, np.int32), num_iters=np.empty(B, np.int32), qp_solves=np.empty(B, np.int32),
                   cond=np.empty((B, 3), f32), cost=np.empty((B, self.M), f32))
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        tm = _ffi.TimingT()
        t0 = time.time()
        rc = self._lib.dgsqp_solve_batch_f32(self._h, B, fp(x0), fp(u_ws), fp(out['u']), fp(out['l']), fp(out['x']), _ffi.iptr(out['status']),
                                             _ffi.iptr(out['num_iters']), _ffi.iptr(out['qp_solves']), fp(out['cond']), fp(out['cost']), C.byref(tm))
        if rc != 0:
            raise RuntimeError(f'dgsqp_solve_batch_f32 failed ({rc}): {self._lib.dgsqp_last_error(self._h).decode()}')
        out['time'] = time.time() - t0
        out['kernel_ms'] = tm.kernel_ms
        out['msg'] = [_ffi.STATUS_MSG[s] for s in out['status']]
        out['converged'] = out['status'] <= 1
        out['u_pred'] = self._to_time_major(out['u'])
        return out

    def set_trace(self, pairs_per_scenario: int):
        """Test hook: record the SQP event log of subsequent solves (0 disables)."""
        self._trace_cap = int(pairs_per_scenario)
        self._lib.dgsqp_set_trace(self._h, self._trace_cap)

    def fetch_trace(self, B: int):
        """List of [(code, value)] arrays, one per scenario of the last solve_batch (``B`` = its batch size)."""
        cap = getattr(self, '_trace_cap', 0)
        raw = np.zeros((B, 1 + 2 * cap))
        rc = self._lib.dgsqp_fetch_trace(self._h, _ffi.dptr(raw), raw.size)
        if rc != 0:
            raise RuntimeError(self._lib.dgsqp_last_error(self._h).decode())
        if (raw[:, 0] > cap).any():
            raise RuntimeError(f'event log truncated: {int(raw[:, 0].max())} events, capacity {cap} (set_trace with a larger value)')
        return [raw[b, 1:1 + 2 * int(raw[b, 0])].reshape(-1, 2) for b in range(B)]

    def set_cooperative(self, mode: int):
        """Cooperative line search (include/dgsqp.h: dgsqp_set_cooperative): 0 never, 1 synchronous calls only (default), 2 every
        launch.  Results are bit-identical in every mode; only the time a launch spends behind its slowest scenario changes."""
        if self._lib.dgsqp_set_cooperative(self._h, int(mode)) != 0:
            raise ValueError(f'bad cooperative mode {mode}')
        self._coop_mode = int(mode)

    def coop_stats(self) -> dict:
        """Counters of the last cooperative launch (include/dgsqp.h: dgsqp_coop_stats)."""
        out = (C.c_uint64 * 6)()
        if self._lib.dgsqp_coop_stats(self._h, out) != 0:
            raise RuntimeError(self._lib.dgsqp_last_error(self._h).decode())
        return dict(helped=int(out[0]), helper_registrations=int(out[1]), finished=int(out[2]), idle=int(out[3]), used=int(out[4]), mismatches=int(out[5]))

    def set_deferral(self, min_iters: int = 8, factor: float = 2.0):
        """Deferral of long scenarios in cooperative launches (include/dgsqp.h: dgsqp_set_deferral): a scenario still iterating after
        max(min_iters, factor x mean iterations of the finished ones) is set aside while fresh scenarios wait and resumed, bit for
        bit, once the queue is empty.  min_iters = 0 switches it off."""
        if self._lib.dgsqp_set_deferral(self._h, int(min_iters), float(factor)) != 0:
            raise ValueError(f'bad deferral setting {min_iters}, {factor}')

    def deferral_stats(self) -> dict:
        out = (C.c_uint64 * 2)()
        if self._lib.dgsqp_deferral_stats(self._h, out) != 0:
            raise RuntimeError(self._lib.dgsqp_last_error(self._h).decode())
        return dict(deferred=int(out[0]), resumed=int(out[1]))

    def deferral_log(self, max_rows: int = 1 << 16) -> np.ndarray:
        """Rows (ticket, iterations, QPs, cost in 100 MHz ticks when set aside, ticks since launch start when set aside / resumed /
        finished, final iterations, final QPs, the three convergence measures when set aside) of the last cooperative launch's deferred
        scenarios (diagnostic), as float64."""
        out = np.zeros((max_rows, 11), np.uint64)
        n = self._lib.dgsqp_deferral_log(self._h, out.ctypes.data_as(C.POINTER(C.c_uint64)), max_rows)
        if n < 0:
            raise RuntimeError('dgsqp_deferral_log failed')
        r = out[:n, :8].astype(np.int64)
        return np.column_stack([r[:, :7], r[:, 7] & 0xffffffff, r[:, 7] >> 32, out[:n, 8:].copy().view(np.float64)])

    def set_iterate_log(self, records_per_scenario: int):
        """Keep (u, l) after every SQP iteration of subsequent solves (what ``solve()`` reports in ``iter_data``); 0 disables."""
        self._itlog_cap = int(records_per_scenario)
        if self._lib.dgsqp_set_iterate_log(self._h, self._itlog_cap) != 0:
            raise RuntimeError(self._lib.dgsqp_last_error(self._h).decode())

    def fetch_iterate_log(self, B: int):
        """Per scenario of the last solve_batch: (u [records, n] agent-major, l [records, n_c]); record 0 is the start (u_ws, l0)."""
        cap = getattr(self, '_itlog_cap', 0)
        w = self.n + self.n_c_total
        raw = np.zeros((B, 1 + cap * w))
        rc = self._lib.dgsqp_fetch_iterate_log(self._h, _ffi.dptr(raw), raw.size)
        if rc != 0:
            raise RuntimeError(self._lib.dgsqp_last_error(self._h).decode())
        out = []
        for b in range(B):
            k = min(int(raw[b, 0]), cap)
            rec = raw[b, 1:1 + k * w].reshape(k, w)
            out.append((rec[:, :self.n].copy(), rec[:, self.n:].copy()))
        return out

    # ---- test hooks ----------------------------------------------------------------------------
    def evaluate_batch(self, x0, u, l=None):
        """One ``_evaluate(u, l, x0, up=0, hessian=True)`` per row (DGSQP.py:509-533) + dual init."""
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        u = np.ascontiguousarray(u, dtype=np.float64)
        B = x0.shape[0]
        l = None if l is None else np.ascontiguousarray(l, dtype=np.float64)
        n, nc = self.n, self.n_c_total
        out = dict(q=np.empty((B, n)), g=np.empty((B, nc)), G=np.empty((B, nc, n)), Q=np.empty((B, n, n)),
                   x=np.empty((B, self.N + 1, self.n_q)), l0=np.empty((B, nc)))
        rc = self._lib.dgsqp_evaluate_batch(self._h, B, _ffi.dptr(x0), _ffi.dptr(u), _ffi.dptr(l), _ffi.dptr(out['q']),
                                            _ffi.dptr(out['g']), _ffi.dptr(out['G']), _ffi.dptr(out['Q']), _ffi.dptr(out['x']),
                                            _ffi.dptr(out['l0']))
        if rc != 0:
            raise RuntimeError(f'dgsqp_evaluate_batch failed ({rc}): {self._lib.dgsqp_last_error(self._h).decode()}')
        return out

    def qp_batch(self, x0, u, l, want_Qpd=True):
        """One ``_solve_qp`` (DGSQP.py:232-266) per row at the linearisation point (u, l).  ``want_Qpd=False`` leaves the projected
        Hessian on the device -- the path the solve itself takes (the certified ``_nearestPD`` shortcut only runs then)."""
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        u = np.ascontiguousarray(u, dtype=np.float64)
        l = np.ascontiguousarray(l, dtype=np.float64)
        B = x0.shape[0]
        out = dict(du=np.empty((B, self.n)), lhat=np.empty((B, self.n_c_total)), Qpd=np.empty((B, self.n, self.n)) if want_Qpd else None,
                   flag=np.empty(B, np.int32), info=np.zeros((B, 8)))
        rc = self._lib.dgsqp_qp_batch_info(self._h, B, _ffi.dptr(x0), _ffi.dptr(u), _ffi.dptr(l), _ffi.dptr(out['du']),
                                           _ffi.dptr(out['lhat']), _ffi.dptr(out['Qpd']), _ffi.iptr(out['flag']), _ffi.dptr(out['info']))
        if rc != 0:
            raise RuntimeError(f'dgsqp_qp_batch failed ({rc}): {self._lib.dgsqp_last_error(self._h).decode()}')
        return out           # info: OSQP's diagnostics with qp_method='osqp' (include/dgsqp.h: dgsqp_qp_batch_info), zeros otherwise

    def pid_warm_start_batch(self, q0, u_max=(2.1, 0.436), du_max=(10.0, 4.5), substeps=10, want_trajectories=False):
        """PID lane-follower warm start of a batch on the device (chicane.py:411-447 with PID.py; collision check
        chicane.py:38-43).  q0 [B, n_q] -> dict(u_ws [B, N, n_u] time-major as ``set_warm_start`` / ``solve_batch`` take it,
        collide [B] bool, optionally q_ws [B, N+1, n_q])."""
        q0 = np.ascontiguousarray(q0, dtype=np.float64)
        if q0.ndim != 2 or q0.shape[1] != self.n_q:
            raise RuntimeError(f'q0 must be [B, {self.n_q}]')
        B = q0.shape[0]
        pid = _ffi.PidT()
        pid.kp_v, pid.kp_s, pid.ki_s, pid.ey_gain, pid.ei_max = 1.0, 1.0, 0.005, 5.0, 100.0
        pid.u_max[0], pid.u_max[1] = float(u_max[0]), float(u_max[1])
        pid.du_max[0], pid.du_max[1] = float(du_max[0]), float(du_max[1])
        pid.substeps = int(substeps)
        u_am = np.empty((B, self.n))
        q_ws = np.empty((B, self.N + 1, self.n_q)) if want_trajectories else None
        col = np.empty(B, np.int32)
        rc = self._lib.dgsqp_pid_warm_start_batch(self._h, B, _ffi.dptr(q0), C.byref(pid), _ffi.dptr(u_am), _ffi.dptr(q_ws),
                                                  _ffi.iptr(col))
        if rc != 0:
            raise RuntimeError(f'dgsqp_pid_warm_start_batch failed ({rc}): {self._lib.dgsqp_last_error(self._h).decode()}')
        nua = self.n_u // self.M
        u_tm = u_am.reshape(B, self.M, self.N, nua).transpose(0, 2, 1, 3).reshape(B, self.N, self.n_u)
        out = dict(u_ws=np.ascontiguousarray(u_tm), collide=col.astype(bool))
        if want_trajectories:
            out['q_ws'] = q_ws
        return out

    def sample_batch(self, game, B: int, seed: int = 1, u_max=(2.1, 0.436), du_max=None, substeps=10, stage=False, fetch=True):
        """The game's rejection sampler on the device (``dgsqp_sample_batch``; counter-based random numbers, mirrored bit for bit
        by ``dgsqp_amd.sampler``): placement, PID warm start (zero inputs for the merge), collision rejection, the first ``B``
        accepted candidates in candidate order.  ``stage=True`` leaves the batch staged on the handle (``solve_staged`` next);
        ``fetch=False`` skips the copy to the host.  Returns dict(x0, u_ws [B, N, n_u] time-major, candidates)."""
        from . import sampler as smp
        spec = smp.sampler_spec(game, seed)
        pid = _ffi.PidT()
        pid.kp_v, pid.kp_s, pid.ki_s, pid.ey_gain, pid.ei_max = 1.0, 1.0, 0.005, 5.0, 100.0
        if du_max is None:
            rl = game.agent_constraints[0] if game.agent_constraints else None
            du_max = (10.0, 4.5) if (rl is None or not hasattr(rl, 'rate_max')) else tuple(rl.rate_max)
        pid.u_max[0], pid.u_max[1] = float(u_max[0]), float(u_max[1])
        pid.du_max[0], pid.du_max[1] = float(du_max[0]), float(du_max[1])
        pid.substeps = int(substeps)
        x0 = np.empty((B, self.n_q)) if fetch else None
        u_am = np.empty((B, self.n)) if fetch else None
        used = C.c_int64(0)
        rc = self._lib.dgsqp_sample_batch(self._h, int(B), C.byref(spec), C.byref(pid), _ffi.dptr(x0), _ffi.dptr(u_am), C.byref(used), 1 if stage else 0)
        if rc != 0:
            raise RuntimeError(f'dgsqp_sample_batch failed ({rc}): {self._lib.dgsqp_last_error(self._h).decode()}')
        out = dict(candidates=int(used.value))
        if fetch:
            nua = self.n_u // self.M
            out['x0'] = x0
            out['u_ws'] = np.ascontiguousarray(u_am.reshape(B, self.M, self.N, nua).transpose(0, 2, 1, 3).reshape(B, self.N, self.n_u))
        return out

    # ---- reference single-scenario surface -----------------------------------------------------
    def solve(self, states: List[VehicleState], parameters: np.ndarray = np.array([])) -> dict:
        solve_start = time.time()
        self.u_prev = np.zeros(self.n_u)
        x0 = self.joint_dynamics.state2q(states)
        u_init = copy.copy(self.u_ws)
        self.print_method(self.solver_name)
        want_iters = bool(self.save_iter_data)
        iters = int(self._cparams.sqp_iters)
        if want_iters:       # worst case per iteration: 3 + 4 + 1 events, 7 watchdog QPs, 3 line searches of line_search_iters trials
            self.set_trace((iters + 1) * (16 + 6 * int(self._cparams.line_search_iters)))
        self.set_iterate_log(iters + 2)
        try:
            res = self.solve_batch(x0[None, :], u_init[None, :])
            events = self.fetch_trace(1)[0] if want_iters else None
            u_log, l_log = self.fetch_iterate_log(1)[0]
        finally:
            if want_iters:
                self.set_trace(0)
            self.set_iterate_log(0)
        self.q_pred = res['x'][0]
        self.u_pred = res['u_pred'][0]
        self._last_u_agent_major = res['u'][0]
        self.l_pred = res['l'][0]
        msg = res['msg'][0]
        cond = dict(p_feas=float(res['cond'][0, 0]), comp=float(res['cond'][0, 1]), stat=float(res['cond'][0, 2]))
        solve_dur = time.time() - solve_start
        self.print_method(f'Solve status: {msg}')
        self.print_method(f'Solve iters: {int(res["num_iters"][0])}')
        self.print_method(f'Solve time: {solve_dur:.2f}')
        self.print_method(str(res['cost'][0]))
        # iter_data (DGSQP.py:386-451): one record per SQP iteration with the optimality measures at its start and its QP
        # solves, rebuilt from the kernel's event log (codes 1-3 and 40), and the iterates (u_sol, l_sol) the iteration
        # ended with, from the kernel's iterate log.
        iter_data = []
        if want_iters:
            cur = None
            for code, val in events:
                code = int(code)
                if code == 1:
                    cur = dict(cond=dict(stat=float(val)), u_sol=None, l_sol=None, qp_solves=0, it_time=None)
                elif cur is not None and code == 2:
                    cur['cond']['p_feas'] = float(val)
                elif cur is not None and code == 3:
                    cur['cond']['comp'] = float(val)
                elif cur is not None and code == 40:
                    cur['qp_solves'] = int(val)
                    iter_data.append(cur)
                    cur = None
            n_it = max(len(iter_data), 1)
            for d in iter_data:
                d['it_time'] = solve_dur / n_it
            for i, d in enumerate(iter_data):
                if i + 1 < len(u_log):
                    d['u_sol'], d['l_sol'] = u_log[i + 1], l_log[i + 1]
        return dict(time=solve_dur, num_iters=int(res['num_iters'][0]), status=bool(res['converged'][0]),
                    cost=[float(c) for c in res['cost'][0]], cond=cond, iter_data=iter_data, msg=msg,
                    init=dict(u=u_log[0] if len(u_log) else None, l=l_log[0] if len(l_log) else None))

    def step(self, states: List[VehicleState], parameters: np.ndarray = np.array([])):
        info = self.solve(states, parameters)
        self.joint_dynamics.qu2state(states, None, self.u_pred[0])
        self.joint_dynamics.qu2prediction(self.state_input_predictions, self.q_pred, self.u_pred)
        for q in self.state_input_predictions:
            q.t = states[0].t
        self.u_prev = self.u_pred[0]
        if info['msg'] not in ['diverged', 'qp_fail']:
            self.set_warm_start(np.vstack((self.u_pred[1:], self.u_pred[-1])))
        return info

    def get_prediction(self) -> List[VehiclePrediction]:
        return self.state_input_predictions
