"""ctypes mirror of include/dgsqp.h and the loader of the HIP library.

There is deliberately no CPU fallback: if ``libdgsqp_hip.so`` is missing or
cannot be loaded, ``load_library()`` raises."""
from __future__ import annotations

import ctypes as C
import os
import pathlib

MAX_KNOTS = 1152
MAX_AGENTS = 6
MAX_SEGS = 16
MAX_NQA = 8
MAX_LANES = 2
NUA = 2

STATUS_MSG = ['conv_abs_tol', 'conv_rel_tol', 'max_it', 'diverged', 'qp_fail', 'time_limit']

dbl2 = C.c_double * NUA


class LaneT(C.Structure):
    _fields_ = [('brk', C.c_double), ('n_lo', dbl2), ('n_hi', dbl2), ('anchor', dbl2), ('r', C.c_double)]


class AgentT(C.Structure):
    _fields_ = [
        ('model', C.c_int32), ('tire_model', C.c_int32), ('drive_wheels', C.c_int32), ('simple_slip', C.c_int32),
        ('L_f', C.c_double), ('L_r', C.c_double), ('mass', C.c_double), ('I_z', C.c_double), ('gravity', C.c_double),
        ('c_dr', C.c_double), ('c_da', C.c_double), ('c_s', C.c_double), ('c_r', C.c_double), ('p_r', C.c_double),
        ('pac_Bf', C.c_double), ('pac_Br', C.c_double), ('pac_Cf', C.c_double), ('pac_Cr', C.c_double),
        ('pac_Df', C.c_double), ('pac_Dr', C.c_double), ('lin_Bf', C.c_double), ('lin_Br', C.c_double),
        ('w_in', dbl2), ('w_rate', dbl2), ('w_prog', C.c_double), ('w_comp', C.c_double),
        ('comp_type', C.c_int32), ('_pad0', C.c_int32),
        ('w_block', C.c_double), ('w_obs', C.c_double), ('obs_cost_r', C.c_double),
        ('has_rate', C.c_int32), ('_pad1', C.c_int32),
        ('rate_ub', dbl2), ('rate_lb', dbl2), ('in_ub', dbl2), ('in_lb', dbl2),
        ('st_ub', C.c_double * MAX_NQA), ('st_lb', C.c_double * MAX_NQA),
        ('radius', C.c_double),
        ('w_goal', C.c_double * MAX_NQA), ('goal', C.c_double * MAX_NQA), ('goal_term_mult', C.c_double),
        ('n_lane', C.c_int32), ('_pad2', C.c_int32), ('lane', LaneT * MAX_LANES),
    ]


class ProblemT(C.Structure):
    _fields_ = [
        ('M', C.c_int32), ('N', C.c_int32), ('integrator', C.c_int32), ('substeps', C.c_int32),
        ('dt', C.c_double),
        ('n_segs', C.c_int32), ('obstacle_rows', C.c_int32),
        ('track_L', C.c_double),
        ('seg_s', C.c_double * (MAX_SEGS + 1)), ('seg_curv', C.c_double * MAX_SEGS),
        ('seg_ang', C.c_double * (MAX_SEGS + 1)),
        ('agents', AgentT * MAX_AGENTS),
        ('track_kind', C.c_int32), ('n_knots', C.c_int32), ('spline', C.c_uint64),     # const double*: kept as an integer so that the POD stays copyable / picklable
    ]


class ParamsT(C.Structure):
    _fields_ = [
        ('beta', C.c_double), ('tau', C.c_double), ('p_tol', C.c_double), ('d_tol', C.c_double), ('reg', C.c_double),
        ('line_search_iters', C.c_int32), ('nonmono_ls', C.c_int32), ('sqp_iters', C.c_int32),
        ('merit_function', C.c_int32), ('rel_tol_req', C.c_int32), ('lsqr_iter_lim', C.c_int32),
        ('lsqr_atol', C.c_double), ('lsqr_btol', C.c_double),
        ('qp_warm_start', C.c_int32), ('hessian_bfgs', C.c_int32),
        ('eig_floor', C.c_double), ('time_limit', C.c_double),
        ('snap_active_bounds', C.c_int32), ('variant', C.c_int32),
        ('nms', C.c_int32), ('nms_frequency', C.c_int32), ('nms_memory_size', C.c_int32), ('merit_decrease_condition', C.c_int32),
        ('qp_method', C.c_int32), ('osqp_rho_carry', C.c_int32), ('mixed_precision', C.c_int32), ('reserved_', C.c_int32),
        ('reg_decay', C.c_double), ('delta_decay', C.c_double), ('merit_decrease', C.c_double), ('merit_parameter', C.c_double),
    ]


class PidT(C.Structure):
    _fields_ = [
        ('kp_v', C.c_double), ('kp_s', C.c_double), ('ki_s', C.c_double), ('ey_gain', C.c_double), ('ei_max', C.c_double),
        ('u_max', C.c_double * 2), ('du_max', C.c_double * 2), ('substeps', C.c_int32), ('reserved_', C.c_int32),
    ]


class StatRecordT(C.Structure):
    _fields_ = [('status', C.c_int32), ('iters', C.c_int32), ('qp_solves', C.c_int32), ('rank', C.c_int32),
                ('p_feas', C.c_double), ('comp', C.c_double), ('stat', C.c_double), ('cost', C.c_double * MAX_AGENTS)]


class DimsT(C.Structure):
    _fields_ = [('M', C.c_int32), ('N', C.c_int32), ('n_q', C.c_int32), ('n_u', C.c_int32), ('n', C.c_int32),
                ('n_c', C.c_int32), ('n_dense', C.c_int32), ('lds_bytes', C.c_int32),
                ('workspace_bytes', C.c_int64), ('layout', C.c_int32), ('reserved_', C.c_int32)]


class TimingT(C.Structure):
    _fields_ = [('h2d_ms', C.c_double), ('kernel_ms', C.c_double), ('d2h_ms', C.c_double), ('total_ms', C.c_double),
                ('grid', C.c_int32), ('block', C.c_int32)]


_PD = C.POINTER(C.c_double)
_PI = C.POINTER(C.c_int32)
_LIB = None
_LIBS = {}          # workgroups per CU -> CDLL (1: the product build; 2: libdgsqp_hip_b256.so, 256-thread workgroups and half the LDS arena)


def library_path(workgroups_per_cu: int = 1) -> pathlib.Path:
    env = os.environ.get('DGSQP_HIP_LIB')
    if env:
        return pathlib.Path(env)
    name = {1: 'libdgsqp_hip.so', 2: 'libdgsqp_hip_b256.so'}[int(workgroups_per_cu)]
    return pathlib.Path(__file__).resolve().parent / 'csrc' / name


def load_library(workgroups_per_cu: int = 1) -> C.CDLL:
    """Load the HIP solver library; raise loudly if it is absent (no CPU fallback).  ``workgroups_per_cu=2`` loads the build with
    256-thread workgroups, two per CU (row N1: large batches of n <= 64 games; the two libraries can live in one process)."""
    global _LIB
    if int(workgroups_per_cu) not in (1, 2):
        raise ValueError('workgroups_per_cu: 1 (the product build) or 2 (libdgsqp_hip_b256.so)')
    if int(workgroups_per_cu) in _LIBS:
        return _LIBS[int(workgroups_per_cu)]
    path = library_path(workgroups_per_cu)
    if not path.exists():
        raise RuntimeError(f'HIP solver library {path} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                           f'(hipcc --offload-arch=gfx950). There is no CPU fallback.')
    lib = C.CDLL(str(path))
    H = C.c_void_p
    lib.dgsqp_create.argtypes = [C.POINTER(ProblemT), C.POINTER(ParamsT), C.c_int, C.POINTER(H)]
    lib.dgsqp_create.restype = C.c_int
    lib.dgsqp_destroy.argtypes = [H]
    lib.dgsqp_destroy.restype = None
    lib.dgsqp_dims.argtypes = [H, C.POINTER(DimsT)]
    lib.dgsqp_dims.restype = C.c_int
    lib.dgsqp_plan.argtypes = [C.POINTER(ProblemT), C.POINTER(ParamsT), C.POINTER(DimsT), C.c_char_p, C.c_int]
    lib.dgsqp_plan.restype = C.c_int
    lib.dgsqp_last_error.argtypes = [H]
    lib.dgsqp_last_error.restype = C.c_char_p
    lib.dgsqp_backend_info.argtypes = [C.c_char_p, C.c_int]
    lib.dgsqp_backend_info.restype = C.c_int
    lib.dgsqp_solve_batch.argtypes = [H, C.c_int64, _PD, _PD, _PD, _PD, _PD, _PI, _PI, _PI, _PD, _PD, C.POINTER(TimingT)]
    lib.dgsqp_solve_batch.restype = C.c_int
    lib.dgsqp_stage_inputs.argtypes = [H, C.c_int64, _PD, _PD]
    lib.dgsqp_stage_inputs.restype = C.c_int
    lib.dgsqp_solve_staged.argtypes = [H, C.POINTER(TimingT)]
    lib.dgsqp_solve_staged.restype = C.c_int
    lib.dgsqp_launch_staged.argtypes = [H]
    lib.dgsqp_launch_staged.restype = C.c_int
    lib.dgsqp_draining.argtypes = [H]
    lib.dgsqp_draining.restype = C.c_int
    lib.dgsqp_wait.argtypes = [H, C.POINTER(TimingT)]
    lib.dgsqp_wait.restype = C.c_int
    lib.dgsqp_fetch_results.argtypes = [H, _PD, _PD, _PD, _PI, _PI, _PI, _PD, _PD]
    lib.dgsqp_fetch_results.restype = C.c_int
    lib.dgsqp_evaluate_batch.argtypes = [H, C.c_int64, _PD, _PD, _PD, _PD, _PD, _PD, _PD, _PD, _PD]
    lib.dgsqp_evaluate_batch.restype = C.c_int
    lib.dgsqp_qp_batch.argtypes = [H, C.c_int64, _PD, _PD, _PD, _PD, _PD, _PD, _PI]
    lib.dgsqp_qp_batch.restype = C.c_int
    lib.dgsqp_qp_batch_info.argtypes = [H, C.c_int64, _PD, _PD, _PD, _PD, _PD, _PD, _PI, _PD]
    lib.dgsqp_qp_batch_info.restype = C.c_int
    lib.dgsqp_pid_warm_start_batch.argtypes = [H, C.c_int64, _PD, C.POINTER(PidT), _PD, _PD, _PI]
    lib.dgsqp_pid_warm_start_batch.restype = C.c_int
    lib.dgsqp_set_trace.argtypes = [H, C.c_int]
    lib.dgsqp_set_trace.restype = C.c_int
    lib.dgsqp_fetch_trace.argtypes = [H, _PD, C.c_int64]
    lib.dgsqp_fetch_trace.restype = C.c_int
    lib.dgsqp_set_iterate_log.argtypes = [H, C.c_int]
    lib.dgsqp_set_iterate_log.restype = C.c_int
    lib.dgsqp_fetch_iterate_log.argtypes = [H, _PD, C.c_int64]
    lib.dgsqp_fetch_iterate_log.restype = C.c_int
    _PF = C.POINTER(C.c_float)
    lib.dgsqp_solve_batch_f32.argtypes = [H, C.c_int64, _PF, _PF, _PF, _PF, _PF, _PI, _PI, _PI, _PF, _PF, C.POINTER(TimingT)]
    lib.dgsqp_solve_batch_f32.restype = C.c_int
    lib.dgsqp_launch_staged_group.argtypes = [C.POINTER(H), C.c_int]
    lib.dgsqp_launch_staged_group.restype = C.c_int
    lib.dgsqp_finished.argtypes = [H]
    lib.dgsqp_finished.restype = C.c_int
    lib.dgsqp_set_cooperative.argtypes = [H, C.c_int]
    lib.dgsqp_set_cooperative.restype = C.c_int
    lib.dgsqp_coop_stats.argtypes = [H, C.POINTER(C.c_uint64)]
    lib.dgsqp_coop_stats.restype = C.c_int
    lib.dgsqp_osqp_counters.argtypes = [H, C.POINTER(C.c_uint64), C.c_int]
    lib.dgsqp_osqp_counters.restype = C.c_int
    lib.dgsqp_set_deferral.argtypes = [H, C.c_int32, C.c_double]
    lib.dgsqp_set_deferral.restype = C.c_int
    lib.dgsqp_reserve_deferral.argtypes = [H, C.c_int64]
    lib.dgsqp_reserve_deferral.restype = C.c_int
    lib.dgsqp_deferral_stats.argtypes = [H, C.POINTER(C.c_uint64)]
    lib.dgsqp_deferral_stats.restype = C.c_int
    lib.dgsqp_deferral_log.argtypes = [H, C.POINTER(C.c_uint64), C.c_int64]
    lib.dgsqp_deferral_log.restype = C.c_int
    lib.dgsqp_sample_batch.argtypes = [H, C.c_int64, C.c_void_p, C.POINTER(PidT), _PD, _PD, C.POINTER(C.c_int64), C.c_int]
    lib.dgsqp_sample_batch.restype = C.c_int
    lib.dgsqp_synchronize.argtypes = [H]
    lib.dgsqp_synchronize.restype = C.c_int
    lib.dgsqp_comm_unique_id.argtypes = [C.c_char_p]
    lib.dgsqp_comm_unique_id.restype = C.c_int
    lib.dgsqp_comm_init.argtypes = [H, C.c_char_p, C.c_int, C.c_int]
    lib.dgsqp_comm_init.restype = C.c_int
    lib.dgsqp_comm_destroy.argtypes = [H]
    lib.dgsqp_comm_destroy.restype = C.c_int
    lib.dgsqp_gather_stats.argtypes = [H, C.c_int64, C.c_void_p]
    lib.dgsqp_gather_stats.restype = C.c_int
    lib.dgsqp_comm_barrier.argtypes = [H]
    lib.dgsqp_comm_barrier.restype = C.c_int
    lib.dgsqp_comm_allreduce_max.argtypes = [H, _PD, C.c_int]
    lib.dgsqp_comm_allreduce_max.restype = C.c_int
    _LIBS[int(workgroups_per_cu)] = lib
    if int(workgroups_per_cu) == 1:
        _LIB = lib
    return lib


EXPORTED_SYMBOLS = ['dgsqp_create', 'dgsqp_destroy', 'dgsqp_dims', 'dgsqp_plan', 'dgsqp_last_error', 'dgsqp_backend_info',
                    'dgsqp_solve_batch', 'dgsqp_stage_inputs', 'dgsqp_solve_staged', 'dgsqp_fetch_results',
                    'dgsqp_evaluate_batch', 'dgsqp_qp_batch', 'dgsqp_qp_batch_info', 'dgsqp_set_trace', 'dgsqp_fetch_trace',
                    'dgsqp_pid_warm_start_batch', 'dgsqp_launch_staged', 'dgsqp_wait', 'dgsqp_draining',
                    'dgsqp_set_iterate_log', 'dgsqp_fetch_iterate_log', 'dgsqp_synchronize', 'dgsqp_finished', 'dgsqp_launch_staged_group', 'dgsqp_solve_batch_f32', 'dgsqp_comm_unique_id', 'dgsqp_comm_init',
                    'dgsqp_comm_destroy', 'dgsqp_gather_stats', 'dgsqp_comm_barrier', 'dgsqp_comm_allreduce_max',
                    'dgsqp_set_cooperative', 'dgsqp_coop_stats', 'dgsqp_osqp_counters', 'dgsqp_sample_batch', 'dgsqp_set_deferral', 'dgsqp_reserve_deferral', 'dgsqp_deferral_stats', 'dgsqp_deferral_log']


def dptr(a):
    return None if a is None else a.ctypes.data_as(_PD)


def iptr(a):
    return None if a is None else a.ctypes.data_as(_PI)
