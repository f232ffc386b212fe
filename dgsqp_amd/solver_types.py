"""Solver parameter dataclasses (reference DGSQP/solvers/solver_types.py:7-51
``ControllerConfig``/``PIDParams``, :91-127 ``DGSQPParams``).  Field names and
defaults are kept identical; tests/test_types.py diffs them against the
reference when /root/reference is present."""
from dataclasses import dataclass

from .types import PythonMsg


@dataclass
class ControllerConfig(PythonMsg):
    dt: float = 0.1


@dataclass
class PIDParams(ControllerConfig):
    Kp: float = 2.0
    Ki: float = 0.0
    Kd: float = 0.0
    int_e_max: float = 100
    int_e_min: float = -100
    u_max: float = None
    u_min: float = None
    du_max: float = None
    du_min: float = None
    u_ref: float = 0.0
    x_ref: float = 0.0
    noise: bool = False
    noise_max: float = 0.1
    noise_min: float = -0.1
    periodic_disturbance: bool = False
    disturbance_amplitude: float = 0.1
    disturbance_period: float = 1.0


@dataclass
class DGSQPParams(ControllerConfig):
    N: int = 10
    beta: float = 0.25
    tau: float = 0.5
    p_tol: float = 1e-3
    d_tol: float = 1e-3
    reg: float = 1e-3
    line_search_iters: int = 50
    nonmono_ls: bool = False
    sqp_iters: int = 50
    merit_function: str = 'stat_l1'
    verbose: bool = False
    save_iter_data: bool = True
    solver_name: str = 'DGSQP'
    time_limit: float = None
    qp_interface: str = 'casadi'
    qp_solver: str = 'osqp'
    conv_approx: bool = True
    hessian_approximation: str = 'none'
    code_gen: bool = False
    jit: bool = False
    opt_flag: str = 'O0'
    enable_jacobians: bool = True
    solver_dir: str = None
    so_name: str = None
    debug: bool = False
    debug_plot: bool = False
    pause_on_plot: bool = False
    local_pos: bool = False


@dataclass
class DGSQPV2Params(ControllerConfig):
    """Reference DGSQP/solvers/solver_types.py:130-175 (field names and defaults identical)."""
    N: int = 10
    beta: float = 0.25
    tau: float = 0.5
    p_tol: float = 1e-4
    d_tol: float = 1e-4
    reg: float = 1e2
    reg_decay: float = 0.95
    line_search_iters: int = 50
    nms: bool = True
    nms_frequency: int = 5
    nms_memory_size: int = 3
    sqp_iters: int = 500
    merit_function: str = 'stat_l1'
    merit_parameter: float = None
    merit_decrease: float = 0.01
    merit_decrease_condition: str = 'armijo'
    approximation_eval: str = 'always'
    delta_decay: float = 0.95
    verbose: bool = False
    save_iter_data: bool = False
    save_qp_data: bool = False
    time_limit: float = None
    code_gen: bool = False
    jit: bool = False
    opt_flag: str = 'O0'
    enable_jacobians: bool = True
    solver_name: str = 'DGSQP'
    solver_dir: str = None
    so_name: str = None
    qp_interface: str = 'casadi'
    qp_solver: str = 'osqp'
    hessian_approximation: str = 'none'
    debug: bool = False
    debug_plot: bool = False
    pause_on_plot: bool = False
    save_plot: bool = False
    show_ts: bool = False
    local_pos: bool = False
