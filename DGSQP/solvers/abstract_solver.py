"""Reference path DGSQP/solvers/abstract_solver.py (``AbstractSolver`` :9) -> dgsqp_amd.solver.AbstractSolver."""
from dgsqp_amd.solver import AbstractSolver  # noqa: F401
