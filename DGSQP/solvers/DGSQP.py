"""Reference path DGSQP/solvers/DGSQP.py (class ``DGSQP``, :24-507) -> dgsqp_amd.solver.DGSQP (HIP library behind it)."""
from dgsqp_amd.solver import DGSQP  # noqa: F401
