"""Reference path DGSQP/solvers/DGSQP_v2.py (class ``DGSQP`` taking ``DGSQPV2Params``, :52-720) -> dgsqp_amd.solver_v2."""
from dgsqp_amd.solver_v2 import DGSQP  # noqa: F401
