"""Reference path DGSQP/solvers/solver_types.py -> dgsqp_amd.solver_types (``PIDParams`` :12, ``DGSQPParams`` :92,
``DGSQPV2Params`` :130).  The parameter classes of the solvers that are out of scope (IBR, ALGAMES, CA_LTV_MPC, PATH-MCP)
are not provided: importing them raises ImportError, as asking for those solvers should."""
from dgsqp_amd.solver_types import ControllerConfig, PIDParams, DGSQPParams, DGSQPV2Params  # noqa: F401
