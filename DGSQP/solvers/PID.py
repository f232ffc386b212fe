"""Reference path DGSQP/solvers/PID.py (``PID`` :13-138, ``PIDLaneFollower`` :185-238) -> dgsqp_amd.pid."""
from dgsqp_amd.pid import PID, PIDLaneFollower  # noqa: F401
