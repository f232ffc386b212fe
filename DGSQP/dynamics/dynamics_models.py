"""Reference path DGSQP/dynamics/dynamics_models.py -> dgsqp_amd.dynamics: the declarative model objects that stand in for
the CasADi models on the hot path (``CasadiKinematicUnicycle`` :306, ``CasadiKinematicBicycleCombined`` :997,
``CasadiDynamicBicycleCombined`` :1945, ``CasadiDecoupledMultiAgentDynamicsModel`` :2482).  The other model classes of the
reference file (Frenet-only, progress-augmented, point mass) belong to solvers that are out of scope and are not provided."""
from dgsqp_amd.dynamics import (CasadiKinematicBicycleCombined, CasadiDynamicBicycleCombined, CasadiKinematicUnicycle,  # noqa: F401
                                CasadiDecoupledMultiAgentDynamicsModel)
