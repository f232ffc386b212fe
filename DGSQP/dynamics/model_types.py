"""Reference path DGSQP/dynamics/model_types.py -> dgsqp_amd.dynamics (``DynamicBicycleConfig`` :34, ``KinematicBicycleConfig``
:88, ``UnicycleConfig`` :116, ``MultiAgentModelConfig`` :124)."""
from dgsqp_amd.dynamics import (ModelConfig, DynamicsConfig, KinematicBicycleConfig, DynamicBicycleConfig, UnicycleConfig,  # noqa: F401
                                MultiAgentModelConfig)
