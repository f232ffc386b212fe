"""Reference path DGSQP/types.py -> dgsqp_amd.types (field-for-field restatement of the message classes)."""
from dgsqp_amd.types import *  # noqa: F401,F403
from dgsqp_amd.types import (PythonMsg, Position, VehicleActuation, BodyLinearVelocity, BodyAngularVelocity,  # noqa: F401
                             BodyLinearAcceleration, BodyAngularAcceleration, OrientationEuler, ParametricPose,
                             ParametricVelocity, VehicleState, VehiclePrediction)
