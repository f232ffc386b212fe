"""dgsqp_amd -- MI355X-native batched Dynamic-Game-SQP (the Monte-Carlo hot path of
zhu-edward/DGSQP behind the reference's DGSQPParams / solve() surface)."""
from .solver_types import DGSQPParams, PIDParams  # noqa: F401
from .types import (VehicleState, VehiclePrediction, VehicleActuation, Position, ParametricPose,  # noqa: F401
                    OrientationEuler, BodyLinearVelocity, BodyAngularVelocity)
from .game import (RacingCost, InputRateLimits, CollisionAvoidance, GoalTrackingCost, LaneBoundaries,  # noqa: F401
                   LaneHalfPlane)
from .dynamics import (KinematicBicycleConfig, DynamicBicycleConfig, UnicycleConfig, MultiAgentModelConfig,  # noqa: F401
                       CasadiKinematicBicycleCombined, CasadiDynamicBicycleCombined, CasadiKinematicUnicycle,
                       CasadiDecoupledMultiAgentDynamicsModel)
from .tracks import CurveTrack, ChicaneTrack, StraightTrack, RadiusArclengthTrack, get_track  # noqa: F401


def __getattr__(name):
    if name == 'DGSQP':          # lazy: importing the solver does not require the HIP library
        from .solver import DGSQP
        return DGSQP
    raise AttributeError(name)
