"""PID warm-start controller (reference DGSQP/solvers/PID.py:13-138 ``PID``,
:185-238 ``PIDLaneFollower``).  Host-side only; it produces ``u_ws`` for
``DGSQP.set_warm_start`` exactly the way the Monte-Carlo scripts do
(scripts/DGSQP_ALGAMES_monte_carlo_chicane.py:411-463)."""
from __future__ import annotations

import numpy as np

from .solver_types import PIDParams
from .types import VehicleState


class PID:
    """Scalar PID with rate and magnitude saturation (PID.py:74-138)."""

    def __init__(self, params: PIDParams = None):
        p = params or PIDParams()
        self.dt = p.dt
        self.Kp, self.Ki, self.Kd = p.Kp, p.Ki, p.Kd
        self.int_e_max, self.int_e_min = p.int_e_max, p.int_e_min
        self.u_max, self.u_min, self.du_max, self.du_min = p.u_max, p.u_min, p.du_max, p.du_min
        self.x_ref, self.u_ref = p.x_ref, p.u_ref
        self.u_prev = 0
        self.e = self.de = self.ei = 0.0

    def set_x_ref(self, x_ref: float):
        self.x_ref = x_ref
        self.ei = 0.0
        self.e = 0.0

    def solve(self, x: float, u_prev: float = None):
        if u_prev is None:
            u_prev = 0 if self.u_prev is None else self.u_prev
        e_t = x - self.x_ref
        de_t = (e_t - self.e) / self.dt
        ei_t = min(max(self.ei + e_t * self.dt, self.int_e_min), self.int_e_max)
        u = -(self.Kp * e_t + self.Ki * ei_t + self.Kd * de_t) + self.u_ref
        du = u - u_prev
        if self.du_max is not None:
            du = np.minimum(du, self.du_max)
        if self.du_min is not None:
            du = np.maximum(du, self.du_min)
        u = du + u_prev
        if self.u_max is not None:
            u = np.minimum(u, self.u_max)
        if self.u_min is not None:
            u = np.maximum(u, self.u_min)
        self.e, self.de, self.ei = e_t, de_t, ei_t
        self.u_prev = u
        return u, {'success': True}


class PIDLaneFollower:
    """Speed PID + lane-offset PID (PID.py:185-238): steering tracks
    ``5*(x_tran - lat_ref) + e_psi`` to zero, throttle tracks ``v_long`` to its reference."""

    def __init__(self, dt: float, steer_pid_params: PIDParams = None, speed_pid_params: PIDParams = None):
        if steer_pid_params is None:
            steer_pid_params = PIDParams(dt=dt, Kp=1, Ki=0.0005 / dt, Kd=0, u_min=-0.35, u_max=0.35, du_min=-4 * dt, du_max=4 * dt)
        if speed_pid_params is None:
            speed_pid_params = PIDParams(dt=dt, Kp=1, Ki=0, Kd=0, u_min=-2, u_max=2, du_min=-10 * dt, du_max=10 * dt)
        self.dt = dt
        steer_pid_params.dt = dt
        speed_pid_params.dt = dt
        self.steer_pid = PID(steer_pid_params)
        self.speed_pid = PID(speed_pid_params)
        self.lat_ref = steer_pid_params.x_ref
        self.steer_pid.set_x_ref(0)

    def step(self, vehicle_state: VehicleState, env_state=None):
        vehicle_state.u.u_a, _ = self.speed_pid.solve(vehicle_state.v.v_long)
        vehicle_state.u.u_steer, _ = self.steer_pid.solve(5.0 * (vehicle_state.p.x_tran - self.lat_ref) + 1.0 * vehicle_state.p.e_psi)
