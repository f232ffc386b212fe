"""Declarative vehicle models of the hot path.

The reference builds CasADi graphs (DGSQP/dynamics/dynamics_models.py); CasADi
``Function`` objects cannot cross a C boundary, so here a model is a plain
description (type + parameters + track) that ``DGSQP`` lowers into the POD
``dgsqp_problem_t`` consumed by the HIP kernels.  Names, constructor
signatures and the state/input layout follow the reference:

* ``KinematicBicycleConfig`` / ``DynamicBicycleConfig`` / ``MultiAgentModelConfig``
  (DGSQP/dynamics/model_types.py:8-117),
* ``CasadiKinematicBicycleCombined`` (dynamics_models.py:997-1150),
  state ``[x, y, v, e_psi, s, e_y]``, input ``[a, delta]``,
* ``CasadiDynamicBicycleCombined`` (dynamics_models.py:1945-2179),
  state ``[x, y, v_x, v_y, psidot, e_psi, s, e_y]``,
* ``CasadiDecoupledMultiAgentDynamicsModel`` (dynamics_models.py:2482-2632).

The only arithmetic kept on the host is a numpy ``fc`` used by the PID
warm-start plant (the reference's ``model.step``, dynamics_models.py:161-186);
the solver's dynamics live in csrc/.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Tuple

import numpy as np

from .types import PythonMsg, VehicleState, VehiclePrediction


# ---------------------------------------------------------------------------------------------
# configuration dataclasses (model_types.py)
# ---------------------------------------------------------------------------------------------
@dataclass
class ModelConfig(PythonMsg):
    model_name: str = 'model'
    use_mx: bool = False
    enable_jacobians: bool = True
    compute_hessians: bool = False
    verbose: bool = False
    code_gen: bool = False
    jit: bool = True
    opt_flag: str = 'O0'
    install: bool = True
    install_dir: str = '~/.dgsqp_models'


@dataclass
class DynamicsConfig(ModelConfig):
    track_name: str = None
    dt: float = 0.01
    discretization_method: str = 'euler'
    M: int = 10
    noise: bool = False
    noise_cov: np.ndarray = None


@dataclass
class KinematicBicycleConfig(DynamicsConfig):
    wheel_dist_front: float = 0.13
    wheel_dist_rear: float = 0.13
    wheel_dist_center_front: float = 0.1
    wheel_dist_center_rear: float = 0.1
    bump_dist_front: float = 0.15
    bump_dist_rear: float = 0.15
    bump_dist_center: float = 0.1
    bump_dist_top: float = 0.1
    com_height: float = 0.05
    mass: float = 2.366
    drag_coefficient: float = 0.0
    damping_coefficient: float = 0.0
    slip_coefficient: float = 0.0
    rolling_resistance: float = 0.0
    rolling_resistance_exponent: float = 0.5


@dataclass
class DynamicBicycleConfig(DynamicsConfig):
    wheel_dist_front: float = 0.13
    wheel_dist_rear: float = 0.13
    wheel_dist_center_front: float = 0.1
    wheel_dist_center_rear: float = 0.1
    bump_dist_front: float = 0.15
    bump_dist_rear: float = 0.15
    bump_dist_center: float = 0.1
    bump_dist_top: float = 0.1
    com_height: float = 0.05
    mass: float = 2.2187
    gravity: float = 9.81
    yaw_inertia: float = 0.02723
    pitch_inertia: float = 0.03
    roll_inertia: float = 0.03
    drag_coefficient: float = 0.0
    damping_coefficient: float = 0.0
    rolling_resistance: float = 0.0
    rolling_resistance_exponent: float = 0.0
    tire_model: str = 'pacejka'
    drive_wheels: str = 'all'
    wheel_friction: float = 0.9
    pacejka_b_front: float = 5.0
    pacejka_b_rear: float = 5.0
    pacejka_c_front: float = 2.28
    pacejka_c_rear: float = 2.28
    pacejka_d_front: float = None
    pacejka_d_rear: float = None
    linear_bf: float = 1.0
    linear_br: float = 1.0
    simple_slip: bool = False

    def __post_init__(self):
        wb = self.wheel_dist_rear + self.wheel_dist_front
        if self.pacejka_d_front is None:
            self.pacejka_d_front = self.wheel_friction * self.mass * self.gravity * self.wheel_dist_rear / wb
        if self.pacejka_d_rear is None:
            self.pacejka_d_rear = self.wheel_friction * self.mass * self.gravity * self.wheel_dist_front / wb


@dataclass
class UnicycleConfig(DynamicsConfig):
    """model_types.py:107-113."""
    mass: float = 2.366
    damping_coefficient: float = 0.0
    drag_coefficient: float = 0.0
    rolling_resistance: float = 0.0
    rolling_resistance_exponent: float = 0.5


@dataclass
class MultiAgentModelConfig(DynamicsConfig):
    use_mx: bool = False


# ---------------------------------------------------------------------------------------------
# models
# ---------------------------------------------------------------------------------------------
MODEL_KIN_BICYCLE = 0
MODEL_DYN_BICYCLE = 1
MODEL_UNICYCLE = 2
INTEGRATORS = {'euler': 0, 'rk4': 1, 'rk3': 2, 'rk2': 3}


def _ca_abs(x):
    return x if x > 0 else -x


def _ca_sign(x):
    return x / np.sqrt(x * x + 1e-6)


class _FrenetBicycle:
    """Shared host-side plumbing of the two Frenet-frame bicycle models."""
    curvature_model = True
    n_u = 2

    def __init__(self, t0: float, model_config, track=None):
        self.t0 = t0
        self.model_config = model_config
        if getattr(model_config, 'track_name', None) is not None and track is None:
            from .tracks import get_track
            track = get_track(model_config.track_name)
        self.track = track
        self.dt = model_config.dt
        self.M = model_config.M
        self.h = self.dt / self.M
        self.use_mx = model_config.use_mx

    def state2qu(self, state: VehicleState) -> Tuple[np.ndarray, np.ndarray]:
        return self.state2q(state), np.array([state.u.u_a, state.u.u_steer])

    def input2u(self, inp) -> np.ndarray:
        return np.array([inp.u_a, inp.u_steer])

    def u2input(self, inp, u):
        inp.u_a, inp.u_steer = u[0], u[1]

    def step(self, vehicle_state: VehicleState, method: str = 'RK45'):
        """Noise-free plant step over dt with adaptive RK45 (dynamics_models.py:161-186)."""
        from scipy.integrate import solve_ivp
        q, u = self.state2qu(vehicle_state)
        t = vehicle_state.t - self.t0
        sol = solve_ivp(lambda _t, z: self.fc(z, u), (0, self.dt), q, t_eval=[self.dt], method=method)
        q_n = sol.y.squeeze()
        self.qu2state(vehicle_state, q_n, u)
        vehicle_state.t = t + self.dt + self.t0
        if self.track is not None:
            self.track.local_to_global_typed(vehicle_state)


class CasadiKinematicBicycleCombined(_FrenetBicycle):
    """Frenet-frame kinematic bicycle (dynamics_models.py:997-1150)."""
    model_id = MODEL_KIN_BICYCLE
    n_q = 6
    s_idx, ey_idx = 4, 5

    def __init__(self, t0: float, model_config: KinematicBicycleConfig = None, track=None):
        super().__init__(t0, model_config or KinematicBicycleConfig(), track)
        c = self.model_config
        self.L_f, self.L_r, self.m = c.wheel_dist_front, c.wheel_dist_rear, c.mass
        self.c_dr, self.c_da, self.c_s = c.drag_coefficient, c.damping_coefficient, c.slip_coefficient
        self.c_r, self.p_r = c.rolling_resistance, c.rolling_resistance_exponent

    def fc(self, q, u) -> np.ndarray:
        """Continuous dynamics (dynamics_models.py:1046-1070), numpy restatement for the PID plant."""
        _, _, v, epsi, s, ey = q
        ua, us = u
        beta = np.arctan2(np.tan(us) * self.L_r, self.L_f + self.L_r)
        psidot = v / self.L_r * np.sin(beta)
        F_ext = -self.c_da * v - self.c_dr * v * _ca_abs(v) - self.c_s * psidot ** 2
        if self.c_r != 0:
            F_ext -= self.c_r * _ca_abs(v) ** self.p_r * _ca_sign(v)
        c = self.track.get_curvature(s)
        psi_t = self.track.get_tangent_angle(s)
        den = 1 - ey * c
        return np.array([v * np.cos(beta + psi_t + epsi), v * np.sin(beta + psi_t + epsi), ua + F_ext / self.m,
                         psidot - c * v * np.cos(beta + epsi) / den, v * np.cos(beta + epsi) / den,
                         v * np.sin(beta + epsi)])

    def state2q(self, state: VehicleState) -> np.ndarray:
        return np.array([state.x.x, state.x.y, state.v.v_long, state.p.e_psi, state.p.s, state.p.x_tran])

    def q2state(self, state: VehicleState, q):
        state.x.x, state.x.y, state.v.v_long, state.p.e_psi, state.p.s, state.p.x_tran = (float(v) for v in q[:6])

    def qu2state(self, state: VehicleState, q=None, u=None):
        if q is not None:
            self.q2state(state, q)
            if u is not None:
                state.w.w_psi = q[2] / self.L_r * np.sin(np.arctan(np.tan(u[1]) * self.L_f / (self.L_f + self.L_r)))
                state.v.v_tran = state.w.w_psi * self.L_r
        if u is not None:
            state.u.u_a, state.u.u_steer = float(u[0]), float(u[1])

    def qu2prediction(self, prediction: VehiclePrediction, q=None, u=None):
        import array
        if prediction is None:
            prediction = VehiclePrediction()
        if q is not None:
            for name, col in (('x', 0), ('y', 1), ('v_long', 2), ('e_psi', 3), ('s', 4), ('x_tran', 5)):
                setattr(prediction, name, array.array('d', q[:, col]))
            if u is not None:
                psidot = q[:-1, 2] * self.L_r * np.sin(np.arctan(np.tan(u[:, 1]) * self.L_f / (self.L_f + self.L_r)))
                psidot = np.append(psidot, psidot[-1])
                prediction.psidot = array.array('d', psidot)
                prediction.v_tran = array.array('d', psidot * self.L_r)
        if u is not None:
            prediction.u_a = array.array('d', u[:, 0])
            prediction.u_steer = array.array('d', u[:, 1])
        return prediction


class CasadiDynamicBicycleCombined(_FrenetBicycle):
    """Frenet-frame dynamic bicycle with Pacejka / linear tyres (dynamics_models.py:1945-2179)."""
    model_id = MODEL_DYN_BICYCLE
    n_q = 8
    s_idx, ey_idx = 6, 7

    def __init__(self, t0: float, model_config: DynamicBicycleConfig = None, track=None):
        super().__init__(t0, model_config or DynamicBicycleConfig(), track)
        c = self.model_config
        self.L_f, self.L_r, self.m, self.I_z, self.g = c.wheel_dist_front, c.wheel_dist_rear, c.mass, c.yaw_inertia, c.gravity
        self.c_dr, self.c_da = c.drag_coefficient, c.damping_coefficient
        self.c_r, self.p_r = c.rolling_resistance, c.rolling_resistance_exponent
        if c.tire_model not in ('pacejka', 'linear'):
            raise ValueError("Tire model must be 'linear' or 'pacejka'")

    def fc(self, q, u) -> np.ndarray:
        """Continuous dynamics (dynamics_models.py:2008-2062), numpy restatement for the PID plant."""
        c = self.model_config
        _, _, vx, vy, w, epsi, s, ey = q
        ua, us = u
        curv = self.track.get_curvature(s)
        psi_t = self.track.get_tangent_angle(s)
        if c.simple_slip:
            a_f = -np.arctan2(vy + self.L_f * w, vx) + us
        else:
            a_f = -np.arctan2((vy + self.L_f * w) * np.cos(us) - vx * np.sin(us),
                              vx * np.cos(us) + (vy + self.L_f * w) * np.sin(us))
        a_r = -np.arctan2(vy - self.L_r * w, vx)
        if c.tire_model == 'pacejka':
            fyf = c.pacejka_d_front * np.sin(c.pacejka_c_front * np.arctan(c.pacejka_b_front * a_f))
            fyr = c.pacejka_d_rear * np.sin(c.pacejka_c_rear * np.arctan(c.pacejka_b_rear * a_r))
        else:
            fyf = c.linear_bf * self.m * self.g * self.L_r / (self.L_f + self.L_r) * a_f
            fyr = c.linear_br * self.m * self.g * self.L_f / (self.L_f + self.L_r) * a_r
        F_ext = -self.c_da * vx - self.c_dr * vx * _ca_abs(vx)
        if self.c_r != 0:
            F_ext -= self.c_r * _ca_abs(vx) ** self.p_r * _ca_sign(vx)
        ar, af = (ua / 2, ua / 2) if c.drive_wheels == 'all' else (ua, 0.0)
        ax = ar + af * np.cos(us) + (F_ext - fyf * np.sin(us)) / self.m
        ay = af * np.sin(us) + (fyf * np.cos(us) + fyr) / self.m
        az = (self.L_f * fyf * np.cos(us) - self.L_r * fyr) / self.I_z
        den = 1 - ey * curv
        vlon = vx * np.cos(epsi) - vy * np.sin(epsi)
        return np.array([vx * np.cos(epsi + psi_t) - vy * np.sin(epsi + psi_t),
                         vy * np.cos(epsi + psi_t) + vx * np.sin(epsi + psi_t),
                         ax + w * vy, ay - w * vx, az, w - curv * vlon / den, vlon / den,
                         vx * np.sin(epsi) + vy * np.cos(epsi)])

    def state2q(self, state: VehicleState) -> np.ndarray:
        return np.array([state.x.x, state.x.y, state.v.v_long, state.v.v_tran, state.w.w_psi,
                         state.p.e_psi, state.p.s, state.p.x_tran])

    def q2state(self, state: VehicleState, q):
        (state.x.x, state.x.y, state.v.v_long, state.v.v_tran, state.w.w_psi,
         state.p.e_psi, state.p.s, state.p.x_tran) = (float(v) for v in q[:8])

    def qu2state(self, state: VehicleState, q=None, u=None):
        if q is not None:
            self.q2state(state, q)
        if u is not None:
            state.u.u_a, state.u.u_steer = float(u[0]), float(u[1])

    def qu2prediction(self, prediction: VehiclePrediction, q=None, u=None):
        import array
        if prediction is None:
            prediction = VehiclePrediction()
        if q is not None:
            for name, col in (('x', 0), ('y', 1), ('v_long', 2), ('v_tran', 3), ('psidot', 4), ('e_psi', 5),
                              ('s', 6), ('x_tran', 7)):
                setattr(prediction, name, array.array('d', q[:, col]))
        if u is not None:
            prediction.u_a = array.array('d', u[:, 0])
            prediction.u_steer = array.array('d', u[:, 1])
        return prediction


class CasadiKinematicUnicycle(_FrenetBicycle):
    """Global-frame kinematic unicycle (dynamics_models.py:306-390): state [x, y, v, psi], input [F, omega].
    The merge script builds it from a plain ``DynamicsConfig`` (DGSQP_merge_monte_carlo.py:95-101), which has no ``mass``
    field; ``UnicycleConfig`` (mass 2.366) is the config the class reads."""
    model_id = MODEL_UNICYCLE
    curvature_model = False
    n_q = 4

    def __init__(self, t0: float, model_config: UnicycleConfig = None, track=None):
        super().__init__(t0, model_config or UnicycleConfig(), track=track)
        self.m = getattr(self.model_config, 'mass', 2.366)

    def fc(self, q, u) -> np.ndarray:
        return np.array([q[2] * np.cos(q[3]), q[2] * np.sin(q[3]), u[0] / self.m, u[1]])

    def state2q(self, state: VehicleState) -> np.ndarray:
        return np.array([state.x.x, state.x.y, state.v.v_long, state.e.psi])

    def q2state(self, state: VehicleState, q):
        state.x.x, state.x.y, state.v.v_long, state.e.psi = (float(v) for v in q[:4])

    def qu2state(self, state: VehicleState, q=None, u=None):
        if q is not None:
            self.q2state(state, q)
        if u is not None:
            state.u.u_a, state.u.u_steer = float(u[0]), float(u[1])

    def qu2prediction(self, prediction: VehiclePrediction, q=None, u=None):
        if prediction is None:
            prediction = VehiclePrediction()
        if q is not None:
            for name, col in (('x', 0), ('y', 1), ('v_long', 2), ('psi', 3)):
                setattr(prediction, name, array.array('d', q[:, col]))
        if u is not None:
            prediction.u_a = array.array('d', u[:, 0])
            prediction.u_steer = array.array('d', u[:, 1])
        return prediction

    def fd(self, q, u):
        """One step of the model's own discretisation (dynamics_models.py:88-125, rk3 :200-211)."""
        q = np.asarray(q, float)
        meth, h = self.model_config.discretization_method, self.dt / self.M
        if meth == 'euler':
            return q + self.dt * self.fc(q, u)
        for _ in range(self.M):
            if meth == 'rk4':
                a1 = self.fc(q, u); a2 = self.fc(q + h / 2 * a1, u); a3 = self.fc(q + h / 2 * a2, u); a4 = self.fc(q + h * a3, u)
                q = q + h * (a1 + 2 * a2 + 2 * a3 + a4) / 6
            elif meth == 'rk3':
                a1 = h * self.fc(q, u); a2 = h * self.fc(q + a1 / 2, u); a3 = h * self.fc(q - a1 + 2 * a2, u)
                q = q + (a1 + 4 * a2 + a3) / 6
            else:
                a1 = self.fc(q, u); a2 = self.fc(q + h * a1, u)
                q = q + h * (a1 + a2) / 2
        return q


class CasadiDecoupledMultiAgentDynamicsModel:
    """Joint model = concatenation of independent agents; ITS config fixes the
    discretisation used by the solver (dynamics_models.py:2482-2530)."""

    def __init__(self, t0: float, dynamics_models: List[_FrenetBicycle], model_config: MultiAgentModelConfig = None):
        self.t0 = t0
        self.dynamics_models = list(dynamics_models)
        self.model_config = model_config or MultiAgentModelConfig()
        self.n_a = len(self.dynamics_models)
        self.n_q = sum(m.n_q for m in self.dynamics_models)
        self.n_u = sum(m.n_u for m in self.dynamics_models)
        self.dt = self.model_config.dt
        self.M = self.model_config.M
        if self.model_config.discretization_method not in INTEGRATORS:
            raise ValueError('Discretization method of %s not recognized' % self.model_config.discretization_method)
        tracks = {id(m.track) for m in self.dynamics_models}
        if len(tracks) != 1:
            raise ValueError('all agents of a joint model must share one track object')
        self.track = self.dynamics_models[0].track

    def state2q(self, vehicle_states: List[VehicleState]) -> np.ndarray:
        return np.concatenate([m.state2q(s) for m, s in zip(self.dynamics_models, vehicle_states)])

    def state2qu(self, vehicle_states):
        qs, us = zip(*[m.state2qu(s) for m, s in zip(self.dynamics_models, vehicle_states)])
        return np.concatenate(qs), np.concatenate(us)

    def qu2state(self, vehicle_states, q_joint=None, u_joint=None):
        if vehicle_states is None:
            vehicle_states = [VehicleState() for _ in range(self.n_a)]
        qi = ui = 0
        for m, st in zip(self.dynamics_models, vehicle_states):
            q = None if q_joint is None else q_joint[qi:qi + m.n_q]
            u = None if u_joint is None else u_joint[ui:ui + m.n_u]
            m.qu2state(st, q, u)
            qi += m.n_q
            ui += m.n_u
        return vehicle_states

    def qu2prediction(self, state_predictions, q_pred=None, u_pred=None):
        if state_predictions is None:
            state_predictions = [VehiclePrediction() for _ in range(self.n_a)]
        qi = ui = 0
        for m, pr in zip(self.dynamics_models, state_predictions):
            q = None if q_pred is None else q_pred[:, qi:qi + m.n_q]
            u = None if u_pred is None else u_pred[:, ui:ui + m.n_u]
            m.qu2prediction(pr, q, u)
            qi += m.n_q
            ui += m.n_u
        return state_predictions
