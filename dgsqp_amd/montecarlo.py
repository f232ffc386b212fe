"""Monte-Carlo drivers: game presets and batched scenario samplers.

Counterpart of the callers of the hot path,
scripts/DGSQP_ALGAMES_monte_carlo_{chicane,curve}.py: the same game
definitions (bounds :80-109, cost weights :111-122, radii :126-129, solver
parameters :161-174), the same rejection sampler for initial conditions
(:384-404) and the same PID-rollout warm start with collision rejection
(:411-467) -- vectorised over a batch so that B scenarios can be handed to
``DGSQP.solve_batch`` at once.  The plant used for the warm start is a
fixed-step RK4 of the continuous model (the reference integrates the same
``fc`` with adaptive RK45 ``solve_ivp``, dynamics_models.py:170).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List

import numpy as np

from .dynamics import (CasadiDecoupledMultiAgentDynamicsModel, CasadiDynamicBicycleCombined,
                       CasadiKinematicBicycleCombined, CasadiKinematicUnicycle, DynamicBicycleConfig,
                       KinematicBicycleConfig, MultiAgentModelConfig, UnicycleConfig)
from .game import (CollisionAvoidance, GoalTrackingCost, InputRateLimits, LaneBoundaries, LaneHalfPlane, RacingCost)
from .solver_types import DGSQPParams, DGSQPV2Params
from .tracks import ChicaneTrack, CurveTrack, get_track
from .types import (BodyAngularVelocity, BodyLinearVelocity, OrientationEuler, ParametricPose, Position,
                    VehicleActuation, VehicleState)


@dataclass
class Game:
    """Everything ``DGSQP(...)`` takes, plus what the sampler needs."""
    joint_model: CasadiDecoupledMultiAgentDynamicsModel
    costs: List[RacingCost]
    agent_constraints: list
    shared_constraints: CollisionAvoidance
    bounds: dict
    params: DGSQPParams
    track: object
    half_width: float
    obs_d: float
    name: str = ''
    sampler: str = 'first_segment'      # 'first_segment' (chicane.py/curve.py/agents.py) or 'circuit' (comp.py)
    second_car_ahead: bool = False      # the sampler also rejects placements with car 2 behind car 1 (ablation.py:387)

    def solver_args(self):
        return (self.joint_model, self.costs, self.agent_constraints, self.shared_constraints, self.bounds, self.params)


def _bounds(half_width, M, u_a=2.1, u_steer=0.436):
    inf = np.inf
    ub = [VehicleState(x=Position(x=inf, y=inf), p=ParametricPose(s=inf, x_tran=half_width, e_psi=inf),
                       e=OrientationEuler(psi=inf), v=BodyLinearVelocity(v_long=inf, v_tran=inf),
                       w=BodyAngularVelocity(w_psi=inf), u=VehicleActuation(u_a=u_a, u_steer=u_steer)) for _ in range(M)]
    lb = [VehicleState(x=Position(x=-inf, y=-inf), p=ParametricPose(s=-inf, x_tran=-half_width, e_psi=-inf),
                       e=OrientationEuler(psi=-inf), v=BodyLinearVelocity(v_long=-inf, v_tran=-inf),
                       w=BodyAngularVelocity(w_psi=-inf), u=VehicleActuation(u_a=-u_a, u_steer=-u_steer)) for _ in range(M)]
    return {'ub': ub, 'lb': lb}


def _track(kind, theta_deg, half_width):
    th = theta_deg * np.pi / 180
    if kind == 'chicane':   # chicane.py:140-149
        return ChicaneTrack(enter_straight_length=1, curve1_length=4, curve1_swept_angle=th, mid_straight_length=1,
                            curve2_length=4, curve2_swept_angle=th, exit_straight_length=5,
                            width=half_width * 2, slack=0.8, mirror=False)
    if kind == 'curve':     # curve.py:140-146
        return CurveTrack(enter_straight_length=1, curve_length=8, curve_swept_angle=th, exit_straight_length=5,
                          width=half_width * 2, slack=0.8, ccw=True)
    raise ValueError(kind)


def kinematic_racing_game(track_kind='chicane', theta_deg=45, N=25, reg=1e-3, M=2, nonmono_ls=True,
                          merit_function='stat_l1') -> Game:
    """2-agent kinematic-bicycle race of chicane.py / curve.py (euler, dt 0.1)."""
    dt, half_width = 0.1, 1.0
    track = _track(track_kind, theta_deg, half_width)
    cfg = lambda: KinematicBicycleConfig(dt=dt, model_name='kinematic_bicycle_cl', noise=False,
                                         discretization_method='euler', wheel_dist_front=0.13, wheel_dist_rear=0.13,
                                         drag_coefficient=0.1, slip_coefficient=0.1, code_gen=False)
    models = [CasadiKinematicBicycleCombined(0, cfg(), track=track) for _ in range(M)]
    joint = CasadiDecoupledMultiAgentDynamicsModel(0, models, MultiAgentModelConfig(
        dt=dt, discretization_method='euler', use_mx=True, code_gen=False, verbose=True, compute_hessians=True))
    chicane = track_kind == 'chicane'
    steer_rate = np.pi if chicane else 4.5          # chicane.py:92-93 vs curve.py:94-95
    r = 0.4 if chicane else 0.2                     # chicane.py:126-127 vs curve.py:128-129
    params = DGSQPParams(solver_name='SQGAMES', dt=dt, N=N, reg=reg, nonmono_ls=nonmono_ls, line_search_iters=50,
                         sqp_iters=50, p_tol=1e-3, d_tol=1e-3, beta=0.01, tau=0.5, verbose=False,
                         merit_function=merit_function)
    cost = lambda: RacingCost(input_weight=(1.0, 1.0), input_rate_weight=(1.0, 1.0), comp_weights=(10.0, 5.0),
                              comp_type='atan', blocking_weight=0, obs_weight=0, obs_r=0.3)
    return Game(joint, [cost() for _ in range(M)],
                [InputRateLimits((10.0, steer_rate), (-10.0, -steer_rate)) for _ in range(M)],
                CollisionAvoidance([r] * M), _bounds(half_width, M), params, track, half_width, 2 * r,
                name=f'kb_{track_kind}_N{N}')


def ablation_racing_game(N=25, nonmono_ls=True, merit_function='stat_l1', theta_deg=90) -> Game:
    """The game of the ablation study, scripts/DGSQP_monte_carlo_ablation.py: curve track with a 90 degree turn (:145-153), horizons
    15 / 20 / 25 (:140), steering-rate limit pi (:101-102, :115-116), radii 0.2 (:134-135), car 2 with blocking weight 1, obstacle
    weight 5 and the competition weights swapped (:123-131), reg 1e-3; the study switches ``nonmono_ls`` and ``merit_function``
    ('stat_l1' with the watchdog: :166-180, 'stat' without: :183-197).  Its sampler also rejects car 2 behind car 1 (:387)."""
    dt, half_width, r, M = 0.1, 1.0, 0.2, 2
    track = _track('curve', theta_deg, half_width)
    cfg = lambda: KinematicBicycleConfig(dt=dt, model_name='kinematic_bicycle_cl', noise=False,
                                         discretization_method='euler', wheel_dist_front=0.13, wheel_dist_rear=0.13,
                                         drag_coefficient=0.1, slip_coefficient=0.1, code_gen=False)
    models = [CasadiKinematicBicycleCombined(0, cfg(), track=track) for _ in range(M)]
    joint = CasadiDecoupledMultiAgentDynamicsModel(0, models, MultiAgentModelConfig(
        dt=dt, discretization_method='euler', use_mx=True, code_gen=False, verbose=True, compute_hessians=True))
    params = DGSQPParams(solver_name='sqgames_all' if nonmono_ls else 'sqgames_none', dt=dt, N=N, reg=1e-3, nonmono_ls=nonmono_ls,
                         merit_function=merit_function, line_search_iters=50, sqp_iters=50, p_tol=1e-3, d_tol=1e-3, beta=0.01, tau=0.5,
                         verbose=False)
    costs = [RacingCost(input_weight=(1.0, 1.0), input_rate_weight=(1.0, 1.0), comp_weights=(10.0, 5.0), comp_type='atan',
                        blocking_weight=0, obs_weight=0, obs_r=0.3),
             RacingCost(input_weight=(1.0, 1.0), input_rate_weight=(1.0, 1.0), comp_weights=(5.0, 10.0), comp_type='atan',
                        blocking_weight=1, obs_weight=5, obs_r=0.3)]
    return Game(joint, costs, [InputRateLimits((10.0, np.pi), (-10.0, -np.pi)) for _ in range(M)],
                CollisionAvoidance([r] * M), _bounds(half_width, M), params, track, half_width, 2 * r,
                name=f'ablation_N{N}_{"nms" if nonmono_ls else "ls"}_{merit_function}', second_car_ahead=True)


def dynamic_racing_game(track_kind='curve', theta_deg=45, N=25, reg=1e-3, rk4_substeps=10, game_def='exact_dynamic', solver='v1') -> Game:
    """BASELINE.json config 2: 2-agent dynamic-bicycle (Pacejka) race on the curve track (curve.py:140-146), rk4 with
    M=10 sub-steps (comparison_study_barc/globals.py:17-18), vehicle of exact_dynamic_game_dynamic.py:26-66.

    ``game_def='exact_dynamic'`` (default) is the reference's own dynamic-bicycle game,
    comparison_study_barc/exact_dynamic_game_dynamic.py with its default ``cost_setting 0``
    (argument_parser.py): input / input-rate weights 1 (:100-105), terminal ``-1*s_a + 5*(s_b - s_a)`` -- linear
    competition term (:146-147, :167-168) --, NO agent constraint rows (``car*_constrs = [None]*(N+1)``, :197,201), obstacle
    rows from k=1 with radii 0.23 (globals.py:12-13, :193, :206-213), boxes |u_a| <= 2.1, |steer| <= 0.436, |e_y| <= H
    (:68-95); rows per stage 8 / 13 / 5.  ``game_def='curve'`` is round 1's synthetic variant -- the costs, rate rows and
    radii of the kinematic curve.py game (progress weight 10, atan competition) on the Pacejka vehicle; a third of its
    scenarios drive the open-loop rollout into |dx/du| ~ 1e17 (profiles/r02_dyn_curve_divergence.txt), it is kept as a
    stress case.  ``solver='v2'``: the DG-SQP v2 parameters of the reference's study (comparison_study_barc/globals.py:27-55:
    reg 1e2 decaying by 0.95, nms with frequency / memory 10, 20 line-search trials, tolerance 1e-4, up to 500 m-steps);
    ``reg`` is then ignored.  ``track_kind='barc'`` puts the game on the study's own L_track_barc circuit."""
    dt, half_width, M = 0.1, 1.0, 2
    if track_kind == 'barc':
        track = get_track('L_track_barc')
        half_width = track.half_width
    else:
        track = _track(track_kind, theta_deg, half_width)
    cfg = lambda: DynamicBicycleConfig(dt=dt, model_name='dynamic_bicycle', noise=False, discretization_method='rk4',
                                       simple_slip=False, tire_model='pacejka', mass=2.2187, yaw_inertia=0.02723,
                                       wheel_friction=0.9, pacejka_b_front=5.0, pacejka_b_rear=5.0,
                                       pacejka_c_front=2.28, pacejka_c_rear=2.28, M=rk4_substeps)
    models = [CasadiDynamicBicycleCombined(0, cfg(), track=track) for _ in range(M)]
    joint = CasadiDecoupledMultiAgentDynamicsModel(0, models, MultiAgentModelConfig(
        dt=dt, discretization_method='rk4', use_mx=False, code_gen=False, verbose=False, compute_hessians=True,
        M=rk4_substeps))
    if solver == 'v2':
        params = DGSQPV2Params(solver_name='DGSQP', dt=dt, N=N, nms=True, nms_frequency=10, nms_memory_size=10, line_search_iters=20,
                               sqp_iters=500, p_tol=1e-4, d_tol=1e-4, reg=1e2, reg_decay=0.95, delta_decay=0.99, merit_decrease=0.01,
                               beta=0.01, tau=0.5, time_limit=600, verbose=False, merit_function='stat_l1',
                               merit_decrease_condition='armijo', merit_parameter=None)
    else:
        params = DGSQPParams(solver_name='DGSQP', dt=dt, N=N, reg=reg, nonmono_ls=True, line_search_iters=50,
                             sqp_iters=50, p_tol=1e-3, d_tol=1e-3, beta=0.01, tau=0.5, verbose=False)
    if game_def == 'exact_dynamic':
        r = 0.23
        cost = lambda: RacingCost(input_weight=(1.0, 1.0), input_rate_weight=(1.0, 1.0), comp_weights=(1.0, 5.0),
                                  comp_type='linear')
        rows = [None] * M
    elif game_def == 'curve':
        r = 0.2
        cost = lambda: RacingCost(input_weight=(1.0, 1.0), input_rate_weight=(1.0, 1.0), comp_weights=(10.0, 5.0),
                                  comp_type='atan')
        rows = [InputRateLimits((10.0, 4.5), (-10.0, -4.5)) for _ in range(M)]
    else:
        raise ValueError(game_def)
    return Game(joint, [cost() for _ in range(M)], rows, CollisionAvoidance([r] * M), _bounds(half_width, M), params, track,
                half_width, 2 * r, name=f'dyn_{track_kind}_N{N}' + ('' if game_def == 'exact_dynamic' else '_' + game_def) + ('_v2' if solver == 'v2' else ''),
                sampler='circuit' if track_kind == 'barc' else 'first_segment')


def barc_racing_game(N=15, M=2, reg=0.0) -> Game:
    """Head-to-head race on the L_track_barc circuit, scripts/DGSQP_comp_monte_carlo.py: kinematic bicycles (euler,
    dt 0.1, :79-97), lateral bounds +-(H-0.1) (:108-134), weights (:141-146), radii 0.2 (:149-150), solver parameters with
    reg=0 (:157-171), rate limits (:254-261), obstacle rows from k=1 (:285-292).  M=3 with N=25 is BASELINE configs[2]
    (its 150 unknowns exceed what the LDS-resident device layout holds; M=3 fits up to N=16)."""
    dt = 0.1
    track = get_track('L_track_barc')
    H = track.half_width
    cfg = lambda: KinematicBicycleConfig(dt=dt, model_name='kinematic_bicycle', noise=False,
                                         discretization_method='euler', wheel_dist_front=0.13, wheel_dist_rear=0.13,
                                         code_gen=False)
    models = [CasadiKinematicBicycleCombined(0, cfg(), track=track) for _ in range(M)]
    joint = CasadiDecoupledMultiAgentDynamicsModel(0, models, MultiAgentModelConfig(
        dt=dt, discretization_method='euler', use_mx=False, code_gen=False, verbose=True, compute_hessians=True))
    r = 0.2
    params = DGSQPParams(solver_name='DGSQP', dt=dt, N=N, reg=reg, nonmono_ls=True, line_search_iters=50, sqp_iters=50,
                         p_tol=1e-3, d_tol=1e-3, beta=0.01, tau=0.5, verbose=False, merit_function='stat_l1')
    cost = lambda: RacingCost(input_weight=(0.1, 0.1), input_rate_weight=(1.0, 1.0), comp_weights=(0.0, 1.0),
                              comp_type='atan')
    return Game(joint, [cost() for _ in range(M)], [InputRateLimits((10.0, 4.5), (-10.0, -4.5)) for _ in range(M)],
                CollisionAvoidance([r] * M), _bounds(H - 0.1, M), params, track, H - 0.1, 2 * r,
                name=f'kb_barc_M{M}_N{N}', sampler='circuit')


def f1_racing_game(N=50, M=2, reg=1e-3, model='kinematic', rk4_substeps=10) -> Game:
    """BASELINE.json config 4: head-to-head race on the F1 track (``f1_austin_tenth_scale``, a ``CasadiBSplineTrack`` in the
    reference: track_lib.get_track :112-113, scripts/comparison_study_f1), long horizon N = 50.  Costs, rate limits, radii and
    solver parameters are those of the circuit race DGSQP_comp_monte_carlo.py (as ``barc_racing_game``); lateral bounds
    +-(half width - 0.1); ``model='dynamic'`` puts the Pacejka bicycle of the comparison studies on it (rk4)."""
    dt = 0.1
    track = get_track('f1_austin_tenth_scale')
    H = track.half_width
    if model == 'kinematic':
        cfg = lambda: KinematicBicycleConfig(dt=dt, model_name='kinematic_bicycle', noise=False, discretization_method='euler',
                                             wheel_dist_front=0.13, wheel_dist_rear=0.13, code_gen=False)
        models = [CasadiKinematicBicycleCombined(0, cfg(), track=track) for _ in range(M)]
        method, sub = 'euler', 1
    else:
        cfg = lambda: DynamicBicycleConfig(dt=dt, model_name='dynamic_bicycle', noise=False, discretization_method='rk4',
                                           simple_slip=False, tire_model='pacejka', mass=2.2187, yaw_inertia=0.02723,
                                           wheel_friction=0.9, pacejka_b_front=5.0, pacejka_b_rear=5.0,
                                           pacejka_c_front=2.28, pacejka_c_rear=2.28, M=rk4_substeps)
        models = [CasadiDynamicBicycleCombined(0, cfg(), track=track) for _ in range(M)]
        method, sub = 'rk4', rk4_substeps
    joint = CasadiDecoupledMultiAgentDynamicsModel(0, models, MultiAgentModelConfig(
        dt=dt, discretization_method=method, use_mx=False, code_gen=False, verbose=False, compute_hessians=True, M=sub))
    r = 0.2
    params = DGSQPParams(solver_name='DGSQP', dt=dt, N=N, reg=reg, nonmono_ls=True, line_search_iters=50, sqp_iters=50,
                         p_tol=1e-3, d_tol=1e-3, beta=0.01, tau=0.5, verbose=False, merit_function='stat_l1')
    cost = lambda: RacingCost(input_weight=(0.1, 0.1), input_rate_weight=(1.0, 1.0), comp_weights=(0.0, 1.0), comp_type='atan')
    return Game(joint, [cost() for _ in range(M)], [InputRateLimits((10.0, 4.5), (-10.0, -4.5)) for _ in range(M)],
                CollisionAvoidance([r] * M), _bounds(H - 0.1, M), params, track, H - 0.1, 2 * r,
                name=f'{"kb" if model == "kinematic" else "dyn"}_f1_M{M}_N{N}', sampler='circuit')


_MERGE_GOAL_X = (4.0, 4.5, 4.25, 4.75, 5.25, 5.0)          # merge.py:85-87 for the first three cars
_MERGE_X_NOM = (0.0, 0.5, 0.25, 1.0, 1.5, -0.35)          # merge.py:430,443,456: cars 1, 2 on the straight lane, car 3 on the ramp


def merge_game(N=20, reg=0.0, M=3) -> Game:
    """Highway merge of scripts/DGSQP_merge_monte_carlo.py: kinematic unicycles (rk3, one sub-step, dt 0.1, :95-123), cars on
    the straight lane and on the ramp (lane geometry :40-74), goal-tracking costs (:253-303), lane rows at every stage
    (:316-342), obstacle rows for every pair from k=1 (:344-356), |v| <= 2, |F| <= 2, |omega| <= 4.5 (:126-159), radii 0.1
    (:162-164), solver parameters with reg=0 (:176-190).  No warm start: the script solves from zero inputs.
    ``M = 3`` is the script; up to 6 cars (BASELINE configs[4]'s family) repeat its pattern -- every third car on the ramp."""
    dt = 0.1
    cfg = lambda: UnicycleConfig(dt=dt, model_name='kinematic_bicycle', noise=False, discretization_method='rk3', code_gen=False, M=1)
    models = [CasadiKinematicUnicycle(0, cfg()) for _ in range(M)]
    joint = CasadiDecoupledMultiAgentDynamicsModel(0, models, MultiAgentModelConfig(
        dt=dt, discretization_method='rk3', use_mx=False, code_gen=False, verbose=True, compute_hessians=True, M=1))
    lw, mw, mp, th, r = 0.3, 0.3, 1.5, np.pi / 12, 0.1
    ns, nm = (0.0, 1.0), (-np.sin(th), np.cos(th))
    x1, x3 = (0.0, lw), (0.0, 0.0)
    x6 = (mp + lw / np.tan(th), lw)
    x7 = (mp + mw / np.sin(th), 0.0)
    neg = lambda v: (-v[0], -v[1])
    straight = lambda: LaneBoundaries([LaneHalfPlane(n_lo=ns, anchor=x1, r=r), LaneHalfPlane(n_lo=neg(ns), anchor=x3, r=r)])
    ramp = lambda: LaneBoundaries([LaneHalfPlane(n_lo=nm, n_hi=ns, brk=x6[0], anchor=x6, r=r),
                                   LaneHalfPlane(n_lo=neg(nm), n_hi=neg(ns), brk=x7[0], anchor=x7, r=r)])
    costs = [GoalTrackingCost(input_weight=(0.1, 0.1), state_weight=(1.0, 10.0, 1.0, 1.0), goal=(_MERGE_GOAL_X[i], 0.15, 0.3, 0.0),
                              terminal_multiplier=10.0) for i in range(M)]
    inf = np.inf
    ub = [VehicleState(x=Position(x=inf, y=inf), e=OrientationEuler(psi=inf), v=BodyLinearVelocity(v_long=2.0, v_tran=inf),
                       w=BodyAngularVelocity(w_psi=inf), u=VehicleActuation(u_a=2.0, u_steer=4.5)) for _ in range(M)]
    lb = [VehicleState(x=Position(x=-inf, y=-inf), e=OrientationEuler(psi=-inf), v=BodyLinearVelocity(v_long=-2.0, v_tran=-inf),
                       w=BodyAngularVelocity(w_psi=-inf), u=VehicleActuation(u_a=-2.0, u_steer=-4.5)) for _ in range(M)]
    params = DGSQPParams(solver_name='DGSQP', dt=dt, N=N, reg=reg, merit_function='stat_l1', nonmono_ls=True,
                         line_search_iters=50, sqp_iters=50, p_tol=1e-3, d_tol=1e-3, beta=0.01, tau=0.5, verbose=False)
    return Game(joint, costs, [ramp() if i % 3 == 2 else straight() for i in range(M)], CollisionAvoidance([0.1] * M),
                {'ub': ub, 'lb': lb}, params, None, 0.0, 0.2, name=f'merge_N{N}' if M == 3 else f'merge_M{M}_N{N}', sampler='merge')


# ---------------------------------------------------------------------------------------------
# vectorised plant for the PID warm start
# ---------------------------------------------------------------------------------------------
def _track_lookup(track, s):
    return track.lookup(s)


def _fc_batch(model, q, u):
    """Continuous dynamics for a batch: q [B, n_q], u [B, 2] (same equations as model.fc)."""
    c = model.model_config
    ua, us = u[:, 0], u[:, 1]
    absv = lambda x: np.where(x > 0, x, -x)
    if model.model_id == 0:
        v, epsi, s, ey = q[:, 2], q[:, 3], q[:, 4], q[:, 5]
        beta = np.arctan2(np.tan(us) * model.L_r, model.L_f + model.L_r)
        psidot = v / model.L_r * np.sin(beta)
        F = -model.c_da * v - model.c_dr * v * absv(v) - model.c_s * psidot ** 2
        curv, psi_t = _track_lookup(model.track, s)
        den = 1 - ey * curv
        return np.stack([v * np.cos(beta + psi_t + epsi), v * np.sin(beta + psi_t + epsi), ua + F / model.m,
                         psidot - curv * v * np.cos(beta + epsi) / den, v * np.cos(beta + epsi) / den,
                         v * np.sin(beta + epsi)], axis=1)
    vx, vy, w, epsi, s, ey = q[:, 2], q[:, 3], q[:, 4], q[:, 5], q[:, 6], q[:, 7]
    curv, psi_t = _track_lookup(model.track, s)
    a_f = -np.arctan2((vy + model.L_f * w) * np.cos(us) - vx * np.sin(us), vx * np.cos(us) + (vy + model.L_f * w) * np.sin(us))
    a_r = -np.arctan2(vy - model.L_r * w, vx)
    fyf = c.pacejka_d_front * np.sin(c.pacejka_c_front * np.arctan(c.pacejka_b_front * a_f))
    fyr = c.pacejka_d_rear * np.sin(c.pacejka_c_rear * np.arctan(c.pacejka_b_rear * a_r))
    F = -model.c_da * vx - model.c_dr * vx * absv(vx)
    ar, af = (ua / 2, ua / 2) if c.drive_wheels == 'all' else (ua, 0 * ua)
    ax = ar + af * np.cos(us) + (F - fyf * np.sin(us)) / model.m
    ay = af * np.sin(us) + (fyf * np.cos(us) + fyr) / model.m
    az = (model.L_f * fyf * np.cos(us) - model.L_r * fyr) / model.I_z
    den = 1 - ey * curv
    vlon = vx * np.cos(epsi) - vy * np.sin(epsi)
    return np.stack([vx * np.cos(epsi + psi_t) - vy * np.sin(epsi + psi_t), vy * np.cos(epsi + psi_t) + vx * np.sin(epsi + psi_t),
                     ax + w * vy, ay - w * vx, az, w - curv * vlon / den, vlon / den,
                     vx * np.sin(epsi) + vy * np.cos(epsi)], axis=1)


def _plant_step(model, q, u, dt, substeps=10):
    h = dt / substeps
    for _ in range(substeps):
        k1 = _fc_batch(model, q, u)
        k2 = _fc_batch(model, q + h / 2 * k1, u)
        k3 = _fc_batch(model, q + h / 2 * k2, u)
        k4 = _fc_batch(model, q + h * k3, u)
        q = q + h * (k1 + 2 * k2 + 2 * k3 + k4) / 6
    return q


def pid_warm_start(model, q0, N, dt, u_max=(2.1, 0.436), du=(10.0, 4.5)):
    """PID lane follower rollout (chicane.py:411-447) for a batch.  q0 [B, n_q] -> (q_ws [B,N+1,n_q], u_ws [B,N,2]).
    Gains: speed Kp=1; steer Kp=1, Ki=0.005 on 5*(e_y - e_y0) + e_psi; du/u saturation as the script passes them."""
    B = q0.shape[0]
    v_idx = 2
    epsi_idx, ey_idx = (3, 5) if model.model_id == 0 else (5, 7)
    v_ref, lat_ref = q0[:, v_idx].copy(), q0[:, ey_idx].copy()
    ei = np.zeros(B)
    u_prev = np.zeros((B, 2))
    q = q0.copy()
    qs, us = [q0.copy()], []
    for _ in range(N):
        ua = -(1.0 * (q[:, v_idx] - v_ref))
        e = 5.0 * (q[:, ey_idx] - lat_ref) + q[:, epsi_idx]
        ei = np.clip(ei + e * dt, -100, 100)
        ust = -(1.0 * e + 0.005 * ei)
        u = np.stack([ua, ust], axis=1)
        for j in range(2):                       # PID.solve saturation order: rate first, then magnitude
            d = np.clip(u[:, j] - u_prev[:, j], -du[j], du[j])
            u[:, j] = np.clip(d + u_prev[:, j], -u_max[j], u_max[j])
        u_prev = u
        q = _plant_step(model, q, u, dt)
        qs.append(q.copy())
        us.append(u.copy())
    return np.stack(qs, axis=1), np.stack(us, axis=1)


def sample_scenarios(game: Game, B: int, seed: int = 1, max_rounds: int = 200, solver=None):
    """Rejection-sample B two-agent scenarios (chicane.py:384-404, :465-467).
    Returns x0 [B, n_q] and u_ws [B, N, n_u] (time-major, as ``set_warm_start`` expects).
    With ``solver`` (a ``dgsqp_amd.solver.DGSQP`` of this game) the PID warm starts and the collision check run on the
    device (``dgsqp_pid_warm_start_batch``); the random draws are the same either way."""
    if game.sampler == 'circuit':
        return _sample_scenarios_circuit(game, B, seed, max_rounds)
    if game.sampler == 'merge':
        return _sample_scenarios_merge(game, B, seed)
    if game.joint_model.n_a != 2:
        return _sample_scenarios_independent(game, B, seed, max_rounds)
    rng = np.random.default_rng(seed)
    track, hw, obs_d = game.track, game.half_width, game.obs_d
    N, dt = game.params.N, game.params.dt
    first_seg_len = track.cl_segs[0, 0]
    models = game.joint_model.dynamics_models
    x0s, uws = [], []
    have = 0
    for _ in range(max_rounds):
        if have >= B:
            break
        n = max(64, 2 * (B - have))
        s1 = np.maximum(0.1, rng.random(n) * first_seg_len)
        ey1 = rng.random(n) * hw * 2 - hw
        v1 = rng.random(n) + 2
        d = 2 * np.pi * rng.random(n)
        s2 = s1 + 1.2 * obs_d * np.cos(d)
        ey2 = ey1 + 1.2 * obs_d * np.sin(d)
        v2 = rng.random(n) + 2
        ok = (s2 >= 0) & (np.abs(ey2) <= hw)
        if game.second_car_ahead:
            ok &= s2 >= s1
        s1, ey1, v1, s2, ey2, v2 = (a[ok] for a in (s1, ey1, v1, s2, ey2, v2))
        q0 = []
        for mdl, s, ey, v in ((models[0], s1, ey1, v1), (models[1], s2, ey2, v2)):
            xy = np.array([track.local_to_global((si, ei_, 0.0))[:2] for si, ei_ in zip(s, ey)]).reshape(-1, 2)
            q = np.zeros((len(s), mdl.n_q))
            q[:, 0], q[:, 1], q[:, 2] = xy[:, 0], xy[:, 1], v
            q[:, mdl.s_idx], q[:, mdl.ey_idx] = s, ey
            q0.append(q)
        rl = game.agent_constraints[0]
        du = (10.0, 4.5) if rl is None else tuple(rl.rate_max)
        if solver is not None:
            dev = solver.pid_warm_start_batch(np.concatenate(q0, axis=1), du_max=du)
            keep = ~dev['collide']
            x0s.append(np.concatenate([q0[0][keep], q0[1][keep]], axis=1))
            uws.append(dev['u_ws'][keep])
            have += int(keep.sum())
            continue
        q_ws, u_ws = zip(*[pid_warm_start(m, q, N, dt, du=du) for m, q in zip(models, q0)])
        dist = np.linalg.norm(q_ws[0][:, :, :2] - q_ws[1][:, :, :2], axis=2)
        keep = ~(dist < obs_d).any(axis=1)           # check_collision (chicane.py:38-43)
        x0s.append(np.concatenate([q0[0][keep], q0[1][keep]], axis=1))
        uws.append(np.concatenate([u_ws[0][keep], u_ws[1][keep]], axis=2))
        have += int(keep.sum())
    x0 = np.concatenate(x0s)[:B]
    u = np.concatenate(uws)[:B]
    if x0.shape[0] < B:
        raise RuntimeError('sampler did not produce enough collision-free scenarios')
    return np.ascontiguousarray(x0), np.ascontiguousarray(u)


def _sample_scenarios_independent(game: Game, B: int, seed: int, max_rounds: int):
    """M-agent sampler of scripts/DGSQP_monte_carlo_agents.py:262-308: every agent is placed independently on the first
    track segment (s, e_y uniform, v in [2,3]), PID warm starts, rejection on any pairwise collision along the horizon."""
    rng = np.random.default_rng(seed)
    track, hw = game.track, game.half_width
    N, dt = game.params.N, game.params.dt
    first_seg_len = track.cl_segs[0, 0]
    models = game.joint_model.dynamics_models
    M = len(models)
    radii = list(game.shared_constraints.radii) if game.shared_constraints is not None else [game.obs_d / 2] * M
    x0s, uws = [], []
    have = 0
    for _ in range(max_rounds):
        if have >= B:
            break
        n = max(64, 4 * (B - have))
        q0, q_ws, u_ws = [], [], []
        for mdl in models:
            s_ = np.maximum(0.1, rng.random(n) * first_seg_len)
            ey = rng.random(n) * hw * 2 - hw
            v = rng.random(n) + 2
            xy = np.array([track.local_to_global((si, ei_, 0.0))[:2] for si, ei_ in zip(s_, ey)]).reshape(-1, 2)
            q = np.zeros((n, mdl.n_q))
            q[:, 0], q[:, 1], q[:, 2] = xy[:, 0], xy[:, 1], v
            q[:, mdl.s_idx], q[:, mdl.ey_idx] = s_, ey
            rl = game.agent_constraints[0]
            du = (10.0, 4.5) if rl is None else tuple(rl.rate_max)
            qw, uw = pid_warm_start(mdl, q, N, dt, du=du)
            q0.append(q); q_ws.append(qw); u_ws.append(uw)
        keep = np.ones(n, bool)
        for i in range(M):
            for j in range(i + 1, M):
                dist = np.linalg.norm(q_ws[i][:, :, :2] - q_ws[j][:, :, :2], axis=2)
                keep &= ~(dist < radii[i] + radii[j]).any(axis=1)
        x0s.append(np.concatenate([q[keep] for q in q0], axis=1))
        uws.append(np.concatenate([u[keep] for u in u_ws], axis=2))
        have += int(keep.sum())
    x0 = np.concatenate(x0s)[:B]
    u = np.concatenate(uws)[:B]
    if x0.shape[0] < B:
        raise RuntimeError('sampler did not produce enough collision-free scenarios')
    return np.ascontiguousarray(x0), np.ascontiguousarray(u)


def _sample_scenarios_circuit(game: Game, B: int, seed: int, max_rounds: int):
    """Sampler of scripts/DGSQP_comp_monte_carlo.py:365-382 (the script seeds with 0): the first car anywhere on the
    circuit, every further car within 1.2 obstacle distances along the track, speeds within 25 %, headings within 5 deg;
    PID warm starts; rejection on a collision along the warm start (:448)."""
    rng = np.random.default_rng(seed)
    track, hw, obs_d = game.track, game.half_width, game.obs_d
    N, dt = game.params.N, game.params.dt
    L = track.track_length
    models = game.joint_model.dynamics_models
    M = len(models)
    radii = list(game.shared_constraints.radii)
    x0s, uws = [], []
    have = 0
    rl = game.agent_constraints[0]
    du = (10.0, 4.5) if rl is None else tuple(rl.rate_max)
    for _ in range(max_rounds):
        if have >= B:
            break
        n = max(64, 2 * (B - have))
        s1 = L * rng.random(n)
        place = [(s1, hw * (2 * rng.random(n) - 1), 2.0 + (rng.random(n) - 0.5), 5.0 * (2 * rng.random(n) - 1) * np.pi / 180)]
        for _a in range(1, M):
            place.append((s1 + 1.2 * obs_d * (2 * rng.random(n) - 1), hw * (2 * rng.random(n) - 1),
                          (1 + 0.25 * (2 * rng.random(n) - 1)) * place[0][2], 5.0 * (2 * rng.random(n) - 1) * np.pi / 180))
        q0, q_ws, u_ws = [], [], []
        for mdl, (s_, ey, v, ep) in zip(models, place):
            xy = np.array([track.local_to_global((si, ei_, pi_))[:2] for si, ei_, pi_ in zip(s_, ey, ep)]).reshape(-1, 2)
            q = np.zeros((n, mdl.n_q))
            q[:, 0], q[:, 1], q[:, 2] = xy[:, 0], xy[:, 1], v
            q[:, mdl.s_idx], q[:, mdl.ey_idx], q[:, 3 if mdl.model_id == 0 else 5] = s_, ey, ep      # e_psi: state 3 (kinematic) / 5 (dynamic)
            qw, uw = pid_warm_start(mdl, q, N, dt, du=du)
            q0.append(q); q_ws.append(qw); u_ws.append(uw)
        keep = np.ones(n, bool)
        for i in range(M):
            for j in range(i + 1, M):
                dist = np.linalg.norm(q_ws[i][:, :, :2] - q_ws[j][:, :, :2], axis=2)
                keep &= ~(dist < radii[i] + radii[j]).any(axis=1)
        x0s.append(np.concatenate([q[keep] for q in q0], axis=1))
        uws.append(np.concatenate([u[keep] for u in u_ws], axis=2))
        have += int(keep.sum())
    x0 = np.concatenate(x0s)[:B]
    u = np.concatenate(uws)[:B]
    if x0.shape[0] < B:
        raise RuntimeError('sampler did not produce enough collision-free scenarios')
    return np.ascontiguousarray(x0), np.ascontiguousarray(u)


def _sample_scenarios_merge(game: Game, B: int, seed: int):
    """Vectorised form of ``_sample_scenarios_merge_scalar`` (same draws in the same order from the same generator, the same rk3
    arithmetic on arrays, the first B accepted candidates): 65,536 six-car scenarios in a second instead of two minutes."""
    rng = np.random.default_rng(seed)
    N = game.params.N
    models = game.joint_model.dynamics_models
    M = len(models)
    radii = list(game.shared_constraints.radii)
    mw, mp, th = 0.3, 1.5, np.pi / 12
    x5, x7 = mp, mp + mw / np.sin(th)
    out, have = [], 0
    while have < B:
        K = max(256, 2 * (B - have))
        r = rng.random((K, M, 4))                       # candidate k, car i: the four draws of the scalar sampler, in its order
        q = np.zeros((K, M, 4))
        for i in range(M):
            x_nom = _MERGE_X_NOM[i]
            if i % 3 != 2:
                q[:, i, 0] = x_nom + 0.5 * r[:, i, 0] - 0.25
                q[:, i, 1] = 0.15 + 0.1 * r[:, i, 1] - 0.05
                q[:, i, 2] = 0.3 * (1 + 0.06 * r[:, i, 2] - 0.03)
                q[:, i, 3] = (5 * r[:, i, 3] - 2.5) * np.pi / 180
            else:
                y_nom = -((x7 + x5) / 2 - x_nom) * np.tan(th)
                s_rand, ey_rand = 0.5 * r[:, i, 0] - 0.25, 0.1 * r[:, i, 1] - 0.05
                q[:, i, 0] = x_nom + s_rand * np.cos(th) - ey_rand * np.sin(th)
                q[:, i, 1] = y_nom + s_rand * np.sin(th) + ey_rand * np.cos(th)
                q[:, i, 2] = 0.3 * (1 + 0.06 * r[:, i, 2] - 0.03)
                q[:, i, 3] = np.pi / 12 + (5 * r[:, i, 3] - 2.5) * np.pi / 180
        traj = []
        for a, mdl in enumerate(models):
            qa = np.zeros((K, 4)) if (M == 3 and a == 2) else q[:, a, :].copy()
            pts = [qa[:, :2].copy()]
            for _ in range(N):
                qa = _unicycle_fd_zero_input(mdl, qa)
                pts.append(qa[:, :2].copy())
            traj.append(np.stack(pts, axis=1))           # [K, N + 1, 2]
        hit = np.zeros(K, bool)
        for i in range(M):
            for j in range(i + 1, M):
                hit |= (np.linalg.norm(traj[i] - traj[j], axis=2) < radii[i] + radii[j]).any(axis=1)
        acc = q[~hit].reshape(-1, 4 * M)
        out.append(acc)
        have += len(acc)
    return np.ascontiguousarray(np.concatenate(out)[:B]), np.zeros((B, N, 2 * M))


def _unicycle_fd_zero_input(mdl, q):
    """``CasadiKinematicUnicycle.fd`` (dynamics_models.py:88-125, rk3 :200-211) on rows of states with u = 0 -- the same expressions
    evaluated on arrays, so that a row equals the scalar ``mdl.fd(q_row, 0)`` bit for bit."""
    meth, h = mdl.model_config.discretization_method, mdl.dt / mdl.M

    def fc(z):
        return np.stack([z[:, 2] * np.cos(z[:, 3]), z[:, 2] * np.sin(z[:, 3]), np.zeros(len(z)) / mdl.m, np.zeros(len(z))], axis=1)
    if meth == 'euler':
        return q + mdl.dt * fc(q)
    for _ in range(mdl.M):
        if meth == 'rk4':
            a1 = fc(q); a2 = fc(q + h / 2 * a1); a3 = fc(q + h / 2 * a2); a4 = fc(q + h * a3)
            q = q + h * (a1 + 2 * a2 + 2 * a3 + a4) / 6
        elif meth == 'rk3':
            a1 = h * fc(q); a2 = h * fc(q + a1 / 2); a3 = h * fc(q - a1 + 2 * a2)
            q = q + (a1 + 4 * a2 + a3) / 6
        else:
            a1 = fc(q); a2 = fc(q + h * a1)
            q = q + h * (a1 + a2) / 2
    return q


def _sample_scenarios_merge_scalar(game: Game, B: int, seed: int):
    """Sampler of scripts/DGSQP_merge_monte_carlo.py:421-480 (seed 1 there): cars on the straight lane around their nominal
    x, every third car on the ramp; zero warm start; rejection if the zero-input trajectories collide.  For the script's
    three cars its quirk is reproduced -- car 3's check trajectory is left at the origin (``car2_q_ws[0]`` is assigned
    twice, :471-472) --, it decides which samples pass; with more cars every trajectory is checked."""
    rng = np.random.default_rng(seed)
    N = game.params.N
    models = game.joint_model.dynamics_models
    M = len(models)
    radii = list(game.shared_constraints.radii)
    mw, mp, th = 0.3, 1.5, np.pi / 12
    x5, x7 = mp, mp + mw / np.sin(th)
    out = []
    while len(out) < B:
        q = []
        for i in range(M):
            x_nom = _MERGE_X_NOM[i]
            if i % 3 != 2:
                q.append(np.array([x_nom + 0.5 * rng.random() - 0.25, 0.15 + 0.1 * rng.random() - 0.05,
                                   0.3 * (1 + 0.06 * rng.random() - 0.03), (5 * rng.random() - 2.5) * np.pi / 180]))
            else:
                y_nom = -((x7 + x5) / 2 - x_nom) * np.tan(th)
                s_rand, ey_rand = 0.5 * rng.random() - 0.25, 0.1 * rng.random() - 0.05
                q.append(np.array([x_nom + s_rand * np.cos(th) - ey_rand * np.sin(th), y_nom + s_rand * np.sin(th) + ey_rand * np.cos(th),
                                   0.3 * (1 + 0.06 * rng.random() - 0.03), np.pi / 12 + (5 * rng.random() - 2.5) * np.pi / 180]))
        traj = []
        for a, mdl in enumerate(models):
            qa = [np.zeros(4) if (M == 3 and a == 2) else q[a].copy()]
            for _ in range(N):
                qa.append(mdl.fd(qa[-1], np.zeros(2)))
            traj.append(np.array(qa))
        hit = False
        for i in range(M):
            for j in range(i + 1, M):
                hit |= bool((np.linalg.norm(traj[i][:, :2] - traj[j][:, :2], axis=1) < radii[i] + radii[j]).any())
        if not hit:
            out.append(np.concatenate(q))
    return np.ascontiguousarray(np.array(out)), np.zeros((B, N, 2 * M))
