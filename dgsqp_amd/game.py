"""Declarative game definition: the cost / constraint families the reference's
Monte-Carlo scripts build symbolically with CasADi, as plain parameter records.

This is the one intentional deviation from the reference's ``DGSQP(...)``
signature: ``costs`` / ``agent_constraints`` /
``shared_constraints`` are these records instead of ``ca.Function`` lists.

* ``RacingCost``      -- scripts/DGSQP_ALGAMES_monte_carlo_chicane.py:111-122,223-277
                         (input + input-rate quadratic stage cost, optional blocking / soft-obstacle
                         state cost, terminal ``-w_p s_a + w_c * sum_b comp(s_b - s_a)`` with
                         ``comp = atan`` (chicane.py:240) or identity
                         (comparison_study_barc/exact_dynamic_game_dynamic.py:146-147)),
* ``InputRateLimits`` -- chicane.py:282-290 (rows ordered ``[a ub, a lb, steer ub, steer lb]``),
* ``CollisionAvoidance`` -- chicane.py:292-293,322-330: one row per agent pair (i<j),
                         absent at k=0, present at k=1..N.
* ``GoalTrackingCost`` -- scripts/DGSQP_merge_monte_carlo.py:253-261: ``1/2 sum w_u u^2 + 1/2 (q-goal)^T diag(w_q) (q-goal)``
                         per stage, ``terminal_multiplier`` times the state part at k=N,
* ``LaneBoundaries``  -- merge.py:66-74,316-342: half-plane rows ``n(p_x)^T (p - (anchor - r n(p_x))) <= 0`` with a
                         piecewise-constant normal, present at every stage k=0..N.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Sequence


@dataclass
class RacingCost:
    input_weight: Sequence[float] = (1.0, 1.0)
    input_rate_weight: Sequence[float] = (1.0, 1.0)
    comp_weights: Sequence[float] = (10.0, 5.0)   # [progress, competition]
    comp_type: str = 'atan'                       # 'atan' | 'linear'
    blocking_weight: float = 0.0
    obs_weight: float = 0.0
    obs_r: float = 0.3

    @classmethod
    def from_params(cls, params: dict, comp_type: str = 'atan') -> 'RacingCost':
        """Build from the ``*_cost_params`` dicts used by the reference scripts."""
        return cls(input_weight=tuple(params['input_weight'][:2]),
                   input_rate_weight=tuple(params['input_rate_weight'][:2]),
                   comp_weights=tuple(params['comp_weights']), comp_type=comp_type,
                   blocking_weight=params.get('blocking_weight', 0.0),
                   obs_weight=params.get('obs_weight', 0.0), obs_r=params.get('obs_r', 0.3))


@dataclass
class InputRateLimits:
    """(u_k - u_{k-1}) in dt*[rate_min, rate_max]; values are per second, as the
    ``*_state_input_rate_max/min`` VehicleStates of the scripts."""
    rate_max: Sequence[float] = (10.0, 4.5)       # [u_a, u_steer]
    rate_min: Sequence[float] = (-10.0, -4.5)

    @classmethod
    def from_states(cls, rate_max_state, rate_min_state) -> 'InputRateLimits':
        return cls((rate_max_state.u.u_a, rate_max_state.u.u_steer),
                   (rate_min_state.u.u_a, rate_min_state.u.u_steer))


@dataclass
class CollisionAvoidance:
    """Row for pair (i<j): (r_i+r_j)^2 - |p_i-p_j|^2 <= 0."""
    radii: Sequence[float] = field(default_factory=lambda: [0.2, 0.2])


@dataclass
class GoalTrackingCost:
    input_weight: Sequence[float] = (0.1, 0.1)
    state_weight: Sequence[float] = (1.0, 10.0, 1.0, 1.0)     # diagonal of Q over the agent's state
    goal: Sequence[float] = (4.0, 0.15, 0.3, 0.0)
    terminal_multiplier: float = 10.0
    input_rate_weight: Sequence[float] = (0.0, 0.0)


@dataclass
class LaneHalfPlane:
    """``n(p_x) = n_lo`` for ``p_x < brk`` else ``n_hi`` (CasADi ``pw_const``); constant normal when ``brk`` is +inf."""
    n_lo: Sequence[float]
    anchor: Sequence[float]
    r: float = 0.1
    n_hi: Sequence[float] = None
    brk: float = float('inf')


@dataclass
class LaneBoundaries:
    lanes: List[LaneHalfPlane] = field(default_factory=list)
