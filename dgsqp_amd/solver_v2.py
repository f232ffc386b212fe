"""Host-side mirror of DG-SQP v2 (reference DGSQP/solvers/DGSQP_v2.py: class ``DGSQP`` with ``DGSQPV2Params``).

Same constructor, ``set_warm_start`` / ``solve`` / ``step`` surface as the v1 mirror (dgsqp_amd/solver.py); the state machine
(d-steps, m-steps, checkpoints, decaying regularisation, merit memory; DGSQP_v2.py:322-720) runs in the HIP library
(csrc/dgsqp_solve_v2.h), selected by ``dgsqp_params_t.variant``.  ``solve()`` returns the reference's v2 dictionary
(DGSQP_v2.py:616-632): the v1 keys plus ``primal_sol, dual_sol, x_pred, u_pred, conds``; a time-out is reported as
``'time_limit_exceeded'`` (DGSQP_v2.py:411)."""
from __future__ import annotations

from .solver import DGSQP as _DGSQPv1
from .solver_types import DGSQPV2Params


class DGSQP(_DGSQPv1):
    def __init__(self, joint_dynamics, costs, agent_constraints, shared_constraints, bounds, params=None, use_mx=False,
                 print_method=print, xy_plot=None, **knobs):
        params = DGSQPV2Params() if params is None else params
        if not isinstance(params, DGSQPV2Params):
            raise TypeError('DG-SQP v2 takes DGSQPV2Params (reference solver_types.py:130-175)')
        super().__init__(joint_dynamics, costs, agent_constraints, shared_constraints, bounds, params,
                         print_method=print_method, xy_plot=xy_plot, use_mx=use_mx, **knobs)

    def solve(self, states, parameters=None):
        info = super().solve(states)
        if info['msg'] == 'time_limit':
            info['msg'] = 'time_limit_exceeded'
        info['primal_sol'], info['dual_sol'] = self._last_u_agent_major, self.l_pred
        info['x_pred'], info['u_pred'] = self.q_pred, self.u_pred
        info['conds'] = dict(info['cond'])
        return info
