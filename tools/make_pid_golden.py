"""Regenerates tests/golden/pid_ref_*.npz in the BUILD container by IMPORTING the reference's own controller
(/root/reference/DGSQP/solvers/PID.py: ``PID`` :13-138, ``PIDLaneFollower`` :185-238 -- casadi-free, see SURVEY.md
section 8c) and driving it exactly as scripts/DGSQP_ALGAMES_monte_carlo_chicane.py:411-447 does.

Two kinds of vectors (data only -- no reference source is copied):
  * ``open_*``: open-loop -- random (v_long, x_tran, e_psi) sequences in, (u_a, u_steer) of ``PIDLaneFollower.step`` out;
    pins the controller arithmetic (anti-windup, rate-before-magnitude saturation, ``set_x_ref(0)`` of the steering loop);
  * ``kb`` / ``dyn``: closed loop over N = 25 steps from sampled initial states, the REFERENCE controller in the loop with this
    repo's plant step (fixed-step rk4 of f_c; the reference's plant ``step`` needs casadi and integrates the same f_c with
    adaptive RK45, dynamics_models.py:161-186): u_ws and the state trajectory.
The tests compare dgsqp_amd/pid.py, the numpy mirror montecarlo.pid_warm_start and the HIP kernel with these."""
import os, sys, pathlib
import numpy as np
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
GOLD = ROOT / 'tests' / 'golden'
os.environ.setdefault('MPLBACKEND', 'Agg')
sys.path.insert(0, '/root/reference')
from DGSQP.solvers.PID import PIDLaneFollower as RefFollower          # noqa: E402  (the reference's own code)
from DGSQP.solvers.solver_types import PIDParams as RefPIDParams      # noqa: E402
from DGSQP.types import VehicleState as RefState, ParametricPose as RefPose, BodyLinearVelocity as RefVel, VehicleActuation as RefAct  # noqa: E402
sys.path.remove('/root/reference')
import dgsqp_amd.montecarlo as mc                                      # noqa: E402


def follower(dt, x_tran0, v0, u_max, du_max):
    """chicane.py:411-424"""
    steer = RefPIDParams(dt=dt, Kp=1.0, Ki=0.005, x_ref=x_tran0, u_max=u_max[1], u_min=-u_max[1], du_max=du_max[1], du_min=-du_max[1])
    speed = RefPIDParams(dt=dt, Kp=1.0, x_ref=v0, u_max=u_max[0], u_min=-u_max[0], du_max=du_max[0], du_min=-du_max[0])
    return RefFollower(dt, steer, speed)


def open_loop(seed, T=40, dt=0.1):
    rng = np.random.default_rng(seed)
    v = 2.5 + np.cumsum(0.3 * rng.standard_normal(T))
    ey = np.cumsum(0.2 * rng.standard_normal(T))
    epsi = 0.5 * rng.standard_normal(T)
    du = (10.0, 4.5) if seed % 2 == 0 else (0.2, 0.05)          # odd seeds: rate limits tight enough to bind
    ctl = follower(dt, ey[0], v[0], (2.1, 0.436), du)
    out = np.zeros((T, 2))
    for k in range(T):
        st = RefState(p=RefPose(x_tran=ey[k], e_psi=epsi[k]), v=RefVel(v_long=v[k]), u=RefAct())
        ctl.step(st)
        out[k] = st.u.u_a, st.u.u_steer
    return np.stack([v, ey, epsi], axis=1), out, np.array(du)


def closed_loop(game, B, seed, du_max):
    x0, _ = mc.sample_scenarios(game, B, seed=seed)
    N, dt = game.params.N, game.params.dt
    models = game.joint_model.dynamics_models
    nqa = models[0].n_q
    v_idx, epsi_idx, ey_idx = (2, 3, 5) if models[0].model_id == 0 else (2, 5, 7)
    q_ws = np.zeros((B, len(models), N + 1, nqa))
    u_ws = np.zeros((B, len(models), N, 2))
    for b in range(B):
        for a, m in enumerate(models):
            q = x0[b, a * nqa:(a + 1) * nqa].copy()
            ctl = follower(dt, q[ey_idx], q[v_idx], (2.1, 0.436), du_max)
            q_ws[b, a, 0] = q
            for k in range(N):
                st = RefState(p=RefPose(x_tran=q[ey_idx], e_psi=q[epsi_idx]), v=RefVel(v_long=q[v_idx]), u=RefAct())
                ctl.step(st)
                u = np.array([st.u.u_a, st.u.u_steer])
                q = mc._plant_step(m, q[None], u[None], dt)[0]
                u_ws[b, a, k], q_ws[b, a, k + 1] = u, q
    return x0, u_ws, q_ws


if __name__ == '__main__':
    ins, outs, dus = zip(*(open_loop(s) for s in range(8)))
    out = dict(open_in=np.array(ins), open_out=np.array(outs), open_du=np.array(dus))
    for tag, game, du in (('kb', mc.kinematic_racing_game('chicane', N=25), (10.0, np.pi)), ('dyn', mc.dynamic_racing_game(N=25), (10.0, 4.5))):
        x0, u_ws, q_ws = closed_loop(game, 12, 5, du)
        out.update({f'{tag}_x0': x0, f'{tag}_u_ws': u_ws, f'{tag}_q_ws': q_ws, f'{tag}_du': np.array(du)})
        print(tag, 'u range', u_ws.min(), u_ws.max())
    np.savez_compressed(GOLD / 'pid_ref.npz', **out)
