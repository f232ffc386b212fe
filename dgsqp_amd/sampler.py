"""Counter-based rejection samplers: the host mirror of ``dgsqp_sample_batch`` (csrc/dgsqp_sampler.h).

The Monte-Carlo scripts of the reference draw from ``np.random.default_rng(seed)`` in a sequential loop
(scripts/DGSQP_ALGAMES_monte_carlo_chicane.py:384-404, DGSQP_monte_carlo_agents.py:262-308, DGSQP_comp_monte_carlo.py:365-382,
DGSQP_merge_monte_carlo.py:429-473); ``dgsqp_amd.montecarlo.sample_scenarios`` reproduces those draws on the host.  On the device
the candidates of a round are drawn by thousands of lanes at once, so the stream has to be a pure function of (seed, candidate,
draw): Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11).  This module restates the
generator and the placement rules in numpy -- the same 53-bit uniforms bit for bit, the same accept / reject decisions, initial
states equal to rounding of sin / cos -- and is what the tests compare the device against.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _ffi

FIRST_SEGMENT, INDEPENDENT, CIRCUIT, MERGE = 0, 1, 2, 3
_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(ctr, key):
    """ctr [..., 4] uint32, key (k0, k1) -> [..., 4] uint32 (ten rounds)."""
    c = [np.asarray(ctr[..., i], dtype=np.uint64) for i in range(4)]
    k0, k1 = int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = _M0 * c[0], _M1 * c[2]
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & _MASK, p1 >> np.uint64(32), p1 & _MASK
        c = [hi1 ^ c[1] ^ np.uint64(k0), lo1, hi0 ^ c[3] ^ np.uint64(k1), lo0]
        k0, k1 = (k0 + _W0) & 0xFFFFFFFF, (k1 + _W1) & 0xFFFFFFFF
    return np.stack(c, axis=-1).astype(np.uint32)


def uniform(seed: int, cand, k: int):
    """Uniform k of the candidates ``cand`` (array of indices): 53 random bits, ((a >> 5) 2^26 + (b >> 6)) / 2^53."""
    cand = np.asarray(cand, dtype=np.uint64)
    ctr = np.stack([cand & _MASK, cand >> np.uint64(32), np.full(cand.shape, k >> 1, np.uint64), np.zeros(cand.shape, np.uint64)], axis=-1)
    r = philox4x32_10(ctr, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
    a, b = r[..., (k & 1) * 2].astype(np.uint64), r[..., (k & 1) * 2 + 1].astype(np.uint64)
    return ((a >> np.uint64(5)).astype(np.float64) * 67108864.0 + (b >> np.uint64(6)).astype(np.float64)) / 9007199254740992.0


class SamplerT(C.Structure):
    _fields_ = [('kind', C.c_int32), ('n_key', C.c_int32), ('seed', C.c_uint64), ('half_width', C.c_double), ('obs_d', C.c_double),
                ('seg0_len', C.c_double), ('x_nom', C.c_double * _ffi.MAX_AGENTS), ('key_pts', (C.c_double * 6) * (_ffi.MAX_SEGS + 1))]


def sampler_kind(game) -> int:
    if game.sampler == 'merge':
        return MERGE
    if game.sampler == 'circuit':
        return CIRCUIT
    return FIRST_SEGMENT if game.joint_model.n_a == 2 else INDEPENDENT


def sampler_spec(game, seed: int) -> SamplerT:
    """The C-ABI description of the game's sampler (include/dgsqp.h: dgsqp_sampler_t)."""
    from .montecarlo import _MERGE_X_NOM
    if getattr(game, 'second_car_ahead', False):
        raise ValueError("the device sampler has no 'car 2 ahead of car 1' rule (DGSQP_monte_carlo_ablation.py:387): sample this game on the host (montecarlo.sample_scenarios)")
    S = SamplerT()
    S.kind, S.seed = sampler_kind(game), int(seed)
    S.half_width, S.obs_d = float(game.half_width), float(game.obs_d)
    for a in range(_ffi.MAX_AGENTS):
        S.x_nom[a] = _MERGE_X_NOM[a]
    if S.kind != MERGE:
        kp = np.asarray(game.track.key_pts, dtype=float)
        if not hasattr(game.track, 'key_pts') or kp.shape[0] > _ffi.MAX_SEGS + 1:
            raise ValueError('the device sampler places cars on arc tracks (key points); spline tracks are sampled on the host')
        S.n_key = kp.shape[0]
        S.seg0_len = float(game.track.cl_segs[0, 0])
        for i in range(kp.shape[0]):
            for j in range(6):
                S.key_pts[i][j] = float(kp[i, j])
    return S


def place(game, seed: int, cand):
    """Placement of the candidates ``cand`` -> (q0 list per agent [n, n_q^a], ok [n]): the arithmetic of dg_sample_place_kernel."""
    cand = np.asarray(cand, dtype=np.uint64)
    n = len(cand)
    U = lambda k: uniform(seed, cand, k)
    models = game.joint_model.dynamics_models
    M = len(models)
    kind = sampler_kind(game)
    ok = np.ones(n, bool)
    if kind == MERGE:
        from .montecarlo import _MERGE_X_NOM
        mw, mp, th = 0.3, 1.5, np.pi / 12
        x5, x7 = mp, mp + mw / np.sin(th)
        q0 = []
        for a in range(M):
            xn = _MERGE_X_NOM[a]
            q = np.zeros((n, 4))
            if a % 3 != 2:
                q[:, 0] = xn + 0.5 * U(4 * a) - 0.25
                q[:, 1] = 0.15 + 0.1 * U(4 * a + 1) - 0.05
                q[:, 2] = 0.3 * (1 + 0.06 * U(4 * a + 2) - 0.03)
                q[:, 3] = (5 * U(4 * a + 3) - 2.5) * np.pi / 180
            else:
                yn = -((x7 + x5) / 2 - xn) * np.tan(th)
                sr, er = 0.5 * U(4 * a) - 0.25, 0.1 * U(4 * a + 1) - 0.05
                q[:, 0] = xn + sr * np.cos(th) - er * np.sin(th)
                q[:, 1] = yn + sr * np.sin(th) + er * np.cos(th)
                q[:, 2] = 0.3 * (1 + 0.06 * U(4 * a + 2) - 0.03)
                q[:, 3] = np.pi / 12 + (5 * U(4 * a + 3) - 2.5) * np.pi / 180
            q0.append(q)
        return q0, ok
    track, hw, obs_d = game.track, game.half_width, game.obs_d

    def put(mdl, s, ey, v, epsi):
        xy = np.array([track.local_to_global((si, ei, 0.0))[:2] for si, ei in zip(s, ey)]).reshape(-1, 2)
        q = np.zeros((n, mdl.n_q))
        q[:, 0], q[:, 1], q[:, 2] = xy[:, 0], xy[:, 1], v
        q[:, 3 if mdl.model_id == 0 else 5] = epsi
        q[:, mdl.s_idx], q[:, mdl.ey_idx] = s, ey
        return q
    if kind == FIRST_SEGMENT:
        seg0 = track.cl_segs[0, 0]
        s1, ey1, v1 = np.maximum(0.1, U(0) * seg0), U(1) * hw * 2 - hw, U(2) + 2
        d = 2 * np.pi * U(3)
        s2, ey2, v2 = s1 + 1.2 * obs_d * np.cos(d), ey1 + 1.2 * obs_d * np.sin(d), U(4) + 2
        ok = (s2 >= 0) & (np.abs(ey2) <= hw)
        s2s, ey2s = np.where(ok, s2, s1), np.where(ok, ey2, ey1)           # (rejected placements are never used)
        return [put(models[0], s1, ey1, v1, 0.0), put(models[1], s2s, ey2s, v2, 0.0)], ok
    if kind == INDEPENDENT:
        seg0 = track.cl_segs[0, 0]
        return [put(m, np.maximum(0.1, U(3 * a) * seg0), U(3 * a + 1) * hw * 2 - hw, U(3 * a + 2) + 2, 0.0) for a, m in enumerate(models)], ok
    L = track.track_length
    s1, v1 = L * U(0), 2.0 + (U(2) - 0.5)
    q0 = [put(models[0], s1, hw * (2 * U(1) - 1), v1, 5.0 * (2 * U(3) - 1) * np.pi / 180)]
    for a in range(1, M):
        q0.append(put(models[a], s1 + 1.2 * obs_d * (2 * U(4 * a) - 1), hw * (2 * U(4 * a + 1) - 1), (1 + 0.25 * (2 * U(4 * a + 2) - 1)) * v1,
                      5.0 * (2 * U(4 * a + 3) - 1) * np.pi / 180))
    return q0, ok


def sample_scenarios_counter(game, B: int, seed: int = 1, chunk: int = 512, max_candidates: int = 10_000_000):
    """Host mirror of ``dgsqp_sample_batch``: the first B accepted candidates in candidate order.
    Returns x0 [B, n_q], u_ws [B, N, n_u] (time-major) and the number of candidates consumed."""
    from .montecarlo import pid_warm_start
    models = game.joint_model.dynamics_models
    M, N, dt = len(models), game.params.N, game.params.dt
    kind = sampler_kind(game)
    radii = list(game.shared_constraints.radii) if game.shared_constraints is not None else [game.obs_d / 2] * M
    rl = game.agent_constraints[0] if game.agent_constraints else None
    du = (10.0, 4.5) if (rl is None or not hasattr(rl, 'rate_max')) else tuple(rl.rate_max)
    x0s, uws, have, c0, used = [], [], 0, 0, 0
    while have < B:
        if c0 >= max_candidates:
            raise RuntimeError('sampler did not produce enough collision-free scenarios')
        cand = np.arange(c0, c0 + chunk, dtype=np.uint64)
        q0, ok = place(game, seed, cand)
        if kind == MERGE:
            traj = []
            for a, mdl in enumerate(models):
                qa = [np.zeros((chunk, 4)) if (M == 3 and a == 2) else q0[a].copy()]
                for _ in range(N):
                    qa.append(np.array([mdl.fd(q, np.zeros(2)) for q in qa[-1]]))
                traj.append(np.stack(qa, axis=1))
            u_ws = [np.zeros((chunk, N, 2)) for _ in range(M)]
        else:
            traj, u_ws = zip(*[pid_warm_start(m, q, N, dt, du=du) for m, q in zip(models, q0)])
        keep = ok.copy()
        for i in range(M):
            for j in range(i + 1, M):
                dist = np.linalg.norm(traj[i][:, :, :2] - traj[j][:, :, :2], axis=2)
                keep &= ~(dist < radii[i] + radii[j]).any(axis=1)
        idx = np.nonzero(keep)[0]
        take = idx[:B - have]
        x0s.append(np.concatenate([q[take] for q in q0], axis=1))
        uws.append(np.concatenate([u[take] for u in u_ws], axis=2))
        have += len(take)
        if have >= B:
            used = c0 + int(take[-1]) + 1
        c0 += chunk
    return np.ascontiguousarray(np.concatenate(x0s)), np.ascontiguousarray(np.concatenate(uws)), used
