"""Known-answer vectors for the dynamics tensors fAd, fBd, fEd, fFd, fGd (dynamics_models.py:128-144) from EXACT symbolic
differentiation (KAT (1)).  Build container only; writes tests/golden/sympy_fd_{kin,dyn,uni}.npz.

Independent of oracle/ and of dgsqp_amd/csrc: the continuous dynamics f_c of the three models are transcribed here into sympy
straight from the reference's CasADi expressions --

    KinematicUnicycle            dynamics_models.py:331-339
    KinematicBicycleCombined     dynamics_models.py:1046-1070   (ca_abs :228-234 = if_else(x > 0, x, -x); ca_sign :236-238 =
                                                                 x / sqrt(x^2 + eps^2), eps = 1e-3)
    DynamicBicycleCombined       dynamics_models.py:2013-2062

-- the discrete map f_d is composed as the reference composes it (euler: q + dt f_c, :91; rk4 with M substeps of h = dt / M,
:188-198; M = 2 for the unicycle and the kinematic bicycle, 1 for the dynamic bicycle), sympy differentiates f_d once and twice with respect to (q, u), and the derivatives are evaluated at a few points.
Vehicle parameters are the defaults of the REFERENCE's own config classes (DGSQP/dynamics/model_types.py, imported live from
/root/reference: that module needs no casadi); they are stored in the file so that the tests can check that they build the same
vehicle.  The track enters through c(s) and psi_t(s) (radius_arclength_track.py:199-225): inside one arc segment c is a constant
and psi_t(s) = psi0 + c (s - s0), which is what CasADi's pw_const / pw_lin give there (all second derivatives zero); the sample
points keep every integrator stage inside the 8 m arc of the curve track of curve.py:140-146 (c = (pi / 4) / 8, s0 = 1, psi0 = 0).

    usage: python tools/make_sympy_kats.py [kin|dyn|uni ...]
"""
import os
import pathlib
import sys
import time

import numpy as np
import sympy as sp

ROOT = pathlib.Path(__file__).resolve().parent.parent
GOLD = ROOT / 'tests' / 'golden'
os.environ.setdefault('MPLBACKEND', 'Agg')
sys.path.insert(0, '/root/reference')
from DGSQP.dynamics.model_types import KinematicBicycleConfig, DynamicBicycleConfig, UnicycleConfig   # noqa: E402  (the reference's own defaults)
sys.path.remove('/root/reference')

C_ARC, S0_ARC, PSI0_ARC = (np.pi / 4) / 8.0, 1.0, 0.0      # the arc of CurveTrack(1, 8, pi/4, 5, ...) (track_lib.py:27-52)
DT = 0.1
RK4_M = {'uni': 2, 'kin': 2, 'dyn': 1}     # (the Pacejka model with two substeps is beyond what sympy differentiates in an hour)


def ca_abs(x):
    return sp.Piecewise((x, x > 0), (-x, True))


def ca_sign(x, eps=1e-3):
    return x / sp.sqrt(x ** 2 + eps ** 2)


def fc_uni(q, u, p):
    x, y, v, psi = q
    Fx, wz = u
    return [v * sp.cos(psi), v * sp.sin(psi), Fx / p['mass'], wz]


def fc_kin(q, u, p):
    x, y, v, epsi, s, xtran = q
    a, gamma = u
    L_f, L_r, m = p['wheel_dist_front'], p['wheel_dist_rear'], p['mass']
    beta = sp.atan2(sp.tan(gamma) * L_r, L_f + L_r)
    psidot = v / L_r * sp.sin(beta)
    F_ext = (- p['damping_coefficient'] * v - p['drag_coefficient'] * v * ca_abs(v)
             - p['rolling_resistance'] * ca_abs(v) ** p['rolling_resistance_exponent'] * ca_sign(v)
             - p['slip_coefficient'] * psidot ** 2)
    c = sp.Float(C_ARC)
    psi_t = PSI0_ARC + c * (s - S0_ARC)
    den = 1 - xtran * c
    return [v * sp.cos(beta + psi_t + epsi), v * sp.sin(beta + psi_t + epsi), a + F_ext / m,
            psidot - c * v * sp.cos(beta + epsi) / den, v * sp.cos(beta + epsi) / den, v * sp.sin(beta + epsi)]


def fc_dyn(q, u, p):
    x, y, vx, vy, psidot, epsi, s, xtran = q
    a, gamma = u
    L_f, L_r, m, I_z, g = p['wheel_dist_front'], p['wheel_dist_rear'], p['mass'], p['yaw_inertia'], p['gravity']
    c = sp.Float(C_ARC)
    psi_t = PSI0_ARC + c * (s - S0_ARC)
    if p['simple_slip']:
        alpha_f = -sp.atan2(vy + L_f * psidot, vx) + gamma
    else:
        alpha_f = -sp.atan2((vy + L_f * psidot) * sp.cos(gamma) - vx * sp.sin(gamma), vx * sp.cos(gamma) + (vy + L_f * psidot) * sp.sin(gamma))
    alpha_r = -sp.atan2(vy - L_r * psidot, vx)
    assert p['tire_model'] == 'pacejka'
    fyf = p['pacejka_d_front'] * sp.sin(p['pacejka_c_front'] * sp.atan(p['pacejka_b_front'] * alpha_f))
    fyr = p['pacejka_d_rear'] * sp.sin(p['pacejka_c_rear'] * sp.atan(p['pacejka_b_rear'] * alpha_r))
    F_ext = (- p['damping_coefficient'] * vx - p['drag_coefficient'] * vx * ca_abs(vx)
             - p['rolling_resistance'] * ca_abs(vx) ** p['rolling_resistance_exponent'] * ca_sign(vx))
    if p['drive_wheels'] == 'all':
        ar, af = a / 2, a / 2
    else:
        ar, af = a, 0
    ax = ar + af * sp.cos(gamma) + (F_ext - fyf * sp.sin(gamma)) / m
    ay = af * sp.sin(gamma) + (fyf * sp.cos(gamma) + fyr) / m
    alphaz = (L_f * fyf * sp.cos(gamma) - L_r * fyr) / I_z
    vlon = vx * sp.cos(epsi) - vy * sp.sin(epsi)
    den = 1 - xtran * c
    return [vx * sp.cos(epsi + psi_t) - vy * sp.sin(epsi + psi_t), vy * sp.cos(epsi + psi_t) + vx * sp.sin(epsi + psi_t),
            ax + psidot * vy, ay - psidot * vx, alphaz, psidot - c * vlon / den, vlon / den, vx * sp.sin(epsi) + vy * sp.cos(epsi)]


def compose(fc, q, u, p, method, M):
    """dynamics_models.py:88-125: euler :91, rk4 :188-198."""
    if method == 'euler':
        return [qi + DT * fi for qi, fi in zip(q, fc(q, u, p))]
    h = sp.Float(DT) / M
    x = list(q)
    for _ in range(M):
        a1 = fc(x, u, p)
        a2 = fc([xi + (h / 2) * ai for xi, ai in zip(x, a1)], u, p)
        a3 = fc([xi + (h / 2) * ai for xi, ai in zip(x, a2)], u, p)
        a4 = fc([xi + h * ai for xi, ai in zip(x, a3)], u, p)
        x = [xi + h * (b1 + 2 * b2 + 2 * b3 + b4) / 6 for xi, b1, b2, b3, b4 in zip(x, a1, a2, a3, a4)]
    return x


def params_of(cfg, names):
    return {n: getattr(cfg, n) for n in names}


MODELS = {
    'uni': (fc_uni, 4, lambda: params_of(UnicycleConfig(), ['mass'])),
    'kin': (fc_kin, 6, lambda: params_of(KinematicBicycleConfig(), ['wheel_dist_front', 'wheel_dist_rear', 'mass', 'drag_coefficient', 'damping_coefficient',
                                                                     'slip_coefficient', 'rolling_resistance', 'rolling_resistance_exponent'])),
    'dyn': (fc_dyn, 8, lambda: params_of(DynamicBicycleConfig(), ['wheel_dist_front', 'wheel_dist_rear', 'mass', 'yaw_inertia', 'gravity', 'drag_coefficient',
                                                                   'damping_coefficient', 'rolling_resistance', 'rolling_resistance_exponent', 'simple_slip',
                                                                   'tire_model', 'drive_wheels', 'pacejka_b_front', 'pacejka_c_front', 'pacejka_d_front',
                                                                   'pacejka_b_rear', 'pacejka_c_rear', 'pacejka_d_rear'])),
}


def points(kind, rng, n):
    out = []
    for _ in range(n):
        if kind == 'uni':
            q = np.array([rng.normal(), rng.normal(), 1.0 + rng.random(), rng.normal() * 0.5])
            u = np.array([rng.normal(), rng.normal() * 0.3])
        elif kind == 'kin':
            q = np.array([rng.normal(), rng.normal(), 2.0 + rng.random(), rng.normal() * 0.1, 3.0 + 3.0 * rng.random(), rng.normal() * 0.3])
            u = np.array([rng.normal() * 0.5, rng.normal() * 0.2])
        else:
            q = np.array([rng.normal(), rng.normal(), 2.0 + rng.random(), rng.normal() * 0.2, rng.normal() * 0.5, rng.normal() * 0.1,
                          3.0 + 3.0 * rng.random(), rng.normal() * 0.3])
            u = np.array([rng.normal() * 0.5, rng.normal() * 0.2])
        out.append(np.concatenate([q, u]))
    return np.array(out)


def main(kinds):
    for kind in kinds:
        fc, nq, pf = MODELS[kind]
        p = pf()
        z = sp.symbols(f'z0:{nq + 2}', real=True)
        q, u = list(z[:nq]), list(z[nq:])
        pts = points(kind, np.random.default_rng({'uni': 3, 'kin': 1, 'dyn': 2}[kind]), 4)
        out = dict(points=pts, dt=DT, track=np.array([C_ARC, S0_ARC, PSI0_ARC]))
        for k, v in p.items():
            out['param_' + k] = np.array(v)
        for tag, method, M in (('euler', 'euler', 1), ('rk4', 'rk4', RK4_M[kind])):
            t = time.time()
            fd = compose(fc, q, u, p, method, M)
            # first and second derivatives, exactly; cse keeps the evaluation tractable, nothing is simplified or truncated
            jac = [[sp.diff(f, zi) for zi in z] for f in fd]
            hes = [[[sp.diff(jac[i][a], z[b]) if b >= a else None for b in range(nq + 2)] for a in range(nq + 2)] for i in range(nq)]
            flat = list(fd) + [jac[i][a] for i in range(nq) for a in range(nq + 2)] + \
                   [hes[i][a][b] for i in range(nq) for a in range(nq + 2) for b in range(a, nq + 2)]
            fn = sp.lambdify(z, flat, modules='math', cse=True)
            F = np.zeros((len(pts), nq)); J = np.zeros((len(pts), nq, nq + 2)); H = np.zeros((len(pts), nq, nq + 2, nq + 2))
            for k, pt in enumerate(pts):
                vals = fn(*[float(v) for v in pt])
                F[k] = vals[:nq]
                J[k] = np.array(vals[nq:nq + nq * (nq + 2)]).reshape(nq, nq + 2)
                it = iter(vals[nq + nq * (nq + 2):])
                for i in range(nq):
                    for a in range(nq + 2):
                        for b in range(a, nq + 2):
                            H[k, i, a, b] = H[k, i, b, a] = next(it)
            out[f'{tag}_M'] = M
            out[f'{tag}_fd'], out[f'{tag}_jac'], out[f'{tag}_hes'] = F, J, H
            print(f'{kind} {tag} M={M}: {len(flat)} expressions, {time.time() - t:.0f} s, max |H| {np.abs(H).max():.3g}', flush=True)
            np.savez_compressed(GOLD / f'sympy_fd_{kind}.npz', **out)      # (after every integrator: the Pacejka model's rk4 takes an hour)


if __name__ == '__main__':
    main(sys.argv[1:] or ['uni', 'kin', 'dyn'])
