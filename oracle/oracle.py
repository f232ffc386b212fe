"""ctypes front-end of the CPU oracle (oracle/dgsqp_oracle.cpp).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never by dgsqp_amd/.  PARITY UNPINNED (see the
header of dgsqp_oracle.cpp)."""
from __future__ import annotations

import ctypes as C
import pathlib
import subprocess

import numpy as np

from dgsqp_amd import _ffi

_HERE = pathlib.Path(__file__).resolve().parent
_LIB = None
_PD = C.POINTER(C.c_double)
_PI = C.POINTER(C.c_int32)


def build(force: bool = False) -> pathlib.Path:
    so = _HERE / 'liboracle.so'
    src = _HERE / 'dgsqp_oracle.cpp'
    if force or not so.exists() or so.stat().st_mtime < max(src.stat().st_mtime, (_HERE / 'jet.hpp').stat().st_mtime, (_HERE / 'osqp.hpp').stat().st_mtime):
        subprocess.check_call(['make', '-C', str(_HERE), 'liboracle.so'])
    return so


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        so = _HERE / 'liboracle.so'
        if not so.exists():
            build()
        _LIB = C.CDLL(str(so))
    return _LIB


def _d(a):
    return None if a is None else a.ctypes.data_as(_PD)


def _i(a):
    return None if a is None else a.ctypes.data_as(_PI)


def dims(P):
    out = np.zeros(6, np.int32)
    lib().oracle_dims(C.byref(P), _i(out))
    return dict(zip(['M', 'N', 'nq', 'nu', 'n', 'nc'], (int(v) for v in out)))


def rows(P):
    nc = dims(P)['nc']
    out = np.zeros((nc, 5), np.int32)
    lib().oracle_rows(C.byref(P), _i(out))
    return out


def dynamics(P, agent, q, u, derivs=True):
    q = np.ascontiguousarray(q, float)
    u = np.ascontiguousarray(u, float)
    nqa = len(q)
    nv = nqa + 2
    dq, qn = np.zeros(nqa), np.zeros(nqa)
    Jac = np.zeros((nqa, nv)) if derivs else None
    Hes = np.zeros((nqa, nv, nv)) if derivs else None
    lib().oracle_dynamics(C.byref(P), C.c_int(agent), _d(q), _d(u), _d(dq), _d(qn), _d(Jac), _d(Hes))
    return dq, qn, Jac, Hes


def track(P, s):
    c, t, dt = C.c_double(), C.c_double(), C.c_double()
    lib().oracle_track(C.byref(P), C.c_double(s), C.byref(c), C.byref(t), C.byref(dt))
    return c.value, t.value, dt.value


def evaluate(P, x0, u, l=None, hessian=1):
    d = dims(P)
    x0 = np.ascontiguousarray(x0, float)
    u = np.ascontiguousarray(u, float)
    l = None if l is None else np.ascontiguousarray(l, float)
    out = dict(q=np.zeros(d['n']), g=np.zeros(d['nc']), G=np.zeros((d['nc'], d['n'])), Q=np.zeros((d['n'], d['n'])),
               x=np.zeros((d['N'] + 1, d['nq'])), J=np.zeros(d['M']))
    lib().oracle_evaluate(C.byref(P), _d(x0), _d(u), _d(l), C.c_int(hessian), _d(out['q']), _d(out['g']), _d(out['G']),
                          _d(out['Q']), _d(out['x']), _d(out['J']))
    return out


def sum_obj(P, x0, u):
    """(sum_a J^a(u), its gradient w.r.t. all inputs): the objective part of DG-SQP v2's merit 'sum_obj_l1'."""
    d = dims(P)
    obj, grad = C.c_double(0.0), np.zeros(d['n'])
    lib().oracle_sum_obj(C.byref(P), _d(np.ascontiguousarray(x0, float)), _d(np.ascontiguousarray(u, float)), C.byref(obj), _d(grad))
    return obj.value, grad


def dual_init(P, par, x0, u):
    d = dims(P)
    l0 = np.zeros(d['nc'])
    lib().oracle_dual_init(C.byref(P), C.byref(par), _d(np.ascontiguousarray(x0, float)), _d(np.ascontiguousarray(u, float)), _d(l0))
    return l0


def nearest_pd(Q, reg, eig_floor=1e-10):
    Q = np.ascontiguousarray(Q, float)
    out = np.zeros_like(Q)
    lib().oracle_nearest_pd2(C.c_int(Q.shape[0]), _d(Q), C.c_double(reg), C.c_double(eig_floor), _d(out))
    return out


def eigh(A):
    A = np.ascontiguousarray(A, float)
    n = A.shape[0]
    s, U = np.zeros(n), np.zeros((n, n))
    lib().oracle_eigh(C.c_int(n), _d(A), _d(s), _d(U))
    return s, U


def qp(H, c, G, g):
    H, c, G, g = (np.ascontiguousarray(a, float) for a in (H, c, G, g))
    n, m = H.shape[0], G.shape[0]
    x, lam = np.zeros(n), np.zeros(m)
    flag = lib().oracle_qp(C.c_int(n), C.c_int(m), _d(H), _d(c), _d(G), _d(g), _d(x), _d(lam))
    return x, lam, flag


def osqp(H, c, G, g):
    """The C++ restatement of OSQP (oracle/osqp.hpp) on  min 1/2 x'Hx + c'x  s.t.  G x <= -g.  Returns (x, lam, info)."""
    H, c, G, g = (np.ascontiguousarray(a, float) for a in (H, c, G, g))
    n, m = H.shape[0], G.shape[0]
    x, lam, info = np.zeros(n), np.zeros(m), np.zeros(8)
    lib().oracle_osqp(C.c_int(n), C.c_int(m), _d(H), _d(c), _d(G), _d(g), _d(x), _d(lam), _d(info))
    return x, lam, dict(status=int(info[0]), iters=int(info[1]), polished=int(info[2]), rho=info[3], rho_updates=int(info[4]),
                        n_active=int(info[5]), pri_res=info[6], dua_res=info[7])


def lsqr(A, b, atol=1e-6, btol=1e-6, iter_lim=0):
    A, b = np.ascontiguousarray(A, float), np.ascontiguousarray(b, float)
    m, n = A.shape
    x = np.zeros(n)
    itn = C.c_int32()
    istop = lib().oracle_lsqr(C.c_int(m), C.c_int(n), _d(A), _d(b), C.c_double(atol), C.c_double(btol), C.c_int(iter_lim), _d(x), C.byref(itn))
    return x, istop, itn.value


def merit(P, par, Q, q, G, g, l, s, du, dl, mu):
    arrs = [np.ascontiguousarray(a, float) for a in (Q, q, G, g, l, s, du, dl)]
    phi, dphi, mu_out = C.c_double(), C.c_double(), C.c_double()
    lib().oracle_merit(C.byref(P), C.byref(par), *[_d(a) for a in arrs], C.c_double(mu), C.byref(phi), C.byref(dphi), C.byref(mu_out))
    return phi.value, dphi.value, mu_out.value


def solve_batch(P, par, x0, u_ws, literal=0, nthreads=1, timed=False):
    """u_ws agent-major [B, n].  ``timed``: also returns ``seconds`` [B], the wall-clock time each scenario took on its thread."""
    d = dims(P)
    x0 = np.ascontiguousarray(x0, float)
    u_ws = np.ascontiguousarray(u_ws, float)
    B = x0.shape[0]
    secs = np.zeros(B) if timed else None
    lib().oracle_set_scenario_seconds(_d(secs))
    out = dict(u=np.zeros((B, d['n'])), l=np.zeros((B, d['nc'])), x=np.zeros((B, d['N'] + 1, d['nq'])),
               status=np.zeros(B, np.int32), num_iters=np.zeros(B, np.int32), qp_solves=np.zeros(B, np.int32),
               cond=np.zeros((B, 3)), cost=np.zeros((B, d['M'])), l_init=np.zeros((B, d['nc'])))
    lib().oracle_solve_batch(C.byref(P), C.byref(par), C.c_int64(B), _d(x0), _d(u_ws), _d(out['u']), _d(out['l']), _d(out['x']),
                             _i(out['status']), _i(out['num_iters']), _i(out['qp_solves']), _d(out['cond']), _d(out['cost']),
                             _d(out['l_init']), C.c_int(literal), C.c_int(nthreads))
    lib().oracle_set_scenario_seconds(None)
    if timed:
        out['seconds'] = secs
    return out


def solve_trace(P, par, x0, u_ws, max_pairs=20000):
    """Event log [(code, value)] of one solve (see tr() in dgsqp_oracle.cpp)."""
    x0 = np.ascontiguousarray(x0, float)
    u_ws = np.ascontiguousarray(u_ws, float)
    out = np.zeros((max_pairs, 2))
    npairs = C.c_int32()
    lib().oracle_solve_trace(C.byref(P), C.byref(par), _d(x0), _d(u_ws), _d(out), C.c_int32(max_pairs), C.byref(npairs))
    return out[:npairs.value]
