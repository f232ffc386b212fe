"""Development aid (runs on the GPU box): stage-by-stage comparison of the HIP path with the oracle."""
import sys, time, pathlib
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
from dgsqp_amd.montecarlo import kinematic_racing_game, dynamic_racing_game, sample_scenarios
from dgsqp_amd.solver import DGSQP
from dgsqp_amd import _ffi
from oracle import oracle
import ctypes

buf = ctypes.create_string_buffer(256)
_ffi.load_library().dgsqp_backend_info(buf, 256)
print('backend:', buf.value.decode())

which = sys.argv[1] if len(sys.argv) > 1 else 'kb'
N = int(sys.argv[2]) if len(sys.argv) > 2 else 25
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
game = kinematic_racing_game('chicane', N=N) if which == 'kb' else dynamic_racing_game(N=N, rk4_substeps=int(sys.argv[4]) if len(sys.argv) > 4 else 10)
import os
if os.environ.get('LSQR_TOL'):
    import dgsqp_amd.solver as _sv
    _bp = _sv.build_params
    def _bp2(p):
        r = _bp(p); r.lsqr_atol = r.lsqr_btol = float(os.environ['LSQR_TOL']); return r
    _sv.build_params = _bp2
t = time.time(); s = DGSQP(*game.solver_args(), print_method=None); print('create', time.time() - t, 'lds', s.dims.lds_bytes, 'ws', s.dims.workspace_bytes)
P, par = s._problem, s._cparams
x0, uws = sample_scenarios(game, B, seed=1)
u = s._to_agent_major(uws) + 0.01 * np.random.default_rng(0).standard_normal((B, s.n))
rng = np.random.default_rng(1)
l = np.maximum(0, rng.standard_normal((B, s.n_c_total)))

def rel(a, b):
    return np.abs(a - b).max() / max(1e-300, np.abs(b).max())

t = time.time(); ev = s.evaluate_batch(x0, u, l); print('evaluate_batch', time.time() - t)
for b in range(min(B, 3)):
    o = oracle.evaluate(P, x0[b], u[b], l[b], 1)
    l0 = oracle.dual_init(P, par, x0[b], u[b])
    print(b, 'x', rel(ev['x'][b], o['x']), 'q', rel(ev['q'][b], o['q']), 'g', rel(ev['g'][b], o['g']), 'G', rel(ev['G'][b], o['G']),
          'Q', rel(ev['Q'][b], o['Q']), 'l0', rel(ev['l0'][b], l0), np.abs(l0).max())

t = time.time(); qp = s.qp_batch(x0, u, l); print('qp_batch', time.time() - t, 'flags', qp['flag'])
for b in range(min(B, 3)):
    o = oracle.evaluate(P, x0[b], u[b], l[b], 1)
    Qpd = oracle.nearest_pd(o['Q'], par.reg)
    du, lam, flag = oracle.qp(Qpd, o['q'], o['G'], o['g'])
    print(b, 'Qpd', rel(qp['Qpd'][b], Qpd), 'du', rel(qp['du'][b], du), 'lhat', rel(qp['lhat'][b], lam), 'oracle flag', flag,
          'nact', (lam > 0).sum(), (qp['lhat'][b] > 0).sum())
    # KKT of the device answer
    r = Qpd @ qp['du'][b] + o['q'] + o['G'].T @ qp['lhat'][b]
    print('   kkt stat', np.abs(r).max(), 'pfeas', (o['G'] @ qp['du'][b] + o['g']).max(), 'lmin', qp['lhat'][b].min())

t = time.time(); res = s.solve_batch(x0, uws); tg = time.time() - t
print('solve_batch', tg, 'kernel_ms', res['kernel_ms'])
t = time.time(); ref = oracle.solve_batch(P, par, x0, s._to_agent_major(uws), nthreads=8); tc = time.time() - t
print('oracle', tc)
print('status gpu', res['status']); print('status ref', ref['status'])
print('iters gpu', res['num_iters']); print('iters ref', ref['num_iters'])
print('qps gpu', res['qp_solves']); print('qps ref', ref['qp_solves'])
for b in range(B):
    print(b, 'u', rel(res['u'][b], ref['u'][b]), 'l', rel(res['l'][b], ref['l'][b]), 'cond', res['cond'][b], ref['cond'][b])

# ---- event-trace comparison
if '--trace' in sys.argv:
    s.set_trace(4000)
    res2 = s.solve_batch(x0, uws)
    traces = s.fetch_trace(B)
    for b in range(B):
        to = oracle.solve_trace(P, par, x0[b], s._to_agent_major(uws)[b])
        tg = traces[b]
        m = min(len(to), len(tg))
        bad = None
        for i in range(m):
            if to[i, 0] != tg[i, 0] or abs(to[i, 1] - tg[i, 1]) > 1e-6 * max(1e-3, abs(to[i, 1])):
                bad = i
                break
        if bad is None and len(to) == len(tg):
            print(b, 'trace identical, events', len(to))
            continue
        bad = m if bad is None else bad
        print(b, 'first divergence at event', bad, 'of', len(to), len(tg))
        for i in range(max(0, bad - 12), min(m, bad + 4)):
            print('    ', i, 'ref', to[i], 'gpu', tg[i], 'rel', abs(to[i, 1] - tg[i, 1]) / max(1e-300, abs(to[i, 1])))
