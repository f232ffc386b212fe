"""Build container: the numpy restatement of the reference loop with the restated OSQP (oracle/pyref.py, oracle/osqp_restate.py)
on the first B sampled scenarios of a game, next to the C++ oracle (exact active-set QP) with the literal and the floored
_nearestPD.  Writes tests/golden/pyref_osqp_<game>.npz (the statistical yardstick of tests/test_gpu.py) and prints the table
kept under profiles/.   usage: ref_stats.py <game> <B> [nproc]"""
import os, sys, pathlib, time, copy
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "1")
import numpy as np
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
import multiprocessing as mp
from conftest import agent_major
import dgsqp_amd.montecarlo as mc
from dgsqp_amd.solver import build_problem, build_params
from oracle import oracle, pyref

GAMES = {'dyn_curve_N25': (lambda: mc.dynamic_racing_game(N=25, rk4_substeps=10), 1),
         'kb_curve_N25': (lambda: mc.kinematic_racing_game('curve', N=25, reg=0.0), 1),
         'kb_chicane_N25': (lambda: mc.kinematic_racing_game('chicane', N=25), 1),
         'kb_chicane_N15': (lambda: mc.kinematic_racing_game('chicane', N=15), 1),
         'kb_barc2_N15': (lambda: mc.barc_racing_game(N=15, M=2), 0),
         'merge_N20': (lambda: mc.merge_game(N=20), 1)}
CODE = {'conv_abs_tol': 0, 'conv_rel_tol': 1, 'max_it': 2, 'diverged': 3, 'exception': 4}


def one(args):
    name, b = args
    g = GAMES[name][0]()
    P, par = build_problem(*g.solver_args()), build_params(g.params, eig_floor=1e-10)
    x0, uws = mc.sample_scenarios(g, B_, seed=GAMES[name][1])
    u = np.concatenate([uws[:, :, 2 * a:2 * a + 2].reshape(B_, -1) for a in range(uws.shape[2] // 2)], axis=1)
    r = pyref.PyRef(P, par, qp='osqp')
    try:
        with np.errstate(all='ignore'):
            s = r.solve(x0[b], u[b])
    except (ValueError, FloatingPointError, np.linalg.LinAlgError):
        # a diverging run fed inf / NaN into the KKT solve: the reference's run would die with an exception here too
        n_, nc_ = oracle.dims(P)['n'], oracle.dims(P)['nc']
        s = dict(msg='exception', num_iters=0, qp_solves=len(r.qp_log), u=np.full(n_, np.nan), l=np.full(nc_, np.nan))
    log = np.array(r.qp_log).reshape(-1, 3)
    return (CODE[s['msg']], s['num_iters'], s['qp_solves'], s['u'], s['l'], int((log[:, 2] != 1).sum()), int((log[:, 0] != 1).sum()), float(min(0.0, s['l'].min())))


def stats(tag, st, it, qp):
    conv = st <= 1
    return f'{tag:58s} converged {conv.mean():6.3f}  max_it {np.mean(st == 2):5.3f}  qp_fail/exception {np.mean(st == 4):5.3f}  mean iters (conv) {it[conv].mean() if conv.any() else float("nan"):6.2f}  mean QPs {qp.mean():6.2f}'


if __name__ == '__main__':
    name, B_ = sys.argv[1], int(sys.argv[2])
    nproc = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    g = GAMES[name][0]()
    x0, uws = mc.sample_scenarios(g, B_, seed=GAMES[name][1])
    u = np.concatenate([uws[:, :, 2 * a:2 * a + 2].reshape(B_, -1) for a in range(uws.shape[2] // 2)], axis=1)
    t = time.time()
    with mp.Pool(nproc, initializer=lambda: globals().__setitem__('B_', B_)) as pool:
        out = pool.map(one, [(name, b) for b in range(B_)], chunksize=1)
    st, it, qp = (np.array([o[k] for o in out], np.int32) for k in range(3))
    U, Lm = np.array([o[3] for o in out]), np.array([o[4] for o in out])
    print(f'# {name}, first {B_} scenarios of the sampler (seed {GAMES[name][1]}); pyref+OSQP took {time.time() - t:.0f} s on {nproc} processes')
    print(stats('numpy loop + restated OSQP (polish), _nearestPD literal', st, it, qp))
    print(f'    OSQP calls whose polish was rejected: {sum(o[5] for o in out)}, calls not "solved": {sum(o[6] for o in out)}, solves ending with a negative multiplier: {sum(o[7] < -1e-9 for o in out)}')
    P = build_problem(*g.solver_args())
    res = {}
    for tag, fl in (('C++ oracle, exact QP, eig_floor 1e-10 (literal)', 1e-10), ('C++ oracle, exact QP, eig_floor 1e-6', 1e-6)):
        par = build_params(g.params, eig_floor=fl)
        o = oracle.solve_batch(P, par, x0, u, nthreads=nproc)
        res[fl] = o
        same = (o['status'] == st) & (o['num_iters'] == it) & (o['qp_solves'] == qp)
        both = (o['status'] <= 1) & (st <= 1)
        du = [np.abs(o['u'][b] - U[b]).max() / max(1e-300, np.abs(U[b]).max()) for b in np.nonzero(both)[0]]
        print(stats(tag, o['status'], o['num_iters'], o['qp_solves']))
        print(f'    vs restated-OSQP loop: identical (status, iters, QPs) {same.mean():.3f}; same converged flag {np.mean((o["status"] <= 1) == (st <= 1)):.3f}; |iters diff| <= 1 on {np.mean(np.abs(o["num_iters"] - it)[both] <= 1) if both.any() else float("nan"):.3f} of the commonly converged; rel. iterate difference there median {np.median(du) if du else float("nan"):.1e} max {max(du) if du else float("nan"):.1e}')
    np.savez_compressed(ROOT / 'tests' / 'golden' / f'pyref_osqp_{name}.npz', x0=x0, u_ws=uws, status=st, num_iters=it, qp_solves=qp, u=U, l=Lm)
