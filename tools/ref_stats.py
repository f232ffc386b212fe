"""Build container: the numpy restatement of the reference loop with the restated OSQP (oracle/pyref.py, oracle/osqp_restate.py)
on the first B sampled scenarios of a game, next to the C++ oracle (exact active-set QP) with the literal and the floored
_nearestPD.  Writes tests/golden/pyref_osqp_<game>.npz (the statistical yardstick of tests/test_gpu.py) and prints the table
kept under profiles/.   usage: ref_stats.py <game> <B> [nproc]
       ref_stats.py <game> <B> <nproc> --stable K   adds to the committed file the mask `stable` of the scenarios whose (status, iterations,
           QP solves) the numpy loop ITSELF reproduces from inputs perturbed by 1e-13 relative (K re-runs), after checking that an
           unperturbed re-run reproduces the stored results (OMP_NUM_THREADS = 1: the run is deterministic)"""
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
         'merge_N20': (lambda: mc.merge_game(N=20), 1),
         # BASELINE configs[2], [3], [4] at their own sizes (round 5: the yardstick for the XL layouts)
         'kb_barc3_N25': (lambda: mc.barc_racing_game(N=25, M=3), 0),
         'kb_f1_N50': (lambda: mc.f1_racing_game(N=50), 0),
         'merge6_N25': (lambda: mc.merge_game(N=25, M=6), 1)}
CODE = {'conv_abs_tol': 0, 'conv_rel_tol': 1, 'max_it': 2, 'diverged': 3, 'exception': 4}


def one(args):
    name, b = args[:2]
    g = GAMES[name][0]()
    P, par = build_problem(*g.solver_args()), build_params(g.params, eig_floor=1e-10)
    x0, uws = mc.sample_scenarios(g, B_, seed=GAMES[name][1])
    u = np.concatenate([uws[:, :, 2 * a:2 * a + 2].reshape(B_, -1) for a in range(uws.shape[2] // 2)], axis=1)
    if len(args) > 2 and args[2]:          # perturbed re-run k: inputs x (1 + 1e-13 N(0, 1))
        rng = np.random.default_rng(1000 * args[2] + b)
        x0 = x0 * (1 + 1e-13 * rng.standard_normal(x0.shape))
        u = u * (1 + 1e-13 * rng.standard_normal(u.shape))
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


def stability(name, nproc, K):
    path = ROOT / 'tests' / 'golden' / f'pyref_osqp_{name}.npz'
    gold = dict(np.load(path))
    cf0 = np.stack([gold['status'], gold['num_iters'], gold['qp_solves']], axis=1)
    stable = np.ones(len(cf0), bool)
    t = time.time()
    with mp.Pool(nproc, initializer=lambda: globals().__setitem__('B_', B_)) as pool:
        for k in range(K + 1):
            out = pool.map(one, [(name, b, k) for b in range(B_)], chunksize=1)
            cf = np.array([[o[0], o[1], o[2]] for o in out], np.int32)
            same = (cf == cf0).all(axis=1)
            if k == 0:
                print(f'# {name}: unperturbed re-run reproduces the committed results on {same.sum()}/{len(same)} scenarios')
                assert same.all(), np.nonzero(~same)[0]
            else:
                stable &= same
                print(f'# {name}: perturbed re-run {k}: same path on {same.mean():.3f}; stable so far {stable.mean():.3f} ({time.time() - t:.0f} s)', flush=True)
    gold['stable'] = stable
    np.savez_compressed(path, **gold)
    print(f'{name}: the numpy loop + restated OSQP reproduces itself under 1e-13 input perturbations ({K} re-runs) on {stable.sum()}/{len(stable)} = {stable.mean():.3f} of the scenarios')


if __name__ == '__main__':
    name, B_ = sys.argv[1], int(sys.argv[2])
    nproc = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    if '--stable' in sys.argv:
        stability(name, nproc, int(sys.argv[sys.argv.index('--stable') + 1]))
        sys.exit(0)
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
