"""Build container: EVENT LOGS of the numpy restatement of the reference loop with the restated OSQP (oracle/pyref.py + oracle/osqp_restate.py:
scipy's lsqr and numpy's eigh at their defaults, the closest available stand-in for the reference) on the first B sampled scenarios of a
game -- from the nominal inputs and from two perturbed copies (1e-12, 1e-11 relative) -- and, per scenario, the STABLE PREFIX: the events up
to the first one at which a perturbed run takes another decision (another event code), with the mask of the values both perturbed runs
reproduce to 1e-7.  Writes tests/golden/pyref_osqp_trace_<game>.npz, the fixture of
tests/test_gpu.py::test_event_trace_prefix_parity_against_the_numpy_loop (device with qp_method = 'osqp' against the numpy loop, event by
event, as far as the loop itself is reproducible).   usage: ref_trace.py <game> <B> [nproc]"""
import os, sys, pathlib, time
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "1")
import numpy as np
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests')); sys.path.insert(0, str(ROOT / 'tools'))
import multiprocessing as mp
import dgsqp_amd.montecarlo as mc
from dgsqp_amd.solver import build_problem, build_params
from oracle import pyref
from ref_stats import GAMES

EPS = (0.0, 1e-12, 1e-11)


def inputs(name, B):
    g = GAMES[name][0]()
    x0, uws = mc.sample_scenarios(g, B, seed=GAMES[name][1])
    u = np.concatenate([uws[:, :, 2 * a:2 * a + 2].reshape(B, -1) for a in range(uws.shape[2] // 2)], axis=1)
    return g, x0, u


def one(args):
    name, B, b, k = args
    g, x0, u = inputs(name, B)
    P, par = build_problem(*g.solver_args()), build_params(g.params, eig_floor=1e-10)
    if EPS[k] > 0:
        rng = np.random.default_rng(7000 * k + b)
        x0 = x0 * (1 + EPS[k] * rng.standard_normal(x0.shape))
        u = u * (1 + EPS[k] * rng.standard_normal(u.shape))
    r = pyref.PyRef(P, par, qp='osqp')
    t0 = time.time()
    try:
        with np.errstate(all='ignore'):
            r.solve(x0[b], u[b], trace=True)
    except (ValueError, FloatingPointError, np.linalg.LinAlgError):
        pass            # a diverging run fed inf / NaN into the KKT solve: the log ends where the reference's run would die
    return b, k, np.array(r.trace, float).reshape(-1, 2), time.time() - t0


def compare(to, tg, vtol):
    """(first index with different event codes -- or min(len) --, mask: value within vtol relative or below 1e-6 or a merit value of an
    iteration whose mu divides by a rounding-size violation): tests/test_gpu.py::_trace_compare"""
    m = min(len(to), len(tg))
    same = to[:m, 0] == tg[:m, 0]
    k = int(np.argmin(same)) if not same.all() else m
    close = (np.abs(to[:k, 1]) <= 1e-6) | (np.abs(tg[:k, 1] - to[:k, 1]) <= vtol * np.abs(to[:k, 1]))
    pf = 1.0
    for i in range(k):
        if to[i, 0] == 2:
            pf = to[i, 1]
        elif to[i, 0] == 1:
            pf = 1.0
        if pf < 1e-9 and to[i, 0] in (11, 12, 13, 20, 21, 22, 31):
            close[i] = True
    return k, close


if __name__ == '__main__':
    name, B = sys.argv[1], int(sys.argv[2])
    nproc = int(sys.argv[3]) if len(sys.argv) > 3 else max(1, (os.cpu_count() or 2) - 1)
    jobs = [(name, B, b, k) for k in range(3) for b in range(B)]
    logs = {}
    t0 = time.time()
    with mp.Pool(nproc) as pool:
        for b, k, tr, dt in pool.imap_unordered(one, jobs):
            logs[(b, k)] = tr
            print(f'scenario {b} run {k}: {len(tr)} events, {dt:.0f} s  [{len(logs)}/{len(jobs)}, {time.time() - t0:.0f} s]', flush=True)
    g, x0, u = inputs(name, B)
    prefix, firm, off = np.zeros(B, int), [], [0]
    for b in range(B):
        k1, c1 = compare(logs[(b, 0)], logs[(b, 1)], 1e-7)
        k2, c2 = compare(logs[(b, 0)], logs[(b, 2)], 1e-7)
        k = min(k1, k2)
        prefix[b] = k
        firm.append(c1[:k] & c2[:k])
        off.append(off[-1] + len(logs[(b, 0)]))
    trace = np.concatenate([logs[(b, 0)] for b in range(B)])
    whole = np.array([prefix[b] == len(logs[(b, 0)]) == len(logs[(b, 1)]) == len(logs[(b, 2)]) for b in range(B)])
    iters = np.array([int((logs[(b, 0)][:prefix[b], 0] == 1).sum()) for b in range(B)])
    out = ROOT / 'tests' / 'golden' / f'pyref_osqp_trace_{name}.npz'
    np.savez_compressed(out, x0=x0, u=u, trace=trace, off=np.array(off), prefix=prefix, firm=np.concatenate([np.pad(f, (0, len(logs[(b, 0)]) - len(f))) for b, f in enumerate(firm)]), whole=whole)
    print(f'{name}: numpy loop + restated OSQP, {B} scenarios: {int(prefix.sum())} of {len(trace)} events inside the stable prefixes (median {int(np.median(prefix))}, min {int(prefix.min())}; '
          f'{int(whole.sum())} logs stable to their end), {int(iters.sum())} SQP iterations (median {int(np.median(iters))}); wrote {out.relative_to(ROOT)} ({out.stat().st_size} bytes), {time.time() - t0:.0f} s')
