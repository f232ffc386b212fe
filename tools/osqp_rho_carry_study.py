"""What if OSQP keeps its adapted rho from call to call?  Inside CasADi's conic plugin the OSQP workspace persists (SURVEY.md parity
hazard 7), the restatement (oracle/osqp_restate.py) starts every call at rho = 0.1.  On the six-car merge (reg = 0) that decides how many
QPs run into the 4,000-iteration limit.  This study runs the numpy loop (oracle/pyref.py) on the first B scenarios of a game twice: as
committed, and with the rho a call ended with handed to the next call of the same solve (dgsqp_params_t.osqp_rho_carry; the device and the
C++ oracle implement the same option, profiles/r05_osqp_rho_carry.txt holds the GPU measurement).
    usage: python tools/osqp_rho_carry_study.py <game of tools/ref_stats.py> <B> [nproc] [--carry-only]   (the restart-at-0.1 pass is what
    tests/golden/pyref_osqp_<game>.npz holds for the same first scenarios)"""
import os, sys, pathlib
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "1")
import numpy as np
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tools')); sys.path.insert(0, str(ROOT / 'tests'))
import multiprocessing as mp
import warnings
warnings.simplefilter('ignore')
import dgsqp_amd.montecarlo as mc
from dgsqp_amd.solver import build_problem, build_params
from oracle import oracle, pyref, osqp_restate
from ref_stats import GAMES, CODE


def one(args):
    name, b, B, carry = args
    g = GAMES[name][0]()
    P, par = build_problem(*g.solver_args()), build_params(g.params, eig_floor=1e-10)
    x0, uws = mc.sample_scenarios(g, B, seed=GAMES[name][1])
    u = np.concatenate([uws[:, :, 2 * a:2 * a + 2].reshape(B, -1) for a in range(uws.shape[2] // 2)], axis=1)
    par.osqp_rho_carry = 1 if carry else 0          # (oracle/pyref.py hands the previous call's rho to the next one)
    r = pyref.PyRef(P, par, qp='osqp')
    try:
        with np.errstate(all='ignore'):
            s = r.solve(x0[b], u[b])
        msg, it = s['msg'], s['num_iters']
    except (ValueError, FloatingPointError, np.linalg.LinAlgError):
        msg, it = 'exception', 0
    log = np.array(r.qp_log).reshape(-1, 3)
    return CODE[msg], it, len(log), int(log[:, 1].sum()), int((log[:, 0] != 1).sum())


if __name__ == '__main__':
    name, B = sys.argv[1], int(sys.argv[2])
    nproc = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3].isdigit() else 4
    with mp.Pool(nproc) as pool:
        for carry in ((True,) if '--carry-only' in sys.argv else (False, True)):
            out = np.array(pool.map(one, [(name, b, B, carry) for b in range(B)], chunksize=1))
            st = out[:, 0]
            conv = st <= 1
            print(f'{name}, first {B} scenarios, rho {"carried from call to call within a solve" if carry else "restarted at 0.1 in every call (as committed)"}: '
                  f'converged {conv.mean():.3f}, max_it {np.mean(st == 2):.3f}, raises {np.mean(st == 4):.3f}, mean iterations (conv.) {out[conv, 1].mean() if conv.any() else float("nan"):.2f}, '
                  f'QPs per solve {out[:, 2].mean():.1f}, ADMM iterations per QP {out[:, 3].sum() / max(1, out[:, 2].sum()):.0f}, OSQP calls not "solved" {out[:, 4].sum()} of {out[:, 2].sum()}', flush=True)
