"""Development aid: device vs oracle on the first scenarios of a bench workload (status / iterations / QP counts)."""
import sys, time, pathlib, os
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from dgsqp_amd.montecarlo import kinematic_racing_game, dynamic_racing_game, sample_scenarios
from dgsqp_amd.solver import DGSQP
from oracle import oracle
which, N, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
game = kinematic_racing_game('curve' if which == 'kbcurve' else 'chicane', N=N, reg=0.0 if which == 'kbcurve' else 1e-3) if which.startswith('kb') else dynamic_racing_game(N=N, rk4_substeps=10)   # reg of curve.py:161 / chicane.py:164
s = DGSQP(*game.solver_args(), print_method=None, **({'lsqr_tol': 1e-13} if os.environ.get('TIGHT') else {}))     # TIGHT: converged LSQR dual start (conftest.tight_lsqr)
x0, uws = sample_scenarios(game, B, seed=1)
res = s.solve_batch(x0, uws)
oracle.build()
u_am = np.ascontiguousarray(s._to_agent_major(uws))
t = time.time()
o = oracle.solve_batch(s._problem, s._cparams, x0, u_am, nthreads=min(B, os.cpu_count() or 1))
print('oracle time', time.time() - t)
same = (res['status'] == o['status']) & (res['num_iters'] == o['num_iters']) & (res['qp_solves'] == o['qp_solves'])
print('identical (status, iters, qps):', same.mean())
if B <= 64:
    print('gpu    status', res['status'].tolist()); print('oracle status', o['status'].tolist())
    print('gpu    iters ', res['num_iters'].tolist()); print('oracle iters ', o['num_iters'].tolist())
conv = same & (o['status'] <= 1)
relu = [np.abs(res['u'][i] - o['u'][i]).max() / max(1e-300, np.abs(o['u'][i]).max()) for i in np.nonzero(conv)[0]]
print('identical & converged:', int(conv.sum()), 'max rel |du| among them', max(relu) if relu else None, 'median', float(np.median(relu)) if relu else None)
print('mean iters gpu', res['num_iters'].mean(), 'oracle', o['num_iters'].mean())
print('gpu conv', np.mean(res['status'] <= 1), 'oracle conv', np.mean(o['status'] <= 1))
same_conv = (res['status'] <= 1) == (o['status'] <= 1)
print('same converged flag:', same_conv.mean(), '| iterations within 1 among the commonly converged:', float(np.mean(np.abs(res['num_iters'] - o['num_iters'])[(res['status'] <= 1) & (o['status'] <= 1)] <= 1)))
