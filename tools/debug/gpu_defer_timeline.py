"""Development aid (GPU box): timeline of one cooperative grouped launch with deferral -- when the queue ran empty, when the deferred
scenarios were resumed and finished, how well the resume order's key predicts what a scenario still costs.
usage: python tools/debug/gpu_defer_timeline.py [workload] [B] [G] [min_it] [factor] [seed]"""
import os, sys, pathlib
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
import bench
from dgsqp_amd.montecarlo import sample_scenarios
from dgsqp_amd.solver import DGSQP, solve_batches

name = sys.argv[1] if len(sys.argv) > 1 else 'dyn_curve_N25'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
G = int(sys.argv[3]) if len(sys.argv) > 3 else 8
min_it = int(sys.argv[4]) if len(sys.argv) > 4 else 8
factor = float(sys.argv[5]) if len(sys.argv) > 5 else 2.0
seed = int(sys.argv[6]) if len(sys.argv) > 6 else 5
g = bench.make_game(name)
solvers = [DGSQP(*g.solver_args(), print_method=None, qp_method=os.environ.get('DGSQP_QP_METHOD', 'active_set')) for _ in range(G)]
batches = [sample_scenarios(g, B, seed=seed + i) for i in range(G)]
solvers[0].set_deferral(min_it, factor)
for rep in range(2):
    r = solve_batches(solvers, batches)
log = solvers[0].deferral_log()
out_dir = pathlib.Path(__file__).resolve().parent.parent.parent / 'gpurun_out' / 'defer'
out_dir.mkdir(parents=True, exist_ok=True)
np.save(out_dir / f'log_{name}_{G}_{min_it}.npy', log)
ms = lambda t: np.asarray(t) * 1e-5
print(f'{name} B={B} x {G}, deferral min_it {min_it} factor {factor}: kernel {r[0]["kernel_ms"]:.1f} ms, {len(log)} deferred')
tp, tr, td = ms(log[:, 4]), ms(log[:, 5]), ms(log[:, 6])
rem = td - tr
print(f'set aside between {tp.min():.0f} and {tp.max():.0f} ms (median {np.median(tp):.0f}); resumed between {tr.min():.0f} and {tr.max():.0f} ms (median {np.median(tr):.0f}); '
      f'finished between {td.min():.0f} and {td.max():.0f} ms')
print(f'remaining cost of a deferred scenario: mean {rem.mean():.1f} ms, median {np.median(rem):.1f}, max {rem.max():.1f}; sum {rem.sum() / 1e3:.2f} s = {rem.sum() / 256:.0f} ms per CU')
order = np.argsort(tr)
key = ms(log[:, 3])
print('rank correlation key vs remaining cost:', float(np.corrcoef(np.argsort(np.argsort(key)), np.argsort(np.argsort(rem)))[0, 1]))
def rank(v):
    return np.argsort(np.argsort(v))
long_ = log[:, 7] >= 50
def auc(f):          # probability that a scenario that runs to the iteration limit has the larger feature value
    a, b = f[long_], f[~long_]
    if len(a) == 0 or len(b) == 0:
        return float('nan')
    r = rank(np.concatenate([a, b]))
    return float((r[:len(a)].sum() - len(a) * (len(a) - 1) / 2) / (len(a) * len(b)))
feats = {'key (ms so far)': key, 'QPs so far': log[:, 2], 'violation': log[:, 9], 'complementarity': log[:, 10], 'stationarity': log[:, 11]}
print(f'{int(long_.sum())} of the deferred run to the iteration limit; predictors (rank correlation with the remaining cost, AUC for "runs to the limit"):')
for k, f in feats.items():
    print(f'   {k:18s} {float(np.corrcoef(rank(f), rank(rem))[0, 1]):6.3f} {auc(np.asarray(f, float)):6.3f}')
print('the 12 most expensive: (resumed at, finished at, remaining ms, key ms, its at deferral, qps at deferral, final its, final qps)')
for i in np.argsort(-rem)[:12]:
    print(f'   {tr[i]:7.0f} {td[i]:7.0f} {rem[i]:7.1f} {key[i]:6.1f} {int(log[i, 1]):3d} {int(log[i, 2]):3d} {int(log[i, 7]):3d} {int(log[i, 8]):4d} | {log[i, 9]:.1e} {log[i, 10]:.1e} {log[i, 11]:.1e}')
print('the last 8 to finish:')
for i in np.argsort(-td)[:8]:
    print(f'   {tr[i]:7.0f} {td[i]:7.0f} {rem[i]:7.1f} {key[i]:6.1f} {int(log[i, 1]):3d} {int(log[i, 2]):3d} {int(log[i, 7]):3d} {int(log[i, 8]):4d} | {log[i, 9]:.1e} {log[i, 10]:.1e} {log[i, 11]:.1e}')
