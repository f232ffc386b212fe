"""Development aid (GPU box): deferral of long scenarios (dgsqp_set_deferral) in one cooperative grouped launch of G batches --
kernel time with and without, how many scenarios were deferred, and whether the outputs differ (they must not).
usage: python tools/debug/gpu_defer_debug.py [workload] [B] [G] [seed]"""
import os, sys, pathlib
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
import bench
from dgsqp_amd.montecarlo import sample_scenarios
from dgsqp_amd.solver import DGSQP, solve_batches

name = sys.argv[1] if len(sys.argv) > 1 else 'dyn_curve_N25'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
G = int(sys.argv[3]) if len(sys.argv) > 3 else 8
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 5
g = bench.make_game(name)
solvers = [DGSQP(*g.solver_args(), print_method=None) for _ in range(G)]
batches = [sample_scenarios(g, B, seed=seed + i) for i in range(G)]
KEYS = ('status', 'num_iters', 'qp_solves', 'u', 'l', 'x', 'cond', 'cost')


def run(min_it, factor=2.0, **env):
    for k, v in env.items():
        os.environ[k] = str(v)
    solvers[0].set_deferral(min_it, factor)
    r = solve_batches(solvers, batches)
    st = solvers[0].deferral_stats()
    st.update(solvers[0].coop_stats())
    for k in env:
        os.environ.pop(k)
    return r, st


def diff(a, b):
    bad = {}
    for k in KEYS:
        nd = sum(int((~((x[k] == y[k]) | (np.isnan(x[k].astype(float)) & np.isnan(y[k].astype(float))))).reshape(B, -1).any(axis=1).sum()) for x, y in zip(a, b))
        if nd:
            bad[k] = nd
    return bad or 'bit-identical'


ref, st = run(0)
ref2, _ = run(0)
its = np.concatenate([r['num_iters'] for r in ref])
print(f'{name} B={B} x {G} batches in one cooperative launch: no deferral {ref[0]["kernel_ms"]:.1f} ms, again {ref2[0]["kernel_ms"]:.1f} ms '
      f'({B * G / ref2[0]["kernel_ms"] * 1e3:.0f} scen/s); mean iters {its.mean():.2f}, {np.mean(its >= 50):.3f} at 50; {st}')
print('  plain vs plain:', diff(ref2, ref))
for min_it, factor in ((8, 2.0), (8, 1.5), (6, 1.0), (12, 2.0), (8, 3.0)):
    r, st = run(min_it, factor)
    print(f'deferral min_it {min_it} factor {factor}: {r[0]["kernel_ms"]:.1f} ms ({B * G / r[0]["kernel_ms"] * 1e3:.0f} scen/s) deferred {st["deferred"]} resumed {st["resumed"]} '
          f'helped {st["helped"]} used {st["used"]}; vs no deferral: {diff(r, ref)}')
