"""Development aid (GPU box): cooperative line search vs the plain launch on one batch -- kernel times, helper counters, and how
the outputs differ (they must not).  usage: python tools/debug/gpu_coop_debug.py [workload] [B] [seed]"""
import os, sys, pathlib
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
import bench
from dgsqp_amd.montecarlo import sample_scenarios
from dgsqp_amd.solver import DGSQP

name = sys.argv[1] if len(sys.argv) > 1 else 'dyn_curve_N25'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 768
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 5
g = bench.make_game(name)
s = DGSQP(*g.solver_args(), print_method=None)
x0, u_tm = sample_scenarios(g, B, seed=seed)
KEYS = ('status', 'num_iters', 'qp_solves', 'u', 'l', 'x', 'cond', 'cost')


def run(mode, **env):
    for k, v in env.items():
        os.environ[k] = str(v)
    s.set_cooperative(mode)
    r = s.solve_batch(x0, u_tm)
    st = s.coop_stats() if mode else {}
    for k in env:
        os.environ.pop(k)
    return r, st


def diff(a, b, tag):
    out = []
    for k in KEYS:
        d = a[k].astype(float) - b[k].astype(float)
        nd = int((np.abs(d).reshape(B, -1).max(axis=1) > 0).sum())
        out.append(f'{k}: {nd} scen, max {np.abs(d).max():.2e}')
    print(f'{tag}: ' + '; '.join(out))


ref, _ = run(0)
ref2, _ = run(0)
print(f'{name} B={B}: plain launch kernel {ref["kernel_ms"]:.1f} ms, again {ref2["kernel_ms"]:.1f} ms; slowest scenario iters {ref["num_iters"].max()} qps {ref["qp_solves"].max()}')
diff(ref2, ref, 'plain vs plain')
for start in (2, 4, 8):
    r, st = run(1, DGSQP_COOP_START=start)
    print(f'cooperative, start {start}: kernel {r["kernel_ms"]:.1f} ms {st}')
    diff(r, ref, f'  coop(start {start}) vs plain')
r, st = run(1, DGSQP_COOP_VERIFY=1)
print(f'cooperative + verify: kernel {r["kernel_ms"]:.1f} ms {st}')
diff(r, ref, '  coop(verify) vs plain')
