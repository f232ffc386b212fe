"""Development aid (GPU box): kernel time of bench batches with and without the warm start of the active-set QP."""
import sys, pathlib
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
import bench
from dgsqp_amd.montecarlo import sample_scenarios
from dgsqp_amd.solver import DGSQP
name = sys.argv[1] if len(sys.argv) > 1 else 'dyn_curve_N25'
g = bench.make_game(name)
sv = {w: DGSQP(*g.solver_args(), print_method=None, qp_warm_start=w) for w in (1, 0)}
for j in range(3):
    x0, u_tm = sample_scenarios(g, 1024, seed=1 + 1000 * j, solver=sv[1] if g.sampler == 'first_segment' else None)
    out = {}
    for w in (1, 0):
        sv[w].set_cooperative(0)
        r = sv[w].solve_batch(x0, u_tm)
        out[w] = r
        print(f'{name} batch {j} qp_warm_start={w}: kernel {r["kernel_ms"]:.1f} ms; mean QPs {r["qp_solves"].mean():.2f}')
    same = (out[0]['status'] == out[1]['status']) & (out[0]['num_iters'] == out[1]['num_iters']) & (out[0]['qp_solves'] == out[1]['qp_solves'])
    print('   identical control flow warm vs cold:', same.sum(), '/ 1024')
