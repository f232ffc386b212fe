"""Development aid (GPU box): which scenarios are the slow ones?  Solves B scenarios of a workload and saves inputs + iteration / QP
counts (+ the warm-start trajectory's closest approach) so that a cheap cost proxy for longest-first ticket order can be studied."""
import sys, pathlib
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
import bench
from dgsqp_amd.montecarlo import sample_scenarios
from dgsqp_amd.solver import DGSQP
name = sys.argv[1] if len(sys.argv) > 1 else 'dyn_curve_N25'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
out = sys.argv[3]
g = bench.make_game(name)
s = DGSQP(*g.solver_args(), print_method=None)
x0, u_tm = sample_scenarios(g, B, seed=7, solver=s)
ev = s.evaluate_batch(x0[:1], s._to_agent_major(u_tm)[:1])        # (warm up)
res = s.solve_batch(x0, u_tm)
ws = s.pid_warm_start_batch(x0, want_trajectories=True)
np.savez_compressed(out, x0=x0, u_ws=u_tm, status=res['status'], num_iters=res['num_iters'], qp_solves=res['qp_solves'], q_ws=ws['q_ws'])
print(name, B, 'kernel ms', res['kernel_ms'], 'qps mean', res['qp_solves'].mean(), 'max', res['qp_solves'].max())
