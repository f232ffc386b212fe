"""GPU box: the device's OSQP restatement (qp_method='osqp', csrc/dgsqp_osqp.h) against the C++ restatement (oracle/osqp.hpp).
  part 1: single QPs through the test hook (dgsqp_qp_batch_info) at the start point of B scenarios -- status, ADMM iterations, polish
          verdict, rho, x and lambda;
  part 2: full solves, device vs oracle with the same qp_method.
usage: gpu_osqp_check.py <game of tools/ref_stats.py> [B] [B_solve]"""
import pathlib
import sys
import time

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tools')); sys.path.insert(0, str(ROOT / 'tests'))
import dgsqp_amd.montecarlo as mc  # noqa: E402
from dgsqp_amd.solver import DGSQP, build_problem, build_params, plan  # noqa: E402
from oracle import oracle  # noqa: E402
from ref_stats import GAMES  # noqa: E402

name = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
B2 = int(sys.argv[3]) if len(sys.argv) > 3 else B
g = GAMES[name][0]()
P, par = build_problem(*g.solver_args()), build_params(g.params, qp_method='osqp')
s = DGSQP(*g.solver_args(), print_method=None, qp_method='osqp')
print(name, plan(P, par))
x0, uws = mc.sample_scenarios(g, max(B, B2), seed=GAMES[name][1])
u = s._to_agent_major(uws)
l = np.array([oracle.dual_init(P, par, x0[b], u[b]) for b in range(B)])
t = time.time()
qp = s.qp_batch(x0[:B], u[:B], l)
print(f'device: {B} QPs in {time.time() - t:.3f} s')
same = 0
for b in range(B):
    ev = oracle.evaluate(P, x0[b], u[b], l[b], 1)
    Qpd = oracle.nearest_pd(ev['Q'], par.reg, par.eig_floor)
    xo, lo, io = oracle.osqp(Qpd, ev['q'], ev['G'], ev['g'])
    inf = qp['info'][b]
    key_d, key_o = (int(inf[0]), int(inf[1]), int(inf[2])), (io['status'], io['iters'], io['polished'])
    ex = np.abs(qp['du'][b] - xo).max() / max(1e-300, np.abs(xo).max())
    el = np.abs(qp['lhat'][b] - lo).max() / max(1.0, np.abs(lo).max())
    same += key_d == key_o
    print(f'  qp {b}: device (status, iters, polished) {key_d} rho {inf[3]:.4g} nact {int(inf[5])} res {inf[6]:.2e} {inf[7]:.2e} | oracle {key_o} rho {io["rho"]:.4g} nact {io["n_active"]} '
          f'res {io["pri_res"]:.2e} {io["dua_res"]:.2e} | rel dx {ex:.1e} dl {el:.1e} Qpd {np.abs(qp["Qpd"][b] - Qpd).max():.1e} flag {qp["flag"][b]}')
print(f'identical (status, iters, polished): {same}/{B}')
if B2 > 0:
    t = time.time()
    res = s.solve_batch(x0[:B2], uws[:B2])
    td = time.time() - t
    t = time.time()
    ref = oracle.solve_batch(P, par, x0[:B2], u[:B2], nthreads=min(B2, 32))
    print(f'full solves: device {td:.2f} s, oracle {time.time() - t:.1f} s')
    ident = (res['status'] == ref['status']) & (res['num_iters'] == ref['num_iters']) & (res['qp_solves'] == ref['qp_solves'])
    conv = (res['status'] <= 1) & (ref['status'] <= 1) & ident
    err = [np.abs(res['u'][b] - ref['u'][b]).max() / max(1.0, np.abs(ref['u'][b]).max()) for b in np.nonzero(conv)[0]]
    print(f'identical (status, iters, QPs) {ident.sum()}/{B2}; converged device {np.mean(res["status"] <= 1):.3f} oracle {np.mean(ref["status"] <= 1):.3f}; '
          f'iterates of the identical converged: median {np.median(err) if err else float("nan"):.1e} max {max(err) if err else float("nan"):.1e}')
    for b in np.nonzero(~ident)[0][:10]:
        print(f'    scn {b}: device {res["status"][b], res["num_iters"][b], res["qp_solves"][b]} oracle {ref["status"][b], ref["num_iters"][b], ref["qp_solves"][b]}')
