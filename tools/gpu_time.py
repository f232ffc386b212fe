"""Development aid: time solve_batch for a config on the GPU box (with the -DDG_PROF library: per-phase cycle counters).
usage: gpu_time.py <dyn|kbcurve|kbchicane|agents3|kbcurve50|any workload name of bench.py> <N> <B> [rk4 substeps]   (environment: DGSQP_QP_METHOD, DGSQP_MIXED=1)"""
import sys, time, pathlib
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import os
from dgsqp_amd.montecarlo import kinematic_racing_game, dynamic_racing_game, sample_scenarios
from dgsqp_amd.solver import DGSQP
which, N, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
M = int(sys.argv[4]) if len(sys.argv) > 4 else 10
if which in __import__('bench').WORKLOADS:      # a bench.py workload by name (N is the workload's own)
    game = __import__('bench').make_game(which)
elif which == 'agents3':      # XL layout: scripts/DGSQP_monte_carlo_agents.py at M=3, N=25
    game = kinematic_racing_game('curve', N=N, M=3)
elif which == 'kbcurve50':    # XL layout at n = 200 (BASELINE configs[3]'s size on the curve track)
    game = kinematic_racing_game('curve', N=N)
elif which.startswith('kb'):  # reg as in the scripts: curve.py:161 reg=0, chicane.py:164 reg=1e-3
    game = kinematic_racing_game('curve' if which == 'kbcurve' else 'chicane', N=N, reg=0.0 if which == 'kbcurve' else 1e-3)
else:
    game = dynamic_racing_game(N=N, rk4_substeps=M)
import os
s = DGSQP(*game.solver_args(), print_method=None, qp_method=os.environ.get('DGSQP_QP_METHOD') or None,
          mixed_precision=bool(os.environ.get('DGSQP_MIXED')))
t = time.time(); x0, uws = sample_scenarios(game, B, seed=1); print('sample', time.time() - t)
for rep in range(2):
    t = time.time(); res = s.solve_batch(x0, uws); dt = time.time() - t
    st = res['status']
    print(f'{which} N={N} B={B} wall {dt:.3f}s kernel {res["kernel_ms"]:.1f} ms -> {B/dt:.1f} scen/s | conv {np.mean(st<=1):.3f} abs {np.mean(st==0):.3f} maxit {np.mean(st==2):.3f} qpfail {np.mean(st==4):.3f} div {np.mean(st==3):.3f} | mean iters {res["num_iters"].mean():.2f} (conv {res["num_iters"][st<=1].mean():.2f}) qps {res["qp_solves"].mean():.2f}')
import ctypes, os
lib = s._lib
if hasattr(lib, 'dgsqp_prof_read'):
    buf = (ctypes.c_ulonglong * 256)()
    nph = lib.dgsqp_prof_read(buf, 256)
    names = ['rollout', 'deriv1', 'deriv2', 'chains', 'dp', 'jacobi', 'pform', 'qp', 'merit', 'lsqr', 'qtmul', 'gi_adds/drops', 'wgtotal', 'wgmax(clk,wall100MHz)', 'q_scan', 'q_y', 'q_dir', 'q_step', 'q_upd', 'q_refine', 'q_warm', 'w_build', 'w_mult', 'w_x', 'e_tri', 'e_bis', 'e_vec', 'e_back', 'e_kneg', 'c_nprev', 'c_mbuild', 'c_mwarm', 'c_mfinal', 'c_pruned_trials', 'h_inj', 'h_costate', 'h_contract', 'h_rows', 'osqp_scale', 'osqp_w', 'osqp_kinv', 'osqp_admm', 'osqp_iters(count)', 'osqp_check', 'osqp_pol_inv', 'osqp_pol_rows', 'osqp_pol_solve', 'osqp_nact(count)', 'osqp_it_gt', 'osqp_it_pmul', 'osqp_it_gs', 'osqp_it_upd', 'tri_column', 'tri_product', 'tri_w', 'tri_update', 'qwarm_blocks', 'qwarm_rows', 'qwarm_final', 'qwarm_nprev(count)', 'qwarm_drops(count)', 'psd_pd(LDS: shortcut tried / hit rate; XL: calls / share without a negative eigenvalue)']
    tot = sum(buf[2 * i] for i in range(nph))
    for i in range(nph):
        if buf[2 * i + 1]:
            print(f'  {names[i]:8s} cycles {buf[2*i]:>16d} calls {buf[2*i+1]:>9d} per call {buf[2*i]/buf[2*i+1]:>12.0f} share {buf[2*i]/max(tot,1):.3f}')

if hasattr(lib, 'dgsqp_prof_scn'):
    sc = (ctypes.c_ulonglong * B)()
    if lib.dgsqp_prof_scn(sc, B) == 0:
        cyc = np.array(sc[:], dtype=np.float64)
        order = np.argsort(-cyc)
        print('total Mcycles', cyc.sum() / 1e6, 'mean', cyc.mean() / 1e6, 'max', cyc.max() / 1e6)
        for i in order[:12]:
            print(f'  scn {i:5d} Mcyc {cyc[i]/1e6:9.1f} status {st[i]} iters {res["num_iters"][i]} qps {res["qp_solves"][i]} Mcyc/qp {cyc[i]/1e6/max(res["qp_solves"][i],1):.2f}')
        qps = res['qp_solves'].astype(float)
        print('corr Mcyc/qp overall', cyc.sum() / 1e6 / qps.sum())
        for lo, hi in ((0, 10), (10, 20), (20, 50), (50, 100), (100, 1000)):
            m = (qps >= lo) & (qps < hi)
            if m.any(): print(f'  qps in [{lo},{hi}): n={m.sum()} Mcyc/qp {cyc[m].sum()/1e6/qps[m].sum():.2f} share of total {cyc[m].sum()/cyc.sum():.3f}')
