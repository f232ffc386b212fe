"""Development aid: phase profile of selected scenarios of a sampled batch (prof build)."""
import sys, time, pathlib, ctypes
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
from dgsqp_amd.montecarlo import kinematic_racing_game, dynamic_racing_game, sample_scenarios
from dgsqp_amd.solver import DGSQP
which, N, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
sel = [int(v) for v in sys.argv[4].split(',')]
game = kinematic_racing_game('curve' if which == 'kbcurve' else 'chicane', N=N) if which.startswith('kb') else dynamic_racing_game(N=N, rk4_substeps=10)
s = DGSQP(*game.solver_args(), print_method=None)
x0, uws = sample_scenarios(game, B, seed=1)
lib = s._lib
names = ['rollout', 'deriv1', 'deriv2', 'chains', 'dp', 'jacobi', 'pform', 'qp', 'merit', 'lsqr', 'qtmul', 'gi_adds/drops', 'wgtotal', 'wgmax', 'q_scan', 'q_y', 'q_dir', 'q_step', 'q_upd', 'q_refine', 'q_warm', 'w_build', 'w_mult', 'w_x', 'e_tri', 'e_bis', 'e_vec', 'e_back', 'e_kneg', 'c_nprev', 'c_mbuild', 'c_mwarm', 'c_mfinal', 'c_pruned_trials', 'h_inj', 'h_costate', 'h_contract', 'h_rows']
for i in sel:
    buf = (ctypes.c_ulonglong * 128)()
    lib.dgsqp_prof_read(buf, 128)
    res = s.solve_batch(x0[i:i + 1], uws[i:i + 1])
    nph = lib.dgsqp_prof_read(buf, 128)
    print(f'scenario {i}: status {res["status"][0]} iters {res["num_iters"][0]} qps {res["qp_solves"][0]} kernel {res["kernel_ms"]:.1f} ms')
    for p in range(nph):
        if buf[2 * p + 1]:
            print(f'  {names[p]:14s} Mcyc {buf[2*p]/1e6:10.1f} calls {buf[2*p+1]:>8d} per call {buf[2*p]/buf[2*p+1]:>10.0f}')
