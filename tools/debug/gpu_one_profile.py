"""Development aid (GPU box, -DDG_PROF build): per-phase cycles of ONE scenario of the dyn bench batch (the slowest one by default)."""
import sys, pathlib, ctypes
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
from dgsqp_amd.montecarlo import dynamic_racing_game, sample_scenarios
from dgsqp_amd.solver import DGSQP
idx = int(sys.argv[1]) if len(sys.argv) > 1 else 278
game = dynamic_racing_game(N=25, rk4_substeps=10)
s = DGSQP(*game.solver_args(), print_method=None)
x0, uws = sample_scenarios(game, 1024, seed=1)
buf = (ctypes.c_ulonglong * 128)()
s._lib.dgsqp_prof_read(buf, 128)
res = s.solve_batch(x0[idx:idx + 1], uws[idx:idx + 1])
print('scenario', idx, 'status', res['status'], 'iters', res['num_iters'], 'qps', res['qp_solves'], 'kernel ms', res['kernel_ms'])
nph = s._lib.dgsqp_prof_read(buf, 128)
names = ['rollout', 'deriv1', 'deriv2', 'chains', 'dp', 'jacobi', 'pform', 'qp', 'merit', 'lsqr', 'qtmul', 'gi_adds/drops', 'wgtotal', 'wgmax', 'q_scan', 'q_y', 'q_dir', 'q_step', 'q_upd', 'q_refine', 'q_warm', 'w_build', 'w_mult', 'w_x', 'e_tri', 'e_bis', 'e_vec', 'e_back', 'e_kneg', 'c_nprev', 'c_mbuild', 'c_mwarm', 'c_mfinal', 'c_pruned_trials', 'h_inj', 'h_costate', 'h_contract', 'h_rows']
tot = buf[2 * 12]
for i in range(nph):
    if buf[2 * i + 1]:
        print(f'  {names[i]:16s} Mcycles {buf[2*i]/1e6:10.1f} calls {buf[2*i+1]:>7d} per call {buf[2*i]/buf[2*i+1]/1e3:10.1f} k share of the scenario {buf[2*i]/max(tot,1):.3f}')
