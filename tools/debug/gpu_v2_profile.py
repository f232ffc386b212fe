"""Development aid (GPU box, -DDG_PROF library): phase cycles of DG-SQP v2 on the dynamic-bicycle curve game.
usage: DGSQP_HIP_LIB=dgsqp_amd/csrc/libdgsqp_hip_prof.so python tools/debug/gpu_v2_profile.py [B]"""
import ctypes, sys, pathlib
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
import bench
from dgsqp_amd.montecarlo import sample_scenarios
from dgsqp_amd.solver import DGSQP
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = bench.make_game('dyn_curve_N25_v2')
s = DGSQP(*g.solver_args(), print_method=None)
x0, u = sample_scenarios(g, B, seed=1)
lib = s._lib
buf = (ctypes.c_ulonglong * 128)()
lib.dgsqp_prof_read(buf, 128)
r = s.solve_batch(x0, u)
nph = lib.dgsqp_prof_read(buf, 128)
names = ['rollout', 'deriv1', 'deriv2', 'chains', 'dp', 'jacobi', 'pform', 'qp', 'merit', 'lsqr', 'qtmul', 'gi_adds/drops', 'wgtotal', 'wgmax', 'q_scan', 'q_y', 'q_dir', 'q_step', 'q_upd', 'q_refine', 'q_warm', 'w_build', 'w_mult', 'w_x', 'e_tri', 'e_bis', 'e_vec', 'e_back', 'e_kneg', 'c_nprev', 'c_mbuild', 'c_mwarm', 'c_mfinal', 'c_pruned_trials', 'h_inj', 'h_costate', 'h_contract', 'h_rows']
print(f'v2 B={B}: kernel {r["kernel_ms"]:.1f} ms; conv {np.mean(r["status"] <= 1):.3f}; mean iters {r["num_iters"].mean():.1f} qps {r["qp_solves"].mean():.1f}')
its = r['num_iters'].sum()
for p in range(nph):
    if buf[2 * p + 1]:
        print(f'  {names[p]:10s} {buf[2*p]/1e6:12.1f}M calls {buf[2*p+1]:9d} per call {buf[2*p]/buf[2*p+1]:10.0f} per iteration {buf[2*p]/max(its,1)/1e6:7.3f}M')
