"""Development aid (GPU box, -DDG_PROF library): per-phase cycles of single scenarios solved alone.
usage: DGSQP_HIP_LIB=dgsqp_amd/csrc/libdgsqp_hip_prof.so python tools/debug/gpu_scn_profile.py workload seed:index [seed:index ...]"""
import ctypes, sys, pathlib
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
import bench
from dgsqp_amd.montecarlo import sample_scenarios
from dgsqp_amd.solver import DGSQP
name = sys.argv[1]
g = bench.make_game(name)
s = DGSQP(*g.solver_args(), print_method=None)
lib = s._lib
names = ['rollout', 'deriv1', 'deriv2', 'chains', 'dp', 'jacobi', 'pform', 'qp', 'merit', 'lsqr', 'qtmul', 'gi_adds/drops', 'wgtotal', 'wgmax', 'q_scan', 'q_y', 'q_dir', 'q_step', 'q_upd', 'q_refine', 'q_warm', 'w_build', 'w_mult', 'w_x', 'e_tri', 'e_bis', 'e_vec', 'e_back', 'e_kneg', 'c_nprev', 'c_mbuild', 'c_mwarm', 'c_mfinal', 'c_pruned_trials', 'h_inj', 'h_costate', 'h_contract', 'h_rows']
buf = (ctypes.c_ulonglong * 128)()
cache = {}
for spec in sys.argv[2:]:
    seed, i = (int(v) for v in spec.split(':'))
    if seed not in cache:
        cache[seed] = sample_scenarios(g, 1024, seed=seed)
    x0, u_tm = cache[seed]
    for coop in (0, 1):
        s.set_cooperative(coop)
        lib.dgsqp_prof_read(buf, 128)
        r1 = s.solve_batch(x0[i:i + 1], u_tm[i:i + 1])
        nph = lib.dgsqp_prof_read(buf, 128)
        ph = {names[p]: (buf[2 * p], buf[2 * p + 1]) for p in range(nph) if buf[2 * p + 1]}
        print(f'seed {seed} scenario {i} coop {coop}: alone {r1["kernel_ms"]:.1f} ms; status {r1["status"][0]} iters {r1["num_iters"][0]} QPs {r1["qp_solves"][0]}')
        print('   ' + '  '.join(f'{k} {v[0]/1e6:.1f}M/{v[1]}' for k, v in ph.items()))
