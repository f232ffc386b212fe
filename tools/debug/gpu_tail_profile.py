"""Development aid (GPU box, -DDG_PROF library): what is the tail of a launch made of?  Solves batch j of bench.py's own sampling,
lists the slowest scenarios by cycles and prints the per-phase cycles of each of them solved alone.
usage: DGSQP_HIP_LIB=dgsqp_amd/csrc/libdgsqp_hip_prof.so python tools/debug/gpu_tail_profile.py [workload] [batch index] [top]"""
import ctypes, sys, pathlib
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
import bench
from dgsqp_amd.montecarlo import sample_scenarios
from dgsqp_amd.solver import DGSQP
name = sys.argv[1] if len(sys.argv) > 1 else 'dyn_curve_N25'
j = int(sys.argv[2]) if len(sys.argv) > 2 else 0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 5
B = 1024
g = bench.make_game(name)
s = DGSQP(*g.solver_args(), print_method=None)
s.set_cooperative(0)
x0, u_tm = sample_scenarios(g, B, seed=1 + 1000 * j, solver=s if g.sampler == 'first_segment' and s.dims.M == 2 else None)
lib = s._lib
names = ['rollout', 'deriv1', 'deriv2', 'chains', 'dp', 'jacobi', 'pform', 'qp', 'merit', 'lsqr', 'qtmul', 'gi_adds/drops', 'wgtotal', 'wgmax', 'q_scan', 'q_y', 'q_dir', 'q_step', 'q_upd', 'q_refine', 'q_warm', 'w_build', 'w_mult', 'w_x', 'e_tri', 'e_bis', 'e_vec', 'e_back', 'e_kneg', 'c_nprev', 'c_mbuild', 'c_mwarm', 'c_mfinal', 'c_pruned_trials', 'h_inj', 'h_costate', 'h_contract', 'h_rows']
buf = (ctypes.c_ulonglong * 128)()
lib.dgsqp_prof_read(buf, 128)
res = s.solve_batch(x0, u_tm)
sc = (ctypes.c_ulonglong * B)()
assert lib.dgsqp_prof_scn(sc, B) == 0, 'needs the -DDG_PROF library'
cyc = np.array(sc[:], dtype=np.float64)
lib.dgsqp_prof_read(buf, 128)
order = np.argsort(-cyc)
print(f'{name} batch {j}: kernel {res["kernel_ms"]:.1f} ms (diagnostic build); total {cyc.sum()/1e9:.2f} Gcycles = {cyc.sum()/256/2.4e9*1e3:.1f} ms balanced over 256 CUs at 2.4 GHz; slowest {cyc.max()/2.4e6:.1f} ms')
for i in order[:top]:
    lib.dgsqp_prof_read(buf, 128)
    r1 = s.solve_batch(x0[i:i + 1], u_tm[i:i + 1])
    nph = lib.dgsqp_prof_read(buf, 128)
    tot = cyc[i]
    ph = {names[p]: (buf[2 * p], buf[2 * p + 1]) for p in range(nph) if buf[2 * p + 1]}
    main = ['rollout', 'deriv1', 'deriv2', 'chains', 'dp', 'jacobi', 'pform', 'qp', 'merit', 'lsqr', 'qtmul']
    print(f'scenario {i}: {tot/1e6:.0f} Mcycles in the batch, alone {r1["kernel_ms"]:.1f} ms; status {res["status"][i]} iters {res["num_iters"][i]} QPs {res["qp_solves"][i]}')
    print('   ' + '  '.join(f'{k} {ph[k][0]/1e6:.0f}M/{ph[k][1]}' for k in main if k in ph) + f"  | pruned trials {ph.get('c_pruned_trials', (0, 0))[0]}/{ph.get('c_pruned_trials', (0, 0))[1]}")
    if i == order[0]:
        print('   all counters: ' + '  '.join(f'{k} {v[0]/1e6:.1f}M/{v[1]}' for k, v in ph.items() if k not in main))
