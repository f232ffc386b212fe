"""Development aid (GPU box, diagnostic -DDG_PROF library): how busy are the CUs while bench.py's pipelined steps run, and at which
clock?  Replays bench.py's launch schedule (P launches in flight over 3P staged batches, next launch when the previous one drains) and
reads the per-workgroup lifetime counters of the diagnostic build:
    occupancy = sum of workgroup lifetimes / (wall x CUs x clock),   clock = cycles / wall of the longest-lived workgroup.
usage: DGSQP_HIP_LIB=dgsqp_amd/csrc/libdgsqp_hip_prof.so python tools/debug/gpu_pipeline_occupancy.py [steps] [in_flight] [batch]"""
import ctypes as C, sys, time, pathlib
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
from dgsqp_amd import _ffi
from dgsqp_amd.montecarlo import dynamic_racing_game, sample_scenarios
from dgsqp_amd.solver import DGSQP
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 120
P = int(sys.argv[2]) if len(sys.argv) > 2 else 12
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
g = dynamic_racing_game(N=25, rk4_substeps=10)
nb = min(3 * P, steps)
solvers = [DGSQP(*g.solver_args(), print_method=None) for _ in range(nb)]
lib = solvers[0]._lib
for j, s in enumerate(solvers):
    x0, u_tm = sample_scenarios(g, B, seed=1 + 1000 * j, solver=solvers[0])
    u = np.ascontiguousarray(s._to_agent_major(u_tm))
    assert lib.dgsqp_stage_inputs(s._h, B, _ffi.dptr(np.ascontiguousarray(x0)), _ffi.dptr(u)) == 0
H = [s._h for s in solvers]
tm = _ffi.TimingT()
assert lib.dgsqp_solve_staged(H[0], C.byref(tm)) == 0
buf = (C.c_ulonglong * 128)()
lib.dgsqp_prof_read(buf, 128)          # (reading resets the counters: the staging / warm-up launch is discarded)
flying, last, nxt = [], None, 0
def retire_finished(block):
    while True:
        done = [i for i in flying if lib.dgsqp_finished(H[i])]
        for i in done:
            flying.remove(i)
            assert lib.dgsqp_wait(H[i], C.byref(tm)) == 0
        if done or not block or not flying:
            return
        time.sleep(0.0002)
t0 = time.perf_counter()
for step in range(steps):
    if last is not None and P > 1:
        while not lib.dgsqp_draining(H[last]):
            time.sleep(0.0002)
    retire_finished(False)
    while len(flying) >= P:
        retire_finished(True)
    while True:
        i, nxt = nxt, (nxt + 1) % nb
        if i not in flying:
            break
    assert lib.dgsqp_launch_staged(H[i]) == 0
    flying.append(i); last = i
while flying:
    assert lib.dgsqp_wait(H[flying.pop(0)], C.byref(tm)) == 0
wall = time.perf_counter() - t0
lib.dgsqp_prof_read(buf, 128)
PH_WGTOTAL, PH_WGMAX = 12, 13
wg_cycles = buf[2 * PH_WGTOTAL]
print(f'steps {steps} in flight {P} B {B}: wall {wall:.3f} s -> {steps * B / wall:.0f} scen/s (diagnostic build)')
print(f'sum of workgroup lifetimes {wg_cycles / 1e9:.1f} Gcycles; longest workgroup: {buf[2 * PH_WGMAX] / 1e9:.3f} Gcycles in {buf[2 * PH_WGMAX + 1] * 1e-8:.3f} s '
      f'-> clock {buf[2 * PH_WGMAX] / max(buf[2 * PH_WGMAX + 1], 1) / 1e-8 / 1e9:.3f} GHz')
for f in (2.4e9, buf[2 * PH_WGMAX] / max(buf[2 * PH_WGMAX + 1], 1) * 1e8):
    print(f'occupancy of 256 CUs at {f / 1e9:.2f} GHz: {wg_cycles / (wall * 256 * f):.3f}')
