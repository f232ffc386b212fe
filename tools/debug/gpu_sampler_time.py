"""Development aid: host (numpy) vs device PID warm start + collision rejection inside the scenario sampler."""
import sys, time, pathlib
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
from dgsqp_amd.montecarlo import kinematic_racing_game, dynamic_racing_game, sample_scenarios
from dgsqp_amd.solver import DGSQP
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for name, game in (('kb_curve_N25', kinematic_racing_game('curve', N=25)), ('dyn_curve_N25', dynamic_racing_game(N=25, rk4_substeps=10))):
    s = DGSQP(*game.solver_args(), print_method=None)
    sample_scenarios(game, 64, seed=1, solver=s)            # warm up
    t = time.time(); a = sample_scenarios(game, B, seed=1); th = time.time() - t
    t = time.time(); b = sample_scenarios(game, B, seed=1, solver=s); td = time.time() - t
    print(f'{name} B={B}: host sampler {th:.3f} s, device PID + collision {td:.3f} s ({th/td:.1f}x), same x0 {np.array_equal(a[0], b[0])}, max |du_ws| {np.abs(a[1]-b[1]).max():.1e}')
