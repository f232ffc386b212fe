"""Parity of the HIP path (through the C-ABI) with the CPU oracle, on a real MI355X.

Bars: integer outputs (status, iteration counts, QP-solve counts) identical; floating-point iterates within
1e-5 relative (north_star), stage-level quantities far tighter (stated per test).  Full solves are compared
with a converged LSQR dual start (conftest.tight_lsqr) because scipy's default 1e-6 LSQR tolerance leaves a
~1e-4 spread between any two correct implementations (tests/test_oracle.py::test_lsqr_matches_scipy)."""
import pathlib

import numpy as np
import pytest

from conftest import agent_major, assert_control_flow_parity, stable_mask, tight_lsqr

pytestmark = pytest.mark.gpu
GOLD = pathlib.Path(__file__).parent / 'golden'


def rel(a, b):
    return np.abs(a - b).max() / max(1e-300, np.abs(b).max())


@pytest.fixture(scope='module')
def solvers(games):
    from dgsqp_amd.solver import DGSQP
    return {name: DGSQP(*g.solver_args(), print_method=None, lsqr_tol=1e-13) for name, (g, P, par) in games.items()}


def test_native_library_is_loaded(solvers):
    """The product path is the HIP library (in-tree .so), on a gfx950 device."""
    import ctypes
    from dgsqp_amd import _ffi
    buf = ctypes.create_string_buffer(256)
    assert _ffi.load_library().dgsqp_backend_info(buf, 256) == 0
    assert b'gfx950' in buf.value
    assert _ffi.library_path().name == 'libdgsqp_hip.so' and _ffi.library_path().exists()
    s = solvers['kb_chicane_N25']
    assert s.dims.lds_bytes <= 163840 and s.dims.n == 100 and s.dims.n_c == 525 and s.dims.n_dense == 75


@pytest.mark.parametrize('name', ['kb_chicane_N15', 'kb_chicane_N25', 'kb_curve_N10', 'dyn_curve_N15', 'dyn_curve_N25'])
def test_evaluate_parity(oracle, games, solvers, name):
    """_evaluate (DGSQP.py:509-533): rollout, q, g, G, raw Q.  Tolerance 1e-12 relative (fp64, different
    derivative techniques: Taylor directions on device vs dense jets in the oracle)."""
    from dgsqp_amd.montecarlo import sample_scenarios
    g, P, par = games[name]
    s = solvers[name]
    B = 6
    x0, u_tm = sample_scenarios(g, B, seed=21)
    rng = np.random.default_rng(0)
    u = agent_major(u_tm) + 0.01 * rng.standard_normal((B, s.n))
    l = np.maximum(0, rng.standard_normal((B, s.n_c_total)))
    ev = s.evaluate_batch(x0, u, l)
    for b in range(B):
        o = oracle.evaluate(P, x0[b], u[b], l[b], 1)
        for key in ('x', 'q', 'g', 'G', 'Q'):
            assert rel(ev[key][b], o[key]) < 1e-12, (key, b)
        l0 = oracle.dual_init(P, tight_lsqr(par), x0[b], u[b])
        assert rel(ev['l0'][b], l0) < 1e-6, b          # LSQR run to 1e-13: both reach the min-norm solution


@pytest.mark.parametrize('name', ['kb_chicane_N15', 'kb_chicane_N25', 'dyn_curve_N15'])
def test_qp_parity_and_kkt(oracle, games, solvers, name):
    """_nearestPD + _solve_qp (DGSQP.py:232-266, 1290-1296): projected Hessian to 1e-11, primal step to 1e-8,
    multipliers to 1e-6 relative, identical active sets, KKT residuals of the device answer itself."""
    from dgsqp_amd.montecarlo import sample_scenarios
    g, P, par = games[name]
    s = solvers[name]
    B = 6
    x0, u_tm = sample_scenarios(g, B, seed=22)
    u = agent_major(u_tm)
    l = np.array([oracle.dual_init(P, tight_lsqr(par), x0[b], u[b]) for b in range(B)])
    qp = s.qp_batch(x0, u, l)
    for b in range(B):
        o = oracle.evaluate(P, x0[b], u[b], l[b], 1)
        Qpd = oracle.nearest_pd(o['Q'], par.reg)
        du, lam, flag = oracle.qp(Qpd, o['q'], o['G'], o['g'])
        assert qp['flag'][b] == flag
        # absolute error of the projection scales with the RAW Hessian (its large negative part is projected away)
        assert np.abs(qp['Qpd'][b] - Qpd).max() < 1e-11 * max(1.0, np.abs(o['Q']).max())
        if flag != 0:
            continue
        cond_scale = max(1.0, np.abs(o['Q']).max() / max(1e-300, np.abs(Qpd).max()))
        assert rel(qp['du'][b], du) < 1e-8 * cond_scale and rel(qp['lhat'][b], lam) < 1e-6 * cond_scale
        assert np.array_equal(qp['lhat'][b] > 0, lam > 0)
        d, lh = qp['du'][b], qp['lhat'][b]
        scale = max(1.0, np.abs(o['q']).max())
        assert np.abs(Qpd @ d + o['q'] + o['G'].T @ lh).max() < 1e-7 * scale
        assert (o['G'] @ d + o['g']).max() < 1e-12 and lh.min() >= 0
        assert np.abs(lh * (o['G'] @ d + o['g'])).max() < 1e-10 * scale


ABLATION = [f'ablation_N{N}_{nm}_{mf}' for N in (15, 25) for nm in ('nms', 'ls') for mf in ('stat_l1', 'stat')]


@pytest.mark.parametrize('name', ['kb_chicane_N15', 'kb_curve_N10', 'dyn_curve_N15', 'dyn_curve_N25', 'kb_barc2_N15', 'merge_N8'] + ABLATION)
def test_solve_matches_golden_fixtures(solvers, name):
    """Committed oracle solutions (tools/make_golden.py, literal parameters): identical flags / iteration / QP counts on the
    scenarios the oracle itself reproduces under 1e-13 input perturbations (the fixture's ``stable`` mask), iterates of the
    identical ones within 1e-5 relative (north_star); forks on the unstable rest are printed, not hidden.
    ``ablation_N*``: the reference's ablation study (scripts/DGSQP_monte_carlo_ablation.py:166-197, theta = 90 degrees, car 2 with
    blocking and soft-obstacle costs) in all four combinations nonmono_ls x merit_function ('nms' = watchdog, 'ls' = plain
    backtracking line search, the DGSQPParams default), 32 scenarios each at N = 15 and N = 25."""
    gold = np.load(GOLD / f'{name}.npz')
    s = solvers[name]
    res = s.solve_batch(gold['x0'], gold['u_ws'])
    # (reg = 0 games -- kb_barc2_N15, merge_N8 -- run the literal 1e-10 floor and are held to the same bar since both sides polish
    # their QPs, round 3)
    same = assert_control_flow_parity(res, gold, gold['stable'], name, min_stable_same=0.95, max_conv_gap=0.05)
    assert same.mean() >= 0.85            # an absolute floor next to the stable-subset one
    for b in np.where(same & (gold['status'] <= 1))[0]:
        assert rel(res['u'][b], gold['u'][b]) < 1e-5, b
        if gold['status'][b] == 0:
            assert rel(res['l'][b], gold['l'][b]) < 1e-5, b
            assert rel(res['cost'][b], gold['cost'][b]) < 1e-8, b


# identical-path counts measured at the literal LSQR setting (profiles/r05_gpu_tests_parity_lines.txt); the floor of each fixture is ONE scenario below
LITERAL_LSQR_IDENTICAL = {'kb_chicane_N15': (32, 32), 'kb_curve_N10': (32, 32), 'dyn_curve_N15': (16, 16), 'dyn_curve_N25': (56, 64), 'kb_barc2_N15': (32, 32), 'merge_N8': (16, 16)}


@pytest.mark.parametrize('name', ['kb_chicane_N15', 'kb_curve_N10', 'dyn_curve_N15', 'dyn_curve_N25', 'kb_barc2_N15', 'merge_N8'])
def test_identity_at_the_reference_lsqr_setting_is_tracked(oracle, games, name):
    """The strict parity tests above run with a converged LSQR dual start (lsqr_tol = 1e-13); the reference (DGSQP.py:324) and bench.py run
    scipy's default 1e-6.  This test runs the LITERAL default on both sides -- device and oracle on the scenarios of the golden fixture,
    nothing overridden -- and reports the identical-path fraction, so that the number at the reference's own setting is tracked from
    round to round.  Each fixture is held to its OWN measured count minus one scenario (round 5: 100 % on five fixtures, 56/64 on
    dyn_curve_N25 -- at 1e-6 two correct LSQR implementations stop a Lanczos step apart on some scenarios, l0 differs by 1e-4), converged
    fractions within 2 scenarios: a regression of two scenarios anywhere fails (the global 70 % floor of round 5 would have let a
    quarter of them go)."""
    from dgsqp_amd.solver import DGSQP
    gold = np.load(GOLD / f'{name}.npz')
    g, P, par = games[name]
    s = DGSQP(*g.solver_args(), print_method=None)                  # defaults: no lsqr_tol, no eig_floor, exact QP
    res = s.solve_batch(gold['x0'], gold['u_ws'])
    ref = oracle.solve_batch(P, par, gold['x0'], agent_major(gold['u_ws']) if gold['u_ws'].shape[2] == 4 else s._to_agent_major(gold['u_ws']), nthreads=8)
    same = ((res['status'] == ref['status']) & (res['num_iters'] == ref['num_iters']) & (res['qp_solves'] == ref['qp_solves']))
    cd, cr = res['status'] <= 1, ref['status'] <= 1
    idc = np.nonzero(same & cd & cr)[0]
    err = max([rel(res['u'][b], ref['u'][b]) for b in idc], default=0.0)
    print(f'{name} at the reference LSQR setting (1e-6): identical (status, iterations, QPs) on {same.sum()}/{len(same)} scenarios; converged device {cd.mean():.3f} '
          f'oracle {cr.mean():.3f}; largest relative iterate difference of the identical converged ones {err:.1e}')
    measured, total = LITERAL_LSQR_IDENTICAL[name]
    assert len(same) == total and same.sum() >= measured - 1, (int(same.sum()), measured)
    assert abs(int(cd.sum()) - int(cr.sum())) <= 2
    assert err < 5e-3


def test_baseline_config1_dyn_curve_N25_parity(solvers):
    """BASELINE configs[1] at its own size (2-agent dynamic bicycle, Pacejka, rk4 M=10, N=25; 64 committed oracle solutions):
    identical (status, iterations, QP solves) on the oracle-stable scenarios, >= 85 % overall (the oracle reproduces ITSELF on
    57/64 = 89 % under 1e-13 perturbations), converged fraction within 5 points, iterates within 1e-5 relative; and the
    convergence statistics against the numpy loop with the restated OSQP (tests/golden/pyref_osqp_*.npz, tools/ref_stats.py)."""
    gold = np.load(GOLD / 'dyn_curve_N25.npz')
    res = solvers['dyn_curve_N25'].solve_batch(gold['x0'], gold['u_ws'])
    same = assert_control_flow_parity(res, gold, gold['stable'], 'dyn_curve_N25')
    assert same.mean() >= 0.85
    for b in np.where(same & (gold['status'] <= 1))[0]:
        assert rel(res['u'][b], gold['u'][b]) < 1e-5, b
        assert rel(res['l'][b], gold['l'][b]) < 1e-5, b


@pytest.mark.parametrize('name', ['dyn_curve_N25', 'kb_curve_N25', 'kb_chicane_N25', 'kb_barc2_N15', 'merge_N20'])
def test_convergence_statistics_against_the_restated_osqp_loop(name):
    """north_star: "matching reference convergence rate".  The yardstick is the line-by-line numpy restatement of the
    reference loop with the restated OSQP as its QP (oracle/pyref.py + oracle/osqp_restate.py; scipy lsqr at its default
    tolerance, numpy eigh), run in the build container on the first 256 scenarios of each sampler and committed
    (tools/ref_stats.py, profiles/r03_pyref_osqp_vs_oracle.txt).  OSQP stops ADMM at 1e-3 and its polish is accepted on residuals
    alone (negative multipliers included), so individual paths differ; the Monte-Carlo statistics the reference reports
    (process_data_curve.py:99-110) must agree: converged fraction within 5 points, same converged flag on >= 90 % of the scenarios,
    mean iterations of the commonly converged within 0.5."""
    from dgsqp_amd import montecarlo as mc
    from dgsqp_amd.solver import DGSQP
    ref = np.load(GOLD / f'pyref_osqp_{name}.npz')
    g = {'dyn_curve_N25': lambda: mc.dynamic_racing_game(N=25, rk4_substeps=10), 'kb_curve_N25': lambda: mc.kinematic_racing_game('curve', N=25, reg=0.0),
         'kb_chicane_N25': lambda: mc.kinematic_racing_game('chicane', N=25), 'kb_barc2_N15': lambda: mc.barc_racing_game(N=15, M=2),
         'merge_N20': lambda: mc.merge_game(N=20)}[name]()
    assert len(ref['status']) >= 256
    res = DGSQP(*g.solver_args(), print_method=None).solve_batch(ref['x0'], ref['u_ws'])      # all defaults: literal formulas, scipy's LSQR tolerance
    cd, cr = res['status'] <= 1, ref['status'] <= 1
    both = cd & cr
    print(name, 'converged device', cd.mean(), 'restated-OSQP loop', cr.mean(), 'same flag', np.mean(cd == cr),
          'mean iters (commonly converged)', res['num_iters'][both].mean(), ref['num_iters'][both].mean())
    assert abs(cd.mean() - cr.mean()) <= 0.05
    assert np.mean(cd == cr) >= 0.90
    assert abs(res['num_iters'][both].mean() - ref['num_iters'][both].mean()) <= 0.5


# ---- qp_method = 'osqp': the reference's own QP arithmetic (csrc/dgsqp_osqp.h) --------------------------------------------------------
REF_GAMES = {'dyn_curve_N25': lambda mc: mc.dynamic_racing_game(N=25, rk4_substeps=10), 'kb_curve_N25': lambda mc: mc.kinematic_racing_game('curve', N=25, reg=0.0),
             'kb_chicane_N25': lambda mc: mc.kinematic_racing_game('chicane', N=25), 'kb_barc2_N15': lambda mc: mc.barc_racing_game(N=15, M=2),
             'merge_N20': lambda mc: mc.merge_game(N=20)}


@pytest.mark.parametrize('name', sorted(REF_GAMES))
def test_device_osqp_matches_the_cpu_restatement(oracle, name):
    """The ADMM + polish kernel (one workgroup per QP, reduced KKT system through an explicit inverse, polish in unscaled variables)
    against the dense C++ restatement of OSQP (oracle/osqp.hpp, itself held to the numpy restatement by
    tests/test_oracle.py::test_cpp_osqp_follows_the_numpy_restatement) on 64 game QPs per workload: the first 32 scenarios of the
    sampler at two linearisation points each -- (u_ws, dual start) and the point after the first full step (u + du, lhat).
    Same OSQP status, same ADMM iteration count, same polish verdict (accepted / rejected) and same rho on at least 62 of the 64;
    on those, x and lambda within 1e-6 relative (the polished KKT point or, where the polish is rejected, the ADMM iterate)."""
    from concurrent.futures import ThreadPoolExecutor
    from dgsqp_amd import montecarlo as mc
    from dgsqp_amd.solver import DGSQP, build_problem, build_params
    g = REF_GAMES[name](mc)
    ref = np.load(GOLD / f'pyref_osqp_{name}.npz')
    P, par = build_problem(*g.solver_args()), build_params(g.params, qp_method='osqp')
    s = DGSQP(*g.solver_args(), print_method=None, qp_method='osqp')
    B = 32
    x0, u = ref['x0'][:B], s._to_agent_major(ref['u_ws'][:B])
    l = np.array([oracle.dual_init(P, par, x0[b], u[b]) for b in range(B)])
    qp1 = s.qp_batch(x0, u, l)
    ok1 = qp1['flag'] == 0
    u2, l2 = np.where(ok1[:, None], u + qp1['du'], u), np.where(ok1[:, None], qp1['lhat'], l)
    qp2 = s.qp_batch(x0, u2, l2)

    def cpu(args):
        xb, ub, lb = args
        ev = oracle.evaluate(P, xb, ub, lb, 1)
        return oracle.osqp(oracle.nearest_pd(ev['Q'], par.reg, par.eig_floor), ev['q'], ev['G'], ev['g'])
    with ThreadPoolExecutor(8) as ex:
        cpu1 = list(ex.map(cpu, [(x0[b], u[b], l[b]) for b in range(B)]))
        cpu2 = list(ex.map(cpu, [(x0[b], u2[b], l2[b]) for b in range(B)]))
    same, polished, rejected, rho_updates, worst_x, worst_l = 0, 0, 0, 0, 0.0, 0.0
    for qp, cpus in ((qp1, cpu1), (qp2, cpu2)):
        for b in range(B):
            xo, lo, io = cpus[b]
            inf = qp['info'][b]
            if (int(inf[0]), int(inf[1]), int(inf[2])) != (io['status'], io['iters'], io['polished']) or abs(inf[3] - io['rho']) > 1e-6 * io['rho']:
                print(name, 'QP', b, 'device', inf[:6], 'oracle', io)
                continue
            same += 1
            polished += io['polished'] == 1
            rejected += io['polished'] == -1
            rho_updates += io['rho_updates'] > 0
            assert int(inf[5]) == io['n_active'] and (qp['flag'][b] != 0) == (io['status'] in (-3, -4, -10))
            if io['status'] in (-3, -4, -10):
                continue
            worst_x = max(worst_x, np.abs(qp['du'][b] - xo).max() / max(1.0, np.abs(xo).max()))
            worst_l = max(worst_l, np.abs(qp['lhat'][b] - lo).max() / max(1.0, np.abs(lo).max()))
    print(f'{name}: OSQP on the device vs oracle/osqp.hpp: identical (status, iterations, polish verdict, rho) on {same}/64 QPs ({polished} polished, {rejected} polish '
          f'rejected, {rho_updates} with rho updates); x within {worst_x:.1e}, lambda within {worst_l:.1e}')
    assert same >= 62 and polished >= 32
    assert worst_x < 1e-6 and worst_l < 1e-6


@pytest.mark.parametrize('name', sorted(REF_GAMES))
def test_osqp_solves_follow_the_numpy_loop_with_the_restated_osqp(name):
    """Full solves with qp_method='osqp' against the line-by-line numpy restatement of the reference loop with the numpy restatement
    of OSQP as its QP (oracle/pyref.py + oracle/osqp_restate.py; numpy.linalg.eigh, scipy's lsqr), the closest stand-in for the
    reference's own iterates (tests/golden/pyref_osqp_*.npz, 256 scenarios per workload, tools/ref_stats.py).  Identical (status,
    iterations, QP solves) on the scenarios the numpy loop itself reproduces under 1e-13 input perturbations (its `stable` mask; a
    solve that raises in the reference -- status 4 -- has no counts to compare), iterates of the identical converged ones within 1e-5;
    converged fractions within 2 points.  With the exact active-set QP the same comparison gives 52-79 % identical paths and iterate
    differences of 1e-4 (profiles/r04_osqp_vs_pyref.txt)."""
    from dgsqp_amd import montecarlo as mc
    from dgsqp_amd.solver import DGSQP
    ref = np.load(GOLD / f'pyref_osqp_{name}.npz')
    g = REF_GAMES[name](mc)
    res = DGSQP(*g.solver_args(), print_method=None, qp_method='osqp').solve_batch(ref['x0'], ref['u_ws'])
    same = ((res['status'] == ref['status']) & (res['num_iters'] == ref['num_iters']) & (res['qp_solves'] == ref['qp_solves'])) | ((res['status'] == 4) & (ref['status'] == 4))
    stable = ref['stable']
    cd, cr = res['status'] <= 1, ref['status'] <= 1
    idc = np.nonzero(same & cd & cr)[0]
    err = np.array([rel(res['u'][b], ref['u'][b]) for b in idc])
    print(f'{name}: qp_method osqp vs numpy loop + restated OSQP: identical {same.mean():.3f} of all, {same[stable].mean():.3f} of the {stable.sum()} numpy-stable scenarios; '
          f'converged {cd.mean():.3f} vs {cr.mean():.3f}; same flag {np.mean(cd == cr):.3f}; iterates of the identical converged: median {np.median(err):.1e}, '
          f'above 1e-5: {int((err > 1e-5).sum())} of {len(err)}')
    assert stable.mean() >= 0.5
    assert same[stable].mean() >= 0.95
    assert same.mean() >= 0.80
    assert abs(cd.mean() - cr.mean()) <= 0.02 and np.mean(cd == cr) >= 0.95
    assert np.mean(err > 1e-5) <= 0.02 and np.median(err) < 1e-7


# ---- BASELINE configs[2], [3], [4] (XL layout, n = 150 / 200 / 300) against the restated OSQP: dgsqp_osqp_xl.h ---------------------------------
XL_REF_GAMES = {'kb_barc3_N25': lambda mc: mc.barc_racing_game(N=25, M=3), 'kb_f1_N50': lambda mc: mc.f1_racing_game(N=50),
                'merge6_N25': lambda mc: mc.merge_game(N=25, M=6)}


@pytest.mark.parametrize('name', sorted(XL_REF_GAMES))
def test_xl_convergence_statistics_against_the_restated_osqp_loop(name):
    """configs[2], [3], [4] at their own sizes with the DEFAULT (exact active-set) QP against the numpy loop with the restated OSQP
    (tests/golden/pyref_osqp_<game>.npz: the first 64 scenarios of each sampler, tools/ref_stats.py): converged fraction within 5
    points (+ one scenario), the same converged flag on >= 90 %.  On the circuit game 95 % of the solves end in an infeasible QP on
    both sides -- a property of the game (every such verdict is LP-certified, tests/test_oracle.py::test_infeasible_verdicts_are_
    backed_by_an_lp), not of the QP solver: the OSQP loop fails on 97 % of them."""
    from dgsqp_amd import montecarlo as mc
    from dgsqp_amd.solver import DGSQP
    if not (GOLD / f'pyref_osqp_{name}.npz').exists():
        pytest.skip('fixture not generated yet (tools/ref_stats.py: the numpy loop with the numpy OSQP takes hours at this size)')
    ref = np.load(GOLD / f'pyref_osqp_{name}.npz')
    g = XL_REF_GAMES[name](mc)
    assert len(ref['status']) >= 64
    res = DGSQP(*g.solver_args(), print_method=None, qp_method='active_set').solve_batch(ref['x0'], ref['u_ws'])
    cd, cr = res['status'] <= 1, ref['status'] <= 1
    both = cd & cr
    print(name, 'converged device (exact QP)', cd.mean(), 'restated-OSQP loop', cr.mean(), 'same flag', np.mean(cd == cr),
          'mean iters (commonly converged)', res['num_iters'][both].mean() if both.any() else None, ref['num_iters'][both].mean() if both.any() else None,
          'qp_fail / exception', np.mean(res['status'] == 4), np.mean(ref['status'] == 4))
    if name == 'merge6_N25':
        # The one config where the two QPs do NOT give the same statistics: at reg = 0 (eigenvalues floored at 1e-10) the restated OSQP
        # stops at its 4,000-iteration limit in 83 % of its calls (tools/ref_stats.py: 2,401 of 2,896 calls not 'solved') and the SQP
        # continues from unconverged ADMM iterates -- 81 % converged after 14.4 iterations and 45 QPs per solve; the exact QP converges on
        # 98 % after 7.9.  The exact QP is held to "not worse"; the device's OSQP arithmetic is held to the yardstick itself in
        # test_xl_osqp_solves_follow_the_numpy_loop_with_the_restated_osqp.
        assert cd.mean() >= cr.mean() - 0.05 and cd.mean() >= 0.95
        return
    assert abs(cd.mean() - cr.mean()) <= 0.05 + 1.0 / len(cd)
    if name == 'kb_f1_N50':
        # long horizon on the F1 track: WHICH scenarios converge is decided by rounding -- the C++ oracle with the exact QP and the numpy
        # loop agree on the converged flag of 55 % of the scenarios (coin-flip agreement at 52 % converged would be 50 %), commonly
        # converged ones end in different equilibria (median iterate difference 0.6; profiles/r05_pyref_osqp_kb_f1_N50.txt).  Only the
        # Monte-Carlo statistics are comparable: converged fraction (above) and mean iterations of the converged within 20 %.
        assert abs(res['num_iters'][cd].mean() - ref['num_iters'][cr].mean()) <= 0.2 * ref['num_iters'][cr].mean()
        return
    assert np.mean(cd == cr) >= 0.90
    if both.sum() >= 16:
        assert abs(res['num_iters'][both].mean() - ref['num_iters'][both].mean()) <= 0.5


@pytest.mark.parametrize('name', sorted(XL_REF_GAMES))
def test_xl_device_osqp_matches_the_cpu_restatement(oracle, name):
    """dgsqp_osqp_xl.h (OSQP's ADMM + polish with the matrices in the L2 scratch, K^-1 through the blocked elimination) against
    oracle/osqp.hpp on 16 QPs per game: 8 scenarios at (u_ws, dual start) and at the point after the first full step.  Same status, ADMM
    iteration count, polish verdict, rho and number of active rows on at least 15; x and lambda within 1e-6 relative on those."""
    from concurrent.futures import ThreadPoolExecutor
    from dgsqp_amd import montecarlo as mc
    from dgsqp_amd.solver import DGSQP, build_problem, build_params
    g = XL_REF_GAMES[name](mc)
    P, par = build_problem(*g.solver_args()), build_params(g.params, qp_method='osqp')
    s = DGSQP(*g.solver_args(), print_method=None, qp_method='osqp')
    assert s.dims.layout == 2
    B = 8
    x0, u_tm = mc.sample_scenarios(g, B, seed=1 if name == 'merge6_N25' else 0)       # (the first scenarios of tools/ref_stats.py's samplers)
    u = s._to_agent_major(u_tm)
    l = np.array([oracle.dual_init(P, par, x0[b], u[b]) for b in range(B)])
    qp1 = s.qp_batch(x0, u, l)
    ok1 = qp1['flag'] == 0
    u2, l2 = np.where(ok1[:, None], u + qp1['du'], u), np.where(ok1[:, None], qp1['lhat'], l)
    qp2 = s.qp_batch(x0, u2, l2)

    def cpu(args):
        xb, ub, lb = args
        ev = oracle.evaluate(P, xb, ub, lb, 1)
        return oracle.osqp(oracle.nearest_pd(ev['Q'], par.reg, par.eig_floor), ev['q'], ev['G'], ev['g'])
    with ThreadPoolExecutor(8) as ex:
        cpu1 = list(ex.map(cpu, [(x0[b], u[b], l[b]) for b in range(B)]))
        cpu2 = list(ex.map(cpu, [(x0[b], u2[b], l2[b]) for b in range(B)]))
    same, polished, worst_x, worst_l = 0, 0, 0.0, 0.0
    for qp, cpus in ((qp1, cpu1), (qp2, cpu2)):
        for b in range(B):
            xo, lo, io = cpus[b]
            inf = qp['info'][b]
            if (int(inf[0]), int(inf[1]), int(inf[2]), int(inf[5])) != (io['status'], io['iters'], io['polished'], io['n_active']) or abs(inf[3] - io['rho']) > 1e-6 * io['rho']:
                print(name, 'QP', b, 'device', inf[:6], 'oracle', io)
                continue
            same += 1
            polished += io['polished'] == 1
            assert (qp['flag'][b] != 0) == (io['status'] in (-3, -4, -10, 3, 4))
            if io['status'] in (-3, -4, -10, 3, 4):
                continue
            worst_x = max(worst_x, np.abs(qp['du'][b] - xo).max() / max(1.0, np.abs(xo).max()))
            worst_l = max(worst_l, np.abs(qp['lhat'][b] - lo).max() / max(1.0, np.abs(lo).max()))
    print(f'{name}: OSQP on the device (XL layout) vs oracle/osqp.hpp: identical (status, iterations, polish verdict, rho, active rows) on {same}/16 QPs '
          f'({polished} polished); x within {worst_x:.1e}, lambda within {worst_l:.1e}')
    assert same >= 15
    assert worst_x < 1e-6 and worst_l < 1e-6


@pytest.mark.parametrize('name', sorted(XL_REF_GAMES))
def test_xl_osqp_solves_follow_the_numpy_loop_with_the_restated_osqp(name):
    """Full solves of configs[2], [3], [4] with qp_method='osqp' against the numpy loop + numpy OSQP (tests/golden/pyref_osqp_<game>.npz,
    64 scenarios): identical (status, iterations, QP solves) -- a solve that raises in the reference (status 4) is compared by status
    alone -- on >= 95 % of the scenarios the numpy loop itself reproduces under 1e-13 perturbations (`stable`, when the file has it)
    and >= 80 % of all; converged fractions within 5 points; iterates of the identical converged ones within 1e-5."""
    from dgsqp_amd import montecarlo as mc
    from dgsqp_amd.solver import DGSQP
    if not (GOLD / f'pyref_osqp_{name}.npz').exists():
        pytest.skip('fixture not generated yet (tools/ref_stats.py: the numpy loop with the numpy OSQP takes hours at this size)')
    ref = np.load(GOLD / f'pyref_osqp_{name}.npz')
    g = XL_REF_GAMES[name](mc)
    res = DGSQP(*g.solver_args(), print_method=None, qp_method='osqp').solve_batch(ref['x0'], ref['u_ws'])
    same = ((res['status'] == ref['status']) & (res['num_iters'] == ref['num_iters']) & (res['qp_solves'] == ref['qp_solves'])) | ((res['status'] == 4) & (ref['status'] == 4))
    stable = ref['stable'] if 'stable' in ref.files else np.ones(len(same), bool)
    cd, cr = res['status'] <= 1, ref['status'] <= 1
    idc = np.nonzero(same & cd & cr)[0]
    err = np.array([rel(res['u'][b], ref['u'][b]) for b in idc])
    print(f'{name}: qp_method osqp (XL layout) vs numpy loop + restated OSQP: identical {same.mean():.3f} of all, {same[stable].mean():.3f} of the {stable.sum()} numpy-stable scenarios; '
          f'converged {cd.mean():.3f} vs {cr.mean():.3f}; same flag {np.mean(cd == cr):.3f}; iterates of the identical converged: median {np.median(err) if len(err) else float("nan"):.1e}, '
          f'above 1e-5: {int((err > 1e-5).sum())} of {len(err)}')
    if name == 'kb_f1_N50':
        # The chaotic game: the numpy loop reproduces ITSELF under 1e-13 input perturbations on 9 of its 64 scenarios (14 %;
        # profiles/r05_pyref_osqp_stability.txt).  Path identity is demanded on those (one fork allowed); beyond them only the statistics:
        # converged fraction within 12 points (two binomial standard deviations at 64 scenarios), mean iterations of the converged within 40 %
        # (measured: 20.1 against 15.2 over 38 / 33 converged scenarios).  The dual START already differs on this game: at scipy's default
        # 1e-6 the restated LSQR (operator applied as G (G' v)) and scipy's (on the assembled G G') stop a Lanczos step apart on half of
        # the scenarios, l0 differs by up to 5e-3 -- which is also why identical paths end 6e-3 apart here and 1e-10 on the other games.
        assert stable.sum() >= 5 and same[stable].sum() >= stable.sum() - 1
        assert abs(cd.mean() - cr.mean()) <= 0.12
        assert abs(res['num_iters'][cd].mean() - ref['num_iters'][cr].mean()) <= 0.40 * ref['num_iters'][cr].mean()
        return
    assert same[stable].mean() >= 0.95 or ('stable' not in ref.files and same.mean() >= 0.80)
    assert same.mean() >= 0.80
    assert abs(cd.mean() - cr.mean()) <= 0.05 + 1.0 / len(cd) and np.mean(cd == cr) >= 0.90
    if len(err):
        assert np.mean(err > 1e-5) <= 0.05 and np.median(err) < 1e-6


def test_event_trace_parity_with_osqp(oracle, games):
    """The SQP state machine event by event (convergence measures, mu, merit values, watchdog / line-search trials) with OSQP's
    arithmetic on both sides: device (csrc/dgsqp_osqp.h) vs C++ oracle (oracle/osqp.hpp), converged LSQR dual start."""
    import copy
    from dgsqp_amd.montecarlo import sample_scenarios
    from dgsqp_amd.solver import DGSQP
    g, P, par0 = games['kb_chicane_N15']
    par = tight_lsqr(copy.copy(par0))
    par.qp_method = 1
    s = DGSQP(*g.solver_args(), print_method=None, lsqr_tol=1e-13, qp_method='osqp')
    B = 12
    x0, u_tm = sample_scenarios(g, B, seed=23)
    s.set_trace(6000)
    try:
        s.solve_batch(x0, u_tm)
        traces = s.fetch_trace(B)
    finally:
        s.set_trace(0)
    identical = 0
    for b in range(B):
        to = oracle.solve_trace(P, par, x0[b], agent_major(u_tm)[b])
        tg = traces[b]
        if len(to) == len(tg) and np.array_equal(to[:, 0], tg[:, 0]):
            big = np.abs(to[:, 1]) > 1e-6
            if np.all(np.abs(tg[big, 1] - to[big, 1]) <= 1e-5 * np.abs(to[big, 1])):
                identical += 1
    assert identical >= B - 1, identical


@pytest.mark.parametrize('kind', ['kb_chicane_N15', 'curve3_N25_xl'])
def test_osqp_counters_count_the_qp_calls_of_a_solve(games, kind):
    """dgsqp_osqp_counters: {QP calls, ADMM iterations} of the device since the last reset -- what bench.py divides to price the ADMM work of
    its timed region (both OSQP kernels).  Every QP call of a batch is counted (= the sum of the scenarios' qp_solves, plus the calls that
    ended a solve with qp_fail), the iterations per call lie between one check interval and the limit (25 ... 4,000), and a reset clears both."""
    import ctypes as C
    from dgsqp_amd import montecarlo as mc
    from dgsqp_amd.solver import DGSQP
    g = games['kb_chicane_N15'][0] if kind == 'kb_chicane_N15' else mc.kinematic_racing_game('curve', N=25, M=3)
    s = DGSQP(*g.solver_args(), print_method=None, qp_method='osqp')
    x0, u_tm = mc.sample_scenarios(g, 12 if kind == 'kb_chicane_N15' else 6, seed=23)
    cnt = (C.c_uint64 * 2)()
    assert s._lib.dgsqp_osqp_counters(s._h, None, 1) == 0 and s._lib.dgsqp_osqp_counters(s._h, cnt, 0) == 0 and (cnt[0], cnt[1]) == (0, 0)
    res = s.solve_batch(x0, u_tm)
    assert s._lib.dgsqp_osqp_counters(s._h, cnt, 1) == 0
    calls, its = int(cnt[0]), int(cnt[1])
    nq = int(res['qp_solves'].sum())
    assert nq <= calls <= nq + int((res['status'] == 4).sum()) and 25 * calls <= its <= 4000 * calls, (calls, its, nq)
    print(f'{kind}: {calls} OSQP calls, {its / calls:.0f} ADMM iterations per call')
    assert s._lib.dgsqp_osqp_counters(s._h, cnt, 0) == 0 and (cnt[0], cnt[1]) == (0, 0)


@pytest.mark.parametrize('kind', ['kb_chicane_N15', 'curve3_N25_xl'])
def test_osqp_rho_carry_matches_the_oracle(oracle, games, kind):
    """dgsqp_params_t.osqp_rho_carry (opt-in): an OSQP call starts from the rho the scenario's previous call ended with -- what OSQP does
    inside CasADi's persistent conic plugin (SURVEY.md parity hazard 7) -- on the device (both OSQP kernels) and in the C++ oracle: identical
    (status, iterations, QP solves) with a converged LSQR start, iterates of the converged ones to 1e-5; and the option is not a no-op."""
    import copy
    from dgsqp_amd import montecarlo as mc
    from dgsqp_amd.solver import DGSQP, build_problem, build_params
    g = games['kb_chicane_N15'][0] if kind == 'kb_chicane_N15' else mc.kinematic_racing_game('curve', N=25, M=3)
    B = 12 if kind == 'kb_chicane_N15' else 6
    P = build_problem(*g.solver_args())
    par = tight_lsqr(build_params(g.params, qp_method='osqp', osqp_rho_carry=True))
    assert par.osqp_rho_carry == 1
    x0, u_tm = mc.sample_scenarios(g, B, seed=23)
    s = DGSQP(*g.solver_args(), print_method=None, lsqr_tol=1e-13, qp_method='osqp', osqp_rho_carry=True)
    res = s.solve_batch(x0, u_tm)
    u = s._to_agent_major(u_tm)
    ref = oracle.solve_batch(P, par, x0, u, nthreads=min(B, 8))
    same = (res['status'] == ref['status']) & (res['num_iters'] == ref['num_iters']) & (res['qp_solves'] == ref['qp_solves'])
    base = DGSQP(*g.solver_args(), print_method=None, lsqr_tol=1e-13, qp_method='osqp').solve_batch(x0, u_tm)
    changed = (base['num_iters'] != res['num_iters']) | (base['qp_solves'] != res['qp_solves']) | (np.abs(base['u'] - res['u']).max(axis=1) > 1e-9)
    print(f'{kind}: rho carried between the OSQP calls of a solve: device = oracle on {same.sum()}/{B} scenarios; differs from the restart-at-0.1 run on {changed.sum()}')
    assert same.sum() >= B - 1
    for b in np.nonzero(same & (ref['status'] <= 1))[0]:
        assert rel(res['u'][b], ref['u'][b]) < 1e-5
    assert changed.any()


@pytest.mark.parametrize('name', ['curve3_N25', 'kb_f1_N50', 'merge6_N25'])
def test_xl_osqp_mixed_precision_against_fp64(oracle, name):
    """dgsqp_params_t.mixed_precision (opt-in; BASELINE configs[2..4] name fp32): the XL ADMM iteration with K^-1 stored in fp32 against the
    fp64 kernel and, through it, oracle/osqp.hpp.  QP level (8 scenarios, the QP at the dual start): the same verdict (solved /
    infeasible / limit) on every QP both runs settle (statuses 1, -3, -4), the ADMM iteration count within 25 % (or 50), and wherever both
    polishes succeed the same point to 1e-6 -- the polish is fp64 and lands on the KKT point of the same active set.  Solve level, the
    solvable three-car game (24 scenarios): converged counts within 3 scenarios of each other, and the solutions of scenarios converged in
    both within 1e-3 on at least 80 % of them.  F1 is chaotic -- a 24-scenario sample once gave 2 converged against 7 (profiles/r05_mixed_precision.txt),
    single paths do not agree and are not asserted --: the STATISTIC is held over 256 scenarios, converged fractions within 8 points
    (measured 44.9 % vs 47.3 %) and mean QP solves within 15 %.  The six-car merge runs at reg = 0, where the
    kernel keeps K^-1 in fp64 (csrc/dgsqp_osqp_xl.h, ox_build_k: the measured reason): the switch must leave it bit-identical."""
    from dgsqp_amd import montecarlo as mc
    from dgsqp_amd.solver import DGSQP, build_problem, build_params
    g = mc.kinematic_racing_game('curve', N=25, M=3) if name == 'curve3_N25' else XL_REF_GAMES[name](mc)
    P, par = build_problem(*g.solver_args()), build_params(g.params, qp_method='osqp')
    s64 = DGSQP(*g.solver_args(), print_method=None, qp_method='osqp')
    s32 = DGSQP(*g.solver_args(), print_method=None, qp_method='osqp', mixed_precision=True)
    assert s32.dims.layout == 2 and s32._cparams.mixed_precision == 1 and s64._cparams.mixed_precision == 0
    B = 8
    x0, u_tm = mc.sample_scenarios(g, B, seed=1 if name == 'merge6_N25' else 0)
    u = s64._to_agent_major(u_tm)
    l = np.array([oracle.dual_init(P, par, x0[b], u[b]) for b in range(B)])
    a, b_ = s64.qp_batch(x0, u, l), s32.qp_batch(x0, u, l)
    settled, polished, worst = 0, 0, 0.0
    for i in range(B):
        ia, ib = a['info'][i], b_['info'][i]
        if int(ia[0]) in (1, -3, -4) and int(ib[0]) in (1, -3, -4):
            settled += 1
            assert int(ia[0]) == int(ib[0]), (ia[:6], ib[:6])
            assert abs(ia[1] - ib[1]) <= max(50, 0.25 * ia[1]), (ia[:6], ib[:6])
        if int(ia[0]) == 1 and int(ib[0]) == 1 and int(ia[2]) == 1 and int(ib[2]) == 1 and int(ia[5]) == int(ib[5]):
            polished += 1
            worst = max(worst, np.abs(a['du'][i] - b_['du'][i]).max() / max(1.0, np.abs(a['du'][i]).max()))
    print(f'{name}: K^-1 in fp32 vs fp64, QP at the dual start: {settled}/{B} settled in both with the same verdict, {polished} polished in both, x within {worst:.1e}; '
          f'ADMM iterations fp64 {a["info"][:, 1].astype(int).tolist()} fp32 {b_["info"][:, 1].astype(int).tolist()}')
    assert worst < 1e-6
    if name == 'curve3_N25':
        assert settled >= B - 1 and polished >= B // 2
    B = {'merge6_N25': 8, 'kb_f1_N50': 256}.get(name, 24)          # (the merge only has to come out bit-identical: eight solves show it; F1: a statistic)
    x0, u_tm = mc.sample_scenarios(g, B, seed=5)
    r64, r32 = s64.solve_batch(x0, u_tm), s32.solve_batch(x0, u_tm)
    c64, c32 = r64['status'] == 0, r32['status'] == 0
    both = c64 & c32
    close = np.array([rel(r32['u'][i], r64['u'][i]) < 1e-3 for i in np.nonzero(both)[0]])
    print(f'{name}: full solves, K^-1 fp32 vs fp64: converged {c32.sum()} vs {c64.sum()} of {B}, mean QP solves {r32["qp_solves"].mean():.1f} vs {r64["qp_solves"].mean():.1f}; '
          f'same solution (1e-3) on {int(close.sum())}/{int(both.sum())} converged in both')
    if name == 'kb_f1_N50':
        assert abs(c32.mean() - c64.mean()) <= 0.08 and abs(r32['qp_solves'].mean() - r64['qp_solves'].mean()) <= 0.15 * r64['qp_solves'].mean()
    else:
        assert abs(int(c32.sum()) - int(c64.sum())) <= 3
    if name == 'curve3_N25':
        assert both.sum() >= B // 2 and close.mean() >= 0.8
    if name == 'merge6_N25':
        assert g.params.reg == 0
        for k in ('status', 'num_iters', 'qp_solves', 'u'):
            assert np.array_equal(r32[k], r64[k]), k
        assert np.array_equal(a['du'], b_['du'], equal_nan=True)


def test_xl_event_trace_parity_with_osqp(oracle):
    """The same on the XL layout (csrc/dgsqp_osqp_xl.h): the solvable three-car game of configs[2]'s size (n = 150, 825 rows), event
    by event against the C++ oracle with its OSQP, converged LSQR dual start; eight scenarios: the same SEQUENCE of events (iterations, QP
    solves, watchdog steps, line-search trials) on at least seven (measured: 8 of 8), the largest deviation of an event value below 1e-3 in the median
    (measured: 5e-5; up to 7e-2 on single events such as a mu computed from a violation sum near zero) -- the 1e-9
    differences of two polished QP answers grow along a solve of 20 iterations (the LDS-path games keep 1e-5), the decisions do not change."""
    from concurrent.futures import ThreadPoolExecutor
    from dgsqp_amd import montecarlo as mc
    from dgsqp_amd.solver import DGSQP, build_problem, build_params
    g = mc.kinematic_racing_game('curve', N=25, M=3)
    P, par = build_problem(*g.solver_args()), tight_lsqr(build_params(g.params, qp_method='osqp'))
    s = DGSQP(*g.solver_args(), print_method=None, lsqr_tol=1e-13, qp_method='osqp')
    assert s.dims.layout == 2
    B = 8
    x0, u_tm = mc.sample_scenarios(g, B, seed=23)
    u = s._to_agent_major(u_tm)
    s.set_trace(20000)
    try:
        res = s.solve_batch(x0, u_tm)
        traces = s.fetch_trace(B)
    finally:
        s.set_trace(0)
    with ThreadPoolExecutor(8) as ex:
        ref = list(ex.map(lambda b: oracle.solve_trace(P, par, x0[b], u[b], max_pairs=20000), range(B)))
    same_flow, tight, worst = 0, 0, []
    for b in range(B):
        to, tg = ref[b], traces[b]
        if len(to) != len(tg) or not np.array_equal(to[:, 0], tg[:, 0]):
            worst.append(None)
            continue
        same_flow += 1                       # the same sequence of events: iterations, QP solves, watchdog steps, line-search trials
        big = np.abs(to[:, 1]) > 1e-6
        dev = float(np.max(np.abs(tg[big, 1] - to[big, 1]) / np.abs(to[big, 1]))) if big.any() else 0.0
        worst.append(dev)
        tight += dev <= 1e-5
    print(f'XL layout, qp_method osqp: identical event sequences on {same_flow}/{B} scenarios, values within 1e-5 throughout on {tight}; largest relative '
          f'deviation of an event value per scenario: {[None if w is None else float(f"{w:.1e}") for w in worst]}; status {res["status"].tolist()}')
    assert same_flow >= B - 1
    assert np.median([w for w in worst if w is not None]) < 1e-3


@pytest.mark.parametrize('name', ['kb_chicane_N15', 'ablation_N15_ls_stat', 'ablation_N15_ls_stat_l1', 'ablation_N15_nms_stat'])
def test_event_trace_parity(oracle, games, solvers, name):
    """Event-by-event comparison of the SQP state machine (convergence measures, mu, merit values, every
    watchdog / line-search trial): same event codes in the same order, values within 1e-5 relative.  The ablation games run the
    other three combinations of nonmono_ls x merit_function (plain _line_search_3 instead of the watchdog; merit 'stat' without the
    l1 term and with mu = 0).  Without the watchdog a failing line search halves alpha down to 2^-50, where the Armijo test compares
    merits that agree to 13 digits -- whether it takes one more trial there is decided by rounding in the reference as well: a trace
    whose first difference is such a trial (alpha < 1e-9) is exempt (at most 3 of the 12), any other difference is not (at most 1)."""
    from dgsqp_amd.montecarlo import sample_scenarios
    g, P, par = games[name]
    s = solvers[name]
    B = 12
    x0, u_tm = sample_scenarios(g, B, seed=23)
    s.set_trace(6000)
    try:
        res = s.solve_batch(x0, u_tm)
        traces = s.fetch_trace(B)
    finally:
        s.set_trace(0)
    identical, deep = np.zeros(B, bool), np.zeros(B, bool)
    for b in range(B):
        to = oracle.solve_trace(P, tight_lsqr(par), x0[b], agent_major(u_tm)[b])
        tg = traces[b]
        m = min(len(to), len(tg))
        close = (to[:m, 0] == tg[:m, 0]) & ((np.abs(to[:m, 1]) <= 1e-6) | (np.abs(tg[:m, 1] - to[:m, 1]) <= 1e-5 * np.abs(to[:m, 1])))
        # mu = |dphi| / (0.5 sum(g - s)) on a violation of rounding size (p_feas < 1e-9: a quotient by 1e-13 noise, 1e8 and more) is noise on
        # both sides, and so is every merit value mu x sum(g - s) enters in that iteration (codes 11-13, 20-22, 31): there only the event
        # codes are compared -- the control flow they produce
        noisy, pf = np.zeros(m, bool), 1.0
        for k in range(m):
            if to[k, 0] == 2:
                pf = to[k, 1]
            elif to[k, 0] == 1:
                pf = 1.0
            noisy[k] = pf < 1e-9 and to[k, 0] in (11, 12, 13, 20, 21, 22, 31)
        close |= noisy & (to[:m, 0] == tg[:m, 0])
        if len(to) == len(tg) and close.all():
            identical[b] = True
            continue
        # the first differing event: a line search that is still halving at alpha < 1e-9 compares merits that agree to 13 digits -- whether
        # it takes one more trial is decided by rounding (in the reference as well); anything else is a real difference
        k = int(np.argmin(close)) if not close.all() else m
        print(f'  {name} scenario {b}: first differing event {k} of {len(to)} / {len(tg)}: oracle {to[k].tolist() if k < len(to) else None} device {tg[k].tolist() if k < len(tg) else None}')
        alphas = to[:k][to[:k, 0] == 30, 1]
        deep[b] = len(alphas) > 0 and alphas[-1] < 1e-9 and (k >= len(to) or to[k, 0] in (30, 31, 40, 22)) and (k >= len(tg) or tg[k, 0] in (30, 31, 40, 22))
    print(name, 'event traces identical', identical.sum(), 'of', B, '; first difference inside a line search below alpha = 1e-9:', deep.sum())
    assert (identical | deep).sum() >= B - 1 and identical.sum() >= B - 3, (identical, deep)


def _trace_compare(to, tg, vtol):
    """Two SQP event logs [(code, value)]: (index of the first event whose CODES differ, or min(len) when one log is a prefix of the
    other; mask over that many events: value within ``vtol`` relative).  Values below 1e-6 and the merit values of an iteration whose mu
    is a quotient by a rounding-size violation (p_feas < 1e-9) count as close: there only the codes carry information
    (test_event_trace_parity)."""
    m = min(len(to), len(tg))
    same = to[:m, 0] == tg[:m, 0]
    k = int(np.argmin(same)) if not same.all() else m
    close = (np.abs(to[:k, 1]) <= 1e-6) | (np.abs(tg[:k, 1] - to[:k, 1]) <= vtol * np.abs(to[:k, 1]))
    pf = 1.0
    for i in range(k):
        if to[i, 0] == 2:
            pf = to[i, 1]
        elif to[i, 0] == 1:
            pf = 1.0
        if pf < 1e-9 and to[i, 0] in (11, 12, 13, 20, 21, 22, 31):
            close[i] = True
    return k, close


@pytest.mark.parametrize('name', ['kb_f1_N50', 'kb_barc3_N25'])
def test_event_trace_prefix_parity_on_the_chaotic_configs(oracle, name):
    """BASELINE configs[3] (F1, N = 50, n = 200) and configs[2] (three cars on the BARC circuit, N = 25, n = 150) are chaotic / failing
    games: whole-solve identity is only defined on the scenarios the oracle reproduces under 1e-13 input perturbations (39 of 64 and 39 of
    48 in round 5), the others were compared by two loose statistics.  Here EVERY scenario is used up to the point where the oracle
    itself stops being reproducible: the C++ oracle's event log (DGSQP.py:368-398 convergence measures, :559-585 mu, merit values, every
    watchdog / line-search trial) is recorded from the nominal inputs and from two perturbed copies (1e-12, 1e-11 relative); the scenario's STABLE PREFIX
    ends at the first event where a perturbed log takes another decision (another event code).  Over that prefix the device's log must
    carry the same codes in the same order, and every value the perturbed logs themselves reproduce to 1e-7 must agree with the device's
    to 1e-5 (a value the oracle's own perturbation already moves by more than 1e-7 -- the growth that precedes a fork, or a dual start
    through an ill-conditioned G G' -- is compared by code alone).  At most 1 scenario in 16 may leave its prefix early (two re-runs do
    not find every fragile decision: conftest.stable_mask); a departure inside a line search that is still halving below alpha = 1e-9 is
    rounding in the reference as well and exempt, as in test_event_trace_parity (measured on the F1 game: 56 of 64 whole prefixes, 5 such
    departures, 3 others)."""
    from concurrent.futures import ThreadPoolExecutor
    import os
    from dgsqp_amd import montecarlo as mc
    from dgsqp_amd.solver import DGSQP, build_problem, build_params
    g, B = (mc.f1_racing_game(N=50), 64) if name == 'kb_f1_N50' else (mc.barc_racing_game(N=25, M=3), 48)
    P, par = build_problem(*g.solver_args()), build_params(g.params, lsqr_tol=1e-13)
    s = DGSQP(*g.solver_args(), print_method=None, lsqr_tol=1e-13)
    assert s.dims.layout == 2
    x0, u_tm = mc.sample_scenarios(g, B, seed=0)
    u = s._to_agent_major(u_tm)
    cap = 40000
    s.set_trace(cap)
    try:
        s.solve_batch(x0, u_tm)
        dev = s.fetch_trace(B)
    finally:
        s.set_trace(0)
    rng = np.random.default_rng(2024)
    # (perturbations of 1e-12 and 1e-11: these games sum over n = 150 / 200 terms and condition-1e10 factors; two correct implementations
    # differ by that much after the first QP, and the prefix has to end where THAT difference changes a decision)
    pert = [(x0 * (1 + e * rng.standard_normal(x0.shape)), u * (1 + e * rng.standard_normal(u.shape))) for e in (1e-12, 1e-11)]
    jobs = [(x0[b], u[b]) for b in range(B)] + [(px[b], pu[b]) for px, pu in pert for b in range(B)]
    with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 8)) as ex:          # (ctypes releases the GIL: the oracle runs on all host cores)
        logs = list(ex.map(lambda xu: oracle.solve_trace(P, par, xu[0], xu[1], max_pairs=cap), jobs))
    nominal, p1, p2 = logs[:B], logs[B:2 * B], logs[2 * B:]
    prefix, ok, whole, iters, checked, departures = np.zeros(B, int), np.zeros(B, bool), np.zeros(B, bool), np.zeros(B, int), 0, []
    deep = np.zeros(B, bool)
    for b in range(B):
        k1, c1 = _trace_compare(nominal[b], p1[b], 1e-7)
        k2, c2 = _trace_compare(nominal[b], p2[b], 1e-7)
        k = min(k1, k2)
        firm = c1[:k] & c2[:k]                                   # values the oracle reproduces under both perturbations
        prefix[b], whole[b] = k, k == len(nominal[b]) == len(p1[b]) == len(p2[b])
        iters[b] = int((nominal[b][:k, 0] == 1).sum())           # SQP iterations that start inside the prefix
        kd, cd = _trace_compare(nominal[b][:k], dev[b], 1e-5)
        bad = np.nonzero(firm[:kd] & ~cd)[0]
        ok[b] = kd == k and len(bad) == 0
        checked += int(firm[:kd].sum())
        if not ok[b]:
            e = int(bad[0]) if len(bad) else kd
            departures.append((b, e, k))
            ev = lambda t: t[e].tolist() if e < len(t) else None
            # a line search still halving below alpha = 1e-9 compares merits that agree to 13 digits: whether it takes one more trial is
            # decided by rounding, in the reference as well (test_event_trace_parity's exemption)
            alphas = nominal[b][:e][nominal[b][:e, 0] == 30, 1]
            deep[b] = (not len(bad) and len(alphas) > 0 and alphas[-1] < 1e-9 and (e >= len(nominal[b]) or nominal[b][e, 0] in (30, 31, 40, 22))
                       and (e >= len(dev[b]) or dev[b][e, 0] in (30, 31, 40, 22)))
            print(f'  {name} scenario {b}: device leaves at event {e} of a stable prefix of {k} ({"value" if len(bad) else "code"}): oracle {ev(nominal[b])} perturbed {ev(p1[b])} {ev(p2[b])} device {ev(dev[b])}; '
                  f'SQP iteration {int((nominal[b][:e, 0] == 1).sum())} of {int((nominal[b][:k, 0] == 1).sum())} inside the prefix')
    print(f'{name}: stable prefixes of {B} oracle event logs: {int(prefix.sum())} of {sum(len(t) for t in nominal)} events (median {int(np.median(prefix))}, min {int(prefix.min())}; '
          f'{int(whole.sum())} logs stable to their end), {int(iters.sum())} SQP iterations inside them (median {int(np.median(iters))} per scenario); '
          f'device: same events over the whole prefix and every firm value within 1e-5 on {int(ok.sum())}/{B} scenarios ({checked} values compared); '
          f'early departures (scenario, event, prefix): {departures}, of which inside a line search below alpha = 1e-9: {int(deep.sum())}')
    assert prefix.min() >= 3 and np.median(iters) >= 2                   # every prefix holds at least the first convergence test; typically several iterations
    assert (ok | deep).sum() >= B - max(2, B // 16) and ok.sum() >= B - B // 5, departures      # (measured: 61 of 64 and 47 of 48; one scenario of margin)
    assert (~(ok | deep)[whole]).sum() <= 1                              # where the oracle is stable to the end, so is the device (whole-solve identity)


@pytest.mark.parametrize('name', ['kb_f1_N50'])
def test_event_trace_prefix_parity_against_the_numpy_loop(name):
    """The same prefix comparison against the yardstick that shares no code with the C++ oracle: the line-by-line numpy restatement of the
    reference loop with the numpy OSQP (oracle/pyref.py + osqp_restate.py, scipy's lsqr and numpy's eigh at their defaults).  Its event logs
    on the first 64 scenarios of BASELINE configs[3] -- nominal and two perturbed runs, 3.5 CPU-hours -- are committed with each scenario's
    stable prefix (tools/ref_trace.py -> tests/golden/pyref_osqp_trace_kb_f1_N50.npz); the device runs qp_method = 'osqp' with every default
    (nothing overridden: the reference's own setting) and must produce the same events over the prefix, every value the numpy loop itself
    reproduces to 1e-7 within 1e-5.  OSQP's ADMM stops at multiples of 25 iterations and its polish is accepted on a comparison of
    residuals, so a QP answer can differ by 1e-3 between two correct implementations (the numpy loop reproduces ITSELF
    on 9 of these 64 solves, and its stable prefixes hold 269 SQP iterations in all); the floor is the measured count (50 of 64) minus four."""
    from dgsqp_amd import montecarlo as mc
    from dgsqp_amd.solver import DGSQP
    fx = np.load(GOLD / f'pyref_osqp_trace_{name}.npz')
    g = mc.f1_racing_game(N=50)
    B = len(fx['prefix'])
    s = DGSQP(*g.solver_args(), print_method=None, qp_method='osqp')
    s.set_trace(40000)
    try:
        s.solve_batch(fx['x0'], fx['u'])
        dev = s.fetch_trace(B)
    finally:
        s.set_trace(0)
    ok, deep, checked, events, departures = np.zeros(B, bool), np.zeros(B, bool), 0, 0, []
    for b in range(B):
        to = fx['trace'][fx['off'][b]:fx['off'][b + 1]]
        k = int(fx['prefix'][b])
        firm = fx['firm'][fx['off'][b]:fx['off'][b] + k].astype(bool)
        kd, cd = _trace_compare(to[:k], dev[b], 1e-5)
        bad = np.nonzero(firm[:kd] & ~cd)[0]
        ok[b] = kd == k and len(bad) == 0
        checked += int(firm[:kd].sum())
        events += kd if not len(bad) else int(bad[0])
        if not ok[b]:
            e = int(bad[0]) if len(bad) else kd
            alphas = to[:e][to[:e, 0] == 30, 1]
            deep[b] = (not len(bad) and len(alphas) > 0 and alphas[-1] < 1e-9 and (e >= len(to) or to[e, 0] in (30, 31, 40, 22)) and (e >= len(dev[b]) or dev[b][e, 0] in (30, 31, 40, 22)))
            departures.append((b, e, k, 'value' if len(bad) else 'code'))
    its = sum(int((fx['trace'][fx['off'][b]:fx['off'][b] + fx['prefix'][b], 0] == 1).sum()) for b in range(B))
    print(f'{name} against the numpy loop + numpy OSQP: stable prefixes hold {int(fx["prefix"].sum())} events / {its} SQP iterations ({int(fx["whole"].sum())} logs stable to their end); '
          f'device qp_method osqp, all defaults: the whole prefix on {int(ok.sum())}/{B} scenarios (+ {int(deep.sum())} leaving inside a line search below alpha = 1e-9), '
          f'{events} events followed, {checked} values compared; departures (scenario, event, prefix, kind): {departures}')
    assert (ok | deep).sum() >= 46, departures            # measured: 50 of 64 (profiles/r06_gpu_tests_parity_lines.txt); most early departures sit at the first QP's verdict


def test_full_size_properties(games):
    """BASELINE config sizes (2-agent N=25, B=1024): size-independent properties.
    * conv_abs_tol  =>  optimality measures recomputed from (u, l) by an independent kernel are below tolerance
    * multipliers non-negative, iteration counts within [0, sqp_iters]
    * bitwise determinism of a repeated run and invariance to the order of the scenarios in the batch."""
    from dgsqp_amd.montecarlo import sample_scenarios
    from dgsqp_amd.solver import DGSQP
    g, P, par = games['kb_chicane_N25']
    s = DGSQP(*g.solver_args(), print_method=None)
    B = 1024
    x0, u_tm = sample_scenarios(g, B, seed=1)
    res = s.solve_batch(x0, u_tm)
    st = res['status']
    assert set(np.unique(st)).issubset({0, 1, 2, 3, 4}) and (res['num_iters'] <= 50).all() and (res['num_iters'] >= 0).all()
    assert (st <= 1).mean() > 0.5
    ok = np.where(st == 0)[0]
    ev = s.evaluate_batch(x0[ok[:64]], res['u'][ok[:64]], res['l'][ok[:64]])
    for i, b in enumerate(ok[:64]):
        assert max(0.0, ev['g'][i].max()) < par.p_tol
        assert np.abs(ev['g'][i] * res['l'][b]).max() < par.d_tol
        assert np.abs(ev['q'][i] + ev['G'][i].T @ res['l'][b]).max() < par.d_tol
        assert (res['l'][b] >= 0).all()
        np.testing.assert_allclose(ev['x'][i], res['x'][b], rtol=0, atol=1e-12)
    res2 = s.solve_batch(x0, u_tm)
    for k in ('u', 'l', 'status', 'num_iters', 'qp_solves', 'cond', 'cost'):
        assert np.array_equal(res[k], res2[k], equal_nan=True), k
    perm = np.random.default_rng(0).permutation(B)
    res3 = s.solve_batch(x0[perm], u_tm[perm])
    assert np.array_equal(res3['status'], st[perm]) and np.array_equal(res3['u'], res['u'][perm], equal_nan=True)


def test_edge_cases_and_reference_surface(games, oracle):
    """Empty and ragged batches, wrong shapes, the single-scenario reference surface (solve / step / set_warm_start)."""
    from dgsqp_amd.montecarlo import sample_scenarios
    from dgsqp_amd.solver import DGSQP
    from dgsqp_amd.types import VehicleState
    g, P, par = games['kb_curve_N10']
    s = DGSQP(*g.solver_args(), print_method=None)
    x0, u_tm = sample_scenarios(g, 3, seed=6)
    empty = s.solve_batch(x0[:0], u_tm[:0])
    assert empty['u'].shape == (0, s.n) and empty['status'].shape == (0,)
    one = s.solve_batch(x0[:1], u_tm[:1])
    three = s.solve_batch(x0, u_tm)                                  # B=3 < grid: ragged last wave of work
    assert np.array_equal(one['u'][0], three['u'][0])
    with pytest.raises(RuntimeError):
        s.set_warm_start(np.zeros((s.N + 1, s.n_u)))                  # DGSQP.py:272-273
    with pytest.raises(RuntimeError):
        s.solve_batch(x0, u_tm[:, :-1])
    states = [VehicleState(t=0.0), VehicleState(t=0.0)]
    g.joint_model.qu2state(states, x0[0], None)
    s.set_warm_start(u_tm[0])
    info = s.solve(states)
    assert set(info) == {'time', 'num_iters', 'status', 'cost', 'cond', 'iter_data', 'msg', 'init'}     # DGSQP.py:495-502
    assert info['msg'] in ('conv_abs_tol', 'conv_rel_tol', 'max_it', 'diverged', 'qp_fail')
    assert info['num_iters'] == int(three['num_iters'][0]) and s.q_pred.shape == (s.N + 1, s.n_q) and s.u_pred.shape == (s.N, s.n_u)
    np.testing.assert_allclose(s.u_pred, three['u_pred'][0])
    info2 = s.step(states)
    assert states[0].u.u_a == pytest.approx(s.u_pred[0, 0]) and len(s.get_prediction()) == 2
    assert s.get_prediction()[0].x is not None and info2['msg'] in ('conv_abs_tol', 'conv_rel_tol', 'max_it', 'diverged', 'qp_fail')


@pytest.mark.parametrize('kind,method', [('kin', 'euler'), ('kin', 'rk4'), ('dyn', 'euler'), ('dyn', 'rk4')])
def test_one_stage_game_hessian_from_sympy_tensors(kind, method):
    """SURVEY.md section 8c, KAT (1) on the DEVICE: Q of dgsqp_evaluate_batch on a one-stage two-car race against the closed
    formula in sympy's exact dynamics tensors (tools/make_sympy_kats.py -> tests/golden/sympy_fd_*.npz; conftest.sympy_one_stage_Q),
    1e-11 relative; the rollout's x_1 against sympy's f_d to 1e-13.  No oracle in this test: device vs exact symbolic derivatives."""
    from conftest import sympy_one_stage_game, sympy_one_stage_Q
    from dgsqp_amd.solver import DGSQP
    if method == 'rk4' and 'rk4_M' not in np.load(GOLD / f'sympy_fd_{kind}.npz').files:
        pytest.skip('tests/golden/sympy_fd_dyn.npz holds the euler step only (tools/make_sympy_kats.py dyn: rk4 takes about an hour of sympy)')
    g, kat = sympy_one_stage_game(kind, method)
    s = DGSQP(*g.solver_args(), print_method=None)
    nqa = g.joint_model.dynamics_models[0].n_q
    s_idx = g.joint_model.dynamics_models[0].s_idx
    pts = kat['points']
    pairs = ((0, 1), (2, 3), (3, 0), (1, 2))
    x0 = np.array([np.concatenate([pts[k1][:nqa], pts[k2][:nqa]]) for k1, k2 in pairs])
    u = np.array([np.concatenate([pts[k1][nqa:], pts[k2][nqa:]]) for k1, k2 in pairs])
    obs = 16                                              # rows: 2 x (4 rate + 4 input box) at k = 0, then the obstacle row of k = 1
    assert s.n_c_total == 21
    l = np.zeros((len(pairs), s.n_c_total))
    l[:, obs] = 0.7
    ev = s.evaluate_batch(x0, u, l)
    for b, (k1, k2) in enumerate(pairs):
        want = sympy_one_stage_Q(kat, method, k1, k2, 0.7, nqa, s_idx)
        assert np.abs(ev['Q'][b] - want).max() < 1e-11 * np.abs(want).max(), (kind, method, b)
        x1 = np.concatenate([kat[f'{method}_fd'][k1], kat[f'{method}_fd'][k2]])
        np.testing.assert_allclose(ev['x'][b].reshape(2, -1)[1], x1, rtol=1e-13, atol=1e-14)


def _plant_rk4(model, q, u, substeps=10):
    """The plant of the receding-horizon tests: fixed-step rk4 of the model's own f_c over dt (the same function advances the
    device loop and the oracle loop; the reference's simulator integrates the same f_c adaptively, dynamics_models.py:161-186)."""
    q = np.asarray(q, float)
    h = model.dt / substeps
    for _ in range(substeps):
        k1 = model.fc(q, u); k2 = model.fc(q + h / 2 * k1, u); k3 = model.fc(q + h / 2 * k2, u); k4 = model.fc(q + h * k3, u)
        q = q + h / 6 * (k1 + 2 * k2 + 2 * k3 + k4)
    return q


@pytest.mark.parametrize('name', ['kb_curve_N10', 'dyn_curve_N15'])
def test_step_receding_horizon_matches_the_oracle_loop(oracle, games, name):
    """DGSQP.step (DGSQP.py:283-297), five receding-horizon steps per scenario: solve -> apply u_pred[0] -> plant -> shifted warm
    start u_ws <- [u_pred[1:]; u_pred[-1]] (skipped on 'diverged' / 'qp_fail').  The device's step() is compared step by step with
    a loop written here around the ORACLE's solve: identical (msg, iterations) at every step, applied inputs and shifted warm
    starts within 1e-5 relative (north_star's bound on iterates), the applied input written into the VehicleStates
    (qu2state(states, None, u_pred[0])), and every VehiclePrediction field against the formulas of dynamics_models.py:1127-1150
    (kinematic: psidot = v L_r sin(atan(tan(delta) L_f / (L_f + L_r))) with the last entry repeated, v_tran = psidot L_r) and
    :2596-2615 (dynamic: the eight state columns)."""
    from dgsqp_amd.montecarlo import sample_scenarios
    from dgsqp_amd.solver import DGSQP
    from dgsqp_amd._ffi import STATUS_MSG as MSG
    from dgsqp_amd.types import VehicleState
    g, P, par = games[name]
    par = tight_lsqr(par)
    s = DGSQP(*g.solver_args(), print_method=None, lsqr_tol=1e-13)
    models = g.joint_model.dynamics_models
    x0s, u_tms = sample_scenarios(g, 3, seed=31)
    nq = [m.n_q for m in models]
    for b in range(3):
        # --- device loop through the reference surface
        states = [VehicleState(t=0.0) for _ in models]
        g.joint_model.qu2state(states, x0s[b], None)
        s.set_warm_start(u_tms[b])
        # --- oracle loop
        x_o, uws_o = x0s[b].copy(), agent_major(u_tms[b][None])[0]
        for k in range(5):
            x_d = g.joint_model.state2q(states)
            info = s.step(states)
            ref = oracle.solve_batch(P, par, x_o[None], uws_o[None], nthreads=1)
            msg_o = MSG[int(ref['status'][0])]
            assert (info['msg'], info['num_iters']) == (msg_o, int(ref['num_iters'][0])), (name, b, k, info['msg'], info['num_iters'], msg_o, int(ref['num_iters'][0]))
            u_am = ref['u'][0]
            u_tm_o = np.concatenate([u_am[a * s.N * 2:(a + 1) * s.N * 2].reshape(s.N, 2) for a in range(s.M)], axis=1)
            assert rel(s.u_pred, u_tm_o) < 1e-5, (name, b, k)
            if msg_o not in ('diverged', 'qp_fail'):
                uws_o = agent_major(np.vstack((u_tm_o[1:], u_tm_o[-1]))[None])[0]
            assert rel(s.u_ws, uws_o) < 1e-5, (name, b, k)
            if info['msg'] not in ('diverged', 'qp_fail'):      # the shift itself, exactly (DGSQP.py:293-295)
                assert np.array_equal(s.u_ws, agent_major(np.vstack((s.u_pred[1:], s.u_pred[-1]))[None])[0])
            assert np.array_equal(s.u_prev, s.u_pred[0])
            # applied input in the states, prediction messages
            preds = s.get_prediction()
            qi = 0
            for a, (m, st, pr) in enumerate(zip(models, states, preds)):
                assert (st.u.u_a, st.u.u_steer) == (s.u_pred[0, 2 * a], s.u_pred[0, 2 * a + 1])
                q, u = s.q_pred[:, qi:qi + m.n_q], s.u_pred[:, 2 * a:2 * a + 2]
                assert pr.t == states[0].t
                assert np.array_equal(np.array(pr.u_a), u[:, 0]) and np.array_equal(np.array(pr.u_steer), u[:, 1])
                if m.n_q == 6:
                    for fld, col in (('x', 0), ('y', 1), ('v_long', 2), ('e_psi', 3), ('s', 4), ('x_tran', 5)):
                        assert np.array_equal(np.array(getattr(pr, fld)), q[:, col]), fld
                    psidot = q[:-1, 2] * m.L_r * np.sin(np.arctan(np.tan(u[:, 1]) * m.L_f / (m.L_f + m.L_r)))
                    psidot = np.append(psidot, psidot[-1])
                    np.testing.assert_allclose(np.array(pr.psidot), psidot, rtol=0, atol=1e-15)
                    np.testing.assert_allclose(np.array(pr.v_tran), psidot * m.L_r, rtol=0, atol=1e-15)
                else:
                    for fld, col in (('x', 0), ('y', 1), ('v_long', 2), ('v_tran', 3), ('psidot', 4), ('e_psi', 5), ('s', 6), ('x_tran', 7)):
                        assert np.array_equal(np.array(getattr(pr, fld)), q[:, col]), fld
                assert len(pr.x) == s.N + 1 and len(pr.u_a) == s.N
                qi += m.n_q
            # q_pred[0] is the state the solve started from; q_pred follows the rollout of u_pred (oracle's x)
            assert np.array_equal(s.q_pred[0], x_d)
            assert rel(s.q_pred, ref['x'][0].reshape(s.N + 1, s.n_q)) < 1e-5
            # --- the plant, same function for both loops; each loop applies ITS OWN first input
            qi = 0
            x_next_d, x_next_o = [], []
            for a, m in enumerate(models):
                x_next_d.append(_plant_rk4(m, x_d[qi:qi + m.n_q], s.u_pred[0, 2 * a:2 * a + 2]))
                x_next_o.append(_plant_rk4(m, x_o[qi:qi + m.n_q], u_tm_o[0, 2 * a:2 * a + 2]))
                qi += m.n_q
            x_o = np.concatenate(x_next_o)
            g.joint_model.qu2state(states, np.concatenate(x_next_d), None)
            for st in states:
                st.t += g.joint_model.dt
            assert rel(g.joint_model.state2q(states), x_o) < 1e-5


def test_step_keeps_the_warm_start_after_a_failed_qp(oracle, games):
    """DGSQP.py:293: after 'qp_fail' (or 'diverged') step() does NOT shift the warm start.  Car 2 placed 1 m outside the track
    boundary: the linearised bound rows cannot be met within the input box -> the first QP is infeasible, on the oracle alike."""
    from dgsqp_amd.montecarlo import sample_scenarios
    from dgsqp_amd.solver import DGSQP
    from dgsqp_amd._ffi import STATUS_MSG as MSG
    from dgsqp_amd.types import VehicleState
    g, P, par = games['kb_curve_N10']
    s = DGSQP(*g.solver_args(), print_method=None, lsqr_tol=1e-13)
    x0, u_tm = sample_scenarios(g, 1, seed=31)
    x0 = x0[0].copy()
    x0[11] = g.half_width + 1.0                       # e_y of car 2
    states = [VehicleState(t=0.0), VehicleState(t=0.0)]
    g.joint_model.qu2state(states, x0, None)
    s.set_warm_start(u_tm[0])
    before = s.u_ws.copy()
    info = s.step(states)
    ref = oracle.solve_batch(P, tight_lsqr(par), x0[None], agent_major(u_tm), nthreads=1)
    assert info['msg'] == MSG[int(ref['status'][0])] == 'qp_fail' and info['status'] is False
    assert np.array_equal(s.u_ws, before)
    assert (states[0].u.u_a, states[1].u.u_steer) == (s.u_pred[0, 0], s.u_pred[0, 3])      # the input is still written (DGSQP.py:286)


def test_statistical_parity_at_default_lsqr_tolerance(oracle, games):
    """With scipy's default LSQR tolerance (the reference's setting) the dual start of two correct implementations
    differs by ~1e-4, so individual paths may fork; the Monte-Carlo statistics the reference reports
    (scripts/process_data_curve.py:99-110: converged fraction, mean iterations over converged samples) must agree."""
    from dgsqp_amd.montecarlo import sample_scenarios
    from dgsqp_amd.solver import DGSQP
    g, P, par = games['kb_chicane_N15']
    s = DGSQP(*g.solver_args(), print_method=None)
    B = 64
    x0, u_tm = sample_scenarios(g, B, seed=31)
    res = s.solve_batch(x0, u_tm)
    ref = oracle.solve_batch(P, par, x0, agent_major(u_tm), nthreads=8)
    conv_g, conv_r = res['status'] <= 1, ref['status'] <= 1
    assert abs(conv_g.mean() - conv_r.mean()) <= 0.08
    # at this LSQR tolerance the oracle's own control flow moves with a 1e-6 change of the tolerance: the mask is taken from that
    par2 = tight_lsqr(par); par2.lsqr_atol = par2.lsqr_btol = 1.05e-6
    ref2 = oracle.solve_batch(P, par2, x0, agent_major(u_tm), nthreads=8)
    stable = (ref2['status'] == ref['status']) & (ref2['num_iters'] == ref['num_iters']) & (ref2['qp_solves'] == ref['qp_solves'])
    stable &= stable_mask(oracle, P, par, x0, agent_major(u_tm), ref, K=2)
    assert_control_flow_parity(res, ref, stable, 'kb_chicane_N15 at scipy LSQR tolerance', min_stable_same=0.95, max_conv_gap=0.05)
    both = conv_g & conv_r & (res['status'] == 0) & (ref['status'] == 0)
    assert abs(res['num_iters'][both].mean() - ref['num_iters'][both].mean()) <= 0.1 * ref['num_iters'][both].mean()
    for b in np.where(both)[0]:          # converged to the same equilibrium
        assert rel(res['u'][b], ref['u'][b]) < 5e-3, b


def test_certified_nearest_pd_shortcut_against_the_oracle(oracle, games):
    """The _nearestPD shortcut of the LDS path (round 4): when a workgroup's previous call found no negative eigenvalue it first sweeps
    M = B + reg I with checked pivots and certifies it by ||M^-1||_inf < 1 / reg -- the sweep's result IS P.  That branch only runs when
    the projected Hessian is not asked for, so it is compared here directly: one launch of 700 QPs (every workgroup solves two or three in
    a row, the test hook does not clear the flag between them) with and without the Hessian output -- du, lhat identical to 1e-9 --, and
    the first 12 against the oracle's _nearestPD + QP (DGSQP.py:1290-1296, 232-266)."""
    from dgsqp_amd.montecarlo import sample_scenarios
    from dgsqp_amd.solver import DGSQP
    g, P, par = games['dyn_curve_N25']
    s = DGSQP(*g.solver_args(), print_method=None)
    B = 700
    x0, u_tm = sample_scenarios(g, B, seed=23)
    u = agent_major(u_tm)
    l = np.zeros((B, s.n_c_total))
    full = s.qp_batch(x0, u, l)                      # with Qpd: the full path (tridiagonalisation) or the inertia test
    fast = s.qp_batch(x0, u, l, want_Qpd=False)      # without: the certified sweep where the previous QP of the workgroup had a definite Hessian
    assert np.array_equal(full['flag'], fast['flag'])
    ok = full['flag'] == 0
    assert ok.mean() > 0.9
    scale = np.maximum(1.0, np.abs(full['du'][ok]).max(axis=1))
    assert (np.abs(full['du'][ok] - fast['du'][ok]).max(axis=1) / scale).max() < 1e-9
    assert np.abs(full['lhat'][ok] - fast['lhat'][ok]).max() < 1e-7 * max(1.0, np.abs(full['lhat'][ok]).max())
    definite = 0
    for b in range(12):
        ev = oracle.evaluate(P, x0[b], u[b], l[b], 1)
        definite += np.linalg.eigvalsh(0.5 * (ev['Q'] + ev['Q'].T)).min() > 0
        du, lam, flag = oracle.qp(oracle.nearest_pd(ev['Q'], par.reg), ev['q'], ev['G'], ev['g'])
        assert flag == fast['flag'][b]
        if flag == 0:
            assert rel(fast['du'][b], du) < 1e-8 and np.abs(fast['lhat'][b] - lam).max() < 1e-6 * max(1.0, np.abs(lam).max())
    print(f'certified _nearestPD shortcut: 700 QPs with / without the Hessian output agree; {definite} of the first 12 Hessians are positive definite')


def test_qp_warm_start_from_unrelated_active_set(oracle, games, solvers):
    """Warm start of the dual active-set method: with more scenarios than workgroups the QP test hook starts every
    later QP from the final active set of an UNRELATED scenario; the minimiser must not depend on the guess."""
    from dgsqp_amd.montecarlo import sample_scenarios
    g, P, par = games['kb_curve_N10']
    s = solvers['kb_curve_N10']
    B = 1200                                   # > 256 CUs x resident workgroups
    x0, u_tm = sample_scenarios(g, B, seed=31)
    u = agent_major(u_tm)
    rng = np.random.default_rng(5)
    l = np.maximum(0.0, rng.normal(0.0, 0.3, size=(B, s.dims.n_c)))
    qp = s.qp_batch(x0, u, l)
    checked = 0
    for b in list(range(B - 40, B)) + list(range(0, 8)):          # the tail is certainly warm-started
        o = oracle.evaluate(P, x0[b], u[b], l[b], 1)
        Qpd = oracle.nearest_pd(o['Q'], par.reg)
        du, lam, flag = oracle.qp(Qpd, o['q'], o['G'], o['g'])
        assert qp['flag'][b] == flag
        if flag != 0:
            continue
        cond_scale = max(1.0, np.abs(o['Q']).max() / max(1e-300, np.abs(Qpd).max()))
        assert rel(qp['du'][b], du) < 1e-8 * cond_scale and rel(qp['lhat'][b], lam) < 1e-6 * cond_scale
        assert np.array_equal(qp['lhat'][b] > 0, lam > 0)
        checked += 1
    assert checked >= 30


def test_solve_same_with_and_without_qp_warm_start(oracle, games):
    """qp_warm_start only shortens the active-set path: flags, iteration and QP counts and iterates agree with the
    cold-started solver on the well-conditioned scenarios."""
    from dgsqp_amd.montecarlo import sample_scenarios
    from dgsqp_amd.solver import DGSQP
    g, P, par = games['kb_chicane_N15']
    x0, u_tm = sample_scenarios(g, 48, seed=77)
    res = {}
    for ws in (0, 1):
        res[ws] = DGSQP(*g.solver_args(), print_method=None, lsqr_tol=1e-13, qp_warm_start=bool(ws)).solve_batch(x0, u_tm)
    a, b = res[0], res[1]
    ref = oracle.solve_batch(P, tight_lsqr(par), x0, agent_major(u_tm), nthreads=8)
    stable = stable_mask(oracle, P, tight_lsqr(par), x0, agent_major(u_tm), ref)
    same = assert_control_flow_parity(b, a, stable, 'warm vs cold QP start')
    easy = same & (a['num_iters'] < 30) & (a['status'] <= 1)
    assert easy.sum() >= 20
    for i in np.nonzero(easy)[0]:
        assert rel(a['u'][i], b['u'][i]) < 1e-6 and rel(a['l'][i], b['l'][i]) < 1e-5


@pytest.mark.parametrize('M,N', [(1, 10), (3, 10), (4, 8)])
def test_one_three_and_four_agents(oracle, M, N):
    """More than two agents (scripts/DGSQP_monte_carlo_agents.py): evaluation incl. the game Hessian, the QP and whole
    solves against the oracle; exercises the M-agent instantiations of the second-order adjoint rows."""
    from dgsqp_amd.montecarlo import kinematic_racing_game, sample_scenarios
    from dgsqp_amd.solver import DGSQP, build_problem, build_params
    import dgsqp_amd.solver as sv
    g = kinematic_racing_game('curve', N=N, M=M)
    P, par = build_problem(*g.solver_args()), tight_lsqr(build_params(g.params))
    s = DGSQP(*g.solver_args(), print_method=None, lsqr_tol=1e-13)
    B = 8
    x0, u_tm = sample_scenarios(g, B, seed=4)
    u = np.ascontiguousarray(u_tm.reshape(B, N, M, 2).transpose(0, 2, 1, 3).reshape(B, -1))
    rng = np.random.default_rng(1)
    up = u + 0.01 * rng.standard_normal(u.shape)
    l = np.maximum(0, rng.standard_normal((B, s.n_c_total)))
    ev = s.evaluate_batch(x0, up, l)
    for b in range(B):
        o = oracle.evaluate(P, x0[b], up[b], l[b], 1)
        for key in ('x', 'q', 'g', 'G', 'Q'):
            assert rel(ev[key][b], o[key]) < 1e-12, (key, b)
    res = s.solve_batch(x0, u_tm)
    ref = oracle.solve_batch(P, par, x0, u, nthreads=4)
    assert np.array_equal(res['status'], ref['status']) and np.array_equal(res['num_iters'], ref['num_iters'])
    for b in range(B):
        assert rel(res['u'][b], ref['u'][b]) < 1e-6 and rel(res['l'][b], ref['l'][b]) < 1e-5


@pytest.mark.parametrize('name', ['kb_chicane_N15', 'dyn_curve_N15'])
def test_pid_warm_start_on_device(games, solvers, name):
    """Row (f): the PID lane-follower warm start and the collision check of the Monte-Carlo scripts (chicane.py:411-447,
    :38-43) on the device against the numpy mirror: inputs, trajectories, collision flags, and the sampler built on it."""
    from dgsqp_amd.montecarlo import pid_warm_start, sample_scenarios
    g, P, par = games[name]
    s = solvers[name]
    rng = np.random.default_rng(3)
    B = 300
    models = g.joint_model.dynamics_models
    q0 = []
    for m in models:                                   # same placement rules as the sampler, no rejection
        q = np.zeros((B, m.n_q))
        sv, ey = 0.1 + rng.random(B) * 2.0, rng.random(B) * 1.6 - 0.8
        xy = np.array([g.track.local_to_global((a, b, 0.0))[:2] for a, b in zip(sv, ey)])
        q[:, 0], q[:, 1], q[:, 2] = xy[:, 0], xy[:, 1], rng.random(B) + 2
        q[:, m.s_idx], q[:, m.ey_idx] = sv, ey
        q0.append(q)
    du = tuple(g.agent_constraints[0].rate_max)
    ref = [pid_warm_start(m, q, g.params.N, g.params.dt, du=du) for m, q in zip(models, q0)]
    dev = s.pid_warm_start_batch(np.concatenate(q0, axis=1), du_max=du, want_trajectories=True)
    u_ref = np.concatenate([r[1] for r in ref], axis=2)
    q_ref = np.concatenate([r[0] for r in ref], axis=2)
    assert np.abs(dev['u_ws'] - u_ref).max() < 1e-10 and np.abs(dev['q_ws'] - q_ref).max() < 1e-10
    dist = np.linalg.norm(ref[0][0][:, :, :2] - ref[1][0][:, :, :2], axis=2)
    margin = np.abs(dist - g.obs_d).min(axis=1) > 1e-9          # flags may differ only on exact ties
    assert np.array_equal(dev['collide'][margin], (dist < g.obs_d).any(axis=1)[margin]) and dev['collide'].any() and not dev['collide'].all()
    x0_h, u_h = sample_scenarios(g, 64, seed=9)
    x0_d, u_d = sample_scenarios(g, 64, seed=9, solver=s)
    assert np.array_equal(x0_h, x0_d) and np.abs(u_h - u_d).max() < 1e-10


@pytest.mark.parametrize('tag,name', [('kb', 'kb_chicane_N25'), ('dyn', 'dyn_curve_N25')])
def test_pid_kernel_against_vectors_of_the_reference_controller(solvers, tag, name):
    """Row (f1) pinned: the HIP warm-start kernel against closed-loop rollouts of the REFERENCE's own PIDLaneFollower
    (tests/golden/pid_ref.npz, made by importing /root/reference/DGSQP/solvers/PID.py in the build container)."""
    gold = np.load(GOLD / 'pid_ref.npz')
    s = solvers[name]
    dev = s.pid_warm_start_batch(gold[f'{tag}_x0'], du_max=tuple(gold[f'{tag}_du']), want_trajectories=True)
    u_ref = np.concatenate(list(gold[f'{tag}_u_ws'].transpose(1, 0, 2, 3)), axis=2)       # [B, N, 2M] time-major joint
    q_ref = np.concatenate(list(gold[f'{tag}_q_ws'].transpose(1, 0, 2, 3)), axis=2)
    assert np.abs(dev['u_ws'] - u_ref).max() < 1e-10 and np.abs(dev['q_ws'] - q_ref).max() < 1e-10


def test_concurrent_launches_on_two_handles(games):
    """dgsqp_launch_staged / dgsqp_wait: two batches in flight on two handles of the same game (own stream, workspace,
    result buffers) give bit-identical results to one-at-a-time solves."""
    import ctypes as C
    from dgsqp_amd import _ffi
    from dgsqp_amd.montecarlo import sample_scenarios
    from dgsqp_amd.solver import DGSQP
    g, P, par = games['kb_chicane_N15']
    sa, sb = DGSQP(*g.solver_args(), print_method=None), DGSQP(*g.solver_args(), print_method=None)
    xa, ua = sample_scenarios(g, 300, seed=41)
    xb, ub = sample_scenarios(g, 300, seed=42)
    ref_a, ref_b = sa.solve_batch(xa, ua), sb.solve_batch(xb, ub)
    lib = sa._lib
    for s_, x_, u_ in ((sa, xa, ua), (sb, xb, ub)):
        u_am = np.ascontiguousarray(s_._to_agent_major(u_))
        assert lib.dgsqp_stage_inputs(s_._h, x_.shape[0], _ffi.dptr(np.ascontiguousarray(x_)), _ffi.dptr(u_am)) == 0
    assert lib.dgsqp_launch_staged(sa._h) == 0 and lib.dgsqp_launch_staged(sb._h) == 0      # both in flight
    tm = _ffi.TimingT()
    assert lib.dgsqp_wait(sb._h, C.byref(tm)) == 0 and tm.kernel_ms > 0
    assert lib.dgsqp_draining(sb._h) == 1          # nothing queued any more on a finished launch
    assert lib.dgsqp_wait(sa._h, C.byref(tm)) == 0 and tm.kernel_ms > 0
    for s_, ref in ((sa, ref_a), (sb, ref_b)):
        B = ref['u'].shape[0]
        u = np.empty((B, s_.n)); l = np.empty((B, s_.n_c_total)); st = np.empty(B, np.int32); it = np.empty(B, np.int32)
        assert lib.dgsqp_fetch_results(s_._h, _ffi.dptr(u), _ffi.dptr(l), None, _ffi.iptr(st), _ffi.iptr(it), None, None, None) == 0
        assert np.array_equal(st, ref['status']) and np.array_equal(it, ref['num_iters'])
        assert np.array_equal(u, ref['u']) and np.array_equal(l, ref['l'])


def test_grouped_launch_of_three_staged_batches(games):
    """dgsqp_launch_staged_group: ONE launch over the staged batches of three handles (shared ticket queue, own buffers) gives every
    batch bit-identical results to its own launch; members are in flight until their own wait, mismatched groups are refused."""
    import ctypes as C
    from dgsqp_amd import _ffi
    from dgsqp_amd.montecarlo import sample_scenarios
    from dgsqp_amd.solver import DGSQP
    g, P, par = games['kb_chicane_N15']
    solvers = [DGSQP(*g.solver_args(), print_method=None) for _ in range(3)]
    lib = solvers[0]._lib
    data = [sample_scenarios(g, 150, seed=50 + j) for j in range(3)]
    refs = [s_.solve_batch(x_, u_) for s_, (x_, u_) in zip(solvers, data)]
    for s_, (x_, u_) in zip(solvers, data):
        u_am = np.ascontiguousarray(s_._to_agent_major(u_))
        assert lib.dgsqp_stage_inputs(s_._h, x_.shape[0], _ffi.dptr(np.ascontiguousarray(x_)), _ffi.dptr(u_am)) == 0
    arr = (C.c_void_p * 3)(*[s_._h for s_ in solvers])
    assert lib.dgsqp_launch_staged_group(arr, 3) == 0, lib.dgsqp_last_error(solvers[0]._h)
    tm = _ffi.TimingT()
    assert lib.dgsqp_wait(solvers[2]._h, C.byref(tm)) == 0 and tm.kernel_ms > 0 and tm.grid > 0        # a member reports the group's kernel
    assert lib.dgsqp_finished(solvers[1]._h) == 1 and lib.dgsqp_draining(solvers[0]._h) == 1
    for s_, ref in zip(solvers, refs):                         # (fetching waits for the group where dgsqp_wait has not been called)
        B = ref['u'].shape[0]
        u = np.empty((B, s_.n)); l = np.empty((B, s_.n_c_total)); st = np.empty(B, np.int32); it = np.empty(B, np.int32); qs = np.empty(B, np.int32)
        assert lib.dgsqp_fetch_results(s_._h, _ffi.dptr(u), _ffi.dptr(l), None, _ffi.iptr(st), _ffi.iptr(it), _ffi.iptr(qs), None, None) == 0
        assert np.array_equal(st, ref['status']) and np.array_equal(it, ref['num_iters']) and np.array_equal(qs, ref['qp_solves'])
        assert np.array_equal(u, ref['u']) and np.array_equal(l, ref['l'])
    # a handle twice / batches of different sizes: refused, nothing launched
    bad = (C.c_void_p * 2)(solvers[0]._h, solvers[0]._h)
    assert lib.dgsqp_launch_staged_group(bad, 2) != 0
    x_, u_ = sample_scenarios(g, 20, seed=60)
    assert lib.dgsqp_stage_inputs(solvers[1]._h, 20, _ffi.dptr(np.ascontiguousarray(x_)), _ffi.dptr(np.ascontiguousarray(solvers[1]._to_agent_major(u_)))) == 0
    bad = (C.c_void_p * 2)(solvers[0]._h, solvers[1]._h)
    assert lib.dgsqp_launch_staged_group(bad, 2) != 0
    # the solvers are still usable on their own
    again = solvers[1].solve_batch(*data[1])
    assert np.array_equal(again['u'], refs[1]['u'])
    # the Python front end of the same call
    from dgsqp_amd.solver import solve_batches
    outs = solve_batches(solvers, data)
    for o, ref in zip(outs, refs):
        for k in ('status', 'num_iters', 'qp_solves', 'u', 'l', 'x', 'cond', 'cost'):
            assert np.array_equal(o[k], ref[k]), k


@pytest.mark.parametrize('comp_type', ['atan', 'linear'])
def test_blocking_and_obstacle_cost_terms(oracle, comp_type):
    """Cost terms of scripts/DGSQP_monte_carlo_ablation.py:229-262 that the preset games leave at zero weight: blocking
    (e_y coupling), soft obstacle (active hinge) and both competition types -- evaluation incl. Q, and whole solves."""
    import dataclasses
    from dgsqp_amd.game import RacingCost
    from dgsqp_amd.montecarlo import kinematic_racing_game, sample_scenarios
    from dgsqp_amd.solver import DGSQP, build_problem, build_params
    import dgsqp_amd.solver as sv
    g = kinematic_racing_game('curve', N=12, M=3)
    cost = lambda: RacingCost(input_weight=(1.0, 1.0), input_rate_weight=(1.0, 1.0), comp_weights=(10.0, 5.0), comp_type=comp_type,
                              blocking_weight=0.7, obs_weight=3.0, obs_r=0.9)     # hinge active for agents closer than 1.8
    g = dataclasses.replace(g, costs=[cost() for _ in range(3)])
    P, par = build_problem(*g.solver_args()), tight_lsqr(build_params(g.params))
    s = DGSQP(*g.solver_args(), print_method=None, lsqr_tol=1e-13)
    B, N, M = 8, 12, 3
    x0, u_tm = sample_scenarios(g, B, seed=6)
    u = np.ascontiguousarray(u_tm.reshape(B, N, M, 2).transpose(0, 2, 1, 3).reshape(B, -1))
    rng = np.random.default_rng(2)
    up = u + 0.01 * rng.standard_normal(u.shape)
    l = np.maximum(0, rng.standard_normal((B, s.n_c_total)))
    ev = s.evaluate_batch(x0, up, l)
    active = 0
    for b in range(B):
        o = oracle.evaluate(P, x0[b], up[b], l[b], 1)
        for key in ('x', 'q', 'g', 'G', 'Q'):
            assert rel(ev[key][b], o[key]) < 1e-12, (key, b)
        p = o['x'].reshape(N + 1, M, 6)[:, :, :2]
        active += int((np.linalg.norm(p[:, 0] - p[:, 1], axis=1) < 1.8).any())
    assert active > 0                                   # the hinge really was active somewhere
    res = s.solve_batch(x0, u_tm)
    ref = oracle.solve_batch(P, par, x0, u, nthreads=4)
    same = assert_control_flow_parity(res, ref, stable_mask(oracle, P, par, x0, u, ref), 'blocking / obstacle-cost game', max_conv_gap=0.13)
    for b in np.nonzero(same & (ref['status'] <= 1))[0]:
        assert rel(res['u'][b], ref['u'][b]) < 1e-5


@pytest.mark.parametrize('kind,method,msub,over', [
    ('dyn', 'rk3', 2, dict(tire_model='linear')),
    ('dyn', 'rk2', 2, dict(simple_slip=True, drive_wheels='rear')),
    ('dyn', 'euler', 1, dict(rolling_resistance=0.05, rolling_resistance_exponent=0.5, drag_coefficient=0.1, damping_coefficient=0.05)),
    ('kin', 'rk4', 3, dict(rolling_resistance=0.05, rolling_resistance_exponent=0.5)),
    ('kin', 'rk2', 2, dict()),
])
def test_model_and_integrator_variants(oracle, kind, method, msub, over):
    """Vehicle-model options and integrators the preset games do not use (dynamics_models.py:188-219 rk3/rk2, linear tyres,
    simple slip angle, rear-wheel drive, rolling resistance): rollout, q, g, G and the game Hessian against the oracle."""
    import dataclasses
    from dgsqp_amd.dynamics import (CasadiDynamicBicycleCombined, CasadiKinematicBicycleCombined,
                                    CasadiDecoupledMultiAgentDynamicsModel)
    from dgsqp_amd.montecarlo import dynamic_racing_game, kinematic_racing_game, sample_scenarios
    from dgsqp_amd.solver import DGSQP, build_problem
    base = dynamic_racing_game(N=8, rk4_substeps=3, game_def='curve') if kind == 'dyn' else kinematic_racing_game('curve', N=8)
    cls = CasadiDynamicBicycleCombined if kind == 'dyn' else CasadiKinematicBicycleCombined
    models = [cls(0, dataclasses.replace(m.model_config, discretization_method=method, M=msub, **over), track=base.track)
              for m in base.joint_model.dynamics_models]
    joint = CasadiDecoupledMultiAgentDynamicsModel(0, models, dataclasses.replace(base.joint_model.model_config,
                                                                                  discretization_method=method, M=msub))
    g = dataclasses.replace(base, joint_model=joint)
    P = build_problem(*g.solver_args())
    s = DGSQP(*g.solver_args(), print_method=None)
    B = 4
    x0, u_tm = sample_scenarios(base, B, seed=8)
    rng = np.random.default_rng(4)
    u = agent_major(u_tm) + 0.01 * rng.standard_normal((B, s.n))
    l = np.maximum(0, rng.standard_normal((B, s.n_c_total)))
    ev = s.evaluate_batch(x0, u, l)
    for b in range(B):
        o = oracle.evaluate(P, x0[b], u[b], l[b], 1)
        for key in ('x', 'q', 'g', 'G', 'Q'):
            assert rel(ev[key][b], o[key]) < 1e-11, (key, b)


@pytest.mark.parametrize('kind,opts', [('barc2', {}), ('kb_curve_reg0', {}),
                                       ('barc2', dict(eig_floor=1e-6, snap_active_bounds=True)), ('kb_curve_reg0', dict(eig_floor=1e-6, snap_active_bounds=True))])
def test_reg0_games_track_the_oracle(oracle, kind, opts):
    """reg = 0 (curve.py:161, comp.py:169): _nearestPD leaves the clamped eigenvalues at the floor.  Default = the literal 1e-10
    (DGSQP.py:1293; condition ~1e12..1e13, classical J = L^-T active-set kernels on the device); opt-in = floor 1e-6 with the
    active-bound snap (explicit-inverse kernels).  Since round 3 both sides return the POLISHED point -- the KKT point of the final
    active set, as OSQP's polish does for the reference: the dual method's own iterate is off by up to 0.5 |du| on these QPs, the
    polished points are exact to ~1e-10 (profiles/r03_reg0_qp_study.txt, checked against 60-digit KKT solves).  Device and oracle
    must agree on >= 95 % of the scenarios the oracle itself reproduces, converged fractions within 2 points, iterates to 1e-5."""
    from dgsqp_amd.montecarlo import barc_racing_game, kinematic_racing_game, sample_scenarios
    from dgsqp_amd.solver import DGSQP, build_problem, build_params
    g = barc_racing_game(N=15, M=2) if kind == 'barc2' else kinematic_racing_game('curve', N=20, reg=0.0)
    P, par = build_problem(*g.solver_args()), build_params(g.params, lsqr_tol=1e-13, **opts)
    assert par.reg == 0.0 and par.eig_floor == pytest.approx(opts.get('eig_floor', 1e-10))
    s = DGSQP(*g.solver_args(), print_method=None, lsqr_tol=1e-13, **opts)
    literal = not opts
    B = 48 if kind == 'barc2' else 192      # (curve game: 48 scenarios leave ~37 oracle-stable ones, two forks more or less move the fraction by 5 points)
    x0, u_tm = sample_scenarios(g, B, seed=0 if kind == 'barc2' else 1)
    u = agent_major(u_tm)
    l0 = np.array([oracle.dual_init(P, par, x0[b], u[b]) for b in range(4)])
    qp = s.qp_batch(x0[:4], u[:4], l0)
    for b in range(4):     # the projected Hessian itself, floor included; the QP: feasible and as good as the oracle's point
        o = oracle.evaluate(P, x0[b], u[b], l0[b], 1)
        Qpd = oracle.nearest_pd(o['Q'], 0.0, par.eig_floor)
        assert np.abs(qp['Qpd'][b] - Qpd).max() < 1e-11 * max(1.0, np.abs(o['Q']).max())
        du, lam, flag = oracle.qp(Qpd, o['q'], o['G'], o['g'])
        obj = lambda z: 0.5 * z @ Qpd @ z + o['q'] @ z
        assert qp['flag'][b] == flag == 0 and (o['G'] @ qp['du'][b] + o['g']).max() < 1e-4
        assert abs(obj(qp['du'][b]) - obj(du)) < 1e-6 * max(1.0, abs(obj(du)))
        assert rel(qp['du'][b], du) < 1e-6, (b, rel(qp['du'][b], du))
    res = s.solve_batch(x0, u_tm)
    ref = oracle.solve_batch(P, par, x0, u, nthreads=8)
    same = assert_control_flow_parity(res, ref, stable_mask(oracle, P, par, x0, u, ref), f'{kind} {opts}',
                                      min_stable_same=0.95, max_conv_gap=0.02)
    ok = same & (ref['status'] <= 1)
    assert ok.sum() >= B // 3
    worst = max(rel(res['u'][b], ref['u'][b]) for b in np.where(ok)[0])
    print(f'{kind} {opts}: largest relative difference of a converged iterate {worst:.2e}')
    assert worst < 1e-5


def test_classical_qp_storage_split_of_the_triangular_factor(monkeypatch):
    """The classical active-set QP keeps the leading columns of R in LDS and the rest in the scratch (dgsqp_layout.h: c_rcap).  Games of
    this size never fill the LDS part, so the split is forced to 3 and to 0 columns: same arithmetic, different storage -- every
    output must be bitwise identical.  Covers the LDS-resident J (n = 80), the L2-resident J of the big layout (merge, n = 120) and the
    XL layout (3 agents, n = 150; warm-started QPs)."""
    from dgsqp_amd.montecarlo import kinematic_racing_game, merge_game, sample_scenarios
    from dgsqp_amd.solver import DGSQP
    for g, B in ((kinematic_racing_game('curve', N=20, reg=0.0), 24), (merge_game(), 8), (kinematic_racing_game('curve', N=25, M=3), 6)):
        x0, u_tm = sample_scenarios(g, B, seed=3)
        runs = []
        for cap in (None, '3', '0'):
            if cap is None:
                monkeypatch.delenv('DGSQP_RCAP', raising=False)
            else:
                monkeypatch.setenv('DGSQP_RCAP', cap)
            s = DGSQP(*g.solver_args(), print_method=None)
            runs.append(s.solve_batch(x0, u_tm))
        assert (runs[0]['qp_solves'] > 0).all() and (runs[0]['status'] <= 1).mean() > 0.5
        for r in runs[1:]:
            for k in ('status', 'num_iters', 'qp_solves', 'u', 'l', 'cond'):
                assert np.array_equal(r[k], runs[0][k]), k


def test_big_layout_and_merge_game(oracle):
    """Games beyond the LDS-resident layout keep the packed inverse and the reflectors in the workgroup's L2 scratch
    (n up to 128): KB curve N=30 (n=120) and the three-car merge of scripts/DGSQP_merge_monte_carlo.py at its own
    horizon N=20 (unicycles, goal-tracking costs, lane rows, reg=0) against the oracle."""
    from dgsqp_amd.montecarlo import kinematic_racing_game, merge_game, sample_scenarios
    from dgsqp_amd.solver import DGSQP, build_problem, build_params
    import dgsqp_amd.solver as sv
    # ... and the same merge with six cars (BASELINE configs[4]'s family; N = 10 is what n <= 128 allows)
    for g, B, noise in ((kinematic_racing_game('curve', N=30), 24, 0.01), (merge_game(N=20), 32, 0.05), (merge_game(N=12), 16, 0.05),
                        (merge_game(N=10, M=6), 16, 0.05)):
        N, M = g.params.N, g.joint_model.n_a
        P, par = build_problem(*g.solver_args()), tight_lsqr(build_params(g.params))
        s = DGSQP(*g.solver_args(), print_method=None, lsqr_tol=1e-13)
        assert s.dims.lds_bytes <= 163840
        x0, u_tm = sample_scenarios(g, B, seed=1)
        u = agent_major(u_tm)
        rng = np.random.default_rng(1)
        up = u + noise * rng.standard_normal(u.shape)
        l = np.maximum(0, rng.standard_normal((B, s.n_c_total)))
        ev = s.evaluate_batch(x0[:4], up[:4], l[:4])
        for b in range(4):
            o = oracle.evaluate(P, x0[b], up[b], l[b], 1)
            for key in ('x', 'q', 'g', 'G', 'Q'):
                assert rel(ev[key][b], o[key]) < 1e-11, (g.name, key, b)
        res = s.solve_batch(x0, u_tm)
        ref = oracle.solve_batch(P, par, x0, u, nthreads=8)
        same = assert_control_flow_parity(res, ref, stable_mask(oracle, P, par, x0, u, ref, K=2), g.name, min_stable_same=0.95)
        tol = 1e-5
        for b in np.where(same & (ref['status'] <= 1))[0]:
            assert rel(res['u'][b], ref['u'][b]) < tol and rel(res['l'][b], ref['l'][b]) < 10 * tol, (g.name, b)
        if M >= 3:
            assert (res['status'] <= 1).all()
            with pytest.raises(RuntimeError):
                s.pid_warm_start_batch(x0)


def test_xl_layout_three_agents_n150(oracle):
    """BASELINE configs[2] size / scripts/DGSQP_monte_carlo_agents.py at exp_M=[3], exp_N=[25] (:101-102): n = 150 decision
    variables, 825 rows.  Beyond 128 variables the PSD / QP phases run the generic global-memory kernels of dgsqp_xl.h
    (Jacobi eigen-decomposition, classical Goldfarb-Idnani); evaluation, dual start and the SQP logic are the common code."""
    from dgsqp_amd.montecarlo import kinematic_racing_game, sample_scenarios
    from dgsqp_amd.solver import DGSQP, build_problem, build_params
    import dgsqp_amd.solver as sv
    g = kinematic_racing_game('curve', N=25, M=3)
    P, par = build_problem(*g.solver_args()), tight_lsqr(build_params(g.params))
    s = DGSQP(*g.solver_args(), print_method=None, lsqr_tol=1e-13)
    assert (s.n, s.n_c_total) == (150, 825) and s.dims.lds_bytes <= 163840
    B = 8
    x0, u_tm = sample_scenarios(g, B, seed=1)
    u = agent_major(u_tm)
    rng = np.random.default_rng(1)
    up = u + 0.01 * rng.standard_normal(u.shape)
    l = np.maximum(0, rng.standard_normal((B, s.n_c_total)))
    ev = s.evaluate_batch(x0[:2], up[:2], l[:2])
    l0 = [oracle.dual_init(P, par, x0[b], u[b]) for b in range(2)]
    qp = s.qp_batch(x0[:2], u[:2], np.array(l0))
    for b in range(2):
        o = oracle.evaluate(P, x0[b], up[b], l[b], 1)
        for key in ('x', 'q', 'g', 'G', 'Q'):
            assert rel(ev[key][b], o[key]) < 1e-11, (key, b)
        assert rel(ev['l0'][b], oracle.dual_init(P, par, x0[b], up[b])) < 1e-7
        o = oracle.evaluate(P, x0[b], u[b], l0[b], 1)
        Qpd = oracle.nearest_pd(o['Q'], par.reg, par.eig_floor)
        du, lam, flag = oracle.qp(Qpd, o['q'], o['G'], o['g'])
        assert qp['flag'][b] == flag == 0 and np.abs(qp['Qpd'][b] - Qpd).max() < 1e-10 * np.abs(o['Q']).max()
        assert rel(qp['du'][b], du) < 1e-8 and np.array_equal(qp['lhat'][b] > 0, lam > 0)
    res = s.solve_batch(x0, u_tm)
    ref = oracle.solve_batch(P, par, x0, u, nthreads=8)
    same = (res['status'] == ref['status']) & (res['num_iters'] == ref['num_iters']) & (res['qp_solves'] == ref['qp_solves'])
    assert same.sum() >= B - 2, same
    for b in np.where(same & (ref['status'] <= 1))[0]:
        assert rel(res['u'][b], ref['u'][b]) < 1e-6 and rel(res['l'][b], ref['l'][b]) < 1e-5


@pytest.mark.parametrize('kind', ['curve2', 'dyn2', 'agents3', 'circuit3', 'merge3'])
def test_device_sampler_matches_its_host_mirror(kind):
    """Row (f1): the rejection samplers on the device (dgsqp_sample_batch: counter-based placement, PID warm start, collision check,
    compaction in candidate order) against the numpy mirror (dgsqp_amd/sampler.py): same candidates accepted, the same number
    consumed, initial states to 1e-12 (sin / cos of two maths libraries), warm starts to 1e-9; and a batch sampled with stage=True
    solves to the same results as the same batch handed over from the host."""
    from dgsqp_amd import montecarlo as mc, sampler as smp
    from dgsqp_amd.solver import DGSQP
    g = {'curve2': lambda: mc.kinematic_racing_game('curve', N=15), 'dyn2': lambda: mc.dynamic_racing_game(N=10, rk4_substeps=4),
         'agents3': lambda: mc.kinematic_racing_game('curve', N=10, M=3), 'circuit3': lambda: mc.barc_racing_game(N=10, M=3),
         'merge3': lambda: mc.merge_game(N=12)}[kind]()
    s = DGSQP(*g.solver_args(), print_method=None)
    B = 300 if kind in ('curve2', 'dyn2') else 96            # (more than one round of candidates for the two-car games)
    dev = s.sample_batch(g, B, seed=11)
    x0, u_ws, used = smp.sample_scenarios_counter(g, B, seed=11, chunk=128)
    assert dev['candidates'] == used, (dev['candidates'], used)
    assert np.abs(dev['x0'] - x0).max() < 1e-12 and np.abs(dev['u_ws'] - u_ws).max() < 1e-9
    ref = s.solve_batch(dev['x0'], dev['u_ws'])
    s.sample_batch(g, B, seed=11, stage=True, fetch=False)
    from dgsqp_amd import _ffi
    import ctypes
    tm = _ffi.TimingT()
    assert s._lib.dgsqp_solve_staged(s._h, ctypes.byref(tm)) == 0
    st, it, qp = (np.empty(B, np.int32) for _ in range(3))
    u = np.empty((B, s.n))
    assert s._lib.dgsqp_fetch_results(s._h, _ffi.dptr(u), None, None, _ffi.iptr(st), _ffi.iptr(it), _ffi.iptr(qp), None, None) == 0
    assert np.array_equal(st, ref['status']) and np.array_equal(it, ref['num_iters']) and np.array_equal(qp, ref['qp_solves'])
    assert np.array_equal(u, ref['u'], equal_nan=True)


def test_cooperative_line_search_is_bit_identical(games):
    """Workgroups that run out of scenarios evaluate line-search trial points for the ones still solving (dgsqp_set_cooperative;
    default in the synchronous calls).  Same device functions, same reductions: every output of a cooperative launch equals the
    non-cooperative one bit for bit; helpers did evaluate trials; no owner ever gave up waiting for one."""
    from dgsqp_amd.montecarlo import sample_scenarios
    from dgsqp_amd.solver import DGSQP
    import os
    for name, B, force in (('dyn_curve_N25', 768, False), ('kb_chicane_N25', 768, False)):
        g = games[name][0]
        s = DGSQP(*g.solver_args(), print_method=None)
        x0, u_tm = sample_scenarios(g, B, seed=5)
        s.set_cooperative(0)
        ref = s.solve_batch(x0, u_tm)
        s.set_cooperative(1)
        os.environ['DGSQP_COOP_VERIFY'] = '1'            # owners re-evaluate every value a helper hands them and count differing bits
        if force:
            os.environ['DGSQP_COOP_FORCE'] = '1'
        try:
            chk = s.solve_batch(x0, u_tm)
            st_chk = s.coop_stats()
            os.environ.pop('DGSQP_COOP_VERIFY')
            res = s.solve_batch(x0, u_tm)
            st = s.coop_stats()
        finally:
            os.environ.pop('DGSQP_COOP_VERIFY', None)
            os.environ.pop('DGSQP_COOP_FORCE', None)
        print(name, 'kernel ms alone', ref['kernel_ms'], 'cooperative', res['kernel_ms'], st, 'verify run', st_chk)
        for r in (res, chk):
            for k in ('status', 'num_iters', 'qp_solves', 'u', 'l', 'x', 'cond', 'cost'):
                assert np.array_equal(r[k], ref[k], equal_nan=True), (name, k)      # (NaN iterates of qp_fail / diverged runs are outputs too)
        assert st['helper_registrations'] > 0 and st['idle'] == 0 and st['finished'] == B and st['helped'] > 0 and st['used'] > 0, st
        assert st_chk['mismatches'] == 0 and st_chk['used'] > 0, st_chk


def test_deferral_of_long_scenarios_is_bit_identical(games):
    """Cooperative launches set scenarios that iterate much longer than the others aside while fresh ones wait (LDS arena + scratch
    stored in HBM) and resume them, most expensive first, once the queue is empty (dgsqp_set_deferral).  Scheduling only: every
    output equals the plain launch bit for bit, in a grouped launch and in a single one; every deferred scenario was resumed."""
    from dgsqp_amd.montecarlo import sample_scenarios
    from dgsqp_amd.solver import DGSQP, solve_batches
    keys = ('status', 'num_iters', 'qp_solves', 'u', 'l', 'x', 'cond', 'cost')
    for name, B, G in (('dyn_curve_N25', 640, 3), ('kb_chicane_N25', 1536, 1), ('kb_barc2_N15', 768, 2)):
        g = games[name][0]
        solvers = [DGSQP(*g.solver_args(), print_method=None) for _ in range(G)]
        batches = [sample_scenarios(g, B, seed=11 + i) for i in range(G)]
        ref = []
        for s, (x0, u) in zip(solvers, batches):
            s.set_cooperative(0)
            ref.append(s.solve_batch(x0, u))
            s.set_cooperative(1)
        for min_it, factor in ((0, 2.0), (8, 2.0), (4, 0.5)):
            solvers[0].set_deferral(min_it, factor)
            res = solve_batches(solvers, batches)
            st = solvers[0].deferral_stats()
            print(name, 'deferral', (min_it, factor), st, 'kernel ms', res[0]['kernel_ms'], 'plain launches', [round(r['kernel_ms'], 1) for r in ref])
            for r, r0 in zip(res, ref):
                for k in keys:
                    assert np.array_equal(r[k], r0[k], equal_nan=True), (name, min_it, factor, k)
            assert st['deferred'] == st['resumed']
            assert st['deferred'] == 0 if min_it == 0 else (st['deferred'] > 0 or min_it > 4), st      # (the aggressive setting always finds some)
        log = solvers[0].deferral_log()
        assert len(log) == st['deferred'] and (log[:, 6] >= log[:, 5]).all() and (log[:, 5] >= log[:, 4]).all()     # set aside <= resumed <= finished
        solvers[0].set_deferral(8, 2.0)


def test_deferral_of_long_v2_scenarios_is_bit_identical():
    """The same for DG-SQP v2 (round 4): its loop carries more state -- decaying reg, trust radius, both at the checkpoint, m-step and
    checkpoint counters, the merit-memory ring -- which travels in the deferral entry; the iteration records and the previous iterate
    are part of the stored scratch image.  Outputs equal the plain launch bit for bit; every deferred scenario was resumed."""
    from dgsqp_amd.montecarlo import dynamic_racing_game, sample_scenarios
    from dgsqp_amd.solver import DGSQP, solve_batches
    keys = ('status', 'num_iters', 'qp_solves', 'u', 'l', 'x', 'cond', 'cost')
    g = dynamic_racing_game(N=15, rk4_substeps=4, game_def='curve', solver='v2')
    B, G = 512, 4        # (nothing is deferred unless two more rounds of fresh scenarios wait behind the 256 workgroups)
    solvers = [DGSQP(*g.solver_args(), print_method=None) for _ in range(G)]
    batches = [sample_scenarios(g, B, seed=21 + i) for i in range(G)]
    ref = []
    for s, (x0, u) in zip(solvers, batches):
        s.set_cooperative(0)
        ref.append(s.solve_batch(x0, u))
        s.set_cooperative(1)
    for min_it, factor in ((8, 2.0), (4, 0.5)):
        solvers[0].set_deferral(min_it, factor)
        res = solve_batches(solvers, batches)
        st = solvers[0].deferral_stats()
        print('v2 deferral', (min_it, factor), st, 'kernel ms', res[0]['kernel_ms'], 'plain launches', [round(r['kernel_ms'], 1) for r in ref],
              'iterations mean / max', ref[0]['num_iters'].mean(), ref[0]['num_iters'].max())
        for r, r0 in zip(res, ref):
            for k in keys:
                assert np.array_equal(r[k], r0[k], equal_nan=True), (min_it, factor, k)
        assert st['deferred'] == st['resumed'] and (st['deferred'] > 0 or min_it > 4), st
    solvers[0].set_deferral(8, 2.0)


def test_six_agent_merge_n300(oracle):
    """BASELINE configs[4]'s game at its own size: six cars on the highway merge (DGSQP_merge_monte_carlo.py:66-74, 253-261, 316-342
    generalised to six cars), N = 25: n = 300 decision variables, 36 / 63 / 39 rows per stage = 1,587 rows, 837 distinct dense
    gradients (31,200 doubles, in the L2 scratch), tables read from the constant block, XL kernels with five registers per column.
    Stage quantities to 1e-11, _nearestPD + QP against the oracle, full solves identical in (status, iterations, QP solves)."""
    from dgsqp_amd.montecarlo import merge_game, sample_scenarios
    from dgsqp_amd.solver import DGSQP, build_problem, build_params
    g = merge_game(N=25, M=6)
    P, par = build_problem(*g.solver_args()), tight_lsqr(build_params(g.params))
    s = DGSQP(*g.solver_args(), print_method=None, lsqr_tol=1e-13)
    assert (s.n, s.n_c_total, s.dims.n_dense, s.dims.layout) == (300, 1587, 837, 2) and s.dims.lds_bytes <= 163840
    B = 32
    x0, u_tm = sample_scenarios(g, B, seed=1)
    u = agent_major(u_tm)
    rng = np.random.default_rng(1)
    up = u + 0.05 * rng.standard_normal(u.shape)
    l = np.maximum(0, rng.standard_normal((B, s.n_c_total)))
    ev = s.evaluate_batch(x0[:3], up[:3], l[:3])
    for b in range(3):
        o = oracle.evaluate(P, x0[b], up[b], l[b], 1)
        for key in ('x', 'q', 'g', 'G', 'Q'):
            assert rel(ev[key][b], o[key]) < 1e-11, (key, b)
        assert rel(ev['l0'][b], oracle.dual_init(P, par, x0[b], up[b])) < 1e-6, b
    l0 = np.array([oracle.dual_init(P, par, x0[b], u[b]) for b in range(2)])
    qp = s.qp_batch(x0[:2], u[:2], l0)
    for b in range(2):
        o = oracle.evaluate(P, x0[b], u[b], l0[b], 1)
        Qpd = oracle.nearest_pd(o['Q'], par.reg, par.eig_floor)
        du, lam, flag = oracle.qp(Qpd, o['q'], o['G'], o['g'])
        assert qp['flag'][b] == flag == 0 and np.abs(qp['Qpd'][b] - Qpd).max() < 1e-10 * np.abs(o['Q']).max()
        assert np.array_equal(qp['lhat'][b] > 0, lam > 0), b
        # (reg = 0, literal floor: condition 1e12 -- the step agrees to the accuracy either side solves that QP to)
        assert rel(qp['du'][b], du) < 1e-6, (b, rel(qp['du'][b], du))
    res = s.solve_batch(x0, u_tm)
    import os
    ref = oracle.solve_batch(P, par, x0, u, nthreads=min(B, os.cpu_count() or 8))
    same = assert_control_flow_parity(res, ref, stable_mask(oracle, P, par, x0, u, ref, K=1), g.name, min_stable_same=0.95)
    assert (res['status'] <= 1).mean() >= 0.9 and same.sum() >= B - 2
    for b in np.where(same & (ref['status'] <= 1))[0]:
        assert rel(res['u'][b], ref['u'][b]) < 1e-5 and rel(res['x'][b], ref['x'][b]) < 1e-6, b


@pytest.mark.parametrize('N,B', [(15, 24), (25, 48)])
def test_three_agents_on_the_barc_circuit(oracle, N, B):
    """BASELINE configs[2]'s own game: 3 kinematic bicycles on the L_track_barc circuit (DGSQP_comp_monte_carlo.py game with a
    third car, 24 / 33 / 9 rows per stage, reg = 0), at the script's N = 15 (n = 90, LDS layout) and at BASELINE's N = 25
    (n = 150, 825 rows, XL layout): stage quantities to 1e-11, control flow identical on the oracle-stable scenarios.  At N = 25
    DG-SQP v1 itself fails on this game -- 11 of 12 scenarios end in infeasible QPs or at the iteration limit, on the oracle as on
    the device, and only a third of the runs survive a 1e-13 perturbation of the oracle's inputs -- which is what the test pins."""
    from dgsqp_amd.montecarlo import barc_racing_game, sample_scenarios
    from dgsqp_amd.solver import DGSQP, build_problem, build_params
    g = barc_racing_game(N=N, M=3)
    P, par = build_problem(*g.solver_args()), build_params(g.params, lsqr_tol=1e-13)
    s = DGSQP(*g.solver_args(), print_method=None, lsqr_tol=1e-13)
    assert (s.n, s.n_c_total) == (6 * N, 24 + 33 * (N - 1) + 9) and s.dims.layout == (0 if N == 15 else 2)
    x0, u_tm = sample_scenarios(g, B, seed=0)
    u = agent_major(u_tm)
    rng = np.random.default_rng(1)
    up = u + 0.01 * rng.standard_normal(u.shape)
    l = np.maximum(0, rng.standard_normal((B, s.n_c_total)))
    ev = s.evaluate_batch(x0[:2], up[:2], l[:2])
    for b in range(2):
        o = oracle.evaluate(P, x0[b], up[b], l[b], 1)
        for key in ('x', 'q', 'g', 'G', 'Q'):
            assert rel(ev[key][b], o[key]) < 1e-11, (key, b)
    res = s.solve_batch(x0, u_tm)
    import os
    ref = oracle.solve_batch(P, par, x0, u, nthreads=min(B, os.cpu_count() or 8))
    stable = stable_mask(oracle, P, par, x0, u, ref, K=2)
    same = assert_control_flow_parity(res, ref, stable, f'barc3 N={N}', min_stable_same=0.95, max_conv_gap=0.05, min_stable_frac=0.25)
    assert stable.sum() >= 16            # (that many oracle-stable scenarios back the parity claim)
    # (XL layout, reg = 0, n = 150: identical control flow over ~40 iterations of condition-1e12 QPs; measured 1.6e-4 on the one scenario
    # that converges at N = 25)
    # that converges at N = 25: the literal reg = 0 leaves the QP's Hessian with eigenvalues of 1e-10 -- directions along which the iterate is
    # only determined through the active rows, DESIGN.md section 2 -- hence 2e-4 there, not 1e-5; printed so that a drift shows)
    for b in np.where(same & (ref['status'] <= 1))[0]:
        e = rel(res['u'][b], ref['u'][b])
        print(f'barc3 N={N}: scenario {b} (status {ref["status"][b]}, {ref["num_iters"][b]} iterations): iterate difference {e:.1e}')
        assert e < (2e-4 if N == 25 else 1e-5), b


@pytest.mark.parametrize('kind', ['dyn', 'kb', 'kb_sum_obj', 'kb_osqp'])
def test_dgsqp_v2_matches_oracle(oracle, kind):
    """DG-SQP v2 (DGSQP_v2.py:322-720 -- d-steps / m-steps with checkpoints, decaying regularisation,
    merit memory, merit without the complementarity term) on the device against the oracle's restatement: the dynamic-bicycle game
    with the parameters of the reference's study (comparison_study_barc/globals.py) and a kinematic game with the default
    DGSQPV2Params; event logs identical event by event, flags / iteration / QP counts identical, iterates within 1e-5.
    ``kb_osqp``: the kinematic game with OSQP's own arithmetic behind v2's _solve_qp (qp_method='osqp' on both sides; the
    regularisation v2 decays enters the projected Hessian OSQP is handed), six scenarios."""
    import copy
    from dgsqp_amd import montecarlo as mc
    from dgsqp_amd.solver import build_problem, build_params
    from dgsqp_amd.solver_types import DGSQPV2Params
    from dgsqp_amd.solver_v2 import DGSQP as DGSQPv2
    if kind == 'dyn':
        g = mc.dynamic_racing_game(N=10, rk4_substeps=3, solver='v2')
    else:
        g = mc.kinematic_racing_game('curve', N=12)           # DGSQPV2Params defaults: rejected m-steps, checkpoint loads, line searches
        g.params = DGSQPV2Params(dt=0.1, N=12)
        if kind == 'kb_sum_obj':                              # merit 'sum_obj_l1' (DGSQP_v2.py:1161-1164): device = costate sweep, oracle = dense Du_x
            g.params.merit_function = 'sum_obj_l1'
    g.params.time_limit = None
    qpm = 'osqp' if kind == 'kb_osqp' else None
    P, par = build_problem(*g.solver_args()), build_params(g.params, lsqr_tol=1e-13, qp_method=qpm)
    assert par.variant == 1 and par.rel_tol_req == 10 and par.qp_method == (1 if qpm else 0)
    s = DGSQPv2(*g.solver_args(), print_method=None, lsqr_tol=1e-13, qp_method=qpm)
    B = 6 if qpm else 10
    x0, u_tm = mc.sample_scenarios(g, B, seed=2)
    u = agent_major(u_tm)
    s.set_trace(20000)
    try:
        res = s.solve_batch(x0, u_tm)
        traces = s.fetch_trace(B)
    finally:
        s.set_trace(0)
    ref = oracle.solve_batch(P, par, x0, u, nthreads=8)
    same = assert_control_flow_parity(res, ref, stable_mask(oracle, P, par, x0, u, ref, K=2), f'v2 {kind}')
    assert (ref['num_iters'] > 20).all() and (ref['status'] <= 2).all()          # v2 really iterates: reg starts at 100
    if kind != 'kb_sum_obj':
        assert (ref['status'] == 0).all() if kind == 'dyn' else (ref['status'] == 1).any()
    for b in np.where(same)[0]:
        assert rel(res['u'][b], ref['u'][b]) < 1e-5 and rel(res['l'][b], ref['l'][b]) < 1e-4, b
    identical = 0
    for b in range(4):
        to = oracle.solve_trace(P, par, x0[b], u[b], max_pairs=60000)
        tg = traces[b]
        if len(to) == len(tg) and np.array_equal(to[:, 0], tg[:, 0]):
            big = np.abs(to[:, 1]) > 1e-6
            identical += int(np.all(np.abs(tg[big, 1] - to[big, 1]) <= 1e-5 * np.abs(to[big, 1])))
    assert identical >= 3, identical
    # the reference surface: solve() of the v2 class returns the v2 dictionary (DGSQP_v2.py:616-632)
    states = s.joint_dynamics.qu2state(None, x0[0], None)
    s.set_warm_start(u_tm[0])
    info = s.solve(states)
    assert {'primal_sol', 'dual_sol', 'x_pred', 'u_pred', 'conds', 'msg', 'num_iters', 'status'} <= set(info)
    assert info['num_iters'] == int(res['num_iters'][0]) and np.array_equal(info['primal_sol'], res['u'][0])


def test_dgsqp_v2_on_the_xl_layout_with_osqp(oracle):
    """DG-SQP v2 with OSQP's arithmetic on the XL layout (v2_qp -> dev_xl_psd + dev_qp_osqp_xl with v2's decaying regularisation): the
    three-car curve game (n = 150), the iteration limit cut to 12 so that the CPU side stays short; flags, iteration and QP counts
    against the oracle's v2 + OSQP."""
    from dgsqp_amd import montecarlo as mc
    from dgsqp_amd.solver import build_problem, build_params
    from dgsqp_amd.solver_types import DGSQPV2Params
    from dgsqp_amd.solver_v2 import DGSQP as DGSQPv2
    g = mc.kinematic_racing_game('curve', N=25, M=3)
    g.params = DGSQPV2Params(dt=0.1, N=25, sqp_iters=12)
    g.params.time_limit = None
    P, par = build_problem(*g.solver_args()), build_params(g.params, lsqr_tol=1e-13, qp_method='osqp')
    assert par.variant == 1 and par.qp_method == 1
    s = DGSQPv2(*g.solver_args(), print_method=None, lsqr_tol=1e-13, qp_method='osqp')
    assert s.dims.layout == 2
    B = 4
    x0, u_tm = mc.sample_scenarios(g, B, seed=2)
    u = s._to_agent_major(u_tm)
    res = s.solve_batch(x0, u_tm)
    ref = oracle.solve_batch(P, par, x0, u, nthreads=B)
    same = (res['status'] == ref['status']) & (res['num_iters'] == ref['num_iters']) & (res['qp_solves'] == ref['qp_solves'])
    print(f'v2 + OSQP on the XL layout: identical {same.sum()}/{B}; device {res["status"].tolist()} {res["num_iters"].tolist()} {res["qp_solves"].tolist()} oracle {ref["status"].tolist()} {ref["num_iters"].tolist()} {ref["qp_solves"].tolist()}')
    assert same.sum() >= B - 1
    for b in np.nonzero(same)[0]:
        assert rel(res['u'][b], ref['u'][b]) < 1e-5


@pytest.mark.parametrize('model,N,B', [('kinematic', 12, 16), ('dynamic', 8, 8), ('kinematic', 50, 64)])
def test_f1_spline_track_game(oracle, model, N, B):
    """BASELINE configs[3]'s game: two cars on the F1 track, a cubic-spline centre line (CasadiBSplineTrack,
    casadi_bspline_track.py:56-71, :122-149) whose curvature has non-zero derivatives -- evaluated on the device in Taylor
    arithmetic from the spline table.  Stage quantities to 1e-10, the PID warm start on the spline track, full solves against
    the oracle; N = 50 is the configuration's horizon (n = 200, 1,050 rows, XL layout)."""
    from dgsqp_amd import montecarlo as mc
    from dgsqp_amd.solver import DGSQP, build_problem, build_params
    g = mc.f1_racing_game(N=N, model=model, rk4_substeps=3)
    P, par = build_problem(*g.solver_args()), build_params(g.params, lsqr_tol=1e-13)
    s = DGSQP(*g.solver_args(), print_method=None, lsqr_tol=1e-13)
    assert s.dims.layout == (2 if N == 50 else 0) and (N != 50 or (s.n, s.n_c_total) == (200, 1050))
    x0, u_tm = mc.sample_scenarios(g, B, seed=0)
    u = agent_major(u_tm)
    rng = np.random.default_rng(1)
    up = u + 0.02 * rng.standard_normal(u.shape)
    l = np.maximum(0, rng.standard_normal((B, s.n_c_total)))
    ev = s.evaluate_batch(x0[:3], up[:3], l[:3])
    for b in range(3):
        o = oracle.evaluate(P, x0[b], up[b], l[b], 1)
        for key in ('x', 'q', 'g', 'G', 'Q'):
            assert rel(ev[key][b], o[key]) < 1e-10, (key, b)
    dev = s.pid_warm_start_batch(x0, du_max=(10.0, 4.5), want_trajectories=True)
    models = g.joint_model.dynamics_models
    ref_u = np.concatenate([mc.pid_warm_start(m, x0[:, a * m.n_q:(a + 1) * m.n_q], N, 0.1, du=(10.0, 4.5))[1] for a, m in enumerate(models)], axis=2)
    assert np.abs(dev['u_ws'] - ref_u).max() < 1e-9
    res = s.solve_batch(x0, u_tm)
    import os
    ref = oracle.solve_batch(P, par, x0, u, nthreads=min(B, os.cpu_count() or 8))
    stable = stable_mask(oracle, P, par, x0, u, ref, K=2)
    same = assert_control_flow_parity(res, ref, stable, f'f1 {model} N={N}', min_stable_same=0.95, max_conv_gap=0.05, min_stable_frac=0.4)
    assert N != 50 or stable.sum() >= 28            # (that many oracle-stable scenarios back the parity claim at the configuration's own horizon)
    # (N = 50: XL layout, n = 200, reg = 1e-3, up to 47 iterations with identical control flow: the iterate differences of the commonly
    # converged scenarios are printed; all but one are below 2e-5, one run that ends on the relative-tolerance test reaches 7.5e-3)
    ok = np.where(same & (ref['status'] <= 1))[0]
    errs = np.array([rel(res['u'][b], ref['u'][b]) for b in ok])
    print(f'f1 {model} N={N}: iterate differences of {len(ok)} identical converged scenarios: median {np.median(errs):.1e}, max {errs.max():.1e}; above 1e-5: {int((errs > 1e-5).sum())}')
    # exits on the absolute tolerances (status 0) are held to 1e-5 at every horizon; an exit on the relative-tolerance test (status 1: three
    # consecutive steps below p_tol / 2) stops wherever the third short step happens to land, and the two sides may stop 1e-2 apart
    abs_exit = ref['status'][ok] == 0
    assert np.median(errs) < 1e-5 and (not abs_exit.any() or errs[abs_exit].max() < 1e-5) and (abs_exit.all() or errs[~abs_exit].max() < 1e-2)
    assert (errs > 1e-4).sum() <= max(1, len(ok) // 10)


def test_bfgs_hessian_option(oracle):
    """DGSQPParams.hessian_approximation = 'bfgs' (DGSQP.py:353-364, :535-557): exact Hessian at the first iteration, damped
    BFGS updates of the projected Hessian afterwards, against the oracle; more iterations than with exact Hessians."""
    from dgsqp_amd.montecarlo import kinematic_racing_game, sample_scenarios
    from dgsqp_amd.solver import DGSQP, build_problem, build_params
    import dgsqp_amd.solver as sv
    g = kinematic_racing_game('curve', N=15)
    g.params.hessian_approximation = 'bfgs'
    P, par = build_problem(*g.solver_args()), tight_lsqr(build_params(g.params))
    assert par.hessian_bfgs == 1
    s = DGSQP(*g.solver_args(), print_method=None, lsqr_tol=1e-13)
    B = 24
    x0, u_tm = sample_scenarios(g, B, seed=1)
    u = agent_major(u_tm)
    res = s.solve_batch(x0, u_tm)
    ref = oracle.solve_batch(P, par, x0, u, nthreads=8)
    same = assert_control_flow_parity(res, ref, stable_mask(oracle, P, par, x0, u, ref), 'bfgs')
    for b in np.where(same & (ref['status'] <= 1))[0]:
        assert rel(res['u'][b], ref['u'][b]) < 1e-5 and rel(res['l'][b], ref['l'][b]) < 1e-4
    g.params.hessian_approximation = 'none'
    exact = oracle.solve_batch(P, tight_lsqr(build_params(g.params)), x0, u, nthreads=8)
    assert ref['num_iters'].mean() > exact['num_iters'].mean()


def test_xl_layout_long_horizon_n200(oracle):
    """BASELINE configs[3] size: 2 agents, N = 50 -> 200 decision variables, 1,050 rows.  XL layout with the packed constraint
    gradients (82 KB) in the L2 scratch as well; same common code for evaluation / dual start / SQP logic."""
    from dgsqp_amd.montecarlo import kinematic_racing_game, sample_scenarios
    from dgsqp_amd.solver import DGSQP, build_problem, build_params
    import dgsqp_amd.solver as sv
    g = kinematic_racing_game('curve', N=50)
    P, par = build_problem(*g.solver_args()), tight_lsqr(build_params(g.params))
    s = DGSQP(*g.solver_args(), print_method=None, lsqr_tol=1e-13)
    assert (s.n, s.n_c_total) == (200, 1050) and s.dims.layout == 2 and s.dims.lds_bytes <= 163840
    B = 4
    x0, u_tm = sample_scenarios(g, B, seed=1)
    u = agent_major(u_tm)
    rng = np.random.default_rng(1)
    up = u + 0.01 * rng.standard_normal(u.shape)
    l = np.maximum(0, rng.standard_normal((B, s.n_c_total)))
    ev = s.evaluate_batch(x0[:2], up[:2], l[:2])
    for b in range(2):
        o = oracle.evaluate(P, x0[b], up[b], l[b], 1)
        for key in ('x', 'q', 'g', 'G', 'Q'):
            assert rel(ev[key][b], o[key]) < 1e-11, (key, b)
        assert rel(ev['l0'][b], oracle.dual_init(P, par, x0[b], up[b])) < 1e-7
    res = s.solve_batch(x0, u_tm)
    ref = oracle.solve_batch(P, par, x0, u, nthreads=8)
    same = (res['status'] == ref['status']) & (res['num_iters'] == ref['num_iters']) & (res['qp_solves'] == ref['qp_solves'])
    assert same.sum() >= B - 1, (res['status'], ref['status'], res['num_iters'], ref['num_iters'])
    for b in np.where(same & (ref['status'] <= 1))[0]:
        assert rel(res['u'][b], ref['u'][b]) < 1e-6 and rel(res['l'][b], ref['l'][b]) < 1e-5


@pytest.mark.parametrize('M,N', [(3, 30), (5, 25), (2, 64)])
def test_xl_sizes_between_the_configs(oracle, M, N):
    """XL sizes the BASELINE configs do not name: n = 180 (last elimination panel of 4 pivots, last tridiagonalisation panel of 2
    columns), n = 250 (panel of 10; five agents), n = 256 (a multiple of the panel width: no short panel; N = 64 is the horizon limit).
    _nearestPD and the QP against the oracle on the linearisation of perturbed start points, then whole solves."""
    from dgsqp_amd.montecarlo import kinematic_racing_game, sample_scenarios
    from dgsqp_amd.solver import DGSQP, build_problem, build_params
    g = kinematic_racing_game('curve', N=N, M=M)
    P, par = build_problem(*g.solver_args()), tight_lsqr(build_params(g.params))
    s = DGSQP(*g.solver_args(), print_method=None, lsqr_tol=1e-13)
    assert s.n == 2 * N * M and s.dims.layout == 2
    B = 4
    x0, u_tm = sample_scenarios(g, B, seed=4)
    u = agent_major(u_tm)
    l0 = np.array([oracle.dual_init(P, par, x0[b], u[b]) for b in range(2)])
    qp = s.qp_batch(x0[:2], u[:2], l0)
    for b in range(2):
        o = oracle.evaluate(P, x0[b], u[b], l0[b], 1)
        Qpd = oracle.nearest_pd(o['Q'], par.reg, par.eig_floor)
        du, lam, flag = oracle.qp(Qpd, o['q'], o['G'], o['g'])
        assert np.abs(qp['Qpd'][b] - Qpd).max() < 1e-10 * np.abs(o['Q']).max(), (b, np.abs(qp['Qpd'][b] - Qpd).max())
        assert qp['flag'][b] == flag, (b, qp['flag'][b], flag)
        if flag == 0:
            assert rel(qp['du'][b], du) < 1e-6 and np.array_equal(qp['lhat'][b] > 1e-9, lam > 1e-9), (b, rel(qp['du'][b], du))
    res = s.solve_batch(x0, u_tm)
    ref = oracle.solve_batch(P, par, x0, u, nthreads=B)
    same = (res['status'] == ref['status']) & (res['num_iters'] == ref['num_iters']) & (res['qp_solves'] == ref['qp_solves'])
    print(f'xl n={s.n}: identical {same.sum()}/{B}; status device {res["status"]} oracle {ref["status"]}; iterations {res["num_iters"]} {ref["num_iters"]}')
    assert same.sum() >= B - 1, (res['status'], ref['status'], res['num_iters'], ref['num_iters'])
    for b in np.where(same & (ref['status'] <= 1))[0]:
        assert rel(res['u'][b], ref['u'][b]) < 1e-5, (b, rel(res['u'][b], ref['u'][b]))


@pytest.mark.parametrize('M,N', [(3, 30), (5, 25), (2, 64), (4, 17)])
def test_xl_osqp_sizes_between_the_configs(oracle, M, N):
    """qp_method='osqp' on XL sizes the BASELINE configs do not name -- n = 180, 250, 256 (a multiple of the elimination's panel width) and
    136 (the smallest panel remainder, 8; four agents): the LDS placement of dgsqp_layout.h (ox_*) and the short last panels of the blocked
    elimination, three QPs each against oracle/osqp.hpp: same status, ADMM iterations, polish verdict, rho, active rows; x, lambda to 1e-6."""
    from concurrent.futures import ThreadPoolExecutor
    from dgsqp_amd.montecarlo import kinematic_racing_game, sample_scenarios
    from dgsqp_amd.solver import DGSQP, build_problem, build_params
    g = kinematic_racing_game('curve', N=N, M=M)
    P, par = build_problem(*g.solver_args()), build_params(g.params, qp_method='osqp')
    s = DGSQP(*g.solver_args(), print_method=None, qp_method='osqp')
    assert s.n == 2 * N * M and s.dims.layout == 2
    B = 3
    x0, u_tm = sample_scenarios(g, B, seed=4)
    u = s._to_agent_major(u_tm)
    l0 = np.array([oracle.dual_init(P, par, x0[b], u[b]) for b in range(B)])
    qp = s.qp_batch(x0, u, l0)

    def cpu(b):
        ev = oracle.evaluate(P, x0[b], u[b], l0[b], 1)
        return oracle.osqp(oracle.nearest_pd(ev['Q'], par.reg, par.eig_floor), ev['q'], ev['G'], ev['g'])
    with ThreadPoolExecutor(B) as ex:
        refs = list(ex.map(cpu, range(B)))
    for b, (xo, lo, io) in enumerate(refs):
        inf = qp['info'][b]
        assert (int(inf[0]), int(inf[1]), int(inf[2]), int(inf[5])) == (io['status'], io['iters'], io['polished'], io['n_active']), (b, inf[:6], io)
        assert abs(inf[3] - io['rho']) <= 1e-6 * io['rho']
        if io['status'] not in (-3, -4, -10, 3, 4):
            assert np.abs(qp['du'][b] - xo).max() < 1e-6 * max(1.0, np.abs(xo).max()) and np.abs(qp['lhat'][b] - lo).max() < 1e-6 * max(1.0, np.abs(lo).max())


@pytest.mark.parametrize('kind', ['merge6', 'kb_curve_N50'])
def test_xl_nearest_pd_with_many_negative_eigenvalues(oracle, kind):
    """_nearestPD (DGSQP.py:601-626) of the XL layout when MOST of the curvature is negative: with multipliers 100 x the usual size the
    game Hessian of the six-car merge has ~100 negative eigenvalues of 300, the N = 50 race ~90 of 200.  Until round 4 the tridiagonal
    path handled 64 and left the rest to one-sided Jacobi sweeps; now every count goes through the blocked Householder reduction,
    multisection, twisted factorisation, Gram-Schmidt and the rank-k correction on the matrix cores.  Checked against the oracle's
    Jacobi eigh: the projected Hessian to 1e-10 of |Q|, its smallest eigenvalue at the floor."""
    from dgsqp_amd.montecarlo import kinematic_racing_game, merge_game, sample_scenarios
    from dgsqp_amd.solver import DGSQP, build_problem, build_params
    g = merge_game(N=25, M=6) if kind == 'merge6' else kinematic_racing_game('curve', N=50)
    P, par = build_problem(*g.solver_args()), tight_lsqr(build_params(g.params))
    s = DGSQP(*g.solver_args(), print_method=None, lsqr_tol=1e-13)
    assert s.dims.layout == 2
    B = 3
    x0, u_tm = sample_scenarios(g, B, seed=3)
    rng = np.random.default_rng(5)
    u = agent_major(u_tm) + 0.05 * rng.standard_normal((B, s.n))
    l = 100.0 * np.maximum(0, rng.standard_normal((B, s.n_c_total)))
    qp = s.qp_batch(x0, u, l)
    for b in range(B):
        o = oracle.evaluate(P, x0[b], u[b], l[b], 1)
        w = np.linalg.eigvalsh(0.5 * (o['Q'] + o['Q'].T))
        assert (w < 0).sum() > 64, (b, (w < 0).sum())
        Qpd = oracle.nearest_pd(o['Q'], par.reg, par.eig_floor)
        assert np.abs(qp['Qpd'][b] - Qpd).max() < 1e-10 * np.abs(o['Q']).max(), (b, np.abs(qp['Qpd'][b] - Qpd).max(), np.abs(o['Q']).max())
        wd = np.linalg.eigvalsh(0.5 * (qp['Qpd'][b] + qp['Qpd'][b].T))
        assert wd.min() > -1e-9 * np.abs(w).max(), (b, wd.min())


def test_time_limit_status(oracle, games):
    """DGSQPParams.time_limit (DGSQP.py:470): checked at the end of every SQP iteration; with a limit of 0.1 microseconds every
    scenario that is not finished after its first iteration ends with 'time_limit' -- on the device and on the oracle alike."""
    import copy
    from dgsqp_amd.montecarlo import sample_scenarios
    from dgsqp_amd.solver import DGSQP, build_problem, build_params
    g = copy.deepcopy(games['kb_chicane_N15'][0])
    g.params.time_limit = 1e-7
    P, par = build_problem(*g.solver_args()), build_params(g.params)
    s = DGSQP(*g.solver_args(), print_method=None)
    x0, u_tm = sample_scenarios(g, 12, seed=3)
    res = s.solve_batch(x0, u_tm)
    ref = oracle.solve_batch(P, par, x0, agent_major(u_tm))
    assert np.array_equal(res['status'], ref['status']) and np.array_equal(res['num_iters'], ref['num_iters'])
    assert set(res['status']) <= {0, 4, 5} and (res['status'] == 5).sum() >= 10 and (res['num_iters'][res['status'] == 5] == 1).all()
    assert res['msg'][int(np.argmax(res['status'] == 5))] == 'time_limit'
    # time_limit = 0 is a limit (the reference times out after its first iteration), None is none
    g.params.time_limit = 0.0
    r0 = DGSQP(*g.solver_args(), print_method=None).solve_batch(x0, u_tm)
    assert np.array_equal(r0['status'], res['status'])
    # a limit of about one SQP iteration: scenarios cross it in the middle of their run, at different iterations.  The decision
    # is taken once per workgroup (one clock reading shared through LDS); every scenario must end in a valid state, the ones
    # that finish in time exactly as without a limit, and the kernel must come back.
    g.params.time_limit = None
    free = DGSQP(*g.solver_args(), print_method=None).solve_batch(x0, u_tm)
    x0b, u_b = sample_scenarios(g, 600, seed=4)
    free_b = DGSQP(*g.solver_args(), print_method=None).solve_batch(x0b, u_b)
    for lim in (2e-4, 1e-3, 4e-3):
        g.params.time_limit = lim
        rl = DGSQP(*g.solver_args(), print_method=None).solve_batch(x0b, u_b)
        timed_out = rl['status'] == 5
        assert set(rl['status']) <= {0, 1, 2, 3, 4, 5}
        assert (rl['num_iters'][timed_out] >= 1).all() and (rl['num_iters'][timed_out] <= free_b['num_iters'][timed_out] + 0).all()
        fin = ~timed_out
        assert np.array_equal(rl['status'][fin], free_b['status'][fin]) and np.array_equal(rl['num_iters'][fin], free_b['num_iters'][fin])
        assert np.array_equal(rl['u'][fin], free_b['u'][fin])
    assert np.array_equal(free['status'] == 5, np.zeros(12, bool))


def test_handles_of_different_games_do_not_mix_constants(games):
    """The kernels read the game from one per-device constant block: a launch of a different game waits for the launches in
    flight instead of overwriting their constants.  Two different games launched back to back (asynchronously) give exactly
    the results of separate solves."""
    import ctypes as C
    from dgsqp_amd import _ffi
    from dgsqp_amd.montecarlo import sample_scenarios
    from dgsqp_amd.solver import DGSQP
    ga, gb = games['kb_chicane_N15'][0], games['kb_curve_N10'][0]
    sa, sb = DGSQP(*ga.solver_args(), print_method=None), DGSQP(*gb.solver_args(), print_method=None)
    xa, ua = sample_scenarios(ga, 700, seed=21)
    xb, ub = sample_scenarios(gb, 700, seed=22)
    ref_a, ref_b = sa.solve_batch(xa, ua), sb.solve_batch(xb, ub)
    lib = sa._lib
    for s_, x_, u_ in ((sa, xa, ua), (sb, xb, ub)):
        assert lib.dgsqp_stage_inputs(s_._h, x_.shape[0], _ffi.dptr(np.ascontiguousarray(x_)), _ffi.dptr(np.ascontiguousarray(s_._to_agent_major(u_)))) == 0
    assert lib.dgsqp_launch_staged(sa._h) == 0 and lib.dgsqp_launch_staged(sb._h) == 0
    assert lib.dgsqp_wait(sb._h, None) == 0 and lib.dgsqp_wait(sa._h, None) == 0
    for s_, ref in ((sa, ref_a), (sb, ref_b)):
        B = ref['u'].shape[0]
        u = np.empty((B, s_.n)); st = np.empty(B, np.int32); it = np.empty(B, np.int32)
        assert lib.dgsqp_fetch_results(s_._h, _ffi.dptr(u), None, None, _ffi.iptr(st), _ffi.iptr(it), None, None, None) == 0
        assert np.array_equal(st, ref['status']) and np.array_equal(it, ref['num_iters']) and np.array_equal(u, ref['u'])


def test_solve_iter_data_records(games, oracle):
    """solve() with save_iter_data (DGSQP.py:386-451): one record per SQP iteration (the iteration that detects convergence
    included) with the optimality measures at its start and its QP solves; they add up to the solve's totals
    (what scripts/process_data_curve.py:50 sums)."""
    from dgsqp_amd.montecarlo import sample_scenarios
    from dgsqp_amd.solver import DGSQP
    g, P, par = games['kb_chicane_N15']
    par = tight_lsqr(par)
    s = DGSQP(*g.solver_args(), print_method=None, lsqr_tol=1e-13)
    x0, u_tm = sample_scenarios(g, 3, seed=5)
    for b in range(3):
        states = s.joint_dynamics.qu2state(None, x0[b], None)
        s.set_warm_start(u_tm[b])
        info = s.solve(states)
        batch = s.solve_batch(x0[b:b + 1], u_tm[b:b + 1])
        recs = info['iter_data']
        assert info['num_iters'] == int(batch['num_iters'][0]) and info['msg'] == batch['msg'][0]
        assert sum(r['qp_solves'] for r in recs) == int(batch['qp_solves'][0])
        assert len(recs) == info['num_iters'] + (1 if info['msg'] in ('conv_abs_tol', 'diverged', 'qp_fail') else 0)
        assert recs[-1]['cond'] == pytest.approx(info['cond'])
        assert all(set(r['cond']) == {'stat', 'p_feas', 'comp'} for r in recs)
        # per-iteration iterates (u_sol, l_sol of DGSQP.py:386,451) and the start (init, :328): the last record carries the final
        # iterates, record i the point whose optimality measures record i + 1 reports, init the warm start and the LSQR duals
        assert all(r['u_sol'] is not None and r['l_sol'] is not None for r in recs)
        assert np.array_equal(recs[-1]['u_sol'], batch['u'][0]) and np.array_equal(recs[-1]['l_sol'], batch['l'][0])
        assert np.array_equal(info['init']['u'], s._to_agent_major(u_tm[b:b + 1])[0])
        l0 = oracle.dual_init(P, par, x0[b], info['init']['u'])
        assert np.abs(info['init']['l'] - l0).max() < 1e-7
        if len(recs) >= 2:
            ev = oracle.evaluate(P, x0[b], recs[0]['u_sol'], recs[0]['l_sol'], 0)
            d = ev['q'] + ev['G'].T @ recs[0]['l_sol']
            assert np.abs(d).max() == pytest.approx(recs[1]['cond']['stat'], rel=1e-6)


def test_large_batch_equals_small_batches(games):
    """Every scenario of a 3,500-scenario launch (14 per workgroup, dynamic ticket order) comes out exactly as in a small
    launch: no state leaks between the scenarios a workgroup processes (warm-started active sets, trajectory tags)."""
    from dgsqp_amd.montecarlo import sample_scenarios
    from dgsqp_amd.solver import DGSQP
    g, P, par = games['kb_curve_N10']
    s = DGSQP(*g.solver_args(), print_method=None)
    x0, u = sample_scenarios(g, 3500, seed=13)
    big = s.solve_batch(x0, u)
    for lo in (0, 1024, 2900):
        small = s.solve_batch(x0[lo:lo + 600], u[lo:lo + 600])
        for key in ('status', 'num_iters', 'qp_solves', 'u', 'l', 'x', 'cond', 'cost'):
            assert np.array_equal(big[key][lo:lo + 600], small[key]), (key, lo)


def test_single_precision_boundary(games, solvers):
    """float arrays at the boundary (dgsqp_solve_batch_f32).  Inputs exactly representable in fp32 give the very
    same solve as the fp64 entry point -- identical status / iteration / QP counts -- and outputs that are the fp64 results rounded
    to fp32 (the arithmetic stays fp64; fp32 kernels are not built)."""
    from dgsqp_amd.montecarlo import sample_scenarios
    g = games['kb_chicane_N15'][0]
    s = solvers['kb_chicane_N15']
    x0, u_tm = sample_scenarios(g, 40, seed=8)
    x32, u32 = x0.astype(np.float32), u_tm.astype(np.float32)
    r64 = s.solve_batch(x32.astype(np.float64), u32.astype(np.float64))
    r32 = s.solve_batch(x32, u32, dtype=np.float32)
    assert r32['u'].dtype == np.float32 and r32['l'].dtype == np.float32 and r32['x'].dtype == np.float32
    for k in ('status', 'num_iters', 'qp_solves'):
        assert np.array_equal(r32[k], r64[k]), k
    for k in ('u', 'l', 'x', 'cond', 'cost'):
        assert np.array_equal(r32[k], r64[k].astype(np.float32)), k


def test_rccl_gather_single_rank(games):
    """The library-owned RCCL communicator (dgsqp_comm_init / dgsqp_gather_stats, include/dgsqp.h) on one rank: ncclAllGather of the
    88-byte records of the last solve, padding rows, barrier and max-reduction -- no PyTorch involved."""
    from dgsqp_amd.montecarlo import sample_scenarios
    from dgsqp_amd.sharding import Communicator, stats_from_records, pack_stats
    from dgsqp_amd.solver import DGSQP
    g = games['kb_curve_N10'][0]
    s = DGSQP(*g.solver_args(), print_method=None)
    x0, u = sample_scenarios(g, 37, seed=2)
    res = s.solve_batch(x0, u)
    comm = Communicator(s, 0, 1)
    try:
        comm.barrier()
        assert np.array_equal(comm.allreduce_max([1.5, -2.0]), [1.5, -2.0])
        rec = comm.gather_stats(40)
        assert rec.shape == (40,) and (rec['status'][37:] == -1).all() and (rec['rank'] == 0).all()
        assert np.array_equal(stats_from_records(rec), pack_stats(res))
        assert np.array_equal(rec['cost'][:37, :2], res['cost'])
    finally:
        comm.close()


def test_monte_carlo_example_script(tmp_path):
    """examples/monte_carlo_curve.py -- the DG-SQP leg of scripts/DGSQP_ALGAMES_monte_carlo_curve.py on the library: grouped launches
    plus a ragged remainder give the results of one plain solve_batch, and the pickle has the layout process_data_curve.py reads."""
    import pathlib
    import pickle
    import subprocess
    import sys
    from dgsqp_amd.montecarlo import kinematic_racing_game, sample_scenarios
    from dgsqp_amd.solver import DGSQP
    root = pathlib.Path(__file__).resolve().parent.parent
    out_file = tmp_path / 'data_curve.pkl'
    g = kinematic_racing_game('curve', N=12, reg=0.0)
    sv = DGSQP(*g.solver_args(), print_method=None)
    for sampler in ('host', 'device'):
        out = subprocess.run([sys.executable, str(root / 'examples' / 'monte_carlo_curve.py'), '--num-mc', '150', '--N', '12', '--batch', '64',
                              '--seed', '5', '--out', str(out_file)] + (['--host-sampler'] if sampler == 'host' else []),
                             capture_output=True, text=True, timeout=600, cwd=str(root))
        assert out.returncode == 0, out.stderr[-2000:]
        data = pickle.load(open(out_file, 'rb'))
        recs = data['sqgames']
        assert len(recs) == 150 and {'solve_info', 'params', 'init'} <= set(recs[0])
        if sampler == 'host':            # the scripts' sequential draws
            x0, u_ws = sample_scenarios(g, 150, seed=5)
        else:                            # batch j: the counter-based device sampler with seed + j, staged and solved without a host copy
            from dgsqp_amd import sampler as smp
            parts = [smp.sample_scenarios_counter(g, n, seed=5 + j)[:2] for j, n in enumerate((64, 64, 22))]
            x0, u_ws = np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])
        ref = sv.solve_batch(x0, u_ws)
        assert [r['solve_info']['num_iters'] for r in recs] == list(ref['num_iters']), sampler
        assert [r['solve_info']['status'] for r in recs] == list(ref['status'] <= 1), sampler
        if sampler == 'host':
            assert all(np.array_equal(r['solve_info']['iter_data'][0]['u_sol'], ref['u'][b]) for b, r in enumerate(recs))
        else:       # (the mirror's initial states equal the device's to rounding of sin / cos: 1e-12)
            assert max(np.abs(r['solve_info']['iter_data'][0]['u_sol'] - ref['u'][b]).max() for b, r in enumerate(recs)) < 1e-6


@pytest.mark.parametrize('name', ['kb_chicane_N15', 'dyn_curve_N15', 'kb_curve_N10'])
def test_two_workgroups_per_cu_build_matches_the_golden_fixtures(name, tmp_path):
    """Row N1: libdgsqp_hip_b256.so -- the same sources with 256-thread workgroups and half the LDS arena, TWO workgroups per CU -- on the
    golden fixtures of the n <= 60 games, in a process of its own (DGSQP_HIP_LIB): the launch really runs 256-thread blocks on a grid
    of two per CU, and the solves meet the same bar as the product build's (test_solve_matches_golden_fixtures: identical control flow on
    >= 95 % of the oracle-stable scenarios, iterates within 1e-5).  A game it cannot hold (XL layout) is refused loudly."""
    import json
    import pathlib
    import subprocess
    import sys
    root = pathlib.Path(__file__).resolve().parent.parent
    so = root / 'dgsqp_amd' / 'csrc' / 'libdgsqp_hip_b256.so'
    assert so.exists(), 'build it: python dgsqp_amd/csrc/build.py'
    code = f"""
import ctypes as C, json, sys, warnings
import numpy as np
warnings.simplefilter('ignore')
sys.path.insert(0, {str(root)!r})
from dgsqp_amd import _ffi, montecarlo as mc
from dgsqp_amd.solver import DGSQP
g = {{'kb_chicane_N15': lambda: mc.kinematic_racing_game('chicane', N=15), 'kb_curve_N10': lambda: mc.kinematic_racing_game('curve', N=10),
     'dyn_curve_N15': lambda: mc.dynamic_racing_game(N=15, rk4_substeps=4, game_def='curve')}}[{name!r}]()      # (the games of conftest.games)
s = DGSQP(*g.solver_args(), print_method=None, lsqr_tol=1e-13)
gold = np.load({str(GOLD / (name + '.npz'))!r})
res = s.solve_batch(gold['x0'], gold['u_ws'])
tm = _ffi.TimingT()
x0, u = mc.sample_scenarios(g, 2048, seed=3)
assert s._lib.dgsqp_stage_inputs(s._h, 2048, _ffi.dptr(np.ascontiguousarray(x0)), _ffi.dptr(np.ascontiguousarray(s._to_agent_major(u)))) == 0
assert s._lib.dgsqp_solve_staged(s._h, C.byref(tm)) == 0
info = C.create_string_buffer(256); s._lib.dgsqp_backend_info(info, 256)
cus = int(info.value.decode().split('CUs=')[1].split()[0])
try:
    DGSQP(*mc.kinematic_racing_game('curve', N=25, M=3).solver_args(), print_method=None)
    refused = ''
except Exception as e:
    refused = str(e)
np.savez({str(tmp_path / 'out.npz')!r}, **{{k: res[k] for k in ('u', 'l', 'status', 'num_iters', 'qp_solves', 'cost')}})
print(json.dumps(dict(block=int(tm.block), grid=int(tm.grid), cus=cus, lds=int(s.dims.lds_bytes), layout=int(s.dims.layout), refused=refused)))
"""
    import os
    run = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, DGSQP_HIP_LIB=str(so)), capture_output=True, text=True, timeout=900, cwd=str(root))
    assert run.returncode == 0, run.stderr[-2000:]
    d = json.loads(run.stdout.strip().splitlines()[-1])
    assert d['block'] == 256 and d['grid'] == 2 * d['cus'] and d['lds'] <= (163840 - 512) // 2 and d['layout'] == 0, d
    assert '(-4)' in d['refused'] and ('LDS' in d['refused'] or 'DG_BLOCK = 256' in d['refused']), d['refused']      # DGSQP_E_TOO_LARGE: half the arena, or a layout this build does not hold
    gold = np.load(GOLD / f'{name}.npz')
    res = dict(np.load(tmp_path / 'out.npz'))
    same = assert_control_flow_parity(res, gold, gold['stable'], name + ' (two workgroups per CU)', min_stable_same=0.95, max_conv_gap=0.05)
    assert same.mean() >= 0.85
    for b in np.where(same & (gold['status'] <= 1))[0]:
        assert rel(res['u'][b], gold['u'][b]) < 1e-5, b
        if gold['status'][b] == 0:
            assert rel(res['l'][b], gold['l'][b]) < 1e-5, b


def test_workgroups_per_cu_option_in_one_process(games):
    """DGSQP(..., workgroups_per_cu=2) runs that solver on libdgsqp_hip_b256.so while other solvers of the same process stay on the product
    build (two libraries, two sets of kernels and constants side by side): the same scenarios through both give the same control flow and
    the same iterates, the launches have 256- and 512-thread blocks, and a game the half-arena build cannot hold raises."""
    import ctypes as C
    from dgsqp_amd import _ffi, montecarlo as mc
    from dgsqp_amd.solver import DGSQP
    g = games['kb_chicane_N15'][0]
    s1 = DGSQP(*g.solver_args(), print_method=None)
    s2 = DGSQP(*g.solver_args(), print_method=None, workgroups_per_cu=2)
    assert s1._lib is not s2._lib and s2.dims.lds_bytes <= (163840 - 512) // 2
    x0, u = mc.sample_scenarios(g, 256, seed=11)
    r1, r2, r1b = s1.solve_batch(x0, u), s2.solve_batch(x0, u), s1.solve_batch(x0, u)
    same = (r1['status'] == r2['status']) & (r1['num_iters'] == r2['num_iters']) & (r1['qp_solves'] == r2['qp_solves'])
    conv = same & (r1['status'] <= 1)
    print(f'workgroups_per_cu 2 vs 1 in one process: identical (status, iterations, QPs) on {int(same.sum())}/256, iterates of the identical converged ones within '
          f'{max(rel(r2["u"][b], r1["u"][b]) for b in np.nonzero(conv)[0]):.1e}')
    assert same.sum() >= 250 and all(rel(r2['u'][b], r1['u'][b]) < 1e-5 for b in np.nonzero(conv)[0])
    assert all(np.array_equal(r1[k], r1b[k]) for k in ('u', 'l', 'status', 'num_iters'))          # the product build's results do not depend on what the other library did in between
    tm = _ffi.TimingT()
    for sv, block in ((s1, 512), (s2, 256)):
        assert sv._lib.dgsqp_stage_inputs(sv._h, 256, _ffi.dptr(np.ascontiguousarray(x0)), _ffi.dptr(np.ascontiguousarray(sv._to_agent_major(u)))) == 0
        assert sv._lib.dgsqp_solve_staged(sv._h, C.byref(tm)) == 0 and tm.block == block
    with pytest.raises(RuntimeError, match=r'\(-4\)'):
        DGSQP(*games['kb_chicane_N25'][0].solver_args(), print_method=None, workgroups_per_cu=2)


@pytest.mark.parametrize('script', ['chicane', 'comp', 'merge', 'agents', 'ablation'])
def test_monte_carlo_example_drivers(tmp_path, script):
    """examples/monte_carlo_{chicane,comp,merge,agents,ablation}.py -- the DG-SQP legs of scripts/DGSQP_ALGAMES_monte_carlo_chicane.py
    (:487-511), DGSQP_comp_monte_carlo.py (:488-506), DGSQP_merge_monte_carlo.py (:505-525), DGSQP_monte_carlo_agents.py (:323-342) and
    DGSQP_monte_carlo_ablation.py (:480-504) on the library: each writes the pickle(s) its scripts/process_data_*.py reads (file names,
    top-level keys, one ``dict(solve_info, params, init)`` per sample with the keys of DGSQP.py:495-502), and the records are those of a
    plain solve_batch of the same samples (grouped launches + a ragged remainder change nothing)."""
    import pathlib
    import pickle
    import subprocess
    import sys
    from dgsqp_amd import montecarlo as mc
    from dgsqp_amd.solver import DGSQP
    root = pathlib.Path(__file__).resolve().parent.parent
    out = tmp_path / 'out'
    argv = {'chicane': ['--num-mc', '40', '--batch', '16', '--N', '10', '--out', str(out / 'data_c_45_N_10.pkl')],
            'comp': ['--num-mc', '40', '--batch', '16', '--N', '8', '--out', str(out)],
            'merge': ['--num-mc', '40', '--batch', '16', '--N', '8', '--out', str(out)],
            'agents': ['--num-mc', '24', '--batch', '16', '--agents', '2', '3', '--N', '8', '--out', str(out)],
            'ablation': ['--num-mc', '24', '--batch', '16', '--N', '10', '--out', str(out)]}[script]
    run = subprocess.run([sys.executable, str(root / 'examples' / f'monte_carlo_{script}.py')] + argv, capture_output=True, text=True, timeout=900, cwd=str(root))
    assert run.returncode == 0, run.stderr[-2000:]
    load = lambda f: pickle.load(open(f, 'rb'))

    def check(recs, game, n, seed):
        assert len(recs) == n and all({'solve_info', 'params', 'init'} <= set(r) for r in recs)
        si = recs[0]['solve_info']
        assert {'time', 'num_iters', 'status', 'cost', 'cond', 'iter_data', 'msg', 'init'} <= set(si) and {'p_feas', 'comp', 'stat'} <= set(si['cond'])
        assert 'qp_solves' in si['iter_data'][0] and len(recs[0]['init']) == game.joint_model.n_a and hasattr(recs[0]['init'][0].x, 'x')
        x0, u_ws = mc.sample_scenarios(game, n, seed=seed)
        ref = DGSQP(*game.solver_args(), print_method=None).solve_batch(x0, u_ws)
        assert [r['solve_info']['num_iters'] for r in recs] == list(ref['num_iters'])
        assert [r['solve_info']['status'] for r in recs] == list(ref['status'] <= 1) and [r['solve_info']['msg'] for r in recs] == list(ref['msg'])
        assert all(np.array_equal(r['solve_info']['iter_data'][0]['u_sol'], ref['u'][b]) for b, r in enumerate(recs))
        assert abs(recs[0]['init'][0].x.x - x0[0, 0]) < 1e-15 and recs[0]['params'].N == game.params.N

    if script == 'chicane':
        d = load(out / 'data_c_45_N_10.pkl')
        assert {'dgsqp', 'algames', 'track', 'agent_dyn_configs', 'joint_model_config'} <= set(d)        # chicane.py:501-505
        check(d['dgsqp'], mc.kinematic_racing_game('chicane', N=10, reg=1e-3), 40, 1)
    elif script in ('comp', 'merge'):
        files = sorted(out.glob('sample_*.pkl'), key=lambda f: int(f.stem.split('_')[1]))
        assert [f.name for f in files] == [f'sample_{i + 1}.pkl' for i in range(40)]                        # comp.py:503, merge.py:521
        ds = [load(f) for f in files]
        assert all(('dgsqp' in d and 'env' in d) for d in ds) and (script == 'merge' or 'algames' in ds[0])
        check([d['dgsqp'] for d in ds], mc.barc_racing_game(N=8, M=2, reg=0.0) if script == 'comp' else mc.merge_game(N=8, reg=0.0, M=3), 40, 0 if script == 'comp' else 1)
    elif script == 'agents':
        for M in (2, 3):
            d = load(out / f'data_c_45_M_{M}_N_8.pkl')                                                      # agents.py:339
            assert {'sqgames', 'track', 'agent_dyn_configs', 'joint_model_config'} <= set(d) and len(d['agent_dyn_configs']) == M
            check(d['sqgames'], mc.kinematic_racing_game('curve', N=8, M=M, reg=1e-3), 24, 1)
    else:
        d = load(out / 'data_c_90_N_10.pkl')                                                                # ablation.py:501
        assert {'sqgames_all', 'sqgames_none', 'track', 'agent_dyn_configs', 'joint_model_config'} <= set(d)
        check(d['sqgames_all'], mc.ablation_racing_game(N=10, nonmono_ls=True, merit_function='stat_l1'), 24, 1)
        check(d['sqgames_none'], mc.ablation_racing_game(N=10, nonmono_ls=False, merit_function='stat'), 24, 1)
        assert d['sqgames_all'][0]['params'].nonmono_ls and not d['sqgames_none'][0]['params'].nonmono_ls
        assert abs(d['sqgames_all'][3]['init'][1].x.y - d['sqgames_none'][3]['init'][1].x.y) == 0.0         # the same samples for both solvers


def test_bench_line_contract():
    """bench.py prints ONE JSON line with the contract's keys, the roofline object and the CPU baseline (tiny batch): the full record
    (`--line full`, what the extra legs' child processes hand back) and the compact default line of less than 4 KB."""
    import json
    import pathlib
    import subprocess
    import sys
    root = pathlib.Path(__file__).resolve().parent.parent
    out = subprocess.run([sys.executable, str(root / 'bench.py'), '--workload', 'kb_curve_N25', '--batch', '64', '--steps', '3',
                          '--warmup', '1', '--cpu-sample', '2', '--pipeline', '2', '--line', 'full'], capture_output=True, text=True, timeout=600, cwd=str(root))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
                'dtype', 'data', 'config', 'roofline', 'cpu_baseline', 'value_single_launch', 'value_host_inclusive'):
        assert key in d, key
    assert 0 < d['value_host_inclusive'] and 0 < d['value_single_launch'] <= 1.5 * d['value']
    assert d['config']['distinct_batches'] >= 2 and d['roofline']['single_launch']['source'].endswith('one at a time')
    assert 'timed region' in d['roofline']['kernel_ms_source'] and d['roofline']['launches_timed'] >= 1
    assert abs(d['roofline']['solves_per_launch'] * d['roofline']['launches_timed'] - 64 * 3) < 1e-9 and d['config']['qp_method'] == 'active_set'
    assert d['unit'] == 'scenarios/s' and d['n_gpus'] == 1 and d['steps'] == 3 and d['scaling'] == 'weak' and d['dtype'] == 'f64'
    assert d['config']['workload'] == 'kb_curve_N25' and d['config']['batch_per_gpu'] == 64
    assert d['config']['launches_in_flight'] == 2 and d['config']['batches_per_launch'] >= 1      # (steps are issued in grouped launches)
    r = d['roofline']
    # the governing roof: fp64 vector ALU, ALGORITHMIC flops of SURVEY.md section 8(d) (terms in the line) over the HIP-event duration
    assert r['bound'] == 'valu_fp64' and r['peak'] == 78.6 and r['unit'] == 'TFLOP/s' and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-15 and r['kernel_ms'] > 0
    fm = r['flop_model']
    assert abs(fm['flop_per_solve'] - fm['qp_solves_per_scenario'] * sum(fm['per_qp_solve'].values())) < 1e-6 * fm['flop_per_solve']
    assert abs(r['achieved'] - fm['flop_per_solve'] * r['solves_per_launch'] / (r['kernel_ms'] * 1e-3) / 1e12) < 1e-9 * r['achieved']
    assert fm['per_qp_solve']['F_eig'] == 9.0 * 100 ** 3
    # ... and the HBM figure the contract names, secondary
    h = r['hbm']
    assert h['peak'] == 8000.0 and h['unit'] == 'GB/s' and abs(h['frac'] - h['achieved'] / h['peak']) < 1e-15
    assert abs(h['achieved'] - r['algorithmic_bytes_per_launch'] / (r['kernel_ms'] * 1e-3) / 1e9) < 1e-9 * h['achieved']
    assert 'workloads' not in d                      # the other configs ride only on the default invocation
    c = d['cpu_baseline']
    assert c['kind'] == 'port' and c['value'] > 0 and c['cores'] >= 1 and c['value_wall'] > 0 and c['seconds_per_scenario']['max'] >= c['seconds_per_scenario']['mean'] > 0
    assert abs(d['value'] - 64 * 3 / (d['ms_per_step'] * 3e-3)) < 1e-6 * d['value']
    # the default (compact) line of the same workload: everything the contract names, in less than 4 KB, and nothing else on stdout
    out = subprocess.run([sys.executable, str(root / 'bench.py'), '--workload', 'kb_curve_N25', '--batch', '64', '--steps', '3',
                          '--warmup', '1', '--cpu-sample', '2', '--pipeline', '2', '--single-steps', '1', '--host-steps', '1'], capture_output=True, text=True, timeout=600, cwd=str(root))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and len(lines[0]) < 4096
    c = json.loads(lines[0])
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
                'dtype', 'data', 'config', 'roofline', 'cpu_baseline', 'value_single_launch', 'value_host_inclusive', 'value_host_inclusive_grouped'):
        assert key in c, key
    assert {'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'kernel', 'kernel_ms', 'hbm'} <= set(c['roofline']) and abs(c['roofline']['frac'] - c['roofline']['achieved'] / 78.6) < 1e-4 * c['roofline']['frac']
    assert {'value', 'unit', 'cores', 'kind', 'sample'} <= set(c['cpu_baseline']) and c['cpu_baseline']['kind'] == 'port' and c['steps'] == 3 and c['config']['workload'] == 'kb_curve_N25'
    assert 'workloads' not in c and abs(c['value'] - 64 * 3 / (c['ms_per_step'] * 3e-3)) < 1e-6 * c['value']
