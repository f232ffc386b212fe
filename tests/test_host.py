"""Host logic, C-ABI surface and sharding (no GPU needed)."""
import ctypes
import os
import pathlib
import re
import subprocess
import time
import sys

import numpy as np
import pytest

from conftest import agent_major

ROOT = pathlib.Path(__file__).resolve().parent.parent


def test_problem_lowering_and_dims(oracle, games):
    from dgsqp_amd.solver import problem_dims
    for name, (n_q, n_u, n, n_c) in {'kb_chicane_N15': (12, 4, 60, 315), 'kb_chicane_N25': (12, 4, 100, 525),
                                     'dyn_curve_N15': (16, 4, 60, 315)}.items():
        g, P, par = games[name]
        assert problem_dims(P) == (n_q, n_u, n, n_c)
        d = oracle.dims(P)
        assert (d['nq'], d['nu'], d['n'], d['nc']) == (n_q, n_u, n, n_c)
    g, P, par = games['kb_chicane_N25']
    assert P.agents[0].has_rate == 1 and P.agents[0].rate_ub[1] == pytest.approx(np.pi)
    assert P.agents[0].st_ub[5] == 1.0 and np.isinf(P.agents[0].st_ub[0]) and P.agents[0].in_lb[0] == -2.1
    assert (par.beta, par.tau, par.reg, par.nonmono_ls, par.sqp_iters, par.rel_tol_req) == (0.01, 0.5, 1e-3, 1, 50, 3)


def test_unsupported_options_raise():
    from dgsqp_amd.solver import build_params
    from dgsqp_amd.solver_types import DGSQPParams
    with pytest.raises(NotImplementedError):
        build_params(DGSQPParams(conv_approx=False))
    assert build_params(DGSQPParams(hessian_approximation='bfgs')).hessian_bfgs == 1
    with pytest.raises(ValueError):
        build_params(DGSQPParams(hessian_approximation='sr1'))
    with pytest.raises(ValueError):
        build_params(DGSQPParams(merit_function='nope'))


def test_sampler_is_seeded_and_collision_free(games):
    from dgsqp_amd.montecarlo import sample_scenarios
    g, _, _ = games['kb_chicane_N15']
    x0, u = sample_scenarios(g, 16, seed=1)
    x0b, ub = sample_scenarios(g, 16, seed=1)
    assert np.array_equal(x0, x0b) and np.array_equal(u, ub)
    assert x0.shape == (16, 12) and u.shape == (16, 15, 4)
    assert (np.linalg.norm(x0[:, :2] - x0[:, 6:8], axis=1) >= g.obs_d).all()
    assert (np.abs(u[:, :, [0, 2]]) <= 2.1 + 1e-12).all() and (np.abs(u[:, :, [1, 3]]) <= 0.436 + 1e-12).all()
    assert (np.abs(x0[:, [5, 11]]) <= g.half_width).all()


def test_pid_against_vectors_of_the_reference_controller(games):
    """tests/golden/pid_ref.npz is produced by IMPORTING the reference's DGSQP/solvers/PID.py (tools/make_pid_golden.py).
    (a) dgsqp_amd.pid.PIDLaneFollower reproduces its outputs on open-loop random sequences (saturations, anti-windup,
    set_x_ref(0) of the steering loop) bit for bit; (b) the vectorised numpy warm start -- the checker of the HIP kernel --
    reproduces the closed-loop rollouts of the reference controller (chicane.py:411-447) for both vehicle models."""
    from dgsqp_amd import montecarlo as mc
    from dgsqp_amd.pid import PIDLaneFollower
    from dgsqp_amd.solver_types import PIDParams
    from dgsqp_amd.types import BodyLinearVelocity, ParametricPose, VehicleActuation, VehicleState
    gold = np.load(pathlib.Path(__file__).parent / 'golden' / 'pid_ref.npz')
    sat_abs = sat_rel = 0
    for seq, want, du in zip(gold['open_in'], gold['open_out'], gold['open_du']):
        ctl = PIDLaneFollower(0.1, PIDParams(dt=0.1, Kp=1.0, Ki=0.005, x_ref=seq[0, 1], u_max=0.436, u_min=-0.436, du_max=du[1], du_min=-du[1]),
                              PIDParams(dt=0.1, Kp=1.0, x_ref=seq[0, 0], u_max=2.1, u_min=-2.1, du_max=du[0], du_min=-du[0]))
        for k in range(len(seq)):
            st = VehicleState(p=ParametricPose(x_tran=seq[k, 1], e_psi=seq[k, 2]), v=BodyLinearVelocity(v_long=seq[k, 0]), u=VehicleActuation())
            ctl.step(st)
            assert (st.u.u_a, st.u.u_steer) == (want[k, 0], want[k, 1]), k
        sat_abs += int(np.abs(want[:, 1]).max() == 0.436)
        sat_rel += int(np.isclose(np.abs(np.diff(want[:, 1])).max(), du[1]) or np.isclose(np.abs(np.diff(want[:, 0])).max(), du[0]))
    assert sat_abs >= 2 and sat_rel >= 2           # the vectors do exercise both saturations
    for tag, game in (('kb', mc.kinematic_racing_game('chicane', N=25)), ('dyn', mc.dynamic_racing_game(N=25))):
        x0, u_ws, q_ws = gold[f'{tag}_x0'], gold[f'{tag}_u_ws'], gold[f'{tag}_q_ws']
        for a, m in enumerate(game.joint_model.dynamics_models):
            q, u = mc.pid_warm_start(m, x0[:, a * m.n_q:(a + 1) * m.n_q], 25, 0.1, du=tuple(gold[f'{tag}_du']))
            assert np.abs(u - u_ws[:, a]).max() < 1e-12 and np.abs(q - q_ws[:, a]).max() < 1e-12, tag


def test_pid_matches_batched_warm_start(games):
    """The scalar PID mirror (reference PID.py) and the vectorised sampler produce the same first inputs."""
    from dgsqp_amd.montecarlo import pid_warm_start
    from dgsqp_amd.pid import PIDLaneFollower
    from dgsqp_amd.solver_types import PIDParams
    from dgsqp_amd.types import VehicleState
    g, _, _ = games['kb_chicane_N15']
    mdl = g.joint_model.dynamics_models[0]
    q0 = np.array([[0.5, 0.3, 2.4, 0.0, 0.5, 0.3]])
    _, u = pid_warm_start(mdl, q0, 3, 0.1, du=(10.0, np.pi))
    st = VehicleState(t=0.0)
    mdl.q2state(st, q0[0])
    pid = PIDLaneFollower(0.1, PIDParams(dt=0.1, Kp=1.0, Ki=0.005, x_ref=0.3, u_max=0.436, u_min=-0.436, du_max=np.pi, du_min=-np.pi),
                          PIDParams(dt=0.1, Kp=1.0, x_ref=2.4, u_max=2.1, u_min=-2.1, du_max=10.0, du_min=-10.0))
    for k in range(3):
        pid.step(st)
        assert [st.u.u_a, st.u.u_steer] == pytest.approx(list(u[0, k]), abs=1e-9)
        mdl.step(st)                      # adaptive RK45 plant of the reference vs fixed-step RK4 of the sampler
    assert st.p.s > 0.5


def test_library_exports_every_declared_symbol():
    """Every function declared in include/dgsqp.h is exported by the built library (loads without a GPU)."""
    from dgsqp_amd import _ffi
    from dgsqp_amd.csrc.build import build
    so = build()
    lib = ctypes.CDLL(str(so))
    header = (ROOT / 'include' / 'dgsqp.h').read_text()
    declared = set(re.findall(r'\b(dgsqp_[a-z0-9_]+)\s*\(', header))
    assert declared == set(_ffi.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name


def test_half_arena_build_exports_and_plans():
    """libdgsqp_hip_b256.so (-DDG_BLOCK=256: two 256-thread workgroups per CU, row N1) is built next to the product, exports the same C-ABI,
    and plans its games against HALF the LDS arena: the n = 60 games stay LDS-resident, configs[1] (n = 100) only fits as the big layout with
    the packed gradients in the scratch as well (measured 0.5-0.86 x the product build: profiles/r06_n1_two_per_cu.txt), an n = 100 game with
    525 rows does not fit at all (own process each: the library is chosen at load time, DGSQP_HIP_LIB)."""
    from dgsqp_amd import _ffi
    from dgsqp_amd.csrc.build import OUT_B256, build
    build()
    assert OUT_B256.exists()
    lib = ctypes.CDLL(str(OUT_B256))
    for name in _ffi.EXPORTED_SYMBOLS:
        assert hasattr(lib, name), name
    code = ("import sys, warnings; warnings.simplefilter('ignore'); sys.path.insert(0, %r)\n"
            "import bench\nfrom dgsqp_amd.solver import plan, build_problem, build_params\n"
            "for w in ('kb_chicane_N15', 'dyn_curve_N15', 'dyn_curve_N25', 'kb_chicane_N25'):\n"
            "    g = bench.make_game(w)\n"
            "    try:\n        d = plan(build_problem(*g.solver_args()), build_params(g.params, qp_method='active_set')); print(w, d['lds_bytes'], d['layout'])\n"
            "    except ValueError as e:\n        print(w, 'refused', 'LDS' in str(e))\n") % str(ROOT)
    out = {}
    for tag, so in (('b512', ROOT / 'dgsqp_amd' / 'csrc' / 'libdgsqp_hip.so'), ('b256', OUT_B256)):
        r = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, DGSQP_HIP_LIB=str(so)), capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr[-1500:]
        out[tag] = {ln.split()[0]: ln.split()[1:] for ln in r.stdout.splitlines() if ln.strip()}
    half = (163840 - 512) // 2
    assert all(int(out['b512'][w][0]) <= 163840 - 512 and out['b512'][w][1] == '0' for w in out['b512'])          # the product: all four LDS-resident
    assert int(out['b256']['kb_chicane_N15'][0]) <= half and out['b256']['kb_chicane_N15'][1] == '0'
    assert int(out['b256']['dyn_curve_N15'][0]) <= half and out['b256']['dyn_curve_N15'][1] == '0'
    assert int(out['b256']['dyn_curve_N25'][0]) <= half and out['b256']['dyn_curve_N25'][1] == '1'                  # big layout (+ gradients in the scratch)
    assert out['b256']['kb_chicane_N25'] == ['refused', 'True']


def test_struct_layouts_match_the_header():
    """ctypes mirrors and the C structs agree in size (checked by compiling a tiny C program)."""
    from dgsqp_amd import _ffi
    src = '#include <stdio.h>\n#include "dgsqp.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(dgsqp_agent_t), sizeof(dgsqp_problem_t), sizeof(dgsqp_params_t), sizeof(dgsqp_dims_t), sizeof(dgsqp_timing_t), sizeof(dgsqp_pid_t), sizeof(dgsqp_sampler_t));return 0;}\n'
    exe = pathlib.Path('/tmp/dgsqp_sizeof')
    subprocess.run(['gcc', '-x', 'c', '-', '-I', str(ROOT / 'include'), '-o', str(exe)], input=src.encode(), check=True)
    sizes = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    from dgsqp_amd.sampler import SamplerT
    assert sizes == [ctypes.sizeof(_ffi.AgentT), ctypes.sizeof(_ffi.ProblemT), ctypes.sizeof(_ffi.ParamsT),
                     ctypes.sizeof(_ffi.DimsT), ctypes.sizeof(_ffi.TimingT), ctypes.sizeof(_ffi.PidT), ctypes.sizeof(SamplerT)]


def _has_gpu():
    from dgsqp_amd import _ffi
    buf = ctypes.create_string_buffer(256)
    try:
        return _ffi.load_library().dgsqp_backend_info(buf, 256) == 0
    except Exception:
        return False


@pytest.mark.skipif(_has_gpu(), reason='checks the behaviour WITHOUT a GPU')
def test_solver_fails_loudly_without_gpu(games):
    """No CPU fallback: constructing the solver without a HIP device is an error, not a silent slow path."""
    from dgsqp_amd.solver import DGSQP
    g, _, _ = games['kb_chicane_N15']
    with pytest.raises(RuntimeError, match='dgsqp_create failed'):
        DGSQP(*g.solver_args(), print_method=None)


def test_missing_library_is_an_error(monkeypatch, tmp_path):
    from dgsqp_amd import _ffi
    monkeypatch.setattr(_ffi, '_LIB', None)
    monkeypatch.setattr(_ffi, '_LIBS', {})
    monkeypatch.setenv('DGSQP_HIP_LIB', str(tmp_path / 'nope.so'))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        _ffi.load_library()
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        _ffi.load_library(2)
    with pytest.raises(ValueError):
        _ffi.load_library(3)
    monkeypatch.delenv('DGSQP_HIP_LIB')
    assert _ffi.library_path(1).name == 'libdgsqp_hip.so' and _ffi.library_path(2).name == 'libdgsqp_hip_b256.so'      # (workgroups per CU -> build)


def test_product_never_imports_the_oracle():
    for f in (ROOT / 'dgsqp_amd').rglob('*'):
        if f.suffix in ('.py', '.h', '.hip'):
            txt = f.read_text()
            for pat in ('import oracle', 'from oracle', 'liboracle', 'oracle.py', 'dgsqp_oracle', 'oracle/'):
                assert pat not in txt, (f, pat)


def test_shard_ranges_partition_the_batch():
    from dgsqp_amd.sharding import shard_range
    for B in (0, 1, 7, 1024, 1025):
        for w in (1, 2, 3, 8):
            spans = [shard_range(B, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


def test_philox_known_answers_and_counter_based_sampler():
    """The device sampler's random numbers (csrc/dgsqp_sampler.h) are Philox4x32-10; its numpy restatement reproduces the
    published known-answer vectors of the generator (Random123 kat_vectors: zero / all-ones / digits-of-pi counters and keys).
    The sampler built on it is a pure function of (seed, candidate): the accepted scenarios do not depend on how the candidates are
    chunked, obey the placement rules of the scripts, and every one of them is collision-free along its warm start."""
    from dgsqp_amd import sampler as smp
    from dgsqp_amd.montecarlo import kinematic_racing_game, merge_game, barc_racing_game
    hexs = lambda ctr, key: [int(v) for v in smp.philox4x32_10(np.array([ctr], np.uint32), key)[0]]
    assert hexs([0, 0, 0, 0], (0, 0)) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert hexs([0xffffffff] * 4, (0xffffffff, 0xffffffff)) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert hexs([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], (0xa4093822, 0x299f31d0)) == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    u = smp.uniform(7, np.arange(20000), 3)
    assert u.min() >= 0.0 and u.max() < 1.0 and abs(u.mean() - 0.5) < 0.01 and abs(u.var() - 1 / 12) < 0.005
    assert not np.array_equal(u, smp.uniform(8, np.arange(20000), 3)) and not np.array_equal(u, smp.uniform(7, np.arange(20000), 2))
    g = kinematic_racing_game('curve', N=10)
    x0a, ua, ca = smp.sample_scenarios_counter(g, 40, seed=3, chunk=64)
    x0b, ub, cb = smp.sample_scenarios_counter(g, 40, seed=3, chunk=500)
    assert ca == cb and np.array_equal(x0a, x0b) and np.array_equal(ua, ub)
    assert x0a.shape == (40, 12) and ua.shape == (40, 10, 4) and ca >= 40
    assert np.all(x0a[:, 4] >= 0.1) and np.all(np.abs(x0a[:, [5, 11]]) <= g.half_width) and np.all((x0a[:, [2, 8]] >= 2) & (x0a[:, [2, 8]] < 3))
    d = np.hypot(x0a[:, 10] - x0a[:, 4], x0a[:, 11] - x0a[:, 5])
    assert np.allclose(d, 1.2 * g.obs_d, atol=1e-12)                       # car 2 sits 1.2 obstacle distances from car 1 (chicane.py:396-399)
    assert np.all(np.linalg.norm(x0a[:, :2] - x0a[:, 6:8], axis=1) >= g.obs_d)
    gm = merge_game(N=6)
    xm, um, cm = smp.sample_scenarios_counter(gm, 6, seed=1, chunk=16)
    assert xm.shape == (6, 12) and not um.any() and np.all(np.abs(xm[:, [2, 6, 10]] - 0.3) <= 0.3 * 0.03 + 1e-12)
    gc = barc_racing_game(N=6, M=3)
    xc, uc, cc = smp.sample_scenarios_counter(gc, 6, seed=0, chunk=16)
    assert xc.shape == (6, 18) and np.all(np.abs(xc[:, [5, 11, 17]]) <= gc.half_width + 1e-12) and np.all(np.abs(xc[:, [3, 9, 15]]) <= 5 * np.pi / 180 + 1e-12)


def test_interleaved_shards_partition_the_batch():
    """interleaved assignment (rank, rank + world, ...) next to the contiguous blocks; both partition the batch,
    sizes differ by at most one, and ``unshard`` restores the scenario order."""
    from dgsqp_amd.sharding import padded_shard_size, shard_indices, unshard
    for mode in ('contiguous', 'interleaved'):
        for B in (0, 1, 7, 1024, 1025):
            for w in (1, 2, 3, 8):
                parts = [shard_indices(B, r, w, mode) for r in range(w)]
                assert sorted(np.concatenate(parts).tolist()) == list(range(B))
                assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1 and max(len(p) for p in parts) == (padded_shard_size(B, w) if B else 0)
                vals = np.arange(B) * 10.0
                assert np.array_equal(unshard([vals[p] for p in parts], B, w, mode), vals)
    assert shard_indices(10, 1, 4, 'interleaved').tolist() == [1, 5, 9]
    with pytest.raises(ValueError):
        shard_indices(4, 0, 2, 'striped')


_RANK_STUB = r'''
import os, sys, time
sys.path.insert(0, sys.argv[1])
from dgsqp_amd.sharding import exchange_unique_id, rendezvous_path
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
assert os.environ['LOCAL_RANK'] == os.environ['RANK'] and os.environ['MASTER_ADDR'] == '127.0.0.1' and int(os.environ['MASTER_PORT']) > 0
mode = sys.argv[2]
# two rendezvous in a row (communicator re-built in the same launch): the second must not pick up the first one's file
for seq in (0, 1):
    uid = exchange_unique_id(rank, world, lambda: bytes([seq + 1]) * 128, seq=seq, timeout=30.0)
    assert uid == bytes([seq + 1]) * 128, (rank, seq, uid[:4])
if mode == 'fail' and rank == 1:
    sys.exit(7)
if mode == 'fail':
    time.sleep(60)            # a rank stuck in a collective whose peer died: the launcher must take it down
if rank == 0:
    print('{"stub": "ok", "world": %d}' % world, flush=True)
'''


def test_spawn_ranks_environment_rendezvous_and_exit_codes(tmp_path):
    """`bench.py --gpus N` without a launcher: every rank gets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*, a rendezvous file of its own
    (a stale file of an earlier launch under the same name is ignored: wrong launch tag), rank 0's line is relayed, the worst exit
    code is returned, and a dead rank takes the stuck ones down.  Stub ranks: no GPU needed."""
    import importlib.util
    stub = tmp_path / 'rank_stub.py'
    stub.write_text(_RANK_STUB)
    driver = tmp_path / 'driver.py'
    driver.write_text(f'''
import sys
sys.path.insert(0, {str(ROOT)!r})
import bench
bench.spawn_ranks(int(sys.argv[1]), [{str(ROOT)!r}, sys.argv[2]], script={str(stub)!r}, timeout=120.0)
''')
    env = dict(os.environ, TMPDIR=str(tmp_path))
    for world in (3, 8):          # 8: the node the scaling run uses (one rank per GPU)
        ok = subprocess.run([sys.executable, str(driver), str(world), 'ok'], env=env, capture_output=True, text=True, timeout=300)
        assert ok.returncode == 0 and f'"stub": "ok", "world": {world}' in ok.stdout, ok.stdout + ok.stderr
        assert not list(tmp_path.glob('dgsqp_rccl_*.id')), 'rendezvous file left behind'
    t0 = time.time()
    bad = subprocess.run([sys.executable, str(driver), '2', 'fail'], env=env, capture_output=True, text=True, timeout=300)
    assert bad.returncode == 7 and time.time() - t0 < 45, (bad.returncode, bad.stderr[-500:])


def test_stat_record_carries_every_agents_cost():
    """The record of the ONE collective (dgsqp_stat_record_t, 88 bytes) has a cost slot for every agent the library supports: the
    6-car merge (BASELINE configs[4]) keeps all six costs through pack -> pad -> gather -> unpack; struct and numpy dtype agree."""
    import ctypes
    from dgsqp_amd import _ffi
    from dgsqp_amd.sharding import RECORD_DTYPE, costs_from_records, pad_records, records_from_results, stats_from_records
    assert ctypes.sizeof(_ffi.StatRecordT) == RECORD_DTYPE.itemsize == 88 and RECORD_DTYPE['cost'].shape == (_ffi.MAX_AGENTS,) == (6,)
    for f in _ffi.StatRecordT._fields_:
        assert getattr(_ffi.StatRecordT, f[0]).offset == RECORD_DTYPE.fields[f[0]][1], f[0]
    rng = np.random.default_rng(0)
    B, M = 5, 6
    res = dict(status=np.array([0, 1, 2, 4, 0], np.int32), num_iters=np.arange(B, dtype=np.int32), qp_solves=np.arange(B, dtype=np.int32) + 1,
               cond=rng.random((B, 3)), cost=rng.standard_normal((B, M)))
    world = np.concatenate([pad_records(records_from_results({k: v[:3] for k, v in res.items()}, rank=0), 3),
                            pad_records(records_from_results({k: v[3:] for k, v in res.items()}, rank=1), 3)])
    assert len(world) == 6 and (world['status'] == -1).sum() == 1
    assert np.array_equal(costs_from_records(world, M), res['cost'])
    assert np.array_equal(stats_from_records(world)[:, 0], res['status'])
    with pytest.raises(ValueError):
        costs_from_records(world, 7)


def test_rendezvous_ignores_a_stale_file(tmp_path):
    """A file left by a crashed run under an explicit DGSQP_RENDEZVOUS carries another launch's tag: readers wait for THIS launch's."""
    from dgsqp_amd import sharding
    path = str(tmp_path / 'uid')
    os.environ['DGSQP_LAUNCH_TAG'] = 'old-launch'
    try:
        assert sharding.exchange_unique_id(0, 2, lambda: b'a' * 128, path=path) == b'a' * 128
        os.environ['DGSQP_LAUNCH_TAG'] = 'new-launch'
        with pytest.raises(TimeoutError):
            sharding.exchange_unique_id(1, 2, None, path=path, timeout=0.3)
        assert sharding.exchange_unique_id(0, 2, lambda: b'b' * 128, path=path) == b'b' * 128
        assert sharding.exchange_unique_id(1, 2, None, path=path, timeout=5.0) == b'b' * 128
    finally:
        os.environ.pop('DGSQP_LAUNCH_TAG', None)


_GLOO_WORKER = r'''
import os, sys, numpy as np
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist            # test-only transport: gloo stands in for the library's ncclAllGather
from dgsqp_amd.sharding import (RECORD_DTYPE, exchange_unique_id, pad_records, padded_shard_size, records_from_results, shard_range,
                                stats_from_records, pack_stats, summarize)
from oracle import oracle
from dgsqp_amd.montecarlo import kinematic_racing_game, sample_scenarios
from dgsqp_amd.solver import build_problem, build_params
dist.init_process_group('gloo')
rank, world = dist.get_rank(), dist.get_world_size()
# the rendezvous the RCCL communicator uses: rank 0 publishes 128 bytes in a file, the others poll for it
uid = exchange_unique_id(rank, world, lambda: bytes(range(128)), path=sys.argv[2])
assert uid == bytes(range(128))
game = kinematic_racing_game('curve', N=10)
P, par = build_problem(*game.solver_args()), build_params(game.params)
B = 5
x0, u = sample_scenarios(game, B, seed=4)
lo, hi = shard_range(B, rank, world)
B_pad = padded_shard_size(B, world)
u_am = np.concatenate([u[:, :, 2 * a:2 * a + 2].reshape(B, -1) for a in range(2)], axis=1)
res = oracle.solve_batch(P, par, x0[lo:hi], u_am[lo:hi])     # the CPU oracle stands in for the GPU solve here
payload = pad_records(records_from_results(res, rank), B_pad)    # equal-count payload, exactly what dgsqp_gather_stats sends
buf = torch.from_numpy(payload.view(np.uint8).copy())
out = [torch.zeros_like(buf) for _ in range(world)]
dist.all_gather(out, buf)
rec = np.concatenate([o.numpy() for o in out]).view(RECORD_DTYPE)
allstats = stats_from_records(rec)
if rank == 0:
    full = oracle.solve_batch(P, par, x0, u_am)
    assert rec.shape == (world * B_pad,) and (rec['status'] == -1).sum() == world * B_pad - B
    assert allstats.shape == (B, 6), allstats.shape
    assert np.array_equal(allstats, pack_stats(full))
    assert np.array_equal(rec['rank'][rec['status'] >= 0], [0, 0, 0, 1, 1])
    s = summarize(allstats)
    assert s['n'] == B and 0 <= s['converged'] <= 1
    print('GLOO_OK', s['converged'])
dist.barrier()
dist.destroy_process_group()
'''


def test_two_process_gloo_shard_and_gather(tmp_path):
    """world_size=2: contiguous sharding (uneven: 3 + 2), the file rendezvous of the ncclUniqueId, and ONE equal-count all-gather
    of the padded 88-byte records == the single-process result.  gloo is the transport in this CPU test only; on the GPU the
    same payload goes through the library's ncclAllGather (tests/test_gpu.py::test_rccl_gather_single_rank)."""
    script = tmp_path / 'worker.py'
    script.write_text(_GLOO_WORKER)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29533')
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
                          '--master-addr', '127.0.0.1', '--master-port', '29533', str(script), str(ROOT), str(tmp_path / 'uid')],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert 'GLOO_OK' in out.stdout


def test_product_and_bench_do_not_use_torch():
    """north_star: Python host code over a thin ctypes C-ABI, no PyTorch.  (torch.distributed.run may still be the LAUNCHER.)"""
    import re
    for f in [ROOT / 'bench.py'] + sorted((ROOT / 'dgsqp_amd').glob('*.py')):
        src = f.read_text()
        assert not re.search(r'^\s*(import torch|from torch)', src, re.M), f


def test_bench_refuses_a_world_size_mismatch():
    env = dict(os.environ, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    out = subprocess.run([sys.executable, str(ROOT / 'bench.py'), '--gpus', '2'], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 2 and 'WORLD_SIZE=1' in out.stderr


def test_multi_agent_sampler():
    """M-agent rejection sampler (scripts/DGSQP_monte_carlo_agents.py:262-308): shapes, bounds, no collision along the
    PID warm start."""
    from dgsqp_amd.montecarlo import kinematic_racing_game, sample_scenarios
    g = kinematic_racing_game('curve', N=8, M=3)
    x0, u = sample_scenarios(g, 12, seed=2)
    assert x0.shape == (12, 18) and u.shape == (12, 8, 6)
    assert np.all(np.abs(u[..., 0::2]) <= 2.1 + 1e-12) and np.all(np.abs(u[..., 1::2]) <= 0.436 + 1e-12)
    p = x0.reshape(12, 3, 6)[:, :, :2]
    for i in range(3):
        for j in range(i + 1, 3):
            assert np.all(np.linalg.norm(p[:, i] - p[:, j], axis=1) >= 0.4 - 1e-12)


def test_eig_floor_rule_and_oracle_projection(oracle):
    """_nearestPD floor: the host passes the reference's literal 1e-10 (DGSQP.py:1293) by default, at reg = 0 too; a larger
    floor and the active-bound snap are explicit opt-ins; the oracle's projection honours the parameter."""
    from dgsqp_amd.solver import build_params
    from dgsqp_amd.solver_types import DGSQPParams
    assert build_params(DGSQPParams(reg=1e-3)).eig_floor == 1e-10
    assert build_params(DGSQPParams(reg=0.0)).eig_floor == 1e-10 and build_params(DGSQPParams(reg=0.0)).snap_active_bounds == 0
    assert build_params(DGSQPParams(reg=0.0), eig_floor=1e-6, snap_active_bounds=True).eig_floor == pytest.approx(1e-6)
    assert build_params(DGSQPParams(reg=0.0), snap_active_bounds=True).snap_active_bounds == 1
    rng = np.random.default_rng(0)
    Q = rng.standard_normal((12, 12))
    w_ref = np.linalg.eigvalsh(0.5 * (Q + Q.T))
    for floor in (1e-10, 1e-6):
        w = np.linalg.eigvalsh(oracle.nearest_pd(Q, 0.0, floor))
        k = int((w_ref < 0).sum())
        assert np.allclose(w[:k], floor, rtol=1e-3, atol=1e-13) and np.allclose(w[k:], w_ref[k:], atol=1e-12)


def test_barc_circuit_game_and_sampler():
    """scripts/DGSQP_comp_monte_carlo.py: L_track_barc circuit, reg = 0, sampler anywhere on the circuit (s wraps)."""
    from dgsqp_amd.montecarlo import barc_racing_game, sample_scenarios
    from dgsqp_amd.solver import build_problem, problem_dims
    g = barc_racing_game(N=15, M=2)
    assert g.params.reg == 0.0 and g.track.track_length == pytest.approx(17.461, abs=1e-2)
    P = build_problem(*g.solver_args())
    assert problem_dims(P) == (12, 4, 60, 315) and P.n_segs == 9
    x0, u = sample_scenarios(g, 24, seed=0)
    assert x0.shape == (24, 12) and u.shape == (24, 15, 4)
    assert np.all(np.abs(x0[:, [5, 11]]) <= g.half_width + 1e-12) and np.all(np.abs(x0[:, [3, 9]]) <= 5 * np.pi / 180 + 1e-12)
    assert np.all(np.linalg.norm(x0[:, :2] - x0[:, 6:8], axis=1) >= 0.4)
    g3 = barc_racing_game(N=15, M=3)
    assert problem_dims(build_problem(*g3.solver_args())) == (18, 6, 90, 495)     # 24 / 33 / 9 rows per stage


def test_result_records_match_the_post_processing_readers(tmp_path):
    """scripts/process_data_curve.py:44-53 reads solve_info['status'|'msg'|'cond']['p_feas'|'num_iters'|'time'|'iter_data'];
    the records written from a solve_batch result carry exactly those (checked by replaying the reader's loop)."""
    import pickle
    from dgsqp_amd.results import save_monte_carlo, summarize_like_process_data
    from dgsqp_amd.solver_types import DGSQPParams
    from dgsqp_amd import _ffi
    status = np.array([0, 1, 2, 4, 0], np.int32)
    res = dict(status=status, msg=[_ffi.STATUS_MSG[s] for s in status], num_iters=np.array([7, 12, 50, 3, 9], np.int32),
               qp_solves=np.array([7, 15, 80, 4, 9], np.int32), cond=np.arange(15.0).reshape(5, 3) * 1e-4,
               cost=np.ones((5, 2)), u=np.zeros((5, 4)), l=np.zeros((5, 6)))
    path = tmp_path / 'data_c_45_N_25.pkl'
    save_monte_carlo(path, res, DGSQPParams(N=25), wall_time=0.5)
    data = pickle.load(open(path, 'rb'))
    recs = data['sqgames']
    n_conv, n_max, n_div, iters, solves = 0, 0, 0, [], []
    for r in recs:                                   # the reader's loop
        si = r['solve_info']
        if si['status']:
            n_conv += 1
            solves.append(np.sum([d['qp_solves'] for d in si['iter_data']]))
            iters.append(si['num_iters'])
            assert si['time'] == pytest.approx(0.1) and si['cond']['p_feas'] >= 0
        if si['msg'] == 'max_it':
            n_max += 1
        elif si['msg'] in ('diverged', 'qp_fail'):
            n_div += 1
    assert (n_conv, n_max, n_div) == (3, 1, 1) and np.mean(iters) == pytest.approx(28 / 3) and np.mean(solves) == pytest.approx(31 / 3)
    s = summarize_like_process_data(recs)
    assert (s['converged'], s['max_it'], s['failed']) == (3, 1, 1) and s['avg_solves'] == pytest.approx(31 / 3)


def test_layout_plan_of_the_supported_games():
    """dgsqp_plan (host only): which layout dgsqp_create picks and what it needs.  LDS-resident up to n ~ 100, big (P and
    reflectors in L2) up to n = 128, XL (generic kernels; beyond n ~ 160 also the packed gradients in L2) up to n = 256 as long as
    the remaining vectors fit; the 160 KB arena is never exceeded."""
    from dgsqp_amd.montecarlo import barc_racing_game, dynamic_racing_game, kinematic_racing_game, merge_game
    from dgsqp_amd.solver import build_params, build_problem, plan, problem_dims
    cases = [(kinematic_racing_game('chicane', N=25), 0), (dynamic_racing_game(N=25), 0), (kinematic_racing_game('curve', N=10), 0),
             (kinematic_racing_game('curve', N=5, M=1), 0), (barc_racing_game(N=15, M=3), 0), (kinematic_racing_game('curve', N=30), 1),
             (merge_game(N=20), 1), (merge_game(N=10, M=6), 1), (barc_racing_game(N=21, M=3), 1), (kinematic_racing_game('curve', N=16, M=4), 1),
             (kinematic_racing_game('curve', N=25, M=3), 2), (kinematic_racing_game('curve', N=40), 2),
             (kinematic_racing_game('curve', N=50), 2), (kinematic_racing_game('curve', N=20, M=4), 2),      # packed gradients in L2
             (merge_game(N=25, M=6), 2), (kinematic_racing_game('curve', N=24, M=4), 2)]      # BASELINE configs[4] (n = 300, 1,587 rows): tables in the constant block
    for g, layout in cases:
        P, par = build_problem(*g.solver_args()), build_params(g.params)
        d = plan(P, par)
        assert d['layout'] == layout, (g.name, d)
        assert (d['n_q'], d['n_u'], d['n'], d['n_c']) == problem_dims(P)
        assert 0 < d['lds_bytes'] <= 163840 and d['workspace_bytes'] < 8 << 20
    assert plan(build_problem(*merge_game(N=25, M=6).solver_args()), build_params(merge_game(N=25, M=6).params))['n'] == 300
    for g in (merge_game(N=27, M=6), kinematic_racing_game('curve', N=41, M=4)):      # n = 324, 328 > 320
        with pytest.raises(ValueError, match='not supported yet|LDS'):
            plan(build_problem(*g.solver_args()), build_params(g.params))


def test_blocked_elimination_scheme_matches_the_column_form():
    """The panel scheme of the XL layout's elimination (csrc/dgsqp_xl.h: xl_eliminate_blocked -- 16 pivots per pass, rank-16 update of the
    tiles below, garbage above the diagonal) restated in numpy against the column-by-column form it replaces: sizes with a short last
    panel, a full one, and a single panel.  (The device kernel itself is held to the oracle by the -m gpu tests.)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location('xl_proto', ROOT / 'tools' / 'xl_blocked_elimination_proto.py')
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    rng = np.random.default_rng(3)
    for n in (12, 44, 48):
        A = rng.standard_normal((n, n))
        M = A @ A.T + n * np.eye(n)
        a, b = m.column_form(M), m.blocked(M, rng)
        assert np.abs(np.tril(a) - np.tril(b)).max() < 1e-11 * np.abs(a).max(), n


IN_SCOPE_MODULES = ('DGSQP.solvers.DGSQP', 'DGSQP.solvers.DGSQP_v2', 'DGSQP.solvers.PID', 'DGSQP.solvers.solver_types', 'DGSQP.types',
                    'DGSQP.dynamics.dynamics_models', 'DGSQP.dynamics.model_types', 'DGSQP.tracks.track_lib')
OUT_OF_SCOPE_NAMES = {'IBRParams', 'ALGAMESParams', 'CALTVMPCParams', 'PATHMCPParams',                     # parameters of solvers SURVEY section 2 leaves out
                      'CasadiKinematicBicycleProgressAugmented', 'CasadiDynamicBicycleProgressAugmented', 'CasadiDynamicCLBicycle',
                      'CasadiKinematicUnicycleCombined', 'load_tum_raceline', 'load_mpclab_raceline'}


def test_reference_module_paths_resolve_to_this_package():
    """the scripts' import block (scripts/DGSQP_ALGAMES_monte_carlo_curve.py:5-15, minus IBR / ALGAMES /
    casadi, which are out of scope) works unchanged against this repo, and the names are this package's.  Run in a fresh
    interpreter: other tests import the REFERENCE's DGSQP package from /root/reference under the same name."""
    block = '''
from DGSQP.solvers.DGSQP import DGSQP
from DGSQP.solvers.PID import PIDLaneFollower
from DGSQP.solvers.solver_types import DGSQPParams, PIDParams
from DGSQP.types import VehicleState, VehicleActuation, Position, ParametricPose, OrientationEuler, BodyLinearVelocity, BodyAngularVelocity
from DGSQP.dynamics.dynamics_models import CasadiKinematicBicycleCombined, CasadiDecoupledMultiAgentDynamicsModel
from DGSQP.dynamics.model_types import KinematicBicycleConfig, MultiAgentModelConfig
from DGSQP.tracks.track_lib import *
import dgsqp_amd, dgsqp_amd.solver, dgsqp_amd.tracks, dgsqp_amd.pid
assert DGSQP is dgsqp_amd.solver.DGSQP and PIDLaneFollower is dgsqp_amd.pid.PIDLaneFollower
assert DGSQPParams is dgsqp_amd.DGSQPParams and VehicleState is dgsqp_amd.VehicleState
assert CurveTrack is dgsqp_amd.tracks.CurveTrack and get_track is dgsqp_amd.tracks.get_track
from DGSQP.solvers.DGSQP_v2 import DGSQP as DGSQPv2
from DGSQP.solvers.solver_types import DGSQPV2Params
import dgsqp_amd.solver_v2
assert DGSQPv2 is dgsqp_amd.solver_v2.DGSQP
# what the curve script builds before it defines its CasADi cost functions (curve.py:140-172)
track = CurveTrack(enter_straight_length=1, curve_length=8, curve_swept_angle=3.141592653589793 / 4, exit_straight_length=5, width=2.0, slack=0.8)
cfg = KinematicBicycleConfig(dt=0.1, model_name='kinematic_bicycle_cl', noise=False, discretization_method='euler', code_gen=False)
car = CasadiKinematicBicycleCombined(0.0, cfg, track=track)
joint = CasadiDecoupledMultiAgentDynamicsModel(0.0, [car, car], MultiAgentModelConfig(dt=0.1, code_gen=False))
assert (joint.n_q, joint.n_u) == (12, 4)
p = DGSQPParams(dt=0.1, N=25, nonmono_ls=True, reg=0.0, beta=0.01)
print('ok')
'''
    out = subprocess.run([sys.executable, '-c', block], cwd='/tmp', env=dict(os.environ, PYTHONPATH=str(ROOT)), capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip().endswith('ok'), out.stderr[-2000:]


@pytest.mark.skipif(not pathlib.Path('/root/reference/scripts').exists(), reason='reference tree only exists in the build container')
def test_every_in_scope_import_of_the_reference_scripts_resolves():
    """Every ``from DGSQP.<in-scope module> import a, b, c`` line of the reference's scripts resolves against the alias package,
    except the names that belong to components SURVEY.md section 2 leaves out (listed above, each must really be absent)."""
    pat = re.compile(r'^from (DGSQP[\w.]*) import (.+)$')
    wanted = {}
    for f in sorted(pathlib.Path('/root/reference/scripts').rglob('*.py')):
        for line in f.read_text().splitlines():
            m = pat.match(line.strip())
            if m and m.group(1) in IN_SCOPE_MODULES and m.group(2).strip() != '*':
                for nm in m.group(2).split(','):
                    wanted.setdefault(m.group(1), set()).add(nm.split(' as ')[0].strip())
    assert len(wanted) >= 7
    code = 'import importlib, json, sys\nw = json.loads(sys.argv[1])\nmissing = [m + "." + n for m in w for n in w[m] if not hasattr(importlib.import_module(m), n)]\nprint(json.dumps(missing))'
    import json
    out = subprocess.run([sys.executable, '-c', code, json.dumps({k: sorted(v) for k, v in wanted.items()})], cwd='/tmp',
                         env=dict(os.environ, PYTHONPATH=str(ROOT)), capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    missing = {m.rsplit('.', 1)[1] for m in json.loads(out.stdout)}
    assert missing <= OUT_OF_SCOPE_NAMES, missing - OUT_OF_SCOPE_NAMES


def test_qp_solver_is_validated_whatever_qp_method_says():
    """DGSQP.py:183-201 hands params.qp_solver to ca.conic: a solver that is not restated must not be accepted silently because
    the caller also named a qp_method; the default mapping of 'osqp' to the exact QP is announced once."""
    import warnings
    from dgsqp_amd import solver as sv
    from dgsqp_amd.solver_types import DGSQPParams
    for qm in (None, 'active_set', 'osqp'):
        with pytest.raises(ValueError):
            sv.resolve_qp_method(DGSQPParams(qp_solver='superscs'), qm)
    assert sv.resolve_qp_method(DGSQPParams(qp_solver='qrqp'), 'osqp') == sv.QP_METHODS['osqp']
    sv._WARNED_OSQP_DEFAULT = False
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        assert sv.resolve_qp_method(DGSQPParams(), None) == sv.QP_METHODS['active_set']
        assert sv.resolve_qp_method(DGSQPParams(), None) == sv.QP_METHODS['active_set']
    assert len(w) == 1 and "qp_method='osqp'" in str(w[0].message)


def test_vectorised_merge_sampler_equals_the_scalar_one():
    """scripts/DGSQP_merge_monte_carlo.py:421-480 restated per scenario (``_sample_scenarios_merge_scalar``: sequential draws, the
    script's car-3 quirk) and its vectorised form used by ``sample_scenarios``: the same accepted scenarios, bit for bit."""
    import dgsqp_amd.montecarlo as mc
    for M, N in ((3, 20), (6, 25)):
        g = mc.merge_game(N=N, M=M)
        a, ua = mc._sample_scenarios_merge(g, 150, 1)
        b, ub = mc._sample_scenarios_merge_scalar(g, 150, 1)
        assert np.array_equal(a, b) and np.array_equal(ua, ub) and a.shape == (150, 4 * M)


def test_bench_flop_model_fingerprint_and_extra_legs(tmp_path, games):
    """bench.py pieces that need no GPU: SURVEY section 8(d)'s flop formula evaluated on configs[1]'s dimensions (the terms the JSON line
    carries), the source fingerprint that guards stale PMC summaries (comments and white space do not count, statements do), and the extra
    workloads of the default invocation (every one names a known workload; the configs' batch sizes are BASELINE.json's)."""
    import json
    import shutil
    import types
    sys.path.insert(0, str(ROOT))
    import bench
    g, P, par = games['dyn_curve_N25']
    d = types.SimpleNamespace(M=2, N=25, n_q=16, n_u=4, n=100, n_dense=75)
    fm = bench.algorithmic_flops_per_solve(d, P, 10.0, 'active_set')
    t = fm['per_qp_solve']
    assert t['F_eig'] == 9e6 and t['F_eval'] == 2 * (25 * 6 * 16 ** 3 + 2 * 16 * 4 * 625 * 16) + 25 * (2 * 3000.0 * 3.0 * 40)      # rk4, M = 10: 40 f_c per step
    assert t['F_qp'] == 2 * 100 ** 2 * 25 + 100 ** 3 / 3 and fm['flop_per_solve'] == 10.0 * sum(t.values())
    assert bench.algorithmic_flops_per_solve(d, P, 10.0, 'osqp')['per_qp_solve']['F_qp'] > t['F_qp']
    # OSQP: the ADMM term is priced with the MEASURED mean iteration count when the run counted it (dgsqp_osqp_counters), 250 otherwise
    fo, fm_ = bench.algorithmic_flops_per_solve(d, P, 10.0, 'osqp', 3400.0), bench.algorithmic_flops_per_solve(d, P, 10.0, 'osqp')
    assert fo['admm_iterations_per_qp'] == 3400.0 and 'counted' in fo['admm_iterations_source'] and fm_['admm_iterations_per_qp'] == 250 and fm_['admm_iterations_source'] == 'assumed'
    assert abs((fo['per_qp_solve']['F_qp'] - t['F_qp']) / (fm_['per_qp_solve']['F_qp'] - t['F_qp']) - 3400.0 / 250.0) < 1e-12 and fm['admm_iterations_per_qp'] is None
    # fingerprint
    root = tmp_path / 'copy'
    (root / 'dgsqp_amd').mkdir(parents=True)
    shutil.copytree(ROOT / 'dgsqp_amd' / 'csrc', root / 'dgsqp_amd' / 'csrc', ignore=shutil.ignore_patterns('*.so', '__pycache__'))
    shutil.copytree(ROOT / 'include', root / 'include')
    assert bench.source_fingerprint(str(root)) == bench.source_fingerprint()
    f = root / 'dgsqp_amd' / 'csrc' / 'dgsqp_qp.h'
    f.write_text('// a new comment\n' + f.read_text().replace('\n', '\n   ', 3))
    assert bench.source_fingerprint(str(root)) == bench.source_fingerprint()
    f.write_text(f.read_text() + '\nstatic int dg_extra_statement = 1;\n')
    assert bench.source_fingerprint(str(root)) != bench.source_fingerprint()
    # extra legs
    names = json.loads((ROOT / 'BASELINE.json').read_text())['configs']
    assert len(names) == 5
    legs = {leg['tag']: leg for leg in bench.EXTRA_LEGS}
    assert all(leg['workload'] in bench.WORKLOADS for leg in bench.EXTRA_LEGS)
    assert (legs['configs[2] B=4096']['batch'], legs['configs[3] B=16384']['batch'], legs['configs[4] B=65536']['batch']) == (4096, 16384, 65536)
    assert 'batch=4096' in names[2] and 'batch=16384' in names[3] and 'batch=65536' in names[4]


def test_bench_extra_legs_respect_the_wall_budget(monkeypatch, capsys, tmp_path):
    """bench.py's default invocation: ONE stdout line of less than 4 KB (the driver keeps an 8 KB tail: round 5's 30 KB line came back
    unparsed) with the contract's keys, roofline, cpu_baseline and a four-column summary per extra leg; the legs run as CHILD PROCESSES
    with a timeout each (run_leg) after the headline is safe in the side file; a leg that fails or times out is reported in the line, legs
    that would start after --extras-budget seconds are named as skipped -- none takes the headline down (no GPU needed: run_workload and
    run_leg are replaced by stubs that return records as large as the real ones)."""
    import json
    import subprocess
    sys.path.insert(0, str(ROOT))
    import bench
    calls, legs_run = [], []
    big = {'flop_model': {'per_qp_solve': {'F_eval': 1.0, 'F_eig': 2.0, 'F_qp': 3.0}, 'pad': 'x' * 1500}, 'single_launch': {'pad': 'y' * 800}}

    def record(value):
        return {'metric': 'Monte-Carlo scenarios/sec (SQP solves/sec), 2-agent N=25', 'value': value, 'unit': 'scenarios/s', 'n_gpus': 1, 'steps': 20, 'warmup': 5,
                'ms_per_step': 1.234567891234, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
                'config': {'workload': 'dyn_curve_N25', 'description': 'd' * 300, 'batch_per_gpu': 1024, 'batch_total': 1024, 'n': 100, 'n_c': 325, 'parallelism': 'scenario-sharded x1',
                           'layout': 'lds', 'qp_method': 'active_set', 'reg': 1e-3, 'distinct_batches': 20, 'batches_per_launch': 20, 'launches_in_flight': 5,
                           'cooperative_line_search': 'auto', 'sampler': 's' * 200},
                'roofline': dict(big, bound='valu_fp64', achieved=3.9123456789, peak=78.6, unit='TFLOP/s', frac=0.0497753, frac_executed_upper_bound=0.17, traffic=1.2e12,
                                 traffic_note=None, kernel='dg_solve_kernel', kernel_ms=1743.9123456, launches_timed=1, solves_per_launch=20480.0,
                                 hbm={'achieved': 0.09, 'peak': 8000.0, 'unit': 'GB/s', 'frac': 1.1e-5}),
                'cpu_baseline': {'value': 25.3, 'unit': 'scenarios/s', 'cores': 64, 'kind': 'port', 'sample': 'z' * 400, 'sample_short': 'first 512 scenarios', 'value_wall': 11.6, 'value_one_core': 1.8},
                'value_single_launch': 1858.0, 'value_host_inclusive': 1674.0, 'value_host_inclusive_grouped': 11608.0, 'mean_iters': 5.6, 'mean_iters_all': 6.5,
                'mean_qp_solves': 10.5, 'converged_fraction': 0.94, 'status_fractions': {'conv_abs_tol': 0.9}, 'elapsed_s': 1.74, 'elapsed_s_per_rank': [1.74]}

    def stub(args, rank, local_rank, world):
        calls.append((args.workload, args.qp, args.batch))
        return record(11757.123456789)

    def leg_stub(leg, args, timeout):
        legs_run.append((leg['tag'], timeout))
        assert json.load(open(bench.SIDECAR))['headline']['value'] == 11757.123456789      # the headline is on disk before any leg starts
        if leg['workload'] == 'kb_f1_N50' and leg.get('qp') is None:
            raise RuntimeError('stub failure')
        if leg['workload'] == 'merge6_N25' and leg.get('qp') == 'osqp':
            raise subprocess.TimeoutExpired(['bench.py'], timeout)
        return record(float(len(legs_run)))
    monkeypatch.setattr(bench, 'run_workload', stub)
    monkeypatch.setattr(bench, 'run_leg', leg_stub)
    monkeypatch.setattr(bench, 'SIDECAR', str(tmp_path / 'bench_workloads.json'))
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '1', '--steps', '20', '--warmup', '5'])
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        monkeypatch.delenv(k, raising=False)
    bench.main()
    out = [ln for ln in capsys.readouterr().out.splitlines() if ln.strip()]
    assert len(out) == 1 and len(out[0]) < 4096             # the contract: one JSON line, and one the driver's tail can hold
    line = json.loads(out[0])
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config',
                'roofline', 'cpu_baseline', 'value_single_launch', 'value_host_inclusive', 'value_host_inclusive_grouped', 'value_qp_osqp', 'workloads'):
        assert key in line, key
    assert line['value'] == 11757.123456789 and line['config']['workload'] == 'dyn_curve_N25' and 'description' not in line['config']
    assert {'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'kernel', 'kernel_ms', 'hbm', 'frac_executed_upper_bound'} <= set(line['roofline']) and 'flop_model' not in line['roofline']
    assert {'value', 'unit', 'cores', 'kind', 'sample'} <= set(line['cpu_baseline']) and len(line['cpu_baseline']['sample']) < 200
    tags = [w[0] for w in line['workloads']]
    assert tags == ['configs[1] (the headline)'] + [leg['tag'] for leg in bench.EXTRA_LEGS]
    assert calls == [('dyn_curve_N25', 'active_set', 1024)] and [t for t, _ in legs_run] == [leg['tag'] for leg in bench.EXTRA_LEGS]
    assert all(0 < to <= leg['timeout'] for (_, to), leg in zip(legs_run, bench.EXTRA_LEGS))          # every leg has its own timeout
    assert line['value_qp_osqp'] == 1.0                     # (the first leg: --qp osqp on configs[1])
    failed = {w[0]: w[2] for w in line['workloads'] if w[1] is None}
    assert 'stub failure' in failed['configs[3] B=16384'] and 'timeout' in failed['configs[4] --qp osqp, reduced batch B=1024'] and len(failed) == 2
    side = json.load(open(bench.SIDECAR))                   # the full records: headline with its flop model, one record per leg
    assert side['headline']['roofline']['flop_model']['pad'] and len(side['workloads']) == 1 + len(bench.EXTRA_LEGS)
    assert sum('error' in w for w in side['workloads']) == 2 and line['sidecar'].endswith('bench_workloads.json')
    # past the budget: every leg is named, none is started
    calls.clear(), legs_run.clear()
    monkeypatch.setattr(bench, 'T_START', bench.T_START - 1e6)
    bench.main()
    out = [ln for ln in capsys.readouterr().out.splitlines() if ln.strip()]
    line = json.loads(out[-1])
    assert len(out) == 1 and len(out[0]) < 4096
    assert len(calls) == 1 and not legs_run and all(w[1] is None and 'not started' in w[2] for w in line['workloads'][1:]) and line['value'] == 11757.123456789
    monkeypatch.setattr(bench, 'T_START', bench.T_START + 1e6)
    # a non-default invocation times only what it was asked for; `--line full` (what a leg's child prints) is the whole record
    calls.clear()
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--qp', 'osqp'])
    bench.main()
    out = [ln for ln in capsys.readouterr().out.splitlines() if ln.strip()]
    line = json.loads(out[-1])
    assert len(calls) == 1 and 'workloads' not in line and not legs_run and len(out) == 1 and len(out[0]) < 4096
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--workload', 'merge6_N25', '--extras', 'off', '--line', 'full'])
    bench.main()
    line = json.loads(capsys.readouterr().out.splitlines()[-1])
    assert line['roofline']['flop_model']['pad'] and 'status_fractions' in line


def test_bench_compact_line_of_a_multi_rank_run():
    """What rank 0 prints when N > 1 (the driver's SCALE runs): the same compact line -- contract keys, `cpu_baseline` null with its note (timed at
    N = 1 only), every rank's own elapsed time -- under 4 KB also at 8 ranks, no `workloads` (the extra legs ride on the 1-GPU default run)."""
    import json
    sys.path.insert(0, str(ROOT))
    import bench
    full = {'metric': 'Monte-Carlo scenarios/sec (SQP solves/sec), 2-agent N=25', 'value': 90000.123456, 'unit': 'scenarios/s', 'n_gpus': 8, 'steps': 20, 'warmup': 5, 'ms_per_step': 91.0,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'dyn_curve_N25', 'description': 'd' * 300, 'batch_per_gpu': 1024, 'batch_total': 8192, 'n': 100, 'n_c': 325, 'parallelism': 'scenario-sharded x8', 'layout': 'lds',
                       'qp_method': 'active_set', 'reg': 1e-3, 'distinct_batches': 20, 'batches_per_launch': 20, 'launches_in_flight': 5, 'cooperative_line_search': 'auto', 'shard_mode': 'x' * 100},
            'roofline': {'bound': 'valu_fp64', 'achieved': 30.0, 'peak': 78.6, 'unit': 'TFLOP/s', 'frac': 0.38, 'frac_executed_upper_bound': None, 'traffic': None, 'traffic_note': 'n' * 300, 'kernel': 'dg_solve_kernel',
                         'kernel_ms': 1800.0, 'launches_timed': 1, 'solves_per_launch': 20480.0, 'hbm': {'achieved': 0.7, 'peak': 8000.0, 'unit': 'GB/s', 'frac': 9e-5}, 'flop_model': {'pad': 'z' * 2000}},
            'cpu_baseline': None, 'cpu_baseline_note': 'timed on rank 0 of the 1-GPU run only (bench.py --gpus 1)', 'value_single_launch': 15000.0, 'value_host_inclusive': 13000.0,
            'value_host_inclusive_grouped': 88000.0, 'mean_iters': 5.6, 'mean_iters_all': 6.5, 'mean_qp_solves': 10.5, 'converged_fraction': 0.94, 'elapsed_s': 1.82,
            'elapsed_s_per_rank': [1.80, 1.81, 1.79, 1.82, 1.80, 1.78, 1.81, 1.80]}
    out = bench.compact_line(full)
    text = json.dumps(out, separators=(',', ':'))
    assert len(text) < 4096 and out['n_gpus'] == 8 and out['cpu_baseline'] is None
    assert out['cpu_baseline_note'].startswith('timed on rank 0') and len(out['elapsed_s_per_rank']) == 8 and 'workloads' not in out and 'value_qp_osqp' not in out
    assert out['config']['parallelism'] == 'scenario-sharded x8' and 'flop_model' not in out['roofline'] and len(out['roofline']['traffic_note']) <= 120
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config', 'roofline'):
        assert key in out, key


def test_bench_leg_child_process_is_killed_at_its_timeout(monkeypatch, tmp_path):
    """run_leg starts `bench.py` as a child (never an exec) and a child that hangs costs its own record: subprocess.TimeoutExpired after the
    leg's timeout, the child gone; a child that fails raises with its stderr (stand-in scripts, no GPU)."""
    import subprocess
    import types
    sys.path.insert(0, str(ROOT))
    import bench
    args = types.SimpleNamespace(steps=1, warmup=0, batch=8, pipeline=1, coop='auto')
    hang = tmp_path / 'hang.py'
    hang.write_text('import sys, time\nopen(sys.argv[0] + ".args", "w").write(" ".join(sys.argv[1:]))\ntime.sleep(600)\n')
    monkeypatch.setattr(bench, '__file__', str(hang))
    t0 = time.time()
    with pytest.raises(subprocess.TimeoutExpired):
        bench.run_leg(dict(tag='t', workload='merge6_N25', qp='osqp', batch=64, mixed_precision=True), args, 1.5)
    assert time.time() - t0 < 30
    sent = (tmp_path / 'hang.py.args').read_text()
    assert '--workload merge6_N25' in sent and '--qp osqp' in sent and '--batch 64' in sent and '--extras off' in sent and '--line full' in sent and '--mixed-precision' in sent
    assert '--cpu-sample 0' in sent and '--single-steps 0' in sent and '--host-steps 0' in sent
    bad = tmp_path / 'bad.py'
    bad.write_text('import sys\nsys.stderr.write("boom")\nsys.exit(3)\n')
    monkeypatch.setattr(bench, '__file__', str(bad))
    with pytest.raises(RuntimeError, match='exit code 3.*boom'):
        bench.run_leg(dict(tag='t', workload='merge6_N25'), args, 30)
    good = tmp_path / 'good.py'
    good.write_text('print("noise")\nprint(\'{"value": 7.0}\')\n')
    monkeypatch.setattr(bench, '__file__', str(good))
    assert bench.run_leg(dict(tag='t', workload='merge6_N25'), args, 30) == {'value': 7.0}
