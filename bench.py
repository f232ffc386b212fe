#!/usr/bin/env python3
"""Headline benchmark: Monte-Carlo scenarios/s (= DG-SQP solves/s) of the 2-agent N=25 game (BASELINE.json configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W]                   # N > 1 without a launcher: spawns its own N ranks
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one pass of the hot path (DGSQP.solve(), reference DGSQP/solvers/DGSQP.py:302-507) over one batch of synthetic
random-initial-condition scenarios that is resident in HBM before the timed region starts (dgsqp_stage_inputs); consecutive
steps solve DIFFERENT batches (own seed each, (pipeline + 1) x group distinct ones, cycled) and are issued `--group` at a time: ONE
launch solves the staged batches of a group from a shared ticket queue (dgsqp_launch_staged_group; own buffers per batch, results
bit-identical to separate launches), `--pipeline` launches in flight.  The timed region is exactly K steps between two fences
(library stream synchronisation + RCCL barrier), max over ranks.  Scenarios shard over the ranks with no data-path collective
(--scaling weak: fixed batch per GPU; strong: fixed total batch); the only exchange is ONE ncclAllGather of the 88-byte
per-scenario record, issued by the HIP library (dgsqp_gather_stats).  No PyTorch: the launcher only provides RANK / LOCAL_RANK /
WORLD_SIZE.  Rank 0 prints ONE JSON line; next to `value` (K pipelined steps) it carries `value_single_launch` (strictly one
launch at a time -- the only figure in which a launch has the GPU to itself, and where roofline.kernel_ms comes from) and
`value_host_inclusive` (dgsqp_solve_batch from host buffers: H2D + solve + D2H + gather).
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')    # one hardware queue per in-flight batch (HIP default: 4; 32 fails on this stack)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # BASELINE.json configs[1]: the reference's own dynamic-bicycle game (comparison_study_barc/exact_dynamic_game_dynamic.py) on the curve track
    'dyn_curve_N25': dict(desc='2-agent dynamic-bicycle (Pacejka, rk4 M=10) curve track, N=25, game of exact_dynamic_game_dynamic.py (cost_setting 0), fp64', kind='dyn', track='curve', N=25, reg=1e-3),
    # round 1's synthetic variant of it (costs / rate rows of curve.py on the Pacejka vehicle): a third of the scenarios diverge numerically
    'dyn_curve_N25_stress': dict(desc='2-agent dynamic-bicycle curve track, N=25, curve.py costs and rate rows (round-1 definition), fp64', kind='dyn', track='curve', N=25, reg=1e-3, game_def='curve'),
    # DG-SQP v2 on the dynamic-bicycle game, as the reference itself pairs them (comparison_study_barc: exact_dgsqp.py + globals.py), L_track_barc circuit
    'dyn_barc_N25_v2': dict(desc='2-agent dynamic-bicycle (Pacejka, rk4 M=10) race on the L_track_barc circuit, N=25, DG-SQP v2 with the parameters of comparison_study_barc/globals.py, fp64', kind='dyn', track='barc', N=25, reg=None, solver='v2'),
    'dyn_curve_N25_v2': dict(desc='2-agent dynamic-bicycle curve track, N=25, DG-SQP v2 (comparison_study_barc/globals.py parameters), fp64', kind='dyn', track='curve', N=25, reg=None, solver='v2'),
    # the reference's own Monte-Carlo experiment (scripts/DGSQP_ALGAMES_monte_carlo_curve.py), kinematic bicycle
    'kb_curve_N25': dict(desc='2-agent kinematic-bicycle (euler) curve track, N=25, reg=0 (curve.py:161), fp64', kind='kb', track='curve', N=25, reg=0.0),
    'kb_chicane_N25': dict(desc='2-agent kinematic-bicycle (euler) chicane track, N=25, reg=1e-3 (chicane.py:164), fp64', kind='kb', track='chicane', N=25, reg=1e-3),
    # n = 60 games whose arena fits HALF a CU's LDS: the measurement behind row N1 (two 256-thread workgroups per CU, tools/n1_two_per_cu.py)
    'kb_chicane_N15': dict(desc='2-agent kinematic-bicycle (euler) chicane track, N=15, reg=1e-3 (BASELINE configs[0] game), fp64', kind='kb', track='chicane', N=15, reg=1e-3),
    'dyn_curve_N15': dict(desc='2-agent dynamic-bicycle (Pacejka, rk4 M=10) curve track, N=15, game of exact_dynamic_game_dynamic.py, fp64', kind='dyn', track='curve', N=15, reg=1e-3),
    # other Monte-Carlo scripts of the reference at their own sizes (not BASELINE's metric; for the DESIGN.md table)
    'kb_barc2_N15': dict(desc='2-agent kinematic-bicycle race on the L_track_barc circuit, N=15, reg=0 (DGSQP_comp_monte_carlo.py), fp64', kind='barc', M=2, N=15, reg=0.0),
    'kb_barc3_N25': dict(desc='3-agent kinematic-bicycle race on the L_track_barc circuit, N=25, reg=0 (BASELINE configs[2] game), XL layout, fp64', kind='barc', M=3, N=25, reg=0.0),
    'merge_N20': dict(desc='3-car highway merge, kinematic unicycles rk3, N=20, reg=0 (DGSQP_merge_monte_carlo.py), big layout, fp64', kind='merge', N=20, reg=0.0),
    # BASELINE configs[4]: six cars, N = 25 (n = 300, 1,587 rows, 837 distinct dense gradients): XL kernels, tables read from the constant block
    'merge6_N25': dict(desc='6-car highway merge, kinematic unicycles rk3, N=25, reg=0 (DGSQP_merge_monte_carlo.py with six cars = BASELINE configs[4] game), XL layout, fp64', kind='merge', N=25, M=6, reg=0.0),
    'kb_curve_N50': dict(desc='2-agent kinematic-bicycle curve track, N=50, reg=1e-3 (BASELINE configs[3] size on the curve track), XL layout, fp64', kind='kb', track='curve', N=50, reg=1e-3),
    'kb_f1_N50': dict(desc='2-agent kinematic-bicycle race on the F1 track (cubic-spline centre line), N=50, reg=1e-3 (BASELINE configs[3] game), XL layout, fp64', kind='f1', N=50, reg=1e-3),
    'kb_curve3_N25': dict(desc='3-agent kinematic-bicycle curve track, N=25, reg=1e-3 (DGSQP_monte_carlo_agents.py M=3 N=25 = BASELINE configs[2] size), XL layout, fp64', kind='kb', track='curve', N=25, M=3, reg=1e-3),
}


def make_game(name, reg=None):
    from dgsqp_amd.montecarlo import dynamic_racing_game, kinematic_racing_game
    w = WORKLOADS[name]
    reg = w['reg'] if reg is None else reg
    if w['kind'] == 'dyn':
        g = dynamic_racing_game(w['track'], N=w['N'], rk4_substeps=10, reg=reg if reg is not None else 1e-3, game_def=w.get('game_def', 'exact_dynamic'),
                                solver=w.get('solver', 'v1'))
        if w.get('solver') == 'v2':
            g.params.time_limit = None          # wall-clock limits make a benchmark irreproducible; the study's 600 s is never reached here
        return g
    if w['kind'] == 'barc':
        from dgsqp_amd.montecarlo import barc_racing_game
        return barc_racing_game(N=w['N'], M=w['M'], reg=reg)
    if w['kind'] == 'f1':
        from dgsqp_amd.montecarlo import f1_racing_game
        return f1_racing_game(N=w['N'], reg=reg)
    if w['kind'] == 'merge':
        from dgsqp_amd.montecarlo import merge_game
        return merge_game(N=w['N'], reg=reg, M=w.get('M', 3))
    return kinematic_racing_game(w['track'], N=w['N'], reg=reg, M=w.get('M', 2))


def algorithmic_bytes_per_solve(d):
    """inputs x0[n_q] + u_ws[n]; outputs u[n] + l[n_c] + x[(N+1) n_q] + cond[3] + cost[M] (fp64)
    + three int32 (status, iterations, QP solves)."""
    return 8 * (d.n_q + d.n + d.n + d.n_c + (d.N + 1) * d.n_q + 3 + d.M) + 12


FC_COST = {0: 1.0, 1: 3.0, 2: 0.3}          # jet cost of f_c relative to the kinematic bicycle's (dynamic bicycle: 314 instructions against ~100; unicycle ~30)
STAGES = {0: 1, 1: 4, 2: 3, 3: 2}           # f_c evaluations per substep: euler, rk4, rk3, rk2 (dynamics_models.py:88-125)
K_ADMM = 250                                # ADMM iterations per OSQP call in the flop model WHEN the run did not count them (the timed region's own mean, dgsqp_osqp_counters, replaces it: 270 on configs[1], 3,400 on the six-car merge at reg = 0)


def algorithmic_flops_per_solve(d, P, mean_qp_solves, qp_method, k_admm=None):
    """SURVEY.md section 8(d), the compute roof: per QP solve  F_eval + F_eig + F_qp  with
        F_eval = M [N 6 n_q^3 + 2 n_q n_u N^2 n_q] + N C_AD,   C_AD = M x 3,000 x (f_c cost relative to the kinematic bicycle) x (f_c evaluations per step)
        F_eig  = 9 n^3
        F_qp   = 2 n^2 n_act + n^3 / 3 + k_admm (4 n n_dense + 2 n^2),   n_act = n / 4 active rows; k_admm = 0 for the exact active-set QP; for OSQP the
                 MEASURED mean ADMM iterations per QP of the timed region (dgsqp_osqp_counters), 250 when no count is at hand
    times the mean number of QP solves per scenario (every QP solve follows one evaluation with Hessian; the Hessian-free trial
    evaluations of the line searches are not counted).  The survey's constants are ranges (C_AD 2-4 k, F_qp 1e7-1e8 at n = 100); the
    ones used are stated here so that anybody can recompute the figure from the JSON line."""
    M, N, n_q, n_u, n = int(d.M), int(d.N), int(d.n_q), int(d.n_u), int(d.n)
    n_dense = int(d.n_dense)
    model = int(P.agents[0].model)
    stages = STAGES[int(P.integrator)] * (int(P.substeps) if int(P.integrator) != 0 else 1)
    c_ad = M * 3000.0 * FC_COST[model] * stages
    f_eval = M * (N * 6.0 * n_q ** 3 + 2.0 * n_q * n_u * N ** 2 * n_q) + N * c_ad
    f_eig = 9.0 * n ** 3
    k = (K_ADMM if k_admm is None else float(k_admm)) if qp_method == 'osqp' else 0.0
    f_qp = 2.0 * n ** 2 * (n / 4.0) + n ** 3 / 3.0 + k * (4.0 * n * n_dense + 2.0 * n ** 2)
    return dict(per_qp_solve=dict(F_eval=f_eval, F_eig=f_eig, F_qp=f_qp), qp_solves_per_scenario=mean_qp_solves, admm_iterations_per_qp=k if qp_method == 'osqp' else None,
                admm_iterations_source=None if qp_method != 'osqp' else ('counted over the timed region (dgsqp_osqp_counters)' if k_admm is not None else 'assumed'),
                flop_per_solve=mean_qp_solves * (f_eval + f_eig + f_qp))


def source_fingerprint(root=None):
    """sha256 over the kernel sources WITHOUT their comments and white space: a PMC summary under profiles/ is only used for the kernels it
    was collected on (editing a comment does not invalidate it; touching a statement does)."""
    import glob
    import hashlib
    import re
    root = root or ROOT
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(root, 'dgsqp_amd', 'csrc', '*.h')) + glob.glob(os.path.join(root, 'dgsqp_amd', 'csrc', '*.hip')) + [os.path.join(root, 'include', 'dgsqp.h')]):
        text = open(f, encoding='utf-8', errors='replace').read()
        text = re.sub(r'/\*.*?\*/', ' ', text, flags=re.S)
        text = re.sub(r'//[^\n]*', ' ', text)
        h.update(os.path.basename(f).encode())
        h.update(' '.join(text.split()).encode())
    return h.hexdigest()


def spawn_ranks(n, argv, script=None, timeout=3600.0):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes (nothing in this process has
    touched the GPU), relay rank 0's JSON line, exit with the worst return code.  Every spawn has its own rendezvous file (port +
    pid + time, unlinked before the ranks start) and its own launch tag; a rank that dies takes the others down with it
    instead of leaving them in ncclCommInitRank for ever.  `script`: what to run as a rank (tests substitute a stub)."""
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    rdv = os.path.join(os.environ.get('TMPDIR', '/tmp'), f'dgsqp_rccl_{port}_{os.getpid()}_{time.time_ns()}.id')
    for stale in (rdv,):
        try:
            os.remove(stale)
        except OSError:
            pass
    procs = []
    tag = f'{os.getpid()}_{port}_{time.time_ns()}'
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   DGSQP_RENDEZVOUS=rdv, DGSQP_LAUNCH_TAG=tag)
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # rank 0's stdout is read by a thread so that the liveness loop below never blocks on a full pipe
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.time() + timeout
    failed, killed = False, set()
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs) or time.time() > deadline:
            failed = True
            time.sleep(2.0)                 # (let the others notice a broken collective by themselves first)
            for p in procs:
                if p.poll() is None:
                    killed.add(p.pid)
                    p.kill()                # exact PIDs we started
            break
        time.sleep(0.05)
    rcs = [0 if p.pid in killed else p.wait() for p in procs]      # (the code of the rank that failed, not of the ones taken down after it)
    for p in procs:
        p.wait()
    reader.join(10.0)
    sys.stdout.write(b''.join(c for c in chunks if c).decode())
    sys.stdout.flush()
    try:
        os.remove(rdv)
    except OSError:
        pass
    worst = max(abs(rc) for rc in rcs)
    sys.exit(worst if worst else (1 if failed else 0))


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=120, help='timed steps; the default is long enough (about 15 s) for the drain of the last launches -- one slowest scenario, ~0.7 s -- to weigh a few percent')
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--batch', type=int, default=1024, help='scenarios per GPU per step (--scaling weak) or in total per step (strong)')
    ap.add_argument('--scaling', choices=('weak', 'strong'), default='weak')
    ap.add_argument('--workload', default='dyn_curve_N25', choices=sorted(WORKLOADS))
    ap.add_argument('--cpu-sample', type=int, default=64, help='scenarios timed on the host for cpu_baseline (0 disables)')
    ap.add_argument('--pipeline', type=int, default=5,
                    help='launches in flight per GPU (each on its own HIP stream / hardware queue); 1 = strictly one launch at a time')
    ap.add_argument('--group', type=int, default=0, help='staged batches (= steps) solved by ONE launch with a shared ticket queue (dgsqp_launch_staged_group); '
                    '0 = auto: all of them in one cooperative launch when the timed region has at most 24 steps (a launch ends behind its slowest scenarios once, and its '
                    'deferral of long scenarios sees the whole job), otherwise 12 per launch, pipelined')
    ap.add_argument('--batches', type=int, default=0, help='distinct staged batches = handles the steps cycle through (default: (pipeline + 1) x group, so that a group never waits for the tail of a launch that still holds its handles)')
    ap.add_argument('--single-steps', type=int, default=3, help='extra one-launch-at-a-time steps behind value_single_launch / roofline.kernel_ms (0 disables)')
    ap.add_argument('--host-steps', type=int, default=2, help='extra dgsqp_solve_batch calls from host buffers behind value_host_inclusive (0 disables)')
    ap.add_argument('--coop', choices=('auto', 'off'), default='auto',
                    help='cooperative line search (dgsqp_set_cooperative): auto = in the LAST launch of the timed region and in launches that run alone -- idle '
                         'workgroups evaluate line-search trials of the scenarios still solving; results are bit-identical; off = never')
    ap.add_argument('--qp', choices=('active_set', 'osqp'), default='active_set',
                    help="how _solve_qp is computed (dgsqp_params_t.qp_method): 'active_set' = the exact KKT point (dual active-set method + polish), "
                         "'osqp' = OSQP's own ADMM + polish arithmetic as the reference runs it (csrc/dgsqp_osqp.h); the cpu_baseline uses the same method")
    ap.add_argument('--reg', type=float, default=None, help='DGSQPParams.reg (default: the value of the workload)')
    ap.add_argument('--mixed-precision', action='store_true',
                    help="dgsqp_params_t.mixed_precision (off by default): with --qp osqp on the XL layout (configs[2], [3], [4]) the ADMM iteration's "
                         "explicit K^-1 is stored in fp32 (fp64 accumulation); everything else stays fp64.  'dtype' of the line then says 'f64 (K^-1 of the ADMM iteration f32)'")
    ap.add_argument('--wg-per-cu', type=int, choices=(1, 2), default=1,
                    help='2: solve on libdgsqp_hip_b256.so -- 256-thread workgroups, half the LDS arena, two workgroups per CU (row N1: large batches of n <= 64 games; '
                         'explicit-inverse layouts with the active-set QP only)')
    ap.add_argument('--eig-floor', type=float, default=None, help='_nearestPD floor (default: the literal 1e-10, DGSQP.py:1293)')
    ap.add_argument('--snap-active-bounds', action='store_true', help='implementation knob, see include/dgsqp.h (default: literal)')
    ap.add_argument('--extras-budget', type=float, default=300.0,
                    help='seconds since the start of the process by which the extra legs must be over: a leg gets min(its own timeout, what is left), and none is started with less than 15 s left (the legs left out are named in the line)')
    ap.add_argument('--line', choices=('compact', 'full'), default='compact',
                    help='compact: the ONE stdout line holds the contract keys, config, roofline, cpu_baseline and the headline figures in under 4 KB (full records: bench_workloads.json); '
                         'full: the complete record of this workload (what the extra legs, run as child processes, hand back)')
    ap.add_argument('--extras', choices=('auto', 'off'), default='auto',
                    help="auto: the default invocation (configs[1], exact QP, one GPU) also times, in the same run, --qp osqp on configs[1] and BASELINE configs[2], [3], [4] "
                         "at the batch sizes BASELINE.json names -- each leg in a child process with a timeout, full records in bench_workloads.json, a four-column summary "
                         "in the line (key 'workloads'); off: only the workload asked for")
    return ap.parse_args(argv)


def run_workload(args, rank, local_rank, world):
    """One workload through the contract's timed region (plus the one-at-a-time, host-inclusive and cpu_baseline legs the flags ask
    for); returns the JSON record on rank 0, None elsewhere.  Every handle it creates is destroyed before it returns -- also when
    it raises (device buffers, streams, the communicator and the deferral pool's claim go with the handles)."""
    held = {'solvers': [], 'comm': None}
    ok = False
    try:
        line = _run_workload(args, rank, local_rank, world, held)
        ok = True
        return line
    finally:
        if held['comm'] is not None and (ok or world == 1):       # (a failing rank of several must not wait in close()'s barrier for peers that may be gone)
            try:
                held['comm'].close()
            except Exception:
                pass
            held['comm'] = None
        while held['solvers']:
            held['solvers'].pop()           # DGSQP.__del__ -> dgsqp_destroy
        import gc
        gc.collect()


def _run_workload(args, rank, local_rank, world, held):
    from dgsqp_amd import _ffi
    from dgsqp_amd.montecarlo import sample_scenarios
    from dgsqp_amd.sharding import Communicator, padded_shard_size, shard_range, stats_from_records, summarize
    from dgsqp_amd.solver import DGSQP

    game = make_game(args.workload, args.reg)
    mk = lambda: DGSQP(*game.solver_args(), print_method=None, device=local_rank, eig_floor=args.eig_floor,
                       snap_active_bounds=args.snap_active_bounds, qp_method=args.qp, mixed_precision=getattr(args, 'mixed_precision', False),
                       workgroups_per_cu=getattr(args, 'wg_per_cu', 1))
    P = max(1, args.pipeline)
    if args.group <= 0:
        args.group = args.steps if args.steps <= 24 else 12
    n_batches = args.batches if args.batches > 0 else max(P, min(max(3 * P, (P + 1) * max(1, args.group)), args.steps))     # distinct batches = handles; steps cycle through them
    solvers = held['solvers']
    for _ in range(n_batches):
        solvers.append(mk())
    solver = solvers[0]
    d = solver.dims
    lib = solver._lib
    comm = held['comm'] = Communicator(solver, rank, world)          # RCCL communicator owned by handle 0 (world = 1: no peer needed)
    if args.scaling == 'weak':
        B_total, (lo, hi) = args.batch * world, (rank * args.batch, (rank + 1) * args.batch)
    else:
        B_total, (lo, hi) = args.batch, shard_range(args.batch, rank, world)
    B = hi - lo
    B_pad = padded_shard_size(B_total, world)
    # rejection sampling + PID warm starts happen before any timing; batch j of rank r has its own seed
    batches = []
    for j in range(n_batches):
        x0, u_tm = sample_scenarios(game, B, seed=1 + rank + 1000 * j, solver=solver if game.sampler == 'first_segment' and d.M == 2 else None)   # PID warm starts on the device where the sampler supports it
        batches.append((np.ascontiguousarray(x0), np.ascontiguousarray(solver._to_agent_major(u_tm))))
    handles = [sv._h for sv in solvers]
    for hh in handles:
        lib.dgsqp_set_cooperative(hh, 1 if args.coop == 'auto' else 0)
    for hh, (x0, u_am) in zip(handles, batches):
        assert lib.dgsqp_stage_inputs(hh, B, _ffi.dptr(x0), _ffi.dptr(u_am)) == 0, lib.dgsqp_last_error(hh)

    # the deferral pool of the largest cooperative launch of the timed region (all `group` batches in one launch), sized before the warm-up:
    # left to the launch itself, growing it from the warm-up's single batches is a hipFree + hipMalloc of gigabytes inside the timed region
    if args.coop == 'auto':
        assert lib.dgsqp_reserve_deferral(handles[0], B * max(1, min(args.group, n_batches, args.steps))) == 0, lib.dgsqp_last_error(handles[0])
    tm = _ffi.TimingT()
    for w in range(args.warmup):
        hh = handles[w % n_batches]
        assert lib.dgsqp_solve_staged(hh, C.byref(tm)) == 0, lib.dgsqp_last_error(hh)

    def fence():
        for hh in handles:
            assert lib.dgsqp_synchronize(hh) == 0
        comm.barrier()

    def run_steps(steps, in_flight, group=1):
        """`steps` steps (one staged batch each), issued `group` at a time: the batches of a group are solved by ONE launch with a shared
        ticket queue (dgsqp_launch_staged_group; every batch keeps its own buffers, results are bit-identical to separate launches), at
        most `in_flight` launches outstanding.  The next launch is started when the previous one has handed out its last scenario
        (its workgroups begin to exit and free compute units) -- not earlier, or two launches would share the GPU from the start and
        both grow tails.  A launch ends with its slowest scenario (up to ~1 s here, the balanced time of one batch is ~0.1 s), and the
        hardware runs at most 16 launches side by side: launches are retired in the order they FINISH (dgsqp_finished), and fewer,
        longer launches keep the compute units busier than many short ones."""
        kernel_ms, flying, last = [], [], None     # flying: tuples of handle indices, leader first

        def busy():
            return {i for grp in flying for i in grp}

        def retire(grp):
            flying.remove(grp)
            for i in grp:
                assert lib.dgsqp_wait(handles[i], C.byref(tm)) == 0, lib.dgsqp_last_error(handles[i])
            kernel_ms.append(tm.kernel_ms)            # HIP events around that launch on its leader's stream

        def retire_finished(block):
            deadline = time.perf_counter() + 600.0
            while True:
                done = [grp for grp in flying if lib.dgsqp_finished(handles[grp[0]])]
                for grp in done:
                    retire(grp)
                if done or not block or not flying:
                    return
                if time.perf_counter() > deadline:
                    raise RuntimeError('bench.py: a launch did not finish within 600 s')
                time.sleep(0.0002)
        group = max(1, min(group, n_batches))         # a group can never hold more batches than there are handles
        fence()
        t0 = time.perf_counter()
        nxt, step = 0, 0
        while step < steps:
            gsz = min(group, steps - step)
            if last is not None and in_flight > 1:
                deadline = time.perf_counter() + 600.0           # never spin forever on a launch that died
                while not lib.dgsqp_draining(handles[last]) and time.perf_counter() < deadline:
                    time.sleep(0.0002)
            retire_finished(block=False)
            while len(flying) >= in_flight or n_batches - len(busy()) < gsz:
                retire_finished(block=True)
            grp, taken = [], busy()
            while len(grp) < gsz:                     # the next staged batches whose handles are idle (round robin)
                i, nxt = nxt, (nxt + 1) % n_batches
                if i not in taken and i not in grp:
                    grp.append(i)
            arr = (C.c_void_p * gsz)(*[handles[i] for i in grp])
            # the last launch of the region (and any launch that runs alone) is cooperative: nothing else is waiting for the compute
            # units its idle workgroups keep while they help the slowest scenarios
            coop = args.coop == 'auto' and (step + gsz >= steps or in_flight == 1)
            lib.dgsqp_set_cooperative(handles[grp[0]], 2 if coop else 0)
            assert lib.dgsqp_launch_staged_group(arr, gsz) == 0, lib.dgsqp_last_error(handles[grp[0]])
            lib.dgsqp_set_cooperative(handles[grp[0]], 1 if args.coop == 'auto' else 0)
            flying.append(tuple(grp))
            last = grp[0]
            step += gsz
        while flying:
            retire(flying[0])
        mine = time.perf_counter() - t0           # this rank's own time up to its last launch (before the closing fence)
        fence()
        elapsed = float(comm.allreduce_max([time.perf_counter() - t0])[0])
        per_rank = np.zeros(max(world, 1))
        per_rank[rank] = mine
        run_steps.per_rank = comm.allreduce_max(per_rank).tolist() if world <= 64 else None
        return elapsed, kernel_ms, last

    # ---- the timed region of the contract: exactly K steps, fences on both sides, max over ranks
    if args.qp == 'osqp':
        fence()
        lib.dgsqp_osqp_counters(handles[0], None, 1)          # (QP calls, ADMM iterations) of this device from here on
    elapsed, kernel_ms_pipe, last = run_steps(args.steps, P, max(1, args.group))
    k_admm = None
    if args.qp == 'osqp':
        cnt = (C.c_uint64 * 2)()
        if lib.dgsqp_osqp_counters(handles[0], cnt, 0) == 0 and cnt[0] > 0:
            k_admm = cnt[1] / cnt[0]
    elapsed_per_rank = run_steps.per_rank
    value = B_total * args.steps / elapsed
    # the single stats gather: the records of one step (batch 0, solved by handle 0, which owns the communicator)
    rec = comm.gather_stats(B_pad)
    stats = stats_from_records(rec)

    # ---- one launch at a time (no overlap): per-launch kernel time, single-launch throughput
    single = None
    if args.single_steps > 0:
        e1, kms1, _ = run_steps(args.single_steps, 1)
        kms = float(comm.allreduce_max([np.mean(kms1)])[0])
        single = dict(value=B_total * args.single_steps / e1, kernel_ms=kms, ms_per_step=e1 / args.single_steps * 1e3)
    else:
        kms = float(comm.allreduce_max([np.mean(kernel_ms_pipe)])[0])

    # ---- host-inclusive: H2D + solve + D2H of every output + the stats gather, from host buffers
    host = None
    if args.host_steps > 0:
        n, nc, nx = int(d.n), int(d.n_c), int((d.N + 1) * d.n_q)
        ob = dict(u=np.empty((B, n)), l=np.empty((B, nc)), x=np.empty((B, nx)), st=np.empty(B, np.int32), it=np.empty(B, np.int32),
                  qp=np.empty(B, np.int32), cond=np.empty((B, 3)), cost=np.empty((B, int(d.M))))
        fence()
        t0 = time.perf_counter()
        for k in range(args.host_steps):
            x0, u_am = batches[(k + 1) % n_batches]
            rc = lib.dgsqp_solve_batch(handles[0], B, _ffi.dptr(x0), _ffi.dptr(u_am), _ffi.dptr(ob['u']), _ffi.dptr(ob['l']), _ffi.dptr(ob['x']),
                                       _ffi.iptr(ob['st']), _ffi.iptr(ob['it']), _ffi.iptr(ob['qp']), _ffi.dptr(ob['cond']), _ffi.dptr(ob['cost']), C.byref(tm))
            assert rc == 0, lib.dgsqp_last_error(handles[0])
            comm.gather_stats(B_pad)
        fence()
        eh = float(comm.allreduce_max([time.perf_counter() - t0])[0])
        host = dict(value=B_total * args.host_steps / eh, ms_per_step=eh / args.host_steps * 1e3)
        # ... and the same for a GROUP of batches handed over together: H2D of every batch, ONE cooperative launch over all of them, D2H of
        # every output (what a Monte-Carlo driver with more than one batch in hand would call)
        G = max(1, min(args.group, n_batches))
        obs = [dict(u=np.empty((B, n)), l=np.empty((B, nc)), x=np.empty((B, nx)), st=np.empty(B, np.int32), it=np.empty(B, np.int32),
                    qp=np.empty(B, np.int32), cond=np.empty((B, 3)), cost=np.empty((B, int(d.M)))) for _ in range(G)]
        fence()
        t0 = time.perf_counter()
        for k in range(G):
            x0, u_am = batches[k]
            assert lib.dgsqp_stage_inputs(handles[k], B, _ffi.dptr(x0), _ffi.dptr(u_am)) == 0, lib.dgsqp_last_error(handles[k])
        lib.dgsqp_set_cooperative(handles[0], 2 if args.coop == 'auto' else 0)
        arr = (C.c_void_p * G)(*handles[:G])
        assert lib.dgsqp_launch_staged_group(arr, G) == 0, lib.dgsqp_last_error(handles[0])
        lib.dgsqp_set_cooperative(handles[0], 1 if args.coop == 'auto' else 0)
        for k in range(G):
            o = obs[k]
            rc = lib.dgsqp_fetch_results(handles[k], _ffi.dptr(o['u']), _ffi.dptr(o['l']), _ffi.dptr(o['x']), _ffi.iptr(o['st']), _ffi.iptr(o['it']),
                                         _ffi.iptr(o['qp']), _ffi.dptr(o['cond']), _ffi.dptr(o['cost']))
            assert rc == 0, lib.dgsqp_last_error(handles[k])
        comm.gather_stats(B_pad)
        fence()
        eg = float(comm.allreduce_max([time.perf_counter() - t0])[0])
        host['grouped'] = dict(value=B_total * G / eg, batches=G, ms=eg * 1e3)

    if rank == 0:
        # HBM traffic comes from PMC counters collected in separate rocprofv3 --pmc passes (gpurun refuses mixed runs) and summarised
        # under profiles/; so does the fp64 instruction mix (SQ counters) behind the vector-ALU figure.  Latest matching file wins.
        # A summary is used only when it was collected on THESE kernel sources (its `source_sha256`, written by tools/pmc_summary.py, equals
        # the fingerprint of dgsqp_amd/csrc now) and on this QP method; otherwise traffic is null and the line says which file was refused.
        traffic, flop_per_solve, traffic_src, traffic_note, sq_shares = None, None, None, None, None
        try:
            import glob
            fp = source_fingerprint()
            for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r0[5-9]_pmc_*.json'))):
                pm = json.load(open(f))
                if pm.get('workload') == args.workload and pm.get('batch_per_gpu') == B and pm.get('qp_method', 'active_set') == args.qp:
                    if pm.get('source_sha256') != fp:
                        traffic_note = f'{os.path.relpath(f, ROOT)} refused: collected on other kernel sources (stale)'
                        continue
                    if 'traffic_bytes_per_launch' in pm:
                        traffic, traffic_src, traffic_note = pm['traffic_bytes_per_launch'], os.path.relpath(f, ROOT), None
                    flop_per_solve = pm.get('fp64_flop_per_solve_upper_bound', flop_per_solve)
                    sq_shares = pm.get('sq_wave_cycle_shares', sq_shares)
        except Exception as e:
            traffic_note = f'PMC summaries unreadable: {e}'
        # the dominant kernel's launches of the TIMED region: every launch solves `batches_per_launch` staged batches of B scenarios
        # (the last one possibly fewer); HIP events on the launch's own stream (dgsqp_wait -> dgsqp_timing_t.kernel_ms)
        gsz = max(1, min(args.group, n_batches))
        timed_ms = float(np.mean(kernel_ms_pipe))
        solves_per_timed_launch = B * args.steps / max(1, len(kernel_ms_pipe))
        bytes_per_launch = algorithmic_bytes_per_solve(d) * solves_per_timed_launch
        achieved = bytes_per_launch / (timed_ms * 1e-3) / 1e9
        achieved_single = algorithmic_bytes_per_solve(d) * B / (kms * 1e-3) / 1e9
        summ = summarize(stats)
        flops = algorithmic_flops_per_solve(d, solver._problem, float(summ['mean_qp_solves_all']), args.qp, k_admm)
        line = {
            'metric': 'Monte-Carlo scenarios/sec (SQP solves/sec), 2-agent N=25' if args.workload.startswith(('dyn_curve_N25', 'kb_curve_N25', 'kb_chicane_N25'))
                      else 'Monte-Carlo scenarios/sec (SQP solves/sec)', 'value': value, 'unit': 'scenarios/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True, 'scaling': args.scaling, 'vs_baseline': None, 'dtype': 'f64 (K^-1 of the ADMM iteration f32)' if (getattr(args, 'mixed_precision', False) and args.qp == 'osqp' and solver.dims.layout == 2 and float(game.params.reg) >= 1e-4) else 'f64', 'data': 'synthetic',
            'config': {'workload': args.workload, 'description': WORKLOADS[args.workload]['desc'], 'batch_per_gpu': B, 'batch_total': B_total,
                       'n': int(d.n), 'n_c': int(d.n_c), 'parallelism': f'scenario-sharded x{world}',
                       'shard_mode': 'contiguous (weak: every rank samples its own batch_per_gpu scenarios)' if args.scaling == 'weak' else 'contiguous ranges of one batch (sharding.shard_range)',
                       'sampler': {'circuit': 'scripts/DGSQP_comp_monte_carlo.py:365-382, PID warm start',
                                   'merge': 'scripts/DGSQP_merge_monte_carlo.py:421-480, zero warm start'}.get(
                                       game.sampler, 'scripts/DGSQP_ALGAMES_monte_carlo_curve.py:384-467, PID warm start') + ' (seed 1 + rank + 1000 * batch)',
                       'distinct_batches': n_batches, 'batches_per_launch': max(1, args.group), 'layout': {0: 'lds', 1: 'big', 2: 'xl'}[int(d.layout)],
                       'launches_in_flight': P, 'batches_in_flight': P * max(1, args.group), 'workgroups_per_cu': int(getattr(args, 'wg_per_cu', 1)), 'cooperative_line_search': args.coop, 'qp_method': args.qp, 'reg': float(game.params.reg), 'eig_floor': float(solver._cparams.eig_floor),
                       'snap_active_bounds': int(solver._cparams.snap_active_bounds)},
            'elapsed_s': elapsed, 'elapsed_s_per_rank': elapsed_per_rank,        # each rank's own time for its K steps: load imbalance between the shards shows here
            'value_single_launch': single['value'] if single else None,
            'value_host_inclusive': host['value'] if host else None,
            'value_host_inclusive_grouped': host['grouped']['value'] if host else None,      # H2D + one launch + D2H of `batches_per_launch` batches
            'host_inclusive_grouped_batches': host['grouped']['batches'] if host else None,
            'mean_iters': summ['mean_iters_converged'], 'mean_iters_all': summ['mean_iters_all'],
            'mean_qp_solves': summ['mean_qp_solves_all'], 'converged_fraction': summ['converged'],
            'status_fractions': {k: summ[k] for k in ('conv_abs_tol', 'conv_rel_tol', 'max_it', 'diverged', 'qp_fail')},
            # The governing roof is the fp64 VECTOR ALU (per-scenario state lives in LDS for the whole solve, the path is
            # small dense linear algebra + Taylor arithmetic): achieved = ALGORITHMIC flops of the launch (section 8d's formula, constants in
            # algorithmic_flops_per_solve, terms in `flop_model`) / its HIP-event duration, against 256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz
            # = 78.6 TFLOP/s.  The HBM figure the contract also asks for is kept under `hbm`: a tiny fraction of peak by construction.
            'roofline': {'bound': 'valu_fp64', 'achieved': flops['flop_per_solve'] * solves_per_timed_launch / (timed_ms * 1e-3) / 1e12, 'peak': 78.6, 'unit': 'TFLOP/s',
                         'frac': flops['flop_per_solve'] * solves_per_timed_launch / (timed_ms * 1e-3) / 1e12 / 78.6,
                         'flop_model': flops,
                         # executed fp64 flops from the SQ_INSTS_VALU_*_F64 pass (an upper bound: masked lanes count in full), when a summary of these kernels exists
                         'frac_executed_upper_bound': (flop_per_solve * solves_per_timed_launch / (timed_ms * 1e-3) / 1e12 / 78.6) if flop_per_solve is not None else None,
                         'executed_flop_per_solve_upper_bound': flop_per_solve, 'sq_wave_cycle_shares': sq_shares,
                         'hbm': {'achieved': achieved, 'peak': 8000.0, 'unit': 'GB/s', 'frac': achieved / 8000.0},
                         # PMC bytes (FETCH_SIZE x 2 + WRITE_SIZE, separate rocprofv3 --pmc passes over one-batch launches) scaled to the solves of a timed launch
                         'traffic': traffic * solves_per_timed_launch / B if traffic is not None else None, 'traffic_source': traffic_src, 'traffic_note': traffic_note,
                         'kernel': 'dg_solve_kernel', 'kernel_ms': timed_ms,
                         'kernel_ms_source': f'HIP events on the launch streams, mean over the {len(kernel_ms_pipe)} launch(es) of the timed region',
                         'launches_timed': len(kernel_ms_pipe), 'solves_per_launch': solves_per_timed_launch,
                         'algorithmic_bytes_per_solve': algorithmic_bytes_per_solve(d), 'algorithmic_bytes_per_launch': bytes_per_launch,
                         # the same kernel launched one batch at a time (what the rocprof kernel-trace of profiles/ and the PMC passes measure)
                         'single_launch': {'kernel_ms': kms, 'solves_per_launch': B, 'achieved': achieved_single, 'frac': achieved_single / 8000.0, 'traffic': traffic,
                                           'source': 'HIP events, launches one at a time' if single else 'HIP events, overlapping launches'}},
        }
        if world == 1 and args.cpu_sample > 0:
            from oracle import oracle            # checker/baseline only: the CPU restatement, NOT CasADi+OSQP
            oracle.build()
            cores = os.cpu_count() or 1
            # A THROUGHPUT, not the time of the slowest scenario: every thread gets at least 4 scenarios handed out dynamically (default: the
            # first 256 scenarios of the first batch over 64 threads), every scenario's own wall-clock time on its thread is recorded, and
            # `value` = threads / mean seconds per scenario -- what those threads sustain over a long Monte-Carlo run.  `value_wall` is the
            # sample over its wall time, idle tail behind the slowest scenario (tens of seconds for one that never converges) included.
            # (default: 4 scenarios for each of up to 64 threads -- 256 scenarios, ~20 s, the contract's "10-30 s of CPU work"; round 5 took 8 per thread, 42 s;
            # more threads only add contention in the dense oracle:
            # 128 threads on the 256-thread host took 6.4 s per scenario against 0.5 s for one alone, profiles/r04_*)
            ns = min(args.cpu_sample if args.cpu_sample != 64 else 4 * min(cores, 64), B)
            nth = max(1, min(cores, 64, ns // 4))
            x0, u_am = batches[0]
            t1 = time.perf_counter()
            ob = oracle.solve_batch(solver._problem, solver._cparams, x0[:ns], u_am[:ns], nthreads=nth, timed=True)
            dt = time.perf_counter() - t1
            n1 = min(4, ns)
            t1 = time.perf_counter()
            oracle.solve_batch(solver._problem, solver._cparams, x0[:n1], u_am[:n1], nthreads=1)
            dt1 = time.perf_counter() - t1
            sec = ob['seconds']
            line['cpu_baseline'] = {'value': nth / float(sec.mean()), 'unit': 'scenarios/s', 'cores': nth, 'kind': 'port',
                                    'value_wall': ns / dt, 'value_one_core': n1 / dt1, 'scenarios_per_thread': ns / nth,
                                    'sample_short': f'first {ns} scenarios of batch 0, {nth} of {cores} threads, {dt:.0f} s wall; value = threads / mean s per scenario; C++ restatement, not CasADi+OSQP',
                                    'seconds_per_scenario': {'mean': float(sec.mean()), 'median': float(np.median(sec)), 'max': float(sec.max())},
                                    'sample': f'first {ns} scenarios of batch 0 over {nth} of {cores} hardware threads, dynamic hand-out, {ns / nth:.0f} per thread, {dt:.1f} s wall '
                                              f'(mean {sec.mean():.2f} s, slowest {sec.max():.1f} s per scenario); value = threads / mean seconds per scenario; first {n1} alone on one core: {dt1:.1f} s; '
                                              f'oracle/dgsqp_oracle.cpp (dense literal restatement, QP: ' + ('exact active set' if args.qp == 'active_set' else 'restated OSQP, oracle/osqp.hpp') + '; not CasADi+OSQP)'}
        elif world > 1:
            line['cpu_baseline'] = None
            line['cpu_baseline_note'] = 'timed on rank 0 of the 1-GPU run only (bench.py --gpus 1)'
    else:
        line = None
    del comm, handles, solver, solvers
    return line


# The other claims of DESIGN.md / BASELINE.md, timed by the same run as the headline (default invocation at one GPU only): the reference's own
# QP arithmetic on configs[1], and BASELINE configs[2], [3], [4] at the batch sizes BASELINE.json names -- ONE cooperative launch of the
# whole batch each (the XL games' tails need that many scenarios behind them); kb_curve3_N25 is the solvable three-car game of
# configs[2]'s size (DGSQP_monte_carlo_agents.py) next to the circuit game, 95 % of whose solves end in an LP-certified infeasible QP.
# Every leg runs in a FRESH CHILD PROCESS with its own timeout (run_leg), after the parent has measured the headline and destroyed its
# handles; the full records go to the side file bench_workloads.json, the ONE line on stdout stays under 4 KB.
_ONE = dict(steps=1, warmup=0, pipeline=1, batches=1, group=1)
EXTRA_LEGS = (
    dict(tag='configs[1] --qp osqp', workload='dyn_curve_N25', qp='osqp', timeout=90),
    dict(tag='configs[4] B=65536', workload='merge6_N25', batch=65536, timeout=120, **_ONE),
    dict(tag='configs[3] B=16384', workload='kb_f1_N50', batch=16384, timeout=90, **_ONE),
    dict(tag='configs[2] B=4096', workload='kb_barc3_N25', batch=4096, timeout=60, **_ONE),
    dict(tag='configs[2] size, solvable game, B=4096', workload='kb_curve3_N25', batch=4096, timeout=60, **_ONE),
    # ... and the same games with OSQP's own arithmetic (csrc/dgsqp_osqp_xl.h, round 5) at REDUCED batch sizes: at reg = 0 the restated OSQP
    # runs into its 4,000-iteration limit on most QPs of the merge (3,400 ADMM iterations per QP on average), a solve costs 30 x the exact QP's
    dict(tag='configs[2] size, solvable game --qp osqp, B=4096', workload='kb_curve3_N25', qp='osqp', batch=4096, timeout=60, **_ONE),
    dict(tag='configs[4] --qp osqp, reduced batch B=1024', workload='merge6_N25', qp='osqp', batch=1024, timeout=120, **_ONE),
    dict(tag='configs[3] --qp osqp, reduced batch B=1024', workload='kb_f1_N50', qp='osqp', batch=1024, timeout=60, **_ONE),
    # BASELINE configs[0], the reference's own CPU-runnable case: ONE scenario of the N = 15 chicane game per launch, twenty launches one after the
    # other -- `value` is then solves per second of a caller that solves sample by sample as the reference's scripts do (1 / value = latency)
    dict(tag='configs[0] one scenario per launch (1 / value = latency of a solve)', workload='kb_chicane_N15', batch=1, steps=20, warmup=5, pipeline=1, batches=4, group=1, timeout=30),
    # ... and the same game in batches of 1,024 (twenty in one launch, like the headline) on the product build and on the two-workgroups-per-CU build
    # (row N1: libdgsqp_hip_b256.so, DGSQP(workgroups_per_cu=2))
    dict(tag="configs[0]'s game, B=1024 x 20, one 512-thread workgroup per CU", workload='kb_chicane_N15', timeout=30),
    dict(tag="configs[0]'s game, B=1024 x 20, two 256-thread workgroups per CU", workload='kb_chicane_N15', wg_per_cu=2, timeout=30),
    # (the circuit game with OSQP -- 99 % of its solves fail, as the reference's would -- and the opt-in fp32 storage of the ADMM iteration's K^-1
    # are timed by tools/measure_round6.sh, not by every default run: profiles/r06_bench_kb_barc3_N25_B4096_qp_osqp.json, ..._qp_osqp_mixed.json)
)
RECORD_KEYS = ('value', 'unit', 'steps', 'warmup', 'ms_per_step', 'dtype', 'config', 'mean_iters', 'mean_iters_all', 'mean_qp_solves', 'converged_fraction',
               'status_fractions', 'roofline', 'value_single_launch', 'value_host_inclusive', 'value_host_inclusive_grouped', 'elapsed_s')
SIDECAR = os.path.join(ROOT, 'bench_workloads.json')
LINE_LIMIT = 4096           # bytes of the ONE stdout line (the driver keeps an 8 KB tail of the run's output)


T_START = time.perf_counter()


def run_leg(leg, args, timeout):
    """One extra leg as a fresh child process (never an exec: this process has initialised the GPU): `bench.py --workload ... --extras off
    --line full`, no one-at-a-time / host / cpu legs.  Returns the child's full record; raises on a non-zero exit, a timeout (the child and
    only the child is killed) or an unparseable line."""
    cmd = [sys.executable, os.path.abspath(__file__), '--gpus', '1', '--workload', leg['workload'], '--qp', leg.get('qp', 'active_set'),
           '--steps', str(leg.get('steps', args.steps)), '--warmup', str(leg.get('warmup', args.warmup)), '--batch', str(leg.get('batch', args.batch)),
           '--pipeline', str(leg.get('pipeline', args.pipeline)), '--batches', str(leg.get('batches', 0)), '--group', str(leg.get('group', 0)),
           '--coop', args.coop, '--single-steps', '0', '--host-steps', '0', '--cpu-sample', '0', '--extras', 'off', '--line', 'full']
    if leg.get('mixed_precision'):
        cmd.append('--mixed-precision')
    if leg.get('wg_per_cu', 1) != 1:
        cmd += ['--wg-per-cu', str(leg['wg_per_cu'])]
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE')}
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    if out.returncode != 0:
        raise RuntimeError(f'exit code {out.returncode}: {out.stderr.strip()[-300:]}')
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    if not lines:
        raise RuntimeError('the child printed no JSON line')
    return json.loads(lines[-1])


def _sig(x, digits=5):
    return float(f'{x:.{digits}g}') if isinstance(x, float) else x


def compact_line(line, records=None, sidecar=None):
    """The ONE stdout line: the contract's keys, `config`, `roofline` and `cpu_baseline` trimmed to their numbers, the other throughput
    figures of the headline, and -- default invocation -- a summary [tag, scenarios/s, converged fraction, roofline.frac] per extra leg.
    Everything else (flop model terms, single-launch detail, status fractions, the legs' full records) is in the side file."""
    c, r = line['config'], line['roofline']
    out = {k: line[k] for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data')}
    out['config'] = {k: c[k] for k in ('workload', 'batch_per_gpu', 'batch_total', 'n', 'n_c', 'parallelism', 'layout', 'qp_method', 'reg', 'distinct_batches',
                                       'batches_per_launch', 'launches_in_flight', 'cooperative_line_search', 'workgroups_per_cu') if k in c}
    out['roofline'] = {'bound': r['bound'], 'achieved': _sig(r['achieved']), 'peak': r['peak'], 'unit': r['unit'], 'frac': _sig(r['frac']),
                       'frac_is': 'algorithmic flops (SURVEY 8d model) / HIP-event time / peak', 'frac_executed_upper_bound': _sig(r.get('frac_executed_upper_bound')),
                       'traffic': r.get('traffic'), 'kernel': r['kernel'], 'kernel_ms': _sig(r['kernel_ms'], 7), 'launches_timed': r.get('launches_timed'),
                       'solves_per_launch': r.get('solves_per_launch'), 'hbm': {k: _sig(v) for k, v in r['hbm'].items()}}
    if r.get('traffic_note'):
        out['roofline']['traffic_note'] = r['traffic_note'][:120]
    cb = line.get('cpu_baseline')
    if cb:
        out['cpu_baseline'] = {'value': _sig(cb['value']), 'unit': cb['unit'], 'cores': cb['cores'], 'kind': cb['kind'], 'sample': cb.get('sample_short', ''),
                               'value_wall': _sig(cb.get('value_wall')), 'value_one_core': _sig(cb.get('value_one_core'))}
    elif 'cpu_baseline' in line:
        out['cpu_baseline'] = None
        out['cpu_baseline_note'] = line.get('cpu_baseline_note')
    for k in ('value_single_launch', 'value_host_inclusive', 'value_host_inclusive_grouped', 'mean_iters', 'mean_iters_all', 'mean_qp_solves', 'converged_fraction', 'elapsed_s'):
        out[k] = _sig(line.get(k), 6)
    if line.get('elapsed_s_per_rank') and line['n_gpus'] > 1:
        out['elapsed_s_per_rank'] = [_sig(x) for x in line['elapsed_s_per_rank']]
    if records is not None:
        osqp = [w for w in records if w.get('tag') == 'configs[1] --qp osqp' and 'value' in w]
        out['value_qp_osqp'] = _sig(osqp[0]['value'], 6) if osqp else None
        out['workloads'] = [[w['tag'], _sig(w.get('value')), _sig(w.get('converged_fraction')), _sig((w.get('roofline') or {}).get('frac'))] if 'value' in w
                            else [w['tag'], None, (w.get('error') or w.get('skipped') or '')[:60], None] for w in records]
        out['workloads_columns'] = ['tag', 'scenarios/s', 'converged_fraction | why missing', 'roofline.frac']
    if sidecar:
        out['sidecar'] = os.path.relpath(sidecar, ROOT)
    return out


def write_sidecar(path, headline, records):
    try:
        tmp = path + '.tmp'
        with open(tmp, 'w') as f:
            json.dump({'headline': headline, 'workloads': records}, f, indent=1)
        os.replace(tmp, path)
        return path
    except OSError:
        return None


def main():
    args = parse_args()
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        spawn_ranks(args.gpus, sys.argv[1:])
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('DGSQP_BENCH_DEVICE', os.environ.get('LOCAL_RANK', '0')))      # (the override lets a 1-GPU box rehearse spawn + rendezvous of the multi-rank path -- up to ncclCommInitRank, which refuses two ranks on one device)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.gpus != world:
        sys.stderr.write(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run --nproc-per-node {args.gpus} '
                         f'or run `python bench.py --gpus {args.gpus}` without a launcher\n')
        sys.exit(2)
    line = run_workload(args, rank, local_rank, world)
    if rank != 0:
        return
    if args.line == 'full':
        print(json.dumps(line), flush=True)
        return
    extras = (args.extras == 'auto' and world == 1 and args.workload == 'dyn_curve_N25' and args.qp == 'active_set' and args.scaling == 'weak'
              and args.batch == 1024 and args.reg is None and args.eig_floor is None)
    records, sidecar = None, None
    if extras:
        # the headline is measured and safe in the side file before any leg starts; its handles are destroyed (run_workload)
        records = [dict({k: line.get(k) for k in RECORD_KEYS}, tag='configs[1] (the headline)')]
        sidecar = write_sidecar(SIDECAR, line, records)
        for leg in EXTRA_LEGS:
            tag = leg['tag']
            left = args.extras_budget - (time.perf_counter() - T_START)
            if left < 15.0:
                records.append(dict(tag=tag, skipped=f'not started: {time.perf_counter() - T_START:.0f} s into the run, --extras-budget {args.extras_budget:.0f} s'))
                continue
            t0 = time.perf_counter()
            try:
                rec = run_leg(leg, args, min(float(leg.get('timeout', 90)), left))
                records.append(dict({k: rec.get(k) for k in RECORD_KEYS}, tag=tag, wall_s_incl_setup=time.perf_counter() - t0))
            except subprocess.TimeoutExpired as e:
                records.append(dict(tag=tag, error=f'timeout after {e.timeout:.0f} s (child killed)'))
            except Exception as e:           # a leg that fails must not take the headline down with it: say so in the line
                records.append(dict(tag=tag, error=f'{type(e).__name__}: {e}'))
            sidecar = write_sidecar(SIDECAR, line, records)
    out = compact_line(line, records, sidecar)
    text = json.dumps(out, separators=(',', ':'))
    if len(text) > LINE_LIMIT:              # never again a line the driver cannot keep: shed the optional parts, most verbose first
        for k in ('workloads_columns', 'elapsed_s_per_rank', 'workloads'):
            out.pop(k, None)
            text = json.dumps(out, separators=(',', ':'))
            if len(text) <= LINE_LIMIT:
                break
    print(text, flush=True)


if __name__ == '__main__':
    main()
