#!/usr/bin/env python3
"""Headline benchmark: Monte-Carlo scenarios/s (= DG-SQP solves/s) of the 2-agent N=25 game.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one pass of the hot path (DGSQP.solve(), reference DGSQP/solvers/DGSQP.py:302-507) over one batch
of synthetic random-initial-condition scenarios that is already resident in HBM (dgsqp_stage_inputs).  Every rank
owns its own batch of the same size (weak scaling, no data-path collective); the only exchange is ONE all_gather of
the per-scenario convergence record (RCCL over xGMI with the nccl backend).  Rank 0 prints ONE JSON line.

torch is plumbing here (process group, barrier, gather); the solver itself is the ctypes/HIP library.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')     # one hardware queue per in-flight batch (HIP default: 4)

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # BASELINE.json configs[1]
    'dyn_curve_N25': dict(desc='2-agent dynamic-bicycle (Pacejka, rk4 M=10) curve track, N=25, fp64', kind='dyn', track='curve', N=25, reg=1e-3),
    # the reference's own Monte-Carlo experiment (scripts/DGSQP_ALGAMES_monte_carlo_curve.py), kinematic bicycle
    'kb_curve_N25': dict(desc='2-agent kinematic-bicycle (euler) curve track, N=25, reg=0 (curve.py:161), fp64', kind='kb', track='curve', N=25, reg=0.0),
    'kb_chicane_N25': dict(desc='2-agent kinematic-bicycle (euler) chicane track, N=25, reg=1e-3 (chicane.py:164), fp64', kind='kb', track='chicane', N=25, reg=1e-3),
    # other Monte-Carlo scripts of the reference at their own sizes (not BASELINE's metric; for the DESIGN.md table)
    'kb_barc2_N15': dict(desc='2-agent kinematic-bicycle race on the L_track_barc circuit, N=15, reg=0 (DGSQP_comp_monte_carlo.py), fp64', kind='barc', M=2, N=15, reg=0.0),
    'merge_N20': dict(desc='3-car highway merge, kinematic unicycles rk3, N=20, reg=0 (DGSQP_merge_monte_carlo.py), big layout, fp64', kind='merge', N=20, reg=0.0),
    'kb_curve_N50': dict(desc='2-agent kinematic-bicycle curve track, N=50, reg=1e-3 (BASELINE configs[3] size on the curve track), XL layout, fp64', kind='kb', track='curve', N=50, reg=1e-3),
    'kb_curve3_N25': dict(desc='3-agent kinematic-bicycle curve track, N=25, reg=1e-3 (DGSQP_monte_carlo_agents.py M=3 N=25 = BASELINE configs[2] size), XL layout, fp64', kind='kb', track='curve', N=25, M=3, reg=1e-3),
}


def make_game(name, reg=None):
    from dgsqp_amd.montecarlo import dynamic_racing_game, kinematic_racing_game
    w = WORKLOADS[name]
    reg = w['reg'] if reg is None else reg
    if w['kind'] == 'dyn':
        return dynamic_racing_game(w['track'], N=w['N'], rk4_substeps=10, reg=reg)
    if w['kind'] == 'barc':
        from dgsqp_amd.montecarlo import barc_racing_game
        return barc_racing_game(N=w['N'], M=w['M'], reg=reg)
    if w['kind'] == 'merge':
        from dgsqp_amd.montecarlo import merge_game
        return merge_game(N=w['N'], reg=reg)
    return kinematic_racing_game(w['track'], N=w['N'], reg=reg, M=w.get('M', 2))


def algorithmic_bytes_per_solve(d):
    """inputs x0[n_q] + u_ws[n]; outputs u[n] + l[n_c] + x[(N+1) n_q] + cond[3] + cost[M] (fp64)
    + three int32 (status, iterations, QP solves)."""
    return 8 * (d.n_q + d.n + d.n + d.n_c + (d.N + 1) * d.n_q + 3 + d.M) + 12


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--batch', type=int, default=1024, help='scenarios per GPU per step')
    ap.add_argument('--workload', default='dyn_curve_N25', choices=sorted(WORKLOADS))
    ap.add_argument('--cpu-sample', type=int, default=16, help='scenarios timed on the host for cpu_baseline (0 disables)')
    ap.add_argument('--pipeline', type=int, default=5,
                    help='independent batches in flight per GPU (each on its own handle / HIP stream); 1 = strictly one launch at a time')
    ap.add_argument('--reg', type=float, default=None, help='DGSQPParams.reg (default: the value of the workload)')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    import torch
    import torch.distributed as dist
    distributed = world > 1 or os.environ.get('DGSQP_BENCH_FORCE_DIST') == '1'   # the env knob lets a 1-GPU box exercise the RCCL path
    if distributed:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    dev = torch.device('cuda', local_rank)

    from dgsqp_amd import _ffi
    from dgsqp_amd.montecarlo import sample_scenarios
    from dgsqp_amd.sharding import gather_stats, pack_stats, summarize
    from dgsqp_amd.solver import DGSQP
    import ctypes as C

    game = make_game(args.workload, args.reg)
    solver = DGSQP(*game.solver_args(), print_method=None, device=local_rank)
    d = solver.dims
    B = args.batch
    x0, u_tm = sample_scenarios(game, B, seed=1 + rank)          # rejection sampling happens before any timing
    u_am = np.ascontiguousarray(solver._to_agent_major(u_tm))
    lib, h = solver._lib, solver._h
    # Steps are independent batches.  With --pipeline P > 1 they are issued round-robin on P handles (own stream, workspace and
    # result buffers each) without waiting in between: the workgroups of step i+1 take over the compute units while step i
    # drains its slowest scenarios.  Every step still solves its whole batch; all of them are complete at the closing fence.
    P = max(1, args.pipeline)
    solvers = [solver] + [DGSQP(*game.solver_args(), print_method=None, device=local_rank) for _ in range(P - 1)]
    handles = [sv._h for sv in solvers]
    for hh in handles:
        assert lib.dgsqp_stage_inputs(hh, B, _ffi.dptr(x0), _ffi.dptr(u_am)) == 0, lib.dgsqp_last_error(hh)

    tm = _ffi.TimingT()
    for _ in range(args.warmup):
        for hh in handles:
            assert lib.dgsqp_solve_staged(hh, C.byref(tm)) == 0, lib.dgsqp_last_error(hh)

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    kernel_ms = []
    busy = [False] * P
    order = []                                        # handles in launch order (oldest first)

    def wait(i):
        assert lib.dgsqp_wait(handles[i], C.byref(tm)) == 0, lib.dgsqp_last_error(handles[i])
        kernel_ms.append(tm.kernel_ms)            # HIP events around that launch on its own stream
        busy[i] = False
        order.remove(i)

    fence()
    t0 = time.perf_counter()
    last = None
    for step in range(args.steps):
        # start the next batch when the previous launch has handed out its last scenario (its workgroups begin to exit and
        # free compute units) -- not earlier, or two launches would share the GPU from the start and both grow tails
        if last is not None and P > 1:
            deadline = time.perf_counter() + 600.0           # never spin forever on a launch that died
            while not lib.dgsqp_draining(handles[last]) and time.perf_counter() < deadline:
                time.sleep(0.0002)
        if all(busy):
            wait(order[0])
        i = busy.index(False)
        assert lib.dgsqp_launch_staged(handles[i]) == 0, lib.dgsqp_last_error(handles[i])
        busy[i] = True
        order.append(i)
        last = i
    while order:
        wait(order[0])
    fence()
    elapsed = time.perf_counter() - t0
    h = handles[last]                                 # results of the last step
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # results of the last step + the single stats gather
    out = dict(status=np.empty(B, np.int32), num_iters=np.empty(B, np.int32), qp_solves=np.empty(B, np.int32), cond=np.empty((B, 3)))
    assert lib.dgsqp_fetch_results(h, None, None, None, _ffi.iptr(out['status']), _ffi.iptr(out['num_iters']),
                                   _ffi.iptr(out['qp_solves']), _ffi.dptr(out['cond']), None) == 0
    stats = gather_stats(pack_stats(out), device=dev)
    kms = float(np.mean(kernel_ms))
    if distributed:
        t = torch.tensor([kms], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        kms = float(t.item())

    if rank == 0:
        # HBM traffic from PMC counters is collected offline in separate rocprofv3 --pmc passes (gpurun refuses mixed runs)
        # and so is the fp64 instruction mix (SQ counters, profiles/*_pmc_sq_*.json) behind the vector-ALU figure below
        traffic, flop_per_solve = None, None
        try:
            import glob
            for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_*.json'))):
                pm = json.load(open(f))
                if pm.get('workload') == args.workload and pm.get('batch_per_gpu') == B:
                    traffic = pm.get('traffic_bytes_per_launch', traffic)
                    flop_per_solve = pm.get('fp64_flop_per_solve_upper_bound', flop_per_solve)
        except Exception:
            pass
        total = B * world * args.steps
        value = total / elapsed
        bytes_per_launch = algorithmic_bytes_per_solve(d) * B
        achieved = bytes_per_launch / (kms * 1e-3) / 1e9
        summ = summarize(stats)
        line = {
            'metric': 'Monte-Carlo scenarios/sec (SQP solves/sec), 2-agent N=25' if args.workload in ('dyn_curve_N25', 'kb_curve_N25', 'kb_chicane_N25')
                      else 'Monte-Carlo scenarios/sec (SQP solves/sec)', 'value': value, 'unit': 'scenarios/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': args.workload, 'description': WORKLOADS[args.workload]['desc'], 'batch_per_gpu': B,
                       'n': int(d.n), 'n_c': int(d.n_c), 'parallelism': f'scenario-sharded x{world}',
                       'sampler': {'circuit': 'scripts/DGSQP_comp_monte_carlo.py:365-382 (seed 1+rank), PID warm start',
                                   'merge': 'scripts/DGSQP_merge_monte_carlo.py:421-480 (seed 1+rank), zero warm start'}.get(
                                       game.sampler, 'scripts/DGSQP_ALGAMES_monte_carlo_curve.py:384-467 (seed 1+rank), PID warm start'),
                       'layout': {0: 'lds', 1: 'big', 2: 'xl'}[int(d.layout)],
                       'batches_in_flight': P, 'reg': float(game.params.reg), 'eig_floor': float(solver._cparams.eig_floor)},
            'mean_iters': summ['mean_iters_converged'], 'mean_iters_all': summ['mean_iters_all'],
            'mean_qp_solves': summ['mean_qp_solves_all'], 'converged_fraction': summ['converged'],
            'status_fractions': {k: summ[k] for k in ('conv_abs_tol', 'conv_rel_tol', 'max_it', 'diverged', 'qp_fail')},
            # The path is ALU/LDS-bound (state lives in LDS for the whole solve); the HBM figure is reported as the
            # contract asks and is expected to be a tiny fraction of peak.
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': 8000.0, 'unit': 'GB/s', 'frac': achieved / 8000.0,
                         'traffic': traffic, 'kernel': 'dg_solve_kernel', 'kernel_ms': kms,
                         'algorithmic_bytes_per_solve': algorithmic_bytes_per_solve(d)},
        }
        if flop_per_solve is not None:
            # second, honest roof of this path: vector fp64 issue (256 CUs x 4 SIMDs x 32 lanes x 2 flop x 2.4 GHz = 78.6 TF/s).
            # flop per solve from the offline SQ_INSTS_VALU_{FMA,MUL,ADD,TRANS}_F64 pass (an upper bound: masked lanes count)
            line['valu_fp64'] = {'achieved_upper_bound': flop_per_solve * value / world / 1e12, 'peak': 78.6, 'unit': 'TFLOP/s',
                                 'frac_upper_bound': flop_per_solve * value / world / 1e12 / 78.6,
                                 'flop_per_solve_upper_bound': flop_per_solve}
        if world == 1 and args.cpu_sample > 0:
            from oracle import oracle            # checker/baseline only: the CPU restatement, NOT CasADi+OSQP
            oracle.build()
            ns = min(args.cpu_sample, B)
            cores = os.cpu_count() or 1
            t1 = time.perf_counter()
            oracle.solve_batch(solver._problem, solver._cparams, x0[:ns], u_am[:ns], nthreads=min(cores, ns))
            dt = time.perf_counter() - t1
            line['cpu_baseline'] = {'value': ns / dt, 'unit': 'scenarios/s', 'cores': min(cores, ns), 'kind': 'port',
                                    'sample': f'first {ns} scenarios of the same batch, oracle/dgsqp_oracle.cpp, {dt:.1f} s'}
        print(json.dumps(line), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
