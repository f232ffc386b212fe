"""Multi-GPU plumbing: the Monte-Carlo batch shards embarrassingly (every scenario is an independent
``solve()``, reference DGSQP/solvers/DGSQP.py:302-310), one process per GPU, no collective on the data
path; the only exchange is ONE gather of a fixed-size per-scenario stats record for the convergence
statistics (RCCL all_gather over xGMI when the backend is 'nccl', gloo in the CPU tests).

torch.distributed is used as plumbing only and imported lazily, so the solver itself never needs torch.
"""
from __future__ import annotations

import numpy as np

STATS_FIELDS = ('status', 'num_iters', 'qp_solves', 'p_feas', 'comp', 'stat')


def shard_range(B: int, rank: int, world: int):
    """Contiguous block of scenario indices owned by ``rank`` (first ``B % world`` ranks get one extra)."""
    base, rem = divmod(B, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_stats(res: dict) -> np.ndarray:
    """[B_local, 6] float64 record per scenario: status, iterations, QP solves, p_feas, comp, stat."""
    return np.column_stack([res['status'].astype(np.float64), res['num_iters'].astype(np.float64),
                            res['qp_solves'].astype(np.float64), res['cond']]).astype(np.float64)


def gather_stats(local: np.ndarray, device=None) -> np.ndarray:
    """all_gather of the per-scenario stats over the default process group; returns [B_total, 6] in rank order.
    Shards may have different sizes: sizes are exchanged first, payloads are padded to the maximum."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    dev = device if device is not None else torch.device('cpu')
    nloc = torch.tensor([local.shape[0]], dtype=torch.int64, device=dev)
    sizes = [torch.zeros_like(nloc) for _ in range(world)]
    dist.all_gather(sizes, nloc)
    sizes = [int(s.item()) for s in sizes]
    nmax = max(sizes)
    buf = torch.zeros((nmax, local.shape[1]), dtype=torch.float64, device=dev)
    if local.shape[0]:
        buf[:local.shape[0]] = torch.from_numpy(np.ascontiguousarray(local)).to(dev)
    out = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    return np.concatenate([o[:s].cpu().numpy() for o, s in zip(out, sizes)], axis=0)


def summarize(stats: np.ndarray) -> dict:
    """Convergence statistics the way scripts/process_data_curve.py:44-53,99-110 reports them:
    mean iterations / QP solves over CONVERGED samples, plus the all-sample means."""
    status = stats[:, 0].astype(int)
    conv = status <= 1
    out = dict(n=int(len(status)), converged=float(conv.mean()) if len(status) else 0.0,
               conv_abs_tol=float((status == 0).mean()) if len(status) else 0.0,
               conv_rel_tol=float((status == 1).mean()) if len(status) else 0.0,
               max_it=float((status == 2).mean()) if len(status) else 0.0,
               diverged=float((status == 3).mean()) if len(status) else 0.0,
               qp_fail=float((status == 4).mean()) if len(status) else 0.0,
               mean_iters_all=float(stats[:, 1].mean()) if len(status) else 0.0,
               mean_qp_solves_all=float(stats[:, 2].mean()) if len(status) else 0.0)
    out['mean_iters_converged'] = float(stats[conv, 1].mean()) if conv.any() else float('nan')
    out['mean_qp_solves_converged'] = float(stats[conv, 2].mean()) if conv.any() else float('nan')
    return out
