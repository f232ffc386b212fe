"""Multi-GPU plumbing: the Monte-Carlo batch shards embarrassingly (every scenario is an independent
``solve()``, reference DGSQP/solvers/DGSQP.py:302-310), one process per GPU, no collective on the data
path; the only exchange is ONE all-gather of a fixed 88-byte per-scenario record for the convergence
statistics -- ``ncclAllGather`` over xGMI issued by the HIP library itself (``dgsqp_gather_stats``,
include/dgsqp.h; RCCL communicator owned by the solver handle).  No PyTorch anywhere: ranks find each
other through the environment a launcher sets (RANK / LOCAL_RANK / WORLD_SIZE, e.g. ``torch.distributed.run``
or ``bench.py --gpus N`` itself) and exchange the 128-byte ncclUniqueId through a file on the node.
"""
from __future__ import annotations

import ctypes as C
import os
import time

import numpy as np

from . import _ffi

STATS_FIELDS = ('status', 'num_iters', 'qp_solves', 'p_feas', 'comp', 'stat')
RECORD_DTYPE = np.dtype([('status', np.int32), ('iters', np.int32), ('qp_solves', np.int32), ('rank', np.int32),
                         ('p_feas', np.float64), ('comp', np.float64), ('stat', np.float64), ('cost', np.float64, (_ffi.MAX_AGENTS,))])
assert RECORD_DTYPE.itemsize == 88 == C.sizeof(_ffi.StatRecordT)


def shard_range(B: int, rank: int, world: int):
    """Contiguous block of scenario indices owned by ``rank`` (first ``B % world`` ranks get one extra)."""
    base, rem = divmod(B, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_indices(B: int, rank: int, world: int, mode: str = 'contiguous') -> np.ndarray:
    """Scenario indices owned by ``rank``.  ``contiguous``: the block of ``shard_range``.  ``interleaved``: rank, rank + world,
    rank + 2 world, ... (iteration counts that correlate with the sample index -- e.g. a sampler that
    sweeps a parameter -- then spread evenly over the ranks instead of loading one of them).  Both give every rank the same
    number of scenarios up to one, the largest being ``padded_shard_size``."""
    if mode == 'contiguous':
        lo, hi = shard_range(B, rank, world)
        return np.arange(lo, hi)
    if mode == 'interleaved':
        return np.arange(rank, B, world)
    raise ValueError(f'unknown shard mode {mode!r}')


def unshard(parts, B: int, world: int, mode: str = 'contiguous') -> np.ndarray:
    """Inverse of ``shard_indices`` for per-rank result arrays (rank order): the array in the original scenario order."""
    first = np.asarray(parts[0])
    out = np.empty((B,) + first.shape[1:], first.dtype)
    for r, part in enumerate(parts):
        idx = shard_indices(B, r, world, mode)
        out[idx] = np.asarray(part)[:len(idx)]
    return out


def padded_shard_size(B: int, world: int) -> int:
    """Size of the largest shard = the per-rank record count of the (equal-count) all-gather."""
    return -(-B // world)


def pack_stats(res: dict) -> np.ndarray:
    """[B_local, 6] float64 record per scenario: status, iterations, QP solves, p_feas, comp, stat."""
    return np.column_stack([res['status'].astype(np.float64), res['num_iters'].astype(np.float64),
                            res['qp_solves'].astype(np.float64), res['cond']]).astype(np.float64)


def records_from_results(res: dict, rank: int = 0) -> np.ndarray:
    """Host-side construction of the 88-byte records (what ``dg_pack_stats_kernel`` builds on the device)."""
    B = len(res['status'])
    rec = np.zeros(B, RECORD_DTYPE)
    rec['status'], rec['iters'], rec['qp_solves'], rec['rank'] = res['status'], res['num_iters'], res['qp_solves'], rank
    rec['p_feas'], rec['comp'], rec['stat'] = res['cond'][:, 0], res['cond'][:, 1], res['cond'][:, 2]
    if 'cost' in res:
        m = min(_ffi.MAX_AGENTS, res['cost'].shape[1])
        rec['cost'][:, :m] = res['cost'][:, :m]
    return rec


def pad_records(rec: np.ndarray, B_pad: int) -> np.ndarray:
    """Equal-count payload of the all-gather: padding rows carry status -1."""
    out = np.zeros(B_pad, RECORD_DTYPE)
    out['status'] = -1
    out[:len(rec)] = rec
    return out


def stats_from_records(rec: np.ndarray) -> np.ndarray:
    """Gathered records (rank order, padding rows status -1) -> [B_total, 6] float64 table ``summarize`` takes."""
    rec = rec[rec['status'] >= 0]
    return np.column_stack([rec['status'], rec['iters'], rec['qp_solves'], rec['p_feas'], rec['comp'], rec['stat']]).astype(np.float64)


def costs_from_records(rec: np.ndarray, M: int) -> np.ndarray:
    """Gathered records -> [B_total, M] costs of every agent (``f_J``, DGSQP.py:492), padding rows dropped."""
    if not 1 <= M <= _ffi.MAX_AGENTS:
        raise ValueError(f'M must be in 1..{_ffi.MAX_AGENTS}')
    return np.ascontiguousarray(rec[rec['status'] >= 0]['cost'][:, :M])


# ---------------------------------------------------------------------------------------------------------
# rendezvous: rank 0 publishes the ncclUniqueId in a file on the node, the others poll for it
# ---------------------------------------------------------------------------------------------------------
_COMM_SEQ = 0      # communicators this process has built: every rank builds them in the same order, so the number names one rendezvous


def launch_tag() -> bytes:
    """16 bytes naming THIS launch (the common parent of the ranks: pid + start time): written in front of the id, checked by
    the readers -- a file left behind by another launch under the same path is never taken for this launch's id."""
    import hashlib
    ppid = os.getppid()
    try:
        with open(f'/proc/{ppid}/stat') as f:
            started = f.read().rsplit(')', 1)[1].split()[19]
    except (OSError, IndexError):
        started = '0'
    key = os.environ.get('DGSQP_LAUNCH_TAG') or f"{ppid}_{started}_{os.environ.get('MASTER_PORT', '0')}_{os.environ.get('TORCHELASTIC_RUN_ID', 'none')}"
    return hashlib.blake2s(key.encode(), digest_size=16).digest()


def rendezvous_path() -> str:
    explicit = os.environ.get('DGSQP_RENDEZVOUS')
    if explicit:
        return explicit
    # the launcher (torch.distributed.run agent, or bench.py spawning its own ranks) is the common parent of all ranks: its pid plus its
    # start time name this launch and no earlier one -- a file left behind by a crashed run can never be mistaken for this run's id
    ppid = os.getppid()
    try:
        with open(f'/proc/{ppid}/stat') as f:
            started = f.read().rsplit(')', 1)[1].split()[19]      # field 22 (starttime, clock ticks since boot)
    except (OSError, IndexError):
        started = '0'
    tag = f"{os.environ.get('MASTER_PORT', '0')}_{os.environ.get('TORCHELASTIC_RUN_ID', 'none')}_{ppid}_{started}"
    return os.path.join(os.environ.get('TMPDIR', '/tmp'), f'dgsqp_rccl_{tag}.id')


def exchange_unique_id(rank: int, world: int, make_id, path: str = None, timeout: float = 120.0, seq: int = 0) -> bytes:
    """Rank 0 calls ``make_id()`` (-> 128 bytes) and publishes [launch tag (16 bytes), sequence number (8), id (128)] atomically
    (stale files are unlinked first); the others wait for a file that carries THIS launch's tag and THIS rendezvous' number."""
    path = (path or rendezvous_path()) + (f'.{seq}' if seq else '')
    head = launch_tag() + int(seq).to_bytes(8, 'little')
    if rank == 0:
        uid = bytes(make_id())
        assert len(uid) == 128
        try:
            os.remove(path)
        except OSError:
            pass
        tmp = f'{path}.{os.getpid()}.tmp'
        with open(tmp, 'wb') as f:
            f.write(head + uid)
        os.replace(tmp, path)
        return uid
    deadline = time.time() + timeout
    while time.time() < deadline:
        try:
            with open(path, 'rb') as f:
                blob = f.read()
            if len(blob) == 152 and blob[:24] == head:
                return blob[24:]
        except FileNotFoundError:
            pass
        time.sleep(0.01)
    raise TimeoutError(f'rank {rank}: no ncclUniqueId of this launch at {path} after {timeout} s')


class Communicator:
    """The solver handle's RCCL communicator (one process per GPU).  ``world == 1`` needs no peer."""

    def __init__(self, solver, rank: int, world: int, path: str = None):
        self.solver, self.rank, self.world = solver, rank, world
        self._lib, self._h = solver._lib, solver._h
        global _COMM_SEQ
        self._seq, _COMM_SEQ = _COMM_SEQ, _COMM_SEQ + 1
        self._path = path or rendezvous_path()

        def make_id():
            buf = C.create_string_buffer(128)
            if self._lib.dgsqp_comm_unique_id(buf) != 0:
                raise RuntimeError('dgsqp_comm_unique_id failed: ' + (self._lib.dgsqp_last_error(None) or b'').decode())
            return buf.raw
        # RCCL prints a version banner on stdout when it initialises; callers such as bench.py own stdout (ONE JSON line), so
        # the file descriptor is pointed at stderr for the duration of the initialisation.
        import sys
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            uid = exchange_unique_id(rank, world, make_id, self._path, seq=self._seq) if world > 1 else make_id()
            rc = self._lib.dgsqp_comm_init(self._h, uid, rank, world)
            C.CDLL(None).fflush(None)         # RCCL writes through C stdio: flush its buffer while fd 1 still points at stderr
        finally:
            os.dup2(saved, 1)
            os.close(saved)
        if rc != 0:
            raise RuntimeError('dgsqp_comm_init failed: ' + self._lib.dgsqp_last_error(self._h).decode())

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(self._lib.dgsqp_last_error(self._h).decode())

    def barrier(self):
        self._check(self._lib.dgsqp_comm_barrier(self._h))

    def allreduce_max(self, values) -> np.ndarray:
        v = np.ascontiguousarray(values, dtype=np.float64).copy()
        self._check(self._lib.dgsqp_comm_allreduce_max(self._h, _ffi.dptr(v), v.size))
        return v

    def gather_stats(self, B_pad: int) -> np.ndarray:
        """ONE ncclAllGather of the 88-byte records of the handle's last solve; returns the [world * B_pad] record array."""
        out = np.zeros(self.world * B_pad, RECORD_DTYPE)
        self._check(self._lib.dgsqp_gather_stats(self._h, int(B_pad), out.ctypes.data_as(C.c_void_p)))
        return out

    def close(self):
        if self.world > 1:
            self.barrier()
        self._lib.dgsqp_comm_destroy(self._h)
        if self.rank == 0 and self.world > 1:
            try:
                os.remove(self._path + (f'.{self._seq}' if self._seq else ''))
            except OSError:
                pass


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
               time_limit=float((status == 5).mean()) if len(status) else 0.0,
               mean_iters_all=float(stats[:, 1].mean()) if len(status) else 0.0,
               mean_qp_solves_all=float(stats[:, 2].mean()) if len(status) else 0.0)
    out['mean_iters_converged'] = float(stats[conv, 1].mean()) if conv.any() else float('nan')
    out['mean_qp_solves_converged'] = float(stats[conv, 2].mean()) if conv.any() else float('nan')
    return out
