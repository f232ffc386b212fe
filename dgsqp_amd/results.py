"""Per-sample result records of a batched solve in the layout the reference's post-processing reads.

scripts/DGSQP_ALGAMES_monte_carlo_curve.py:480-500 stores, for every Monte-Carlo sample,
``dict(solve_info=<DGSQP.solve() return value>, params=<DGSQPParams>, init=<joint VehicleStates>)`` in a list under the key
``'sqgames'`` and pickles it as ``data_c_<c>_N_<N>.pkl``; scripts/process_data_curve.py:44-53 (and process_data_comp.py,
process_data_merge.py:31-40) read ``solve_info['status' | 'msg' | 'cond']['p_feas'] | 'num_iters' | 'time' | 'iter_data'[i]['qp_solves']``.
``solve_infos`` turns the arrays of ``DGSQP.solve_batch`` into those dictionaries, ``save_monte_carlo`` writes the pickle.
"""
from __future__ import annotations

import copy
import pickle
from typing import List, Optional

import numpy as np


def solve_infos(res: dict, wall_time: Optional[float] = None) -> List[dict]:
    """One ``solve_info`` dictionary (keys of DGSQP.py:495-502) per scenario of a ``solve_batch`` result.
    ``iter_data`` holds one summary record per sample (the kernels keep totals only), so that
    ``sum(d['qp_solves'] for d in iter_data)`` (process_data_curve.py:50) is the sample's number of QP solves;
    ``time`` is the batch's wall time divided by its size when ``wall_time`` is given."""
    B = len(res['status'])
    t = float(wall_time) / B if wall_time is not None else float(res.get('kernel_ms', 0.0)) * 1e-3 / max(B, 1)
    out = []
    for b in range(B):
        cond = dict(p_feas=float(res['cond'][b, 0]), comp=float(res['cond'][b, 1]), stat=float(res['cond'][b, 2]))
        out.append(dict(time=t, num_iters=int(res['num_iters'][b]), status=bool(res['status'][b] <= 1),
                        cost=[float(c) for c in res['cost'][b]], cond=cond,
                        iter_data=[dict(cond=cond, u_sol=res['u'][b], l_sol=res['l'][b], qp_solves=int(res['qp_solves'][b]), it_time=t)],
                        msg=res['msg'][b], init=dict(u=None, l=None)))
    return out


def save_monte_carlo(path, res: dict, params, init_states=None, key: str = 'sqgames', wall_time: Optional[float] = None, extra=None):
    """Pickle ``{key: [dict(solve_info=..., params=..., init=...), ...]}`` (curve.py:486-500).  ``init_states``: per sample
    the joint ``VehicleState`` list the script stores, or None."""
    infos = solve_infos(res, wall_time)
    recs = [dict(solve_info=si, params=copy.deepcopy(params), init=None if init_states is None else init_states[b])
            for b, si in enumerate(infos)]
    data = {key: recs}
    if extra:
        data.update(extra)
    with open(path, 'wb') as f:
        pickle.dump(data, f)
    return data


def summarize_like_process_data(recs: List[dict]) -> dict:
    """The table of process_data_curve.py:37-110 for one solver: counts and the mean / std over CONVERGED samples."""
    conv = [r['solve_info'] for r in recs if r['solve_info']['status']]
    iters = [s['num_iters'] for s in conv]
    solves = [int(np.sum([d['qp_solves'] for d in s['iter_data']])) for s in conv]
    msgs = [r['solve_info']['msg'] for r in recs]
    return dict(converged=len(conv), failed=sum(m in ('diverged', 'qp_fail') for m in msgs), max_it=sum(m == 'max_it' for m in msgs),
                avg_iters=float(np.mean(iters)) if iters else float('nan'), std_iters=float(np.std(iters)) if iters else float('nan'),
                avg_solves=float(np.mean(solves)) if solves else float('nan'), std_solves=float(np.std(solves)) if solves else float('nan'))
