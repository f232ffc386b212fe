"""Build the HIP libraries in-tree with hipcc for gfx950 (cross-compiles without a GPU).

``libdgsqp_hip.so``        the product: 512-thread workgroups, one per CU (every layout, every QP method)
``libdgsqp_hip_b256.so``   the same sources with -DDG_BLOCK=256: 256-thread workgroups and half the LDS arena, TWO per CU (row N1).  It holds
                           the explicit-inverse layouts with the active-set QP only and is what ``DGSQP_HIP_LIB`` selects for large
                           batches of small games (n <= ~64: 1.2-1.5 x the product's throughput, profiles/r06_n1_two_per_cu.txt); a single
                           scenario or an n = 100 game is faster on the product build.
Both are compiled side by side (two hipcc processes)."""
import pathlib
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = pathlib.Path(__file__).resolve().parent
SRC = HERE / 'dgsqp_api.hip'
DEPS = [SRC, HERE / 'dgsqp_comm.h', HERE / 'dgsqp_layout.h', HERE / 'dgsqp_device.h', HERE / 'dgsqp_eval.h', HERE / 'dgsqp_solve.h', HERE / 'dgsqp_qp.h', HERE / 'dgsqp_osqp.h', HERE / 'dgsqp_osqp_xl.h', HERE / 'dgsqp_sampler.h', HERE / 'dgsqp_pid.h', HERE / 'dgsqp_xl.h', HERE / 'dgsqp_solve_v2.h',
        HERE.parent.parent / 'include' / 'dgsqp.h']
OUT = HERE / 'libdgsqp_hip.so'
OUT_B256 = HERE / 'libdgsqp_hip_b256.so'
VARIANTS = ((OUT, []), (OUT_B256, ['-DDG_BLOCK=256']))


def _stale(out: pathlib.Path) -> bool:
    return not out.exists() or out.stat().st_mtime < max(d.stat().st_mtime for d in DEPS)


def _compile(out: pathlib.Path, flags, verbose: bool):
    cmd = ['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC'] + list(flags) + ['-o', str(out), str(SRC)]
    if verbose:
        cmd.append('-Rpass-analysis=kernel-resource-usage')
    subprocess.check_call(cmd, cwd=str(HERE))
    return out


def build(force: bool = False, verbose: bool = False) -> pathlib.Path:
    todo = [(o, f) for o, f in VARIANTS if force or _stale(o)]
    if todo:
        with ThreadPoolExecutor(len(todo)) as ex:
            list(ex.map(lambda of: _compile(of[0], of[1], verbose), todo))
    return OUT


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose='-v' in sys.argv))
