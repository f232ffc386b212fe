"""Build libdgsqp_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import pathlib
import subprocess
import sys

HERE = pathlib.Path(__file__).resolve().parent
SRC = HERE / 'dgsqp_api.hip'
DEPS = [SRC, HERE / 'dgsqp_comm.h', HERE / 'dgsqp_layout.h', HERE / 'dgsqp_device.h', HERE / 'dgsqp_eval.h', HERE / 'dgsqp_solve.h', HERE / 'dgsqp_qp.h', HERE / 'dgsqp_osqp.h', HERE / 'dgsqp_osqp_xl.h', HERE / 'dgsqp_sampler.h', HERE / 'dgsqp_pid.h', HERE / 'dgsqp_xl.h', HERE / 'dgsqp_solve_v2.h',
        HERE.parent.parent / 'include' / 'dgsqp.h']
OUT = HERE / 'libdgsqp_hip.so'


def build(force: bool = False, verbose: bool = False) -> pathlib.Path:
    if not force and OUT.exists() and OUT.stat().st_mtime >= max(d.stat().st_mtime for d in DEPS):
        return OUT
    cmd = ['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-o', str(OUT), str(SRC)]
    if verbose:
        cmd.append('-Rpass-analysis=kernel-resource-usage')
    subprocess.check_call(cmd, cwd=str(HERE))
    return OUT


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose='-v' in sys.argv))
