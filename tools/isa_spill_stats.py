"""Build container: per-FUNCTION register and spill statistics of the HIP library, from the compiler's own assembly
(hipcc -save-temps; the kernel-level `-Rpass-analysis=kernel-resource-usage` block only reports the worst call path).
For every device function: VGPRs, scratch bytes per lane, code size, instruction counts, and how many of its scratch
(spill) loads / stores sit INSIDE loops -- the dynamic cost of a spill is a scratch access per loop trip, a spill outside
the loops costs once per call.  usage: isa_spill_stats.py [min_scratch_bytes]   (writes nothing; redirect to profiles/)"""
import pathlib
import re
import subprocess
import sys
import tempfile

ROOT = pathlib.Path(__file__).resolve().parent.parent
SRC = ROOT / 'dgsqp_amd' / 'csrc' / 'dgsqp_api.hip'
min_scratch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
with tempfile.TemporaryDirectory() as td:
    r = subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-save-temps', '-Rpass-analysis=kernel-resource-usage',
                        '-o', f'{td}/x.so', str(SRC)], cwd=td, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    asm = (pathlib.Path(td) / 'dgsqp_api-hip-amdgcn-amd-amdhsa-gfx950.s').read_text()
    remarks = r.stderr
print('# kernel-resource-usage (worst call path) of the solve kernel')
blk = remarks[remarks.index('Function Name: _Z15dg_solve_kernel'):]
for key in ('TotalSGPRs', 'VGPRs', 'AGPRs', 'ScratchSize [bytes/lane]', 'Occupancy [waves/SIMD]', 'SGPRs Spill', 'VGPRs Spill'):
    m = re.search(re.escape(key) + r': (\d+)', blk)
    print(f'#   {key}: {m.group(1)}')
print('# per function (scratch >= %d B/lane), sorted by scratch: name | VGPRs | scratch B/lane | instructions | fp64 ops | scratch loads / stores | of those inside loops (loads / stores) | share of in-loop instructions' % min_scratch)
rows = []
for m in re.finditer(r'^(\S+):\s*; @\1\n', asm, re.M):
    name = m.group(1)
    end = asm.find('.Lfunc_end', m.end())
    body = asm[m.end():end]
    info = asm[end:end + 6000]
    nv, sc = re.search(r'; NumVgprs: (\d+)', info), re.search(r'; ScratchSize: (\d+)', info)
    if not nv or not sc or int(sc.group(1)) < min_scratch:
        continue
    cur, n, fp, sl, ss, ln, lsl, lss = '', 0, 0, 0, 0, 0, 0, 0
    for line in body.split('\n'):
        lab = re.match(r'^\.(LBB\d+_\d+):\s*;?\s*(.*)$', line)
        if lab:
            cur = lab.group(2)
        t = line.strip()
        if not t or t[0] in ';.':
            continue
        n += 1
        a, b = 'scratch_load' in t, 'scratch_store' in t
        fp += bool(re.match(r'v_(fma|fmac|mul|add)_f64', t))
        sl += a; ss += b
        if 'Loop' in cur:
            ln += 1; lsl += a; lss += b
    dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip().split('(')[0]
    rows.append((int(sc.group(1)), f'{dem[:64]:64s} | {nv.group(1):>3s} | {sc.group(1):>5s} | {n:6d} | {fp:5d} | {sl:4d} / {ss:4d} | {lsl:4d} / {lss:4d} | {100.0 * (lsl + lss) / max(ln, 1):4.1f} % of {ln}'))
for _, r_ in sorted(rows, reverse=True):
    print(r_)
