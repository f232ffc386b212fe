#!/bin/bash
# SQ issue-utilisation counters of dg_solve_kernel (one pass, 8 SQ slots), CSV output.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/round
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS"
rocprofv3 --pmc $C --output-format csv -d $O/pmc_sq_dyn -- python3 $R/bench.py --steps 1 --warmup 0 --pipeline 1 --cpu-sample 0 > $O/pmc_sq_dyn.json 2> $O/pmc_sq_dyn.err
rocprofv3 --pmc $C --output-format csv -d $O/pmc_sq_kb -- python3 $R/bench.py --workload kb_curve_N25 --steps 1 --warmup 0 --pipeline 1 --cpu-sample 0 > $O/pmc_sq_kb.json 2> $O/pmc_sq_kb.err
C2="SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVES"
rocprofv3 --pmc $C2 --output-format csv -d $O/pmc_sq2_dyn -- python3 $R/bench.py --steps 1 --warmup 0 --pipeline 1 --cpu-sample 0 > $O/pmc_sq2_dyn.json 2> $O/pmc_sq2_dyn.err
rocprofv3 --pmc $C2 --output-format csv -d $O/pmc_sq2_kb -- python3 $R/bench.py --workload kb_curve_N25 --steps 1 --warmup 0 --pipeline 1 --cpu-sample 0 > $O/pmc_sq2_kb.json 2> $O/pmc_sq2_kb.err
for d in pmc_sq_dyn pmc_sq_kb pmc_sq2_dyn pmc_sq2_kb; do
  f=$(ls -t $O/$d/*/*_counter_collection.csv 2>/dev/null | head -1)
  echo "== $d $f"; tail -3 $O/$d.err
  [ -n "$f" ] && python3 - $f <<'PY'
import csv, sys, collections
t = collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    if r['Kernel_Name'].startswith('dg_solve_kernel'): t[r['Counter_Name']] += float(r['Counter_Value'])
print(dict(t))
PY
done
