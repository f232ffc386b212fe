#!/bin/bash
# fp64 VALU instruction mix of dg_solve_kernel (one pass), CSV output.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/round
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_INSTS_VALU[A-Z0-9_]*" | sort -u > $O/sq_valu_counters.txt
cat $O/sq_valu_counters.txt | tr '\n' ' '; echo
C="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT"
for w in dyn_curve_N25 kb_curve_N25; do
rocprofv3 --pmc $C --output-format csv -d $O/pmc_f64_$w -- python3 $R/bench.py --workload $w --steps 1 --warmup 0 --pipeline 1 --cpu-sample 0 > $O/pmc_f64_$w.json 2> $O/pmc_f64_$w.err
  f=$(ls -t $O/pmc_f64_$w/*/*_counter_collection.csv 2>/dev/null | head -1)
  echo "== $w $f"; tail -2 $O/pmc_f64_$w.err
  [ -n "$f" ] && python3 - $f <<'PY'
import csv, sys, collections
t = collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    if r['Kernel_Name'].startswith('dg_solve_kernel'): t[r['Counter_Name']] += float(r['Counter_Value'])
print(dict(t))
PY
done
