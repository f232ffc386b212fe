cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/round6e; mkdir -p $O
A="--workload dyn_curve_N25 --steps 2 --warmup 0 --group 1 --pipeline 1 --coop off --extras off --single-steps 0 --host-steps 0 --cpu-sample 0"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $O/pmc_lds1 -- python3 $R/bench.py $A > $O/pmc_lds1.json 2> $O/pmc_lds1.err
rocprofv3 --pmc SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $O/pmc_lds2 -- python3 $R/bench.py $A > $O/pmc_lds2.json 2> $O/pmc_lds2.err
rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS --output-format csv -d $O/pmc_lds3 -- python3 $R/bench.py $A > $O/pmc_lds3.json 2> $O/pmc_lds3.err
python3 - <<'PY'
import csv, glob, collections, os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/round6e'
for d in ('pmc_lds1','pmc_lds2','pmc_lds3'):
    t=collections.defaultdict(float)
    for f in glob.glob(f'{O}/{d}/*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            if r['Kernel_Name'].startswith('dg_solve_kernel'): t[r['Counter_Name']]+=float(r['Counter_Value'])
    print(d, dict(t))
PY
tail -3 $O/pmc_lds1.err
