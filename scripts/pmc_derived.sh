#!/bin/bash
# derived utilisation metrics of the scan kernel (one rocprofv3 pass per metric group)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
i=0
for grp in "VALUBusy SALUBusy" "LDSBankConflict MemUnitStalled" "MeanOccupancyPerCU GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
i=$((i+1)); OUT=$ROOT/gpurun_out/pmcd$i; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc $grp --output-format csv -d $OUT -- python3 $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-e2e --streams 1 ${1:-} > $OUT/log 2>&1
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob('$OUT/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if r['Kernel_Name'].startswith('tps_scan'): acc[r['Counter_Name']].append(float(r['Counter_Value']))
print({k: round(sum(v)/len(v),3) for k,v in sorted(acc.items())} or open('$OUT/log').read()[-400:])
PY
done
