#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_SMEM" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_INST_LEVEL_LDS SQ_WAVES"; do
i=$((i+1)); OUT=$ROOT/gpurun_out/pmci$i; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc $grp --output-format csv -d $OUT -- python3 $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-e2e --streams 1 ${1:-} > $OUT/log 2>&1
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob('$OUT/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if r['Kernel_Name'].startswith('tps_scan'): acc[r['Counter_Name']].append(float(r['Counter_Value']))
print({k: round(sum(v)/len(v),1) for k,v in sorted(acc.items())} or open('$OUT/log').read()[-600:])
PY
done
