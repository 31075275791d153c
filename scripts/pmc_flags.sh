#!/bin/bash
# VALU / SALU / LDS instruction counts of the scan kernel per pipeline prefix (flags 1 = step 1, 3 = + windows,
# 11 = + S_w stores, 15 = everything)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
for fl in 1 11 15; do
OUT=$ROOT/gpurun_out/pmcf$fl; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT -- python3 $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-e2e --streams 1 --flags $fl ${1:-} > $OUT/log 2>&1
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob('$OUT/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if r['Kernel_Name'].startswith('tps_scan'): acc[r['Counter_Name']].append(float(r['Counter_Value']))
print('flags $fl', {k: round(sum(v)/len(v)) for k,v in sorted(acc.items())})
PY
done
