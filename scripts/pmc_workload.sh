#!/bin/bash
# instruction counters of one bench workload: scripts/pmc_workload.sh <workload> [flags]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/pmcw; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
FL=""; [ -n "${2:-}" ] && FL="--flags $2"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT -- python3 $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-e2e --streams 1 --workload $1 $FL > $OUT/log 2>&1
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob('$OUT/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if r['Kernel_Name'].startswith('tps_scan'): acc[r['Counter_Name']].append(float(r['Counter_Value']))
print('$1', '${2:-}', {k: round(sum(v)/len(v)) for k,v in sorted(acc.items())})
PY
