#!/bin/bash
# HBM-side traffic of the scan kernel (FETCH_SIZE, WRITE_SIZE: one --pmc pass each) for one workload under several environments.
# usage: scripts/pmc_traffic_env.sh <workload> "<ENV=V|-> ..."     (KB per launch as rocprofv3 reports them; FETCH is doubled on gfx950)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
W=$1
for kv in $2; do
  for C in FETCH_SIZE WRITE_SIZE; do
    OUT=$ROOT/gpurun_out/pmct; rm -rf $OUT; mkdir -p $OUT
    ( [ "$kv" != "-" ] && for e in ${kv//,/ }; do export "$e"; done
      rocprofv3 --pmc $C --output-format csv -d $OUT -- python3 $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-e2e --streams 1 --workload $W > $OUT/log 2>&1 )
    python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob('$OUT/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if r['Kernel_Name'].startswith('tps_scan'): acc[r['Counter_Name']].append(float(r['Counter_Value']))
print('$W $kv', {k: round(sum(v)/len(v)) for k,v in sorted(acc.items())} or open('$OUT/log').read()[-300:])
PY
  done
done
