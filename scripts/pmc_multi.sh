#!/bin/bash
# instruction / wait counters of several bench workloads in one call: scripts/pmc_multi.sh "<workload>[:flags][+errors] ..."
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
for item in $1; do
  E=""; base=$item; if [[ "$item" == *+* ]]; then E="--errors ${item##*+}"; base=${item%%+*}; fi
  W=${base%%:*}; FL=""; [[ "$base" == *:* ]] && FL="--flags ${base##*:}"
  for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
    OUT=$ROOT/gpurun_out/pmcm; rm -rf $OUT; mkdir -p $OUT
    rocprofv3 --pmc $grp --output-format csv -d $OUT -- python3 $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-e2e --streams 1 --workload $W $FL $E > $OUT/log 2>&1
    python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob('$OUT/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if r['Kernel_Name'].startswith('tps_scan'): acc[r['Counter_Name']].append(float(r['Counter_Value']))
print('$item', {k: round(sum(v)/len(v)) for k,v in sorted(acc.items())} or open('$OUT/log').read()[-300:])
PY
  done
done
