#!/bin/bash
# PMC counter groups of the scan kernel for several bench workloads, one rocprofv3 --pmc pass per group and workload (never together
# with a trace): where a kernel's wave-cycles go (busy / waiting on what).   usage: scripts/pmc_groups.sh OUTTAG "workload[:flags] ..."
#   -> gpurun_out/<OUTTAG>/<workload>_g<i>/ + one summary line per workload and group on stdout
set -u
TAG=$1; shift
LIST=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
G1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
G2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
G3="SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"
for item in $LIST; do
  W=${item%%:*}; FL=""; [[ "$item" == *:* ]] && FL="--flags ${item##*:}"
  i=0
  for G in "$G1" "$G2" "$G3"; do
    i=$((i+1)); D=$OUT/$(echo "$item" | tr ':' '_')_g$i; rm -rf $D; mkdir -p $D
    rocprofv3 --pmc $G --output-format csv -d $D -- python3 $ROOT/bench.py --steps 4 --warmup 1 --min-timed-ms 0 --prime 16 --no-steady --no-cpu-baseline --no-e2e --streams 1 --workload $W $FL > $D/log 2>&1
    python3 - "$D" "$item g$i" <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if r['Kernel_Name'].startswith('tps_scan'):
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
print(sys.argv[2], {k: round(sum(v) / len(v)) for k, v in sorted(acc.items())} or open(sys.argv[1] + '/log').read()[-300:])
PY
    find $D -name "*.csv" -size +2M -delete
  done
done
