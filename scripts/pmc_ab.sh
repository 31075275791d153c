#!/bin/bash
# PMC counters of the scan kernel for several library variants (scripts/build_variant.py), one rocprofv3 --pmc pass per
# counter group and variant (never together with a trace).   usage: scripts/pmc_ab.sh OUTTAG "bench args" variant ...
set -u
TAG=$1; shift
BARGS=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
G1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
G2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
G3="SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"
for v in "$@"; do
  lib=$ROOT/topsicle_amd/libtopsicle_hip_$v.so
  [ "$v" == "main" ] && lib=$ROOT/topsicle_amd/libtopsicle_hip.so
  export TOPSICLE_HIP_LIB=$lib
  i=0
  for G in "$G1" "$G2" "$G3" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    rocprofv3 --pmc $G --output-format csv -d $OUT/${v}_g$i -- python3 $ROOT/bench.py $BARGS --prime 16 --steps 4 --warmup 1 --min-timed-ms 0 --no-cpu-baseline --no-e2e --streams 1 > $OUT/${v}_g$i.log 2>&1
    echo "$v group $i rc=$?"
  done
done
python3 - $OUT "$@" <<'PY'
import collections, csv, glob, os, sys
out, variants = sys.argv[1], sys.argv[2:]
tab = collections.OrderedDict()
for v in variants:
    for f in sorted(glob.glob(os.path.join(out, v + "_g*", "*", "*_counter_collection.csv"))):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith("tps_scan"):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for c, vals in acc.items():
            tab.setdefault(c, {})[v] = sum(vals) / len(vals)
with open(os.path.join(out, "summary.txt"), "w") as h:
    line = "%-28s" % "counter" + "".join("%16s" % v for v in variants)
    print(line); h.write(line + "\n")
    for c, d in tab.items():
        line = "%-28s" % c + "".join("%16.0f" % d.get(v, float("nan")) for v in variants)
        print(line); h.write(line + "\n")
PY
