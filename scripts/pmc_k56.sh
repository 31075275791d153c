#!/bin/bash
# Diagnostic: VALU / SALU / LDS instruction counts of the k = 5 and k = 6 sums-only kernels on chain-free (HiFi-like) reads.
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for w in config5_k5 config5_k6; do
  rm -rf /tmp/pmc_$w
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/pmc_$w -- python3 $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-e2e --workload $w --errors hifi > /tmp/pmc_$w.log 2>&1
  python3 - /tmp/pmc_$w $w <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("tps_scan"):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[2], {k: round(sum(v) / len(v)) for k, v in sorted(acc.items())})
PY
done
