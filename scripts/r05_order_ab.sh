#!/bin/bash
# Round-5 A/B of the dispatch order (tps::plan_dispatch_order): bench lines of the ragged workload per table / output kind, file order against
# longest-first, one stream and two; and the BASELINE workloads (equal reads: one class, file order either way) as the control.
# usage: scripts/r05_order_ab.sh [tag]        -> gpurun_out/<tag>/*.json, one summary line per run on stdout
set -u
TAG=${1:-r05_order}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
line() {
python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")][-1]
    r = d["roofline"]
    print("%-44s two streams %.4f ms/step [p10 %.4f p90 %.4f]  single stream %.4f  kernel alone %.4f ms  frac %.3f  %s" % (sys.argv[2], d["ms_per_step"], d["ms_per_step_p10"], d["ms_per_step_p90"], d["single_stream"]["ms_per_step"], r["kernel_ms_mean"], r["frac"], r["kernel_launch"].split()[0]))
except Exception as e:
    print(sys.argv[2], "FAILED", e, open(sys.argv[1].replace(".json", ".err")).read()[-300:])
PY
}
for item in "ragged_ont:15:4" "ragged_ont:31:4" "ragged_ont:15:5" "ragged_ont:15:6" "ragged_ont:31:6" "config2:15:4" "config5_k6:31:6"; do
  IFS=: read W FL K <<< "$item"
  for order in file longest; do
    name=${W}_k${K}_f${FL}_$order
    dbg=""; [ $order == file ] && dbg="file_order=1"
    TOPSICLE_HIP_DEBUG=$dbg python3 bench.py --workload $W --k $K --flags $FL --steps 200 --warmup 5 --no-cpu-baseline --no-e2e --no-steady > $OUT/$name.json 2> $OUT/$name.err
    line $OUT/$name.json "$W k=$K flags=$FL $order order"
  done
done
