#!/bin/bash
# Round-5 A/B helper: bench lines of the workloads named in $2 for each library variant in $3.. ("main" = the product library;
# other names = topsicle_amd/libtopsicle_hip_<name>.so from scripts/build_variant.py), interleaved, one line per run.
# usage: scripts/r05_ab.sh <tag> "<workload>[:flags] ..." variant ...      -> gpurun_out/<tag>/<variant>_<item>.json
set -u
TAG=${1:-r05_ab}; shift
LIST=${1:-"config5_k4:31 config5_k5:31 config5_k6:31 config5_k6 config5"}; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
for item in $LIST; do
  W=${item%%:*}; FL=""; [[ "$item" == *:* ]] && FL="--flags ${item##*:}"
  for v in "$@"; do
    lib=$ROOT/topsicle_amd/libtopsicle_hip_$v.so
    [ "$v" == "main" ] && lib=$ROOT/topsicle_amd/libtopsicle_hip.so
    name=${v}_$(echo "$item" | tr ':' '_')
    TOPSICLE_HIP_LIB=$lib python3 bench.py --workload $W $FL --steps ${STEPS:-300} --warmup 5 --no-cpu-baseline --no-e2e --no-steady ${STREAMS:+--streams $STREAMS} > $OUT/$name.json 2> $OUT/$name.err
    python3 - "$OUT/$name.json" "$v $item" <<'PY'
import json, sys
try:
    d = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")][-1]
    r = d["roofline"]
    print("%-36s step %.4f ms [p10 %.4f p90 %.4f, %d repeats] (single stream %.4f)  kernel %.4f ms  frac %.3f  %s" % (sys.argv[2], d["ms_per_step"], d["ms_per_step_p10"], d["ms_per_step_p90"], d["repeats"], d["single_stream"]["ms_per_step"], r["kernel_ms_mean"], r["frac"], r["kernel_launch"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e, open(sys.argv[1].replace(".json", ".err")).read()[-400:])
PY
  done
done
