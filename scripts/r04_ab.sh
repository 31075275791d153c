#!/bin/bash
# Round-4 A/B helper: bench lines of the workloads named in $2 (default: the config-5 family + config 2), optional env per run.
# usage: scripts/r04_ab.sh <tag> "<workload>[:flags][@ENV=V,ENV=V] ..."      -> gpurun_out/<tag>/<item>.json
set -u
TAG=${1:-r04_ab}
LIST=${2:-"config2 config5_k4:31 config5_k5 config5_k5:31 config5_k6 config5_k6:31 config5"}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
for item in $LIST; do
  ENVS=""; base=$item
  if [[ "$item" == *@* ]]; then ENVS=${item#*@}; base=${item%%@*}; fi
  W=${base%%:*}; FL=""; [[ "$base" == *:* ]] && FL="--flags ${base##*:}"
  name=$(echo "$item" | tr ':@=,/' '_____')
  ( for kv in ${ENVS//,/ }; do export "$kv"; done
    python3 bench.py --workload $W $FL --steps ${STEPS:-300} --warmup 5 --no-cpu-baseline --no-e2e ${STREAMS:+--streams $STREAMS} > $OUT/$name.json 2> $OUT/$name.err )
  python3 - <<PY
import json
try:
    d = [json.loads(l) for l in open("$OUT/$name.json") if l.startswith("{")][-1]
    r = d["roofline"]
    print("%-44s step %.4f ms (single stream %.4f)  kernel %.4f ms  frac %.3f  %s" % ("$item", d["ms_per_step"], d["single_stream"]["ms_per_step"], r["kernel_ms_mean"], r["frac"], r["kernel_launch"]))
except Exception as e:
    print("$item FAILED", e, open("$OUT/$name.err").read()[-400:])
PY
done
