#!/bin/bash
# A/B of library variants on the GPU box: for every "name[:ENV=VAL,...]" argument runs bench.py against
# topsicle_amd/libtopsicle_hip_<name>.so (name "main" = the product library) with the extra environment, several
# repetitions interleaved so that clock / box drift hits all variants alike.
#   usage: scripts/ab_bench.sh OUTTAG "bench args" variant[:ENV=VAL,..] ...
set -u
TAG=$1; shift
BARGS=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
REPS=${AB_REPS:-3}
for rep in $(seq 1 $REPS); do
  for v in "$@"; do
    name=${v%%:*}
    envs=""
    if [[ "$v" == *:* ]]; then envs=$(echo "${v#*:}" | tr ',' ' '); fi
    lib=$ROOT/topsicle_amd/libtopsicle_hip_$name.so
    [ "$name" == "main" ] && lib=$ROOT/topsicle_amd/libtopsicle_hip.so
    label=$(echo "$v" | tr ':=,' '___')
    env TOPSICLE_HIP_LIB=$lib $envs python3 $ROOT/bench.py $BARGS --no-cpu-baseline --no-e2e --streams 1 > $OUT/${label}_$rep.json 2> $OUT/${label}_$rep.err
    python3 - "$OUT/${label}_$rep.json" "$label" <<'EOF'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(f"{sys.argv[2]:40s} kernel_ms {r['kernel_ms_mean']*1e3:8.2f} us  frac {r['frac']:.3f}  ms/step {d['ms_per_step']*1e3:8.2f} us  {r['kernel_launch']}")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
EOF
  done
done
