#!/bin/bash
# Round 4, VERDICT r3 item 2: the step-1-dominated regime (1 % / 0.1 % of the reads telomeric) measured -- bench lines,
# rocprofv3 kernel stats + PMC (scripts/profile_workloads.sh), the occupancy timeline from the clock stamps of the diagnostics
# build, and a batch-size sweep (one launch of 10 000 reads is ~2 rounds of wave slots; real WGS batches are larger).
# usage: scripts/r04_step1_regime.sh <tag>          -> gpurun_out/<tag>/
set -u
TAG=${1:-r04_step1}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
for W in config4_1pct config4_01pct config4_sample; do
  python3 bench.py --workload $W --steps 400 --warmup 5 --no-cpu-baseline --no-e2e --streams 1 > $OUT/bench_$W.json 2> $OUT/bench_$W.err
  echo "bench $W rc=$?"
done
for N in 40000 100000; do
  python3 bench.py --workload config4_1pct --n-reads $N --steps 100 --warmup 5 --no-cpu-baseline --no-e2e --streams 1 > $OUT/bench_config4_1pct_n$N.json 2> $OUT/bench_config4_1pct_n$N.err
  echo "bench 1pct n=$N rc=$?"
done
bash scripts/profile_workloads.sh ${TAG} "config4_1pct config4_01pct" > $OUT/workloads.csv 2> $OUT/profile_workloads.err
if [ -f topsicle_amd/libtopsicle_hip_diag.so ]; then
  for F in 1 0.01 0.001; do
    TPS_TELO_FRAC=$F TOPSICLE_HIP_LIB=$ROOT/topsicle_amd/libtopsicle_hip_diag.so python3 scripts/stamps_timeline.py 10000 30000 > $OUT/timeline_frac$F.txt 2>&1
    echo "timeline $F rc=$?"
  done
  TPS_TELO_FRAC=0.01 TOPSICLE_HIP_LIB=$ROOT/topsicle_amd/libtopsicle_hip_diag.so python3 scripts/stamps_timeline.py 100000 30000 > $OUT/timeline_frac0.01_n100000.txt 2>&1
fi
