#!/bin/bash
# rocprofv3 evidence for the non-default kernels: per workload (and scan flags) one --kernel-trace --stats run and one
# FETCH_SIZE / WRITE_SIZE PMC pass each (separate runs, as MI355X_MICROARCH.md prescribes).
# usage: scripts/profile_workloads.sh <tag> "<workload>[@seq][:flags] ..."      -> gpurun_out/profw_<tag>/<workload>[_seq]_f<flags>/
# (@seq: bench.py --sequential-tables -- config 5's three table passes back to back on one stream instead of overlapping)
set -u
TAG=${1:-r02}
LIST=${2:-"config3_per_gpu config4_sample config5_k4:31 config5_k5 config5_k5:31 config5_k6 config5_k6:31 config5 config5@seq"}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for item in $LIST; do
  W=${item%%:*}; F=0; [[ "$item" == *:* ]] && F=${item##*:}
  SEQ=""; SEQF=""; if [[ "$W" == *@seq ]]; then W=${W%@seq}; SEQ="_seq"; SEQF="--sequential-tables"; fi
  OUT=$ROOT/gpurun_out/profw_$TAG/${W}${SEQ}_f$F
  rm -rf $OUT; mkdir -p $OUT
  FL=""; [ "$F" != "0" ] && FL="--flags $F"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 400 --warmup 3 --no-cpu-baseline --no-e2e --streams 1 --no-steady $SEQF --workload $W $FL > $OUT/bench.json 2> $OUT/trace.err
  echo "$item trace rc=$?"
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -- python3 $ROOT/bench.py --steps 4 --warmup 1 --min-timed-ms 0 --no-cpu-baseline --no-e2e --streams 1 --no-steady $SEQF --workload $W $FL > $OUT/pmc_$C.log 2>&1
    echo "$item pmc $C rc=$?"
  done
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 $ROOT/bench.py --steps 4 --warmup 1 --min-timed-ms 0 --no-cpu-baseline --no-e2e --streams 1 --no-steady $SEQF --workload $W $FL > $OUT/pmc_sq.log 2>&1
  echo "$item pmc sq rc=$?"
  # keep only the small summaries (the raw traces are tens of MB)
  find $OUT -name "*_kernel_trace.csv" -delete
done
python3 $ROOT/scripts/summarize_workloads.py $ROOT/gpurun_out/profw_$TAG
