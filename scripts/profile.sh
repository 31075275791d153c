#!/bin/bash
# Profiles bench.py on the GPU box: kernel trace + stats, then PMC passes (each in its own run).
# usage: scripts_profile.sh <tag>     (outputs under gpurun_out/prof_<tag>/)
set -u
TAG=${1:-r1}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-e2e --streams 1 --no-steady"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH --steps 2000 > $OUT/trace.log 2>&1   # many steps: the 256 clock-ramp priming launches must not weigh on the average
echo "trace rc=$?"
# ... and the default command (two streams: `value`'s region): its launches overlap, each one's own duration stretches
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_default -- python3 $ROOT/bench.py --steps 2000 --warmup 3 --no-cpu-baseline --no-e2e --no-steady > $OUT/trace_default.log 2>&1
echo "trace_default rc=$?"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -- python3 $ROOT/bench.py --steps 4 --warmup 1 --min-timed-ms 0 --no-cpu-baseline --no-e2e --streams 1 --no-steady > $OUT/pmc_$C.log 2>&1
  echo "pmc $C rc=$?"
done
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc_sq1 -- python3 $ROOT/bench.py --steps 4 --warmup 1 --min-timed-ms 0 --no-cpu-baseline --no-e2e --streams 1 --no-steady > $OUT/pmc_sq1.log 2>&1
echo "pmc sq1 rc=$?"
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/pmc_sq2 -- python3 $ROOT/bench.py --steps 4 --warmup 1 --min-timed-ms 0 --no-cpu-baseline --no-e2e --streams 1 --no-steady > $OUT/pmc_sq2.log 2>&1
echo "pmc sq2 rc=$?"
find $OUT -name "*.csv" | head -40
