for k in k4 k5 k6; do
  for rc in 0 1; do
    AB_REPS=1 scripts/ab_bench.sh r03_h "--steps 300 --workload config5_$k --flags 31 --n-reads 3000 --resident-copies $((rc==1?1:8))" main 2>&1 | sed "s/^/$k copies=$((rc==1?1:8)) n=3000 /"
  done
  AB_REPS=1 scripts/ab_bench.sh r03_h "--steps 300 --workload config5_$k --flags 31" main 2>&1 | sed "s/^/$k full raw /"
  AB_REPS=1 scripts/ab_bench.sh r03_h "--steps 300 --workload config5_$k" main 2>&1 | sed "s/^/$k full sums /"
done
