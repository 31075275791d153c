run() { python bench.py --workload $1 --errors $2 --no-cpu-baseline --no-e2e --steps 300 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2 $3', round(d['roofline']['kernel_ms_mean']*1e3,1), d['roofline']['kernel_launch'])"; }
run config5_k5 hifi base
TPS_WPG=8 TPS_LDS_PAD_BYTES=12288 run config5_k5 hifi shape_of_k6
TPS_WPG=8 run config5_k5 hifi wpg8_only
TPS_LDS_PAD_BYTES=12288 run config5_k5 hifi pad_only_4waves
run config5_k6 hifi base
TPS_WPG=4 run config5_k6 hifi wpg4
