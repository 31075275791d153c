#!/bin/bash
# quick A/B of planner knobs on the GPU box: prints kernel ms per setting
cd ${GRAFT_REPO_ROOT:-/root/repo}
for kb in 40 32 26 20; do
  TPS_LDS_TARGET_KB=$kb python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('lds_target_kb=$kb', 'kernel_ms', round(d['roofline']['kernel_ms_mean'],4), 'ms_per_step', round(d['ms_per_step'],4))"
done
TPS_FORCE_GENERIC=1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('generic', 'kernel_ms', round(d['roofline']['kernel_ms_mean'],4))"
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-store-sums 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('no-store-sums', 'kernel_ms', round(d['roofline']['kernel_ms_mean'],4))"
