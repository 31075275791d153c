#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for sp in 32 64 96 128; do
  TPS_SPANS_PER_TILE=$sp python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('spans=$sp', 'kernel_ms', round(d['roofline']['kernel_ms_mean'],4), 'ms_per_step', round(d['ms_per_step'],4))"
done
TPS_SPANS_PER_TILE=64 python scripts/stamps.py
