#!/bin/bash
# VGPR / SGPR / spill / scratch / occupancy of every kernel in the library (compiler view; no GPU needed)
cd "$(dirname "$0")/../topsicle_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-vectorize -std=c++17 -shared -fPIC -Wno-unused-variable ${TPS_HIPCC_EXTRA:-} \
  -Rpass-analysis=kernel-resource-usage -o /tmp/tps_res.so topsicle_hip.hip 2>&1 | python3 -c '
import re, sys
cur = None
rows = {}
for ln in sys.stdin:
    m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", ln)
    if not m: continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        cur = t.split(":", 1)[1].strip(); rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1); rows[cur][k.strip()] = v.strip()
keys = ["VGPRs", "AGPRs", "TotalSGPRs", "SGPRs Spill", "VGPRs Spill", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]"]
print("%-26s" % "kernel", *["%10s" % k.split()[0][:10] for k in keys])
for n, r in rows.items():
    print("%-26s" % n, *["%10s" % r.get(k, "-") for k in keys])
'
