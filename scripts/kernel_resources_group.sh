#!/bin/bash
# VGPR / SGPR / spill / scratch / occupancy of the kernels of ONE build group (tps_kernels.h: TPS_KGROUP), seconds instead of the
# minutes scripts/kernel_resources.sh takes for the whole library.   usage: kernel_resources_group.sh <group> [extra hipcc flags]
G=${1:-3}; shift
cd "$(dirname "$0")/../topsicle_amd/csrc"
SRC=tps_kernels.hip; [ "$G" = "0" ] && SRC=topsicle_hip.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-vectorize -std=c++17 -fPIC -Wno-unused-variable ${TPS_HIPCC_EXTRA:-} "$@" -DTPS_KGROUP=$G \
  -Rpass-analysis=kernel-resource-usage -c -o /tmp/tps_res_g$G.o $SRC 2>&1 | python3 -c '
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
