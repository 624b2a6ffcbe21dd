#!/bin/bash
# Register / spill / occupancy summary of the k_trace_shade instantiations (CPU box: cross-compile only).
#   tools/kres.sh [extra hipcc flags]      e.g. tools/kres.sh -DRTX_WAVES_PER_EU_NOBVH=5
cd "$(dirname "$0")/../ray-tracer_amd"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden --offload-arch=gfx950 \
  -fno-slp-vectorize -DRT_BUILD "$@" -c csrc/rtx_device.hip -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 |
python3 -c '
import re, sys
cur = {}
rows = []
for line in sys.stdin:
    m = re.search(r"remark: +([A-Za-z ]+?)(?: \[[^\]]*\])?: +(\S+)", line)
    if not m: continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name":
        cur = {"name": v}; rows.append(cur)
    else:
        cur[k] = v
for r in rows:
    n = r["name"]
    m = re.search(r"k_trace_shadeILi(\d)E((?:Lb\dE)+)", n)
    names = ["GEN", "TRACK", "C", "BVH", "STATS", "X6"]
    tag = "trace TAB" + m.group(1) + " " + " ".join(k + v for k, v in zip(names, re.findall(r"Lb(\d)E", m.group(2)))) if m else n[:40]
    print("%-36s VGPR %3s  spill %3s  scratch %5s  occ %s  SGPR %s" % (tag, r.get("VGPRs"), r.get("VGPRs Spill"), r.get("ScratchSize"), r.get("Occupancy"), r.get("TotalSGPRs")))
'
