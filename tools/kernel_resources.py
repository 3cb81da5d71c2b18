#!/usr/bin/env python3
"""VGPRs / spills / occupancy of every kernel in one HIP source (hipcc remarks, no GPU needed).
usage: python tools/kernel_resources.py hash_join_codes_knl_amd/csrc/partition_kernels.hip [filter]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++20", "-c", src, "-o", "/dev/null",
                      "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: +(Function Name|VGPRs|AGPRs|SGPRs Spill|VGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\S+)", line)
    if not m:
        continue
    if m.group(1) == "Function Name":
        cur = subprocess.run(["c++filt", m.group(2)], capture_output=True, text=True).stdout.strip()
        rows[cur] = {}
    elif cur:
        rows[cur][m.group(1).split(" [")[0]] = m.group(2)
for name, r in rows.items():
    if flt in name:
        print("%-90s vgpr %3s  spill v%s s%s  scratch %s  occ %s" % (name[:90], r.get("VGPRs"), r.get("VGPRs Spill"),
              r.get("SGPRs Spill"), r.get("ScratchSize"), r.get("Occupancy")))
