#!/bin/bash
# Register / LDS / occupancy report of every kernel (compiler view), one line per kernel.
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -c -O3 -std=c++17 -ffp-contract=off -fno-fast-math $PT_EXTRA_HIPCC_FLAGS -I include -I pathtracer_amd/csrc/device \
  pathtracer_amd/csrc/device/pt_kernels.hip -o /tmp/pt_kernels_res.o -Rpass-analysis=kernel-resource-usage 2>&1 |
  python3 -c '
import re, sys
cur = {}
for line in sys.stdin:
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: +(\S+) \[-Rpass", line)
    if not m: continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        if cur: print(cur)
        cur = {"fn": v[:24]}
    elif k in ("VGPRs", "AGPRs", "TotalSGPRs", "ScratchSize", "Occupancy", "LDS Size", "VGPRs Spill"):
        cur[k] = v
if cur: print(cur)
'
