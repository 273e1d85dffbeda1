#!/bin/bash
# Compiler view of every kernel of pt_kernels.hip (the exact build): code bytes, VGPRs, scratch bytes per lane (spills + the
# private overflow rows of the traversal stack), occupancy, and the number of scratch load / store sites in its ISA.
# Usage: scripts/kernel_resources.sh [extra hipcc flags...]   (CPU only: hipcc cross-compiles)
cd "$(dirname "$0")/.."
out=${TMPDIR:-/tmp}/pt_kernels_res.s
/opt/rocm/bin/hipcc --offload-arch=gfx950 --cuda-device-only -S -O3 -std=c++17 -ffp-contract=off -fno-fast-math -Wno-unused-command-line-argument \
  $PT_EXTRA_HIPCC_FLAGS "$@" -I include -I pathtracer_amd/csrc/device pathtracer_amd/csrc/device/pt_kernels.hip -o "$out" || exit 1
python3 - "$out" <<'PY'
import re, subprocess, sys
fn, cur, rows = None, {}, []
for line in open(sys.argv[1]):
    m = re.match(r'^(_Z\w+):', line)
    if m:
        fn, cur = m.group(1), {"ld": 0, "st": 0}
    if fn:
        if "scratch_load" in line: cur["ld"] += 1
        if "scratch_store" in line: cur["st"] += 1
    for key in ("codeLenInByte", "NumVgprs", "ScratchSize", "Occupancy", "NumSgprs"):
        m = re.match(r'^; ' + key + r'\s*[:=]\s*(\d+)', line)
        if m: cur[key] = int(m.group(1))
    if line.startswith("; Occupancy") and fn:
        rows.append((fn, cur)); fn = None
names = subprocess.run(["c++filt"] + [r[0] for r in rows], capture_output=True, text=True).stdout.split("\n")
print(f"{'kernel':58s} {'code B':>7s} {'VGPRs':>5s} {'SGPRs':>5s} {'scratch B/lane':>14s} {'occ':>3s} {'scratch ld/st sites':>19s}")
for (fn, c), name in zip(rows, names):
    name = re.sub(r'\(.*', '', name).replace("void ", "")
    print(f"{name[:58]:58s} {c.get('codeLenInByte', 0):7d} {c.get('NumVgprs', 0):5d} {c.get('NumSgprs', 0):5d} {c.get('ScratchSize', 0):14d} {c.get('Occupancy', 0):3d} {c['ld']:9d} /{c['st']:4d}")
PY
