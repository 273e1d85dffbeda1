#!/bin/bash
# Two PMC passes (SQ lane/wait counters; L2 hits) of one bench workload under a set of PT_* environment knobs.
# Usage: scripts/pmc_variant.sh <outdir> "<ENV=V ...>" [bench.py args...]      (the knobs are exported, then python3 runs directly under rocprofv3)
set -u
out=$1; envs=$2; shift; shift
mkdir -p "$out"
export TMPDIR=/tmp
for kv in $envs; do export "$kv"; done
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  timeout -s KILL 240 rocprofv3 --pmc $set --output-format csv -d "$out/pass$i" -- python3 bench.py --steps 1 --warmup 0 --configs none --no-fast-arith --no-cpu-baseline --no-kernel-timing "$@" > "$out/pass$i.log" 2>&1 || { echo "pass $i failed"; tail -n 5 "$out/pass$i.log"; exit 1; }
done <<'SETS'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_INSTS_VMEM_RD
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE
SETS
python3 scripts/pmc_summarize.py "$out" > /dev/null
rm -rf "$out"/pass*/  # raw per-dispatch CSVs are large; the summary is what is kept
python3 - "$out" <<'PY'
import sys, re
txt = open(sys.argv[1] + "/pmc_summary.txt").read()
cur = None; d = {}
for line in txt.splitlines():
    if not line.startswith("  "):
        cur = line.split("(")[0].replace("void ", "").strip(); d[cur] = {}
    else:
        parts = line.split()
        if len(parts) == 2: d[cur][parts[0]] = float(parts[1])
for k, c in d.items():
    if not k.startswith("k_") or "SQ_WAVE_CYCLES" not in c or c["SQ_WAVE_CYCLES"] < 1e9: continue
    lanes = c["SQ_THREAD_CYCLES_VALU"] / max(c["SQ_ACTIVE_INST_VALU"], 1)
    print(f"  {k[:44]:44s} lanes {lanes:5.1f}/64  wait {100*c['SQ_WAIT_ANY']/c['SQ_WAVE_CYCLES']:4.1f} %  L2 hit {100*c.get('TCC_HIT_sum',0)/max(c.get('TCC_REQ_sum',1),1):4.1f} %  "
          f"L1 miss {100*c.get('TCP_TCC_READ_REQ_sum',0)/max(c.get('TCP_TOTAL_CACHE_ACCESSES_sum',1),1):4.1f} %  busy_cycles {c['SQ_BUSY_CYCLES']:.3g}  vmem_rd {c['SQ_INSTS_VMEM_RD']:.3g}")
PY
