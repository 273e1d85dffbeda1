#!/bin/bash
# Hardware-counter passes for bench.py's workload, one rocprofv3 --pmc run per counter set
# (counters are never combined with trace domains on this pool). Usage:
#   scripts/pmc_passes.sh <outdir> [bench.py args...]
set -u
out=$1; shift
mkdir -p "$out"
export TMPDIR=/tmp
export PT_CONCURRENT_BANDS=1  # per-launch figures of whole-frame launches, as in the profiled step of bench.py
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  echo "== pass $i: $set"
  timeout -s KILL 240 rocprofv3 --pmc $set --output-format csv -d "$out/pass$i" -- python3 bench.py --steps 1 --warmup 0 --configs none --no-fast-arith --no-cpu-baseline --no-kernel-timing "$@" > "$out/pass$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$out/pass$i.log"; exit 1; }
done <<'SETS'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY
SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INSTS_LDS
FETCH_SIZE GRBM_GUI_ACTIVE
WRITE_SIZE
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum
SETS
python3 scripts/pmc_summarize.py "$out"
