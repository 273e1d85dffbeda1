#!/bin/bash
# Memory-path counters (TA / TCP / TCC / address translation) of one bench workload, one rocprofv3 --pmc run per set.
# Usage: scripts/pmc_mem.sh <outdir> [bench.py args...]
set -u
out=$1; shift
mkdir -p "$out"
export TMPDIR=/tmp
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  echo "== pass $i: $set"
  timeout -s KILL 240 rocprofv3 --pmc $set --output-format csv -d "$out/pass$i" -- python3 bench.py --steps 1 --warmup 0 --configs none --no-fast-arith --no-cpu-baseline --no-kernel-timing "$@" > "$out/pass$i.log" 2>&1 || { echo "pass $i failed"; tail -n 5 "$out/pass$i.log"; exit 1; }
done <<'SETS'
TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCP_LATENCY_sum TCP_GATE_EN1_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
TCC_BUSY_avr TCC_TAG_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum
TCC_EA0_RDREQ_DRAM_sum TCC_CYCLE_sum TCC_REQ_sum TCC_EA0_RDREQ_32B_sum
TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum TCP_UTCL1_REQUEST_sum TD_TD_BUSY_sum TD_TC_STALL_sum
SETS
python3 scripts/pmc_summarize.py "$out"
rm -rf "$out"/pass*/  # raw per-dispatch CSVs are large; the summary is what is kept
