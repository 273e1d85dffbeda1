#!/bin/bash
# Kernel-trace stats + PMC passes + traffic file of one bench workload, on the GPU box.
# Usage: scripts/profile_workload.sh <tag> <workload>     -> gpurun_out/<tag>_<workload>/{stats,pmc}
set -u
tag=$1; wl=$2
out=gpurun_out/${tag}_${wl}
mkdir -p "$out"
export TMPDIR=/tmp
export PT_CONCURRENT_BANDS=1  # per-launch figures of whole-frame launches, as in the profiled step of bench.py
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 bench.py --workload "$wl" --configs none --no-fast-arith --steps 2 --warmup 0 --no-cpu-baseline > "$out/bench.json" 2> "$out/bench.err" || { echo "stats run failed"; tail -5 "$out/bench.err"; exit 1; }
find "$out/stats" -name '*kernel_stats.csv' -exec cp {} "$out/kernel_stats.csv" \;
scripts/pmc_passes.sh "$out/pmc" --workload "$wl" || exit 1
python3 scripts/pmc_traffic.py "$out/pmc" "$out/traffic_${wl}.json" "$wl" > /dev/null
head -5 "$out/kernel_stats.csv"
