#!/bin/bash
# round-3 profiles: the fast-arithmetic tests first (changed code), then kernel stats + PMC passes + traffic of the three 1080p workloads
set -u
mkdir -p gpurun_out
timeout -k 10 600 python3 -u -m pytest tests/test_gpu_fast_arith.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r03_fast.log 2>&1 || { echo "fast tests failed"; tail -30 gpurun_out/r03_fast.log; exit 1; }
tail -2 gpurun_out/r03_fast.log
for wl in ${PROFILE_WORKLOADS:-atrium1080 rocks1080 cornell1080}; do
  scripts/profile_workload.sh r03 $wl || exit 1
  python3 scripts/show_bench.py gpurun_out/r03_$wl/bench.json
done
