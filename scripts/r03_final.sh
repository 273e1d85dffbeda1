#!/bin/bash
# the whole GPU suite on the final tree, then the driver's round-end sequence (smoke + bench)
set -u
mkdir -p gpurun_out
timeout -k 10 900 python3 -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r03_final_tests.txt 2>&1 || { echo "tests failed"; tail -30 gpurun_out/r03_final_tests.txt; exit 1; }
tail -3 gpurun_out/r03_final_tests.txt
bash scripts/r03_bench.sh
