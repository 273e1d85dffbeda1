#!/bin/bash
# round 3 on the GPU box: guard build on the small parity scenes, the GPU suite on the normal build, then a same-box A/B
# of the round-2 library (variants/base.so) against the working tree's (variants/cur.so)
set -u
mkdir -p gpurun_out
PT_LIB_PATH=$PWD/variants/guard.so timeout -k 10 400 python3 -u -m pytest tests/test_gpu_parity.py -x -q --timeout 180 --timeout-method thread > gpurun_out/r03_guard.log 2>&1 || { echo "guard run failed"; tail -30 gpurun_out/r03_guard.log; exit 1; }
tail -2 gpurun_out/r03_guard.log
grep -c "\[guard\]" gpurun_out/r03_guard.log
timeout -k 10 1000 python3 -u -m pytest tests -m gpu -x -q --timeout 400 --timeout-method thread ${SUITE_ARGS:-} > gpurun_out/r03_gpu_suite.log 2>&1 || { echo "gpu suite failed"; tail -40 gpurun_out/r03_gpu_suite.log; exit 1; }
tail -2 gpurun_out/r03_gpu_suite.log
for wl in ${AB_WORKLOADS:-rocks atrium cornell}; do
  case $wl in
    rocks) args="--workload rocks1080 --spp 64 --steps 2 --warmup 1 --configs none --no-fast-arith";;
    atrium) args="--workload atrium1080 --spp 36 --steps 2 --warmup 1 --configs none --no-fast-arith";;
    soup) args="--workload soup1080 --steps 2 --warmup 1 --configs none --no-fast-arith";;
    textured) args="--workload textured1080 --steps 2 --warmup 1 --configs none --no-fast-arith";;
    cornell) args="--workload cornell1080 --steps 5 --warmup 2 --configs none --no-fast-arith";;
  esac
  scripts/ab.sh "$args" ${AB_VARIANTS:-base cur} || exit 1
done
