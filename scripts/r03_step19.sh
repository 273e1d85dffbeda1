set -u
# stale-entry skip (one-sample merged kernel): parity first, then same-box A/B against the commit before
timeout -k 10 400 python3 -u -m pytest tests/test_gpu_parity.py tests/test_gpu_device_entry.py -x -q --timeout 120 --timeout-method thread > gpurun_out/parity_stale.txt 2>&1 || { echo "parity failed"; tail -30 gpurun_out/parity_stale.txt; exit 1; }
tail -n 1 gpurun_out/parity_stale.txt
scripts/ab.sh "--configs none --no-fast-arith --workload rocks1080 --spp 64 --steps 2 --warmup 1" head base || exit 1
scripts/ab.sh "--configs none --no-fast-arith --workload atrium1080 --spp 36 --steps 2 --warmup 1" head base || exit 1
PT_LIB_PATH=$PWD/variants/base.so scripts/r03_env_ab.sh "--workload rocks1080 --spp 64 --steps 2 --warmup 1" "rows8=PT_MERGED_LDS_STACK=8" "rows9=PT_MERGED_LDS_STACK=9" "rows10=PT_MERGED_LDS_STACK=10" || exit 1
