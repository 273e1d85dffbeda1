set -u
PT_LIB_PATH=$PWD/variants/guard.so timeout -k 10 400 python3 -u -m pytest tests/test_gpu_parity.py -x -q --timeout 120 --timeout-method thread > gpurun_out/parity_guard.txt 2>&1 || { echo "guard parity failed"; grep -m5 "guard\]" gpurun_out/parity_guard.txt; tail -30 gpurun_out/parity_guard.txt; exit 1; }
tail -n 1 gpurun_out/parity_guard.txt; grep -c "guard\]" gpurun_out/parity_guard.txt
timeout -k 10 400 python3 -u -m pytest tests/test_gpu_parity.py tests/test_gpu_wide4.py tests/test_gpu_device_entry.py -x -q --timeout 120 --timeout-method thread > gpurun_out/parity_probe.txt 2>&1 || { echo "parity failed"; tail -30 gpurun_out/parity_probe.txt; exit 1; }
tail -n 1 gpurun_out/parity_probe.txt
scripts/ab.sh "--configs none --no-fast-arith --workload atrium1080 --spp 36 --steps 2 --warmup 1" head base || exit 1
scripts/ab.sh "--configs none --no-fast-arith --workload rocks1080 --spp 64 --steps 2 --warmup 1" head base || exit 1
PT_BVH=wide4 scripts/ab.sh "--configs none --no-fast-arith --workload atrium1080 --spp 36 --steps 2 --warmup 1" head base || exit 1
