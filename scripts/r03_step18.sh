set -u
PT_LIB_PATH=$PWD/variants/base.so timeout -k 10 300 python3 -u -m pytest tests/test_gpu_parity.py tests/test_gpu_wide4.py -x -q --timeout 120 --timeout-method thread > gpurun_out/parity_stash.txt 2>&1 || { echo "parity failed"; tail -20 gpurun_out/parity_stash.txt; exit 1; }
tail -n 1 gpurun_out/parity_stash.txt
scripts/ab.sh "--configs none --no-fast-arith --workload atrium1080 --spp 36 --steps 2 --warmup 1" head base || exit 1
scripts/ab.sh "--configs none --no-fast-arith --workload rocks1080 --spp 64 --steps 2 --warmup 1" head base || exit 1
scripts/ab.sh "--configs none --no-fast-arith --workload soup1080 --steps 2 --warmup 1" head base || exit 1
PT_BVH=wide4 scripts/ab.sh "--configs none --no-fast-arith --workload rocks1080 --spp 64 --steps 2 --warmup 1" head base || exit 1
