set -u
# two-level step (PT_TWO_LEVEL=1, experiment): guard build on the parity scenes, normal build, then A/B
PT_TWO_LEVEL=1 PT_LIB_PATH=$PWD/variants/guard.so timeout -k 10 400 python3 -u -m pytest tests/test_gpu_parity.py -x -q --timeout 120 --timeout-method thread > gpurun_out/parity_guard.txt 2>&1 || { echo "guard parity failed"; grep -m5 "guard\]" gpurun_out/parity_guard.txt; tail -30 gpurun_out/parity_guard.txt; exit 1; }
tail -n 1 gpurun_out/parity_guard.txt; grep -c "guard\]" gpurun_out/parity_guard.txt
PT_TWO_LEVEL=1 timeout -k 10 400 python3 -u -m pytest tests/test_gpu_parity.py -x -q --timeout 120 --timeout-method thread > gpurun_out/parity_two.txt 2>&1 || { echo "parity failed"; tail -30 gpurun_out/parity_two.txt; exit 1; }
tail -n 1 gpurun_out/parity_two.txt
A="--workload atrium1080 --spp 36 --steps 2 --warmup 1"
R="--workload rocks1080 --spp 64 --steps 2 --warmup 1"
scripts/r03_env_ab.sh "$A" "base=PT_TWO_LEVEL=0" "two=PT_TWO_LEVEL=1" "two_rows16=PT_TWO_LEVEL=1 PT_MERGED_LDS_STACK=16" || exit 1
scripts/r03_env_ab.sh "$R" "base=PT_TWO_LEVEL=0" "two=PT_TWO_LEVEL=1" "two_rows20=PT_TWO_LEVEL=1 PT_MERGED_LDS_STACK=20" || exit 1
