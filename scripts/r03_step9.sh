set -u
mkdir -p gpurun_out
PT_LIB_PATH=$PWD/variants/guard.so timeout -k 10 600 python3 -u -m pytest tests/test_gpu_wide4.py tests/test_gpu_parity.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r03_guard.log 2>&1 || { echo "guard run failed"; tail -40 gpurun_out/r03_guard.log; exit 1; }
tail -2 gpurun_out/r03_guard.log; grep -c "\[guard\]" gpurun_out/r03_guard.log
timeout -k 10 600 python3 -u -m pytest tests/test_gpu_wide4.py tests/test_gpu_fast_arith.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r03_wide.log 2>&1 || { echo "wide/fast tests failed"; tail -40 gpurun_out/r03_wide.log; exit 1; }
tail -2 gpurun_out/r03_wide.log
R="--workload rocks1080 --spp 64 --steps 2 --warmup 1"
A="--workload atrium1080 --spp 36 --steps 2 --warmup 1"
S="--workload soup1080 --steps 2 --warmup 1"
for W in "$R" "$A" "$S"; do
scripts/r03_env_ab.sh "$W" "binary=PT_MERGED=1" "wide4=PT_BVH=wide4" || exit 1
done
