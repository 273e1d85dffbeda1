set -u
# one light sample per vertex: 40 B of stash instead of 80 -> room for more LDS stack rows at the same occupancy
PT_MERGED_LDS_STACK=12 timeout -k 10 300 python3 -u -m pytest tests/test_gpu_parity.py -x -q --timeout 120 --timeout-method thread > gpurun_out/parity_stash.txt 2>&1 || { echo "parity failed"; tail -20 gpurun_out/parity_stash.txt; exit 1; }
tail -n 1 gpurun_out/parity_stash.txt
R="--workload rocks1080 --spp 64 --steps 2 --warmup 1"
scripts/r03_env_ab.sh "$R" "rows9=PT_MERGED_LDS_STACK=9" "rows11=PT_MERGED_LDS_STACK=11" "rows13=PT_MERGED_LDS_STACK=13" "rows16=PT_MERGED_LDS_STACK=16" "rows19=PT_MERGED_LDS_STACK=19" "rows20=PT_MERGED_LDS_STACK=20" || exit 1
