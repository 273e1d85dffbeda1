set -u
# far-child touch experiment (PT_TOUCH_FAR): parity on the guard-free build first, then same-box A/B
for v in touch1 touch2; do
  PT_LIB_PATH=$PWD/variants/$v.so timeout -k 10 300 python3 -u -m pytest tests/test_gpu_parity.py -x -q --timeout 120 --timeout-method thread > gpurun_out/parity_$v.txt 2>&1 || { echo "parity failed: $v"; tail -20 gpurun_out/parity_$v.txt; exit 1; }
  tail -n 1 gpurun_out/parity_$v.txt
done
scripts/ab.sh "--configs none --no-fast-arith --workload atrium1080 --spp 36 --steps 2 --warmup 1" base touch1 touch2 || exit 1
scripts/ab.sh "--configs none --no-fast-arith --workload rocks1080 --spp 64 --steps 2 --warmup 1" base touch1 touch2 || exit 1
