set -u
for o in 1 2 3 4; do
  PT_PAIR_ORDER=$o timeout -k 10 300 python3 -u -m pytest tests/test_gpu_parity.py -x -q --timeout 120 --timeout-method thread > gpurun_out/order_$o.txt 2>&1 || { echo "parity failed with order $o"; tail -20 gpurun_out/order_$o.txt; exit 1; }
  tail -1 gpurun_out/order_$o.txt
done
bash scripts/r03_step12.sh
