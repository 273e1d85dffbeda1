#!/bin/bash
# EXPERIMENT: occupancy bound of the two refill kernels (library variants built with -DPT_*_REFILL_MIN_WAVES=n under experiments/libs).
# Usage: scripts/exp_refill_occ.sh <workload> <spp>
wl=$1; spp=$2
run() {
  tag=$1; shift
  env "$@" PT_PIPELINE=wavefront PT_EXTEND_REFILL=16 PT_REFILL_CHUNK=64 timeout -k 10 300 python3 -u bench.py --workload $wl --spp $spp --steps 1 --warmup 0 --configs none --no-cpu-baseline --no-fast-arith > gpurun_out/exp_rocc_${wl}_${tag}.json 2> gpurun_out/exp_rocc_${wl}_${tag}.err || { echo "$tag failed"; tail -n 5 gpurun_out/exp_rocc_${wl}_${tag}.err; return 1; }
  python3 - "$tag" gpurun_out/exp_rocc_${wl}_${tag}.json <<'PY'
import json, sys
j = json.load(open(sys.argv[2]))
k = j.get("kernels", {})
print(f"{sys.argv[1]:28s} {j['value']:9.1f} Mrays/s  step {j['ms_per_step']:8.1f} ms  " + "  ".join(f"{n}: {v['ms']:.1f} ms" for n, v in k.items()) + f"  props_ok={j.get('frame_properties_ok')}")
PY
}
run default &&
run min5 PT_LIB_PATH=$PWD/experiments/libs/lib_s5.so &&
run min6 PT_LIB_PATH=$PWD/experiments/libs/lib_s6.so &&
run min7 PT_LIB_PATH=$PWD/experiments/libs/lib_s7.so
