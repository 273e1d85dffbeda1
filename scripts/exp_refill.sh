#!/bin/bash
# EXPERIMENT: lane refill in k_extend (wavefront pipeline). Usage: scripts/exp_refill.sh <workload> <spp>
wl=$1; spp=$2
run() {
  tag=$1; shift
  env "$@" timeout -k 10 300 python3 -u bench.py --workload $wl --spp $spp --steps 1 --warmup 0 --configs none --no-cpu-baseline > gpurun_out/exp_refill_${wl}_${tag}.json 2> gpurun_out/exp_refill_${wl}_${tag}.err || { echo "$tag failed"; tail -n 5 gpurun_out/exp_refill_${wl}_${tag}.err; return 1; }
  python3 - "$tag" gpurun_out/exp_refill_${wl}_${tag}.json <<'PY'
import json, sys
j = json.load(open(sys.argv[2]))
k = j.get("kernels", {})
print(f"{sys.argv[1]:28s} {j['value']:9.1f} Mrays/s  step {j['ms_per_step']:8.1f} ms  " + "  ".join(f"{n}: {v['ms']:.1f} ms" for n, v in k.items()) + f"  props_ok={j.get('frame_properties_ok')}")
PY
}
run wavefront PT_PIPELINE=wavefront &&
run refill8 PT_PIPELINE=wavefront PT_EXTEND_REFILL=8 &&
run refill16 PT_PIPELINE=wavefront PT_EXTEND_REFILL=16 &&
run refill32 PT_PIPELINE=wavefront PT_EXTEND_REFILL=32 &&
run refill16_c64 PT_PIPELINE=wavefront PT_EXTEND_REFILL=16 PT_REFILL_CHUNK=64 &&
run refill16_c1024 PT_PIPELINE=wavefront PT_EXTEND_REFILL=16 PT_REFILL_CHUNK=1024 &&
run refill16_b512 PT_PIPELINE=wavefront PT_EXTEND_REFILL=16 PT_REFILL_BLOCKS=512 &&
run refill16_b2048 PT_PIPELINE=wavefront PT_EXTEND_REFILL=16 PT_REFILL_BLOCKS=2048
