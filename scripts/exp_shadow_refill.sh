#!/bin/bash
# EXPERIMENT: lane refill in k_shadow as well as k_extend (wavefront pipeline, all-opaque scenes in global memory).
# Usage: scripts/exp_shadow_refill.sh <workload> <spp>
wl=$1; spp=$2
run() {
  tag=$1; shift
  env "$@" timeout -k 10 300 python3 -u bench.py --workload $wl --spp $spp --steps 1 --warmup 0 --configs none --no-cpu-baseline --no-fast-arith > gpurun_out/exp_srefill_${wl}_${tag}.json 2> gpurun_out/exp_srefill_${wl}_${tag}.err || { echo "$tag failed"; tail -n 5 gpurun_out/exp_srefill_${wl}_${tag}.err; return 1; }
  python3 - "$tag" gpurun_out/exp_srefill_${wl}_${tag}.json <<'PY'
import json, sys, hashlib
j = json.load(open(sys.argv[2]))
k = j.get("kernels", {})
print(f"{sys.argv[1]:28s} {j['value']:9.1f} Mrays/s  step {j['ms_per_step']:8.1f} ms  " + "  ".join(f"{n}: {v['ms']:.1f} ms" for n, v in k.items()) + f"  props_ok={j.get('frame_properties_ok')}")
PY
}
run paths PT_PIPELINE=paths &&
run wavefront PT_PIPELINE=wavefront &&
run refill_extend PT_PIPELINE=wavefront PT_EXTEND_REFILL=16 PT_REFILL_CHUNK=64 PT_SHADOW_REFILL=0 &&
run refill_both16 PT_PIPELINE=wavefront PT_EXTEND_REFILL=16 PT_REFILL_CHUNK=64 &&
run refill_both8 PT_PIPELINE=wavefront PT_EXTEND_REFILL=8 PT_REFILL_CHUNK=64 &&
run refill_both32 PT_PIPELINE=wavefront PT_EXTEND_REFILL=32 PT_REFILL_CHUNK=64
