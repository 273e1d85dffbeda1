#!/bin/bash
# EXPERIMENT: short LDS stack (occupancy) x lane refill in the wavefront pipeline. Usage: scripts/exp_occ.sh <workload> <spp>
wl=$1; spp=$2
run() {
  tag=$1; shift
  env "$@" timeout -k 10 300 python3 -u bench.py --workload $wl --spp $spp --steps 1 --warmup 0 --configs none --no-cpu-baseline > gpurun_out/exp_occ_${wl}_${tag}.json 2> gpurun_out/exp_occ_${wl}_${tag}.err || { echo "$tag failed"; tail -n 5 gpurun_out/exp_occ_${wl}_${tag}.err; return 1; }
  python3 - "$tag" gpurun_out/exp_occ_${wl}_${tag}.json <<'PY'
import json, sys
j = json.load(open(sys.argv[2]))
k = j.get("kernels", {})
print(f"{sys.argv[1]:28s} {j['value']:9.1f} Mrays/s  step {j['ms_per_step']:8.1f} ms  " + "  ".join(f"{n}: {v['ms']:.1f} ms" for n, v in k.items()) + f"  props_ok={j.get('frame_properties_ok')}")
PY
}
run paths PT_PIPELINE=paths &&
run paths_L16 PT_PIPELINE=paths PT_LDS_STACK=16 &&
run wf PT_PIPELINE=wavefront &&
run wf_L16 PT_PIPELINE=wavefront PT_LDS_STACK=16 &&
run wf_L12 PT_PIPELINE=wavefront PT_LDS_STACK=12 &&
run wf_L8 PT_PIPELINE=wavefront PT_LDS_STACK=8 &&
run wf_L16_refill PT_PIPELINE=wavefront PT_LDS_STACK=16 PT_EXTEND_REFILL=16 PT_REFILL_CHUNK=64 &&
run wf_L12_refill PT_PIPELINE=wavefront PT_LDS_STACK=12 PT_EXTEND_REFILL=16 PT_REFILL_CHUNK=64 &&
run wf_L12_refill8 PT_PIPELINE=wavefront PT_LDS_STACK=12 PT_EXTEND_REFILL=8 PT_REFILL_CHUNK=64
