#!/bin/bash
# EXPERIMENT: ray re-ordering in the wavefront pipeline. Usage: scripts/exp_sort.sh <workload> <spp>
wl=$1; spp=$2
run() {
  tag=$1; shift
  env "$@" timeout -k 10 300 python3 -u bench.py --workload $wl --spp $spp --steps 1 --warmup 0 --configs none --no-cpu-baseline > gpurun_out/exp_sort_${wl}_${tag}.json 2> gpurun_out/exp_sort_${wl}_${tag}.err || { echo "$tag failed"; tail -n 5 gpurun_out/exp_sort_${wl}_${tag}.err; return 1; }
  python3 - "$tag" gpurun_out/exp_sort_${wl}_${tag}.json <<'PY'
import json, sys
j = json.load(open(sys.argv[2]))
k = j.get("kernels", {})
print(f"{sys.argv[1]:28s} {j['value']:9.1f} Mrays/s  step {j['ms_per_step']:8.1f} ms  " + "  ".join(f"{n}: {v['ms']:.1f} ms" for n, v in k.items()) + f"  props_ok={j.get('frame_properties_ok')}")
PY
}
run paths PT_PIPELINE=paths &&
run wavefront PT_PIPELINE=wavefront &&
run wf_sort1 PT_PIPELINE=wavefront PT_SORT_RAYS=1 &&
run wf_sort2 PT_PIPELINE=wavefront PT_SORT_RAYS=2 &&
run wf_sort1_b4 PT_PIPELINE=wavefront PT_SORT_RAYS=1 PT_SORT_BITS=4 &&
run wf_sort1_b8 PT_PIPELINE=wavefront PT_SORT_RAYS=1 PT_SORT_BITS=8 &&
run wf_sort1_sh PT_PIPELINE=wavefront PT_SORT_RAYS=1 PT_SORT_SHADOW=1 &&
run wf_sort2_sh PT_PIPELINE=wavefront PT_SORT_RAYS=2 PT_SORT_SHADOW=1
