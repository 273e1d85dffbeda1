#!/bin/bash
# EXPERIMENT: N rays per lane advanced together in k_extend (wavefront pipeline). Usage: scripts/exp_multi.sh <workload> <spp>
wl=$1; spp=$2
run() {
  tag=$1; shift
  env "$@" timeout -k 10 300 python3 -u bench.py --workload $wl --spp $spp --steps 1 --warmup 0 --configs none --no-fast-arith --no-cpu-baseline > gpurun_out/exp_multi_${wl}_${tag}.json 2> gpurun_out/exp_multi_${wl}_${tag}.err || { echo "$tag failed"; tail -n 5 gpurun_out/exp_multi_${wl}_${tag}.err; return 1; }
  python3 - "$tag" gpurun_out/exp_multi_${wl}_${tag}.json <<'PY'
import json, sys
j = json.load(open(sys.argv[2]))
k = j.get("kernels", {})
print(f"{sys.argv[1]:28s} {j['value']:9.1f} Mrays/s  step {j['ms_per_step']:8.1f} ms  " + "  ".join(f"{n}: {v['ms']:.1f} ms" for n, v in k.items()) + f"  props_ok={j.get('frame_properties_ok')} visits={j['counters']['node_visits']} tests={j['counters']['tri_tests']}")
PY
}
run wf PT_PIPELINE=wavefront &&
run m2_w4 PT_PIPELINE=wavefront PT_EXTEND_MULTI=2 &&
run m2_w3 PT_PIPELINE=wavefront PT_EXTEND_MULTI=12 &&
run m2_w5 PT_PIPELINE=wavefront PT_EXTEND_MULTI=15 &&
run m3_w3 PT_PIPELINE=wavefront PT_EXTEND_MULTI=3 &&
run m4_w2 PT_PIPELINE=wavefront PT_EXTEND_MULTI=4 &&
run m2_w4_r12 PT_PIPELINE=wavefront PT_EXTEND_MULTI=2 PT_MULTI_ROWS=12 &&
run m2_w4_r4 PT_PIPELINE=wavefront PT_EXTEND_MULTI=2 PT_MULTI_ROWS=4
