#!/bin/bash
# EXPERIMENT: LDS top-of-tree cache + short LDS stack in k_paths. Usage: scripts/exp_cache.sh <workload> <spp>
wl=$1; spp=$2
run() {
  tag=$1; shift
  env "$@" timeout -k 10 300 python3 -u bench.py --workload $wl --spp $spp --steps 1 --warmup 0 --configs none --no-cpu-baseline > gpurun_out/exp_cache_${wl}_${tag}.json 2> gpurun_out/exp_cache_${wl}_${tag}.err || { echo "$tag failed"; tail -n 5 gpurun_out/exp_cache_${wl}_${tag}.err; return 1; }
  python3 - "$tag" gpurun_out/exp_cache_${wl}_${tag}.json <<'PY'
import json, sys
j = json.load(open(sys.argv[2]))
k = j.get("kernels", {})
print(f"{sys.argv[1]:28s} {j['value']:9.1f} Mrays/s  step {j['ms_per_step']:8.1f} ms  " + "  ".join(f"{n}: {v['ms']:.1f} ms" for n, v in k.items()) + f"  props_ok={j.get('frame_properties_ok')}")
PY
}
run base PT_PIPELINE=paths &&
run t256_L16 PT_PATHS_BLOCK_THREADS=256 PT_LDS_STACK=16 &&
run t256_L16_c384 PT_PATHS_BLOCK_THREADS=256 PT_LDS_STACK=16 PT_NODE_CACHE=384 &&
run t512_L16_c704 PT_PATHS_BLOCK_THREADS=512 PT_LDS_STACK=16 PT_NODE_CACHE=704 &&
run t1024_L16_c0 PT_PATHS_BLOCK_THREADS=1024 PT_LDS_STACK=16 &&
run t1024_L16_c1408 PT_PATHS_BLOCK_THREADS=1024 PT_LDS_STACK=16 PT_NODE_CACHE=1408 &&
run t1024_L12_c1600 PT_PATHS_BLOCK_THREADS=1024 PT_LDS_STACK=12 PT_NODE_CACHE=1600 &&
run t1024_L8_c1900 PT_PATHS_BLOCK_THREADS=1024 PT_LDS_STACK=8 PT_NODE_CACHE=1900
