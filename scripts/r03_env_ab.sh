#!/bin/bash
# Same-box A/B of environment knobs on one library. Usage: scripts/r03_env_ab.sh "<bench args>" "<tag>=<ENV=val ENV=val>" ...
args=$1; shift
wl=$(echo "$args" | sed -n 's/.*--workload \([a-z0-9]*\).*/\1/p')
for spec in "$@"; do
  tag=${spec%%=*}; envs=${spec#*=}
  out=gpurun_out/env_${wl}_${tag}
  env $envs timeout -k 10 400 python3 -u bench.py --no-cpu-baseline --configs none --no-fast-arith $args > $out.json 2> $out.err || { echo "$tag failed"; tail -3 $out.err; exit 1; }
  echo "== $tag ($envs)"; python3 scripts/show_bench.py $out.json
done
