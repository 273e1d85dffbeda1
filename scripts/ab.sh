#!/bin/bash
# Same-box A/B: run bench.py for each variants/<name>.so given. Usage:
#   scripts/ab.sh "<bench args>" name1 name2 ...      (results: gpurun_out/ab_<name>_<tag>.json)
args=$1; shift
tag=$(echo "$args" | tr -c 'A-Za-z0-9' '_' | cut -c1-40)
for name in "$@"; do
  PT_LIB_PATH=$PWD/variants/$name.so timeout -k 10 400 python3 -u bench.py --no-cpu-baseline $args > gpurun_out/ab_${name}_${tag}.json 2> gpurun_out/ab_${name}_${tag}.err || { echo "$name failed"; tail -3 gpurun_out/ab_${name}_${tag}.err; exit 1; }
  python3 scripts/show_bench.py gpurun_out/ab_${name}_${tag}.json
done
