#!/bin/bash
# Same-box A/B over environment settings with one library. Usage:
#   scripts/ab_env.sh <variant> "<bench args>" "ENV1=a ENV2=b" "ENV1=c" ...
name=$1; args=$2; shift; shift
i=0
for envs in "$@"; do
  i=$((i+1))
  echo "== $envs"
  env $envs PT_LIB_PATH=$PWD/variants/$name.so timeout -k 10 400 python3 -u bench.py --no-cpu-baseline $args > gpurun_out/abenv_${name}_$i.json 2> gpurun_out/abenv_${name}_$i.err || { echo "failed"; tail -3 gpurun_out/abenv_${name}_$i.err; exit 1; }
  python3 scripts/show_bench.py gpurun_out/abenv_${name}_$i.json
done
