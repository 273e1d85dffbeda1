set -u
R="--workload rocks1080 --spp 64 --steps 2 --warmup 1"
scripts/r03_env_ab.sh "$R" "rows9=PT_MERGED_LDS_STACK=9" "rows8=PT_MERGED_LDS_STACK=8" "rows10=PT_MERGED_LDS_STACK=10" "rows11=PT_MERGED_LDS_STACK=11" "v16=PT_PATHS_VERTICES=16" "v64=PT_PATHS_VERTICES=64" || exit 1
