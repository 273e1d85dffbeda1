set -u
A="--workload atrium1080 --spp 36 --steps 2 --warmup 1"
R="--workload rocks1080 --spp 64 --steps 2 --warmup 1"
scripts/r03_env_ab.sh "$A" "blk_default=PT_MERGED=1" "blk64=PT_PATHS_BLOCK_THREADS=64" "blk128=PT_PATHS_BLOCK_THREADS=128" || exit 1
scripts/r03_env_ab.sh "$R" "blk_default=PT_MERGED=1" "blk64=PT_PATHS_BLOCK_THREADS=64" || exit 1
