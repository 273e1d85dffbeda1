set -u
R="--workload rocks1080 --spp 64 --steps 2 --warmup 1"
A="--workload atrium1080 --spp 36 --steps 2 --warmup 1"
scripts/r03_env_ab.sh "$R" "b2=PT_CONCURRENT_BANDS=2" "b1=PT_CONCURRENT_BANDS=1" "b3=PT_CONCURRENT_BANDS=3" "b4=PT_CONCURRENT_BANDS=4" "blk128=PT_PATHS_BLOCK_THREADS=128" || exit 1
scripts/r03_env_ab.sh "$A" "b2=PT_CONCURRENT_BANDS=2" "b3=PT_CONCURRENT_BANDS=3" "b4=PT_CONCURRENT_BANDS=4" "xcd=PT_XCD_CHUNK=4" || exit 1
