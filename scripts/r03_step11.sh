set -u
A="--workload atrium1080 --spp 36 --steps 2 --warmup 1"
R="--workload rocks1080 --spp 64 --steps 2 --warmup 1"
scripts/r03_env_ab.sh "$A" "exact=PT_MERGED=1" "fast=PT_ARITH=fast" "fast_old=PT_ARITH=fast PT_MERGED=0" "fast_1band=PT_ARITH=fast PT_CONCURRENT_BANDS=1" || exit 1
scripts/r03_env_ab.sh "$R" "exact=PT_MERGED=1" "fast=PT_ARITH=fast" "fast_old=PT_ARITH=fast PT_MERGED=0" || exit 1
