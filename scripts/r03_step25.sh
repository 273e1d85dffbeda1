set -u
A="--workload atrium1080 --spp 36 --steps 2 --warmup 1"
R="--workload rocks1080 --spp 64 --steps 2 --warmup 1"
scripts/r03_env_ab.sh "$A" "merged=PT_MERGED=1" "paths=PT_MERGED=0" "wavefront=PT_PIPELINE=wavefront" || exit 1
scripts/r03_env_ab.sh "$R" "merged=PT_MERGED=1" "paths=PT_MERGED=0" "wavefront=PT_PIPELINE=wavefront" || exit 1
