set -u
# park threshold of all-opaque scenes' merged walk (PT_MERGED_PARK_OPAQUE), same box
scripts/ab.sh "--configs none --no-fast-arith --workload rocks1080 --spp 64 --steps 2 --warmup 1" base parko1 parko1b || exit 1
scripts/ab.sh "--configs none --no-fast-arith --workload soup1080 --steps 2 --warmup 1" base parko1 parko1b || exit 1
