set -u
export PT_TWO_LEVEL=1
scripts/ab.sh "--configs none --no-fast-arith --workload atrium1080 --spp 36 --steps 2 --warmup 1" two4 two5 || exit 1
scripts/ab.sh "--configs none --no-fast-arith --workload rocks1080 --spp 64 --steps 2 --warmup 1" two4 two5 || exit 1
