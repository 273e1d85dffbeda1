set -u
scripts/ab.sh "--configs none --no-fast-arith --workload rocks1080 --spp 64 --steps 2 --warmup 1" base w6 || exit 1
scripts/ab.sh "--configs none --no-fast-arith --workload atrium1080 --spp 36 --steps 2 --warmup 1" base w6 || exit 1
PT_LIB_PATH=$PWD/variants/w6.so PT_MERGED_LDS_STACK=12 scripts/r03_env_ab.sh "--workload rocks1080 --spp 64 --steps 2 --warmup 1" "w6rows12=PT_MERGED_LDS_STACK=12" "w6rows9=PT_MERGED_LDS_STACK=9" || exit 1
