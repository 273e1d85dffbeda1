set -u
# record-order experiment (PT_PAIR_ORDER): same results, other addresses
A="--workload atrium1080 --spp 36 --steps 2 --warmup 1"
R="--workload rocks1080 --spp 64 --steps 2 --warmup 1"
scripts/r03_env_ab.sh "$A" "bfs=PT_PAIR_ORDER=0" "dfs=PT_PAIR_ORDER=1" "top8=PT_PAIR_ORDER=2" "sib=PT_PAIR_ORDER=3" "sib128=PT_PAIR_ORDER=4" || exit 1
scripts/r03_env_ab.sh "$R" "bfs=PT_PAIR_ORDER=0" "dfs=PT_PAIR_ORDER=1" "top8=PT_PAIR_ORDER=2" "top14=PT_PAIR_ORDER=2 PT_PAIR_ORDER_LEVELS=14" "sib=PT_PAIR_ORDER=3" "sib128=PT_PAIR_ORDER=4" || exit 1
