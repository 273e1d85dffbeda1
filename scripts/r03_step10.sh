set -u
mkdir -p gpurun_out/r03_w4
export TMPDIR=/tmp
timeout -k 10 900 python3 -u scripts/exp_collectives_one_rank.py > gpurun_out/r03_collectives_one_rank.txt 2> gpurun_out/r03_collectives_one_rank.err || { echo "collectives experiment failed"; tail -20 gpurun_out/r03_collectives_one_rank.err; exit 1; }
cat gpurun_out/r03_collectives_one_rank.txt
# WRITE_SIZE of the merged kernel compiled for 4 waves per SIMD (128 registers: half the spill sites) against the shipped 5-wave build
for v in cur m4; do
  PT_LIB_PATH=$PWD/variants/$v.so PT_CONCURRENT_BANDS=1 timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r03_w4/$v -- python3 bench.py --workload atrium1080 --steps 1 --warmup 0 --configs none --no-fast-arith --no-cpu-baseline --no-kernel-timing > gpurun_out/r03_w4/$v.json 2> gpurun_out/r03_w4/$v.err || { echo "pmc $v failed"; tail -5 gpurun_out/r03_w4/$v.err; exit 1; }
  python3 - $v <<'PY'
import csv, glob, sys, collections, json
v = sys.argv[1]
tot = collections.defaultdict(float); n = collections.defaultdict(set)
for f in glob.glob(f"gpurun_out/r03_w4/{v}/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        tot[k] += float(row["Counter_Value"]); n[k].add(row["Dispatch_Id"])
d = json.load(open(f"gpurun_out/r03_w4/{v}.json"))
for k in tot:
    if "k_paths" in k:
        print(f"{v}: {k}: WRITE_SIZE {tot[k] * 1024 / len(n[k]) / 1e9:.2f} GB per launch over {len(n[k])} launches; bench {d['value']:.0f} Mrays/s")
PY
done
