#!/bin/bash
# What the fabric-side counters read at the random-gather ceiling (scripts/gather_ceiling.hip): one rocprofv3 --pmc run per set.
set -u
out=gpurun_out/gather_pmc; mkdir -p $out; export TMPDIR=/tmp
timeout -k 10 120 scripts/gather_ceiling.bin 2000 > $out/plain.txt 2>&1 || exit 1
i=0
for set in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $set --output-format csv -d $out/pass$i -- scripts/gather_ceiling.bin 2000 > $out/pass$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/pass$i.log; exit 1; }
done
python3 - $out <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
vals = collections.defaultdict(dict)
for f in sorted(glob.glob(os.path.join(out, "pass*", "**", "*counter_collection.csv"), recursive=True)):
    for row in csv.DictReader(open(f)):
        vals[int(row["Dispatch_Id"])][row["Counter_Name"]] = vals[int(row["Dispatch_Id"])].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
ids = sorted(vals)
plain = [l for l in open(os.path.join(out, "plain.txt")) if l.startswith("2^")]
print("config (plain run) | per dispatch: FETCH_SIZE KiB, TCC hit / miss / req, TCP accesses / TCP->TCC reads")
for n, line in enumerate(plain):
    d = vals[ids[2 * n + 1]]  # the second repetition of each configuration
    visits = 1280 * 4 * int(line.split()[3]) * 2000
    print(line.rstrip()[:96], "| fetch B/visit %.1f  tcc req/visit %.2f miss/visit %.2f  tcp acc/visit %.2f tcp->tcc/visit %.2f" % (
        d.get("FETCH_SIZE", 0) * 1024 / visits, d.get("TCC_REQ_sum", 0) / visits, d.get("TCC_MISS_sum", 0) / visits,
        d.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0) / visits, d.get("TCP_TCC_READ_REQ_sum", 0) / visits))
PY
