#!/usr/bin/env python3
"""HBM-side traffic per launch of every kernel from the FETCH_SIZE / WRITE_SIZE passes of
scripts/pmc_passes.sh, corrected as MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE and
WRITE_SIZE are in KiB; FETCH_SIZE tallies the 128-B requests of wide (16 B/lane) reads at 64 B,
so it is doubled (an upper bound for narrower reads, which the guide leaves uncalibrated).
Usage: pmc_traffic.py <pmc dir> <out.json> [workload name]"""
import csv, glob, json, os, sys, collections
src, out = sys.argv[1], sys.argv[2]
name = sys.argv[3] if len(sys.argv) > 3 else "cornell1080"
tot = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(lambda: collections.defaultdict(set))
for f in sorted(glob.glob(os.path.join(src, "pass*", "**", "*counter_collection.csv"), recursive=True)):
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] not in ("FETCH_SIZE", "WRITE_SIZE"):
            continue
        k = row["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        if k.startswith("ptk_fast::"):
            continue  # the opt-in fast-arithmetic build's leg of bench.py, if it ran
        k = k.replace("ptk_exact::", "")
        tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
        disp[k][row["Counter_Name"]].add(row["Dispatch_Id"])
res = {"workload": name, "source": src, "correction": "bytes = 2 * FETCH_SIZE KiB * 1024 + WRITE_SIZE KiB * 1024", "kernels": {}}
for k in tot:
    n = max(len(v) for v in disp[k].values())
    fetch, write = tot[k].get("FETCH_SIZE", 0.0), tot[k].get("WRITE_SIZE", 0.0)
    res["kernels"][k] = {"launches": n, "fetch_kib_raw": fetch, "write_kib_raw": write,
                         "bytes_per_launch": (2.0 * fetch + write) * 1024.0 / max(n, 1)}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res["kernels"], indent=1))
