"""Sum rocprofv3 --pmc counter_collection CSVs per kernel. Usage: pmc_summarize.py <dir>"""
import csv, glob, os, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(set)
for f in sorted(glob.glob(os.path.join(out, "pass*", "**", "*counter_collection.csv"), recursive=True)):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"].split("(")[0]
            tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
            calls[k].add((f, row["Dispatch_Id"]))
lines = []
for k in sorted(tot):
    n_disp = max(len({d for f2, d in calls[k] if f2 == f0}) for f0 in {f for f, _ in calls[k]})
    lines.append(f"{k}  dispatches={n_disp}")
    for c in sorted(tot[k]):
        lines.append(f"  {c:40s} {tot[k][c]:.6g}")
txt = "\n".join(lines)
print(txt)
open(os.path.join(out, "pmc_summary.txt"), "w").write(txt + "\n")
