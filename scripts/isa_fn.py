#!/usr/bin/env python3
"""Cut one kernel out of the assembly scripts/kernel_resources.sh leaves behind (/tmp/pt_kernels_res.s) and say where its scratch
traffic sits: per basic block with a `Loop` comment, the instruction count and the scratch / global / LDS ops, sorted by loop depth.
Usage: isa_fn.py <fragment of the mangled name> [out.s]"""
import re, sys
frag = sys.argv[1]
lines = open("/tmp/pt_kernels_res.s").read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and frag in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
fn = lines[start:end]
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write("\n".join(fn))
print(lines[start], len(fn), "lines")
blocks, cur = [], None
for i, l in enumerate(fn):
    if re.match(r"^\.LBB\d+_\d+:", l) or l.startswith("; %bb."):
        cur = {"label": l.split(":")[0].strip(), "line": i, "depth": 0, "n": 0, "scratch": 0, "global": 0, "ds": 0}
        m = re.search(r"Depth=(\d+)", l)
        if m: cur["depth"] = int(m.group(1))
        blocks.append(cur)
        continue
    if cur is None: continue
    m = re.search(r";\s+(?:=>)?\s*(?:This )?(?:Inner )?Loop Header: Depth=(\d+)|in Loop: Header=\S+ Depth=(\d+)", l)
    if m and l.lstrip().startswith(";"):
        cur["depth"] = max(cur["depth"], int(m.group(1) or m.group(2)))
        continue
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."): continue
    cur["n"] += 1
    if t.startswith("scratch_"): cur["scratch"] += 1
    if t.startswith("global_"): cur["global"] += 1
    if t.startswith("ds_"): cur["ds"] += 1
by_depth = {}
for b in blocks:
    d = by_depth.setdefault(b["depth"], {"n": 0, "scratch": 0, "global": 0, "ds": 0, "blocks": 0})
    for k in ("n", "scratch", "global", "ds"): d[k] += b[k]
    d["blocks"] += 1
for d in sorted(by_depth):
    print("depth", d, by_depth[d])
deep = max(by_depth)
print("blocks of the two deepest loop levels with scratch ops:")
for b in blocks:
    if b["depth"] >= deep - 1 and b["scratch"]:
        print("  ", b)
