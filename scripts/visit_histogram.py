#!/usr/bin/env python3
"""Diagnostic: which share of the inner-node visits of a workload falls on the first C pair records of the breadth-first
layout (the candidates for an LDS-resident top-of-tree cache)? CPU oracle on a few rows, counters per node.
Usage: visit_histogram.py <workload> [rows]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, oracle_lib
from pathtracer_amd import scenes, lib
from pathtracer_amd.scene import PackedScene

name = sys.argv[1]; rows = int(sys.argv[2]) if len(sys.argv) > 2 else 2
factory, kw, w, h, spp, _ = bench.WORKLOADS[name]
spp = int(os.environ.get("SPP", 16))
L = lib.load(); O = oracle_lib.load()
packed = PackedScene(getattr(scenes, factory)(**kw), L)
rp = packed.render_params(L, w, h, spp, bench.BG.get(name, (0.239,) * 3), bench.SEED)
hist = np.zeros(len(packed.packed_nodes), dtype=np.uint64)
O.orc_set_visit_histogram(hist.ctypes.data_as(C.c_void_p))
ys = [h // 4, h // 2, 3 * h // 4][:max(1, min(3, rows))]
for y in ys:
    oracle_lib.render(O, packed, rp, threads=8, rows=(y, y + max(1, rows // len(ys))), want_ids=False, flags=oracle_lib.GPU_MIRROR)
O.orc_set_visit_histogram(None)
N = packed.packed_nodes
# breadth-first rank of the inner nodes (pt_context.hip's record order)
order = [0] if N[0]["triangle_count"] == 0 else []
head = 0
while head < len(order):
    n = order[head]; head += 1
    for side in (0, 1):
        c = int(N[n]["left_or_begin"]) + side
        if N[c]["triangle_count"] == 0:
            order.append(c)
v = hist[np.array(order)].astype(np.float64)
tot = v.sum(); cum = np.cumsum(v) / tot
print(f"{name}: {len(order)} pair records, {int(tot)} visits sampled")
for c in (64, 128, 256, 384, 512, 704, 1024, 1408, 2048, 4096, 8192, 16384, 65536):
    if c <= len(order):
        print(f"  first {c:6d} records ({c * 64 // 1024:5d} KiB): {100 * cum[c - 1]:5.1f} % of the visits")
