#!/usr/bin/env python3
"""Diagnostic (CPU only): how many node visits of the closest-hit walks are 'stale' — the node was pushed when its entry distance
was below the closest hit, and popped after a nearer hit had been found? The reference visits it anyway; for an INNER node the
visit provably changes nothing (its children's boxes lie inside its own, so both fail the distance test), so a walk that keeps
the entry distance beside each stack entry may skip it and stay exact. Usage: popcull_model.py <workload> [rows]  (SPP=<n>)"""
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
O.orc_popcull_counts.argtypes = [C.c_void_p, C.c_int]
packed = PackedScene(getattr(scenes, factory)(**kw), L)
rp = packed.render_params(L, w, h, spp, bench.BG.get(name, (0.239,) * 3), bench.SEED)
y = h // 2
O.orc_popcull_counts(None, 1)
_, _, _, st = oracle_lib.render(O, packed, rp, threads=1, rows=(y, y + rows), want_ids=False, flags=oracle_lib.GPU_MIRROR)
out = np.zeros(4, dtype=np.uint64)
O.orc_popcull_counts(out.ctypes.data_as(C.c_void_p), 0)
pops, inner, leaf, leaf_hit = (int(x) for x in out)
print(f"{name}: rows {y}..{y + rows}, {spp} spp: {st['node_visits']} node visits in all, {pops} pops of closest-hit walks")
print(f"  stale inner nodes: {inner} = {100.0 * inner / st['node_visits']:.1f} % of all node visits (skippable, exact)")
print(f"  stale leaves     : {leaf} = {100.0 * leaf / max(st['tri_tests'], 1):.1f} % of all triangle tests' leaves; a triangle was accepted in {leaf_hit} of them")
