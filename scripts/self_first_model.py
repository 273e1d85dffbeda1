#!/usr/bin/env python3
"""Diagnostic (CPU only): the shading point's OWN triangle as the first candidate of an any-hit shadow query (all-opaque scenes;
oracle flag ORC_FLAG_SELF_FIRST, an experiment no kernel implements yet). A light on the far side of the surface sends the shadow
ray — which starts NORMAL_DELTA above the triangle — straight back through it, and the walk that finds this takes a full descent
of the tree. Renders a band of rows both ways: the frames must be the same to the bit; the counters say what the walks shrink by.
Usage: self_first_model.py <workload> [rows]   (SPP=<n>, default 16)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, oracle_lib
from pathtracer_amd import scenes, lib
from pathtracer_amd.scene import PackedScene

SELF_FIRST = 4
name = sys.argv[1]; rows = int(sys.argv[2]) if len(sys.argv) > 2 else 8
factory, kw, w, h, spp, _ = bench.WORKLOADS[name]
spp = int(os.environ.get("SPP", 16))
L = lib.load(); O = oracle_lib.load()
packed = PackedScene(getattr(scenes, factory)(**kw), L)
rp = packed.render_params(L, w, h, spp, bench.BG.get(name, (0.239,) * 3), bench.SEED)
y = h // 2
res = {}
for label, flags in (("reference form", 0), ("any-hit (the HIP path)", oracle_lib.GPU_MIRROR), ("any-hit + own triangle first", oracle_lib.GPU_MIRROR | SELF_FIRST)):
    _, accum, ids, st = oracle_lib.render(O, packed, rp, threads=8, rows=(y, y + rows), flags=flags)
    res[label] = (accum[y:y + rows].copy(), ids[y:y + rows].copy(), st)
a0, i0, _ = res["reference form"]
print(f"{name}: rows {y}..{y + rows}, {spp} spp")
base = res["any-hit (the HIP path)"][2]
for label, (a, i, st) in res.items():
    same = bool((a.view(np.uint32) == a0.view(np.uint32)).all() and (i == i0).all())
    print(f"  {label:30s}: frame {'identical' if same else 'DIFFERENT'}; rays {st['rays']}, walked {st['rays_shadow_walked']}, decided by one triangle "
          f"{st['rays_shadow_shortcut']}; node visits {st['node_visits']} ({100.0 * st['node_visits'] / base['node_visits']:.1f} %), "
          f"triangle tests {st['tri_tests']} ({100.0 * st['tri_tests'] / base['tri_tests']:.1f} %)")
