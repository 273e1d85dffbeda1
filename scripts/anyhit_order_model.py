#!/usr/bin/env python3
"""Diagnostic (CPU only): does the ORDER in which an any-hit shadow walk enters two hit children matter? (All-opaque scenes: the
answer is a yes / no, the same for every order; the oracle's experiment switch orc_set_anyhit_order changes only the work.)
Usage: anyhit_order_model.py <workload> [rows]   (SPP=<n>, default 16)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, oracle_lib
from pathtracer_amd import scenes, lib
from pathtracer_amd.scene import PackedScene

name = sys.argv[1]; rows = int(sys.argv[2]) if len(sys.argv) > 2 else 8
factory, kw, w, h, spp, _ = bench.WORKLOADS[name]
spp = int(os.environ.get("SPP", 16))
L = lib.load(); O = oracle_lib.load()
packed = PackedScene(getattr(scenes, factory)(**kw), L)
rp = packed.render_params(L, w, h, spp, bench.BG.get(name, (0.239,) * 3), bench.SEED)
y = h // 2
ref = base = None
print(f"{name}: rows {y}..{y + rows}, {spp} spp, own triangle first (ORC_FLAG_SELF_FIRST) in every line")
for order, label in enumerate(("nearer first (as now)", "farther first", "larger box first", "smaller box first", "leaf before inner node",
                               "longer span first", "shorter span first", "larger t_far first", "leaf first, else farther", "relative span first", "unclamped span first")):
    O.orc_set_anyhit_order(order)
    _, accum, ids, st = oracle_lib.render(O, packed, rp, threads=8, rows=(y, y + rows), flags=oracle_lib.GPU_MIRROR | 4)
    a = accum[y:y + rows].view(np.uint32).copy()
    if ref is None:
        ref, base = a, st
    print(f"  {label:24s}: frame {'identical' if (a == ref).all() else 'DIFFERENT'}; node visits {100.0 * st['node_visits'] / base['node_visits']:6.1f} %, "
          f"triangle tests {100.0 * st['tri_tests'] / base['tri_tests']:6.1f} %")
O.orc_set_anyhit_order(0)
