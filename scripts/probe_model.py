"""Model (oracle only): the transparent-probe scheme on scenes that are not all-opaque — image identical, what the counters become.
Usage: python scripts/probe_model.py [rows] [spp]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib
from pathtracer_amd.scene import PackedScene
from pathtracer_amd import scenes
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 16
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 9
O = oracle_lib.load()
M = oracle_lib
for name, s, w, h in (("atrium(16)", scenes.atrium_scene(detail=16, tex_size=32), 1920, 1080), ("textured(48)", scenes.textured_scene(grid=48), 640, 360),
                      ("cornell+glass", scenes.cornell_scene(True, True, True, True), 320, 180)):
    p = PackedScene(s, O, "orc_")
    rp = p.render_params(O, w, h, spp, (0.5, 0.7, 1.0), 77, prefix="orc_")
    y = h // 2 - rows // 2
    base = None
    for label, flags in (("reference form", 0), ("shipped (round 3)", M.GPU_MIRROR), ("probe", M.GPU_MIRROR | M.TRANSPARENT_PROBE)):
        rgba, accum, ids, st = oracle_lib.render(O, p, rp, threads=8, rows=(y, y + rows), flags=flags)
        if base is None: base = (rgba, accum, ids, st)
        same = rgba.tobytes() == base[0].tobytes() and accum.tobytes() == base[1].tobytes() and (ids == base[2]).all()
        print(f"{name:14s} {label:18s} image identical {same}  rays {st['rays']:>9d} (closest {st['rays_closest']}, walked {st['rays_shadow_walked']}, shortcut {st['rays_shadow_shortcut']}) "
              f"node visits {st['node_visits']:>10d} ({st['node_visits'] / base[3]['node_visits']:.3f}) tri tests {st['tri_tests']:>9d} hits {st['hits_shaded']} taps {st['texture_taps']}")
