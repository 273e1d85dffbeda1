"""Model (oracle only, no GPU): what would any-hit shadow walks be worth on cfg3? The atrium is not all-opaque (cloth banners with
alpha cut-outs), so its shadow walks are closest-hit walks. Upper bound of an exact two-pass scheme (any-hit walk to the first
opaque hit, then a walk of the few sub-trees that hold non-opaque triangles): the same scene with the cloth made opaque, shadow
walks closest-hit (as shipped) against any-hit. Usage: python scripts/anyhit_cfg3_model.py [rows] [spp]"""
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
s = scenes.atrium_scene(detail=16, tex_size=32)
names = s.materials.dtype.names
print("material fields:", names)
cloth = len(s.materials) - 1
print("cloth material:", s.materials[cloth])
s.materials[cloth]["base_color_texture_index"] = -1
p = PackedScene(s, O, "orc_")
rp = p.render_params(O, 1920, 1080, spp, (0.5, 0.7, 1.0), 77, prefix="orc_")
y = 540 - rows // 2
M = oracle_lib
res = {}
for label, flags in (("closest-hit shadow walks (as shipped on cfg3)", M.CULL_BEHIND | M.OPAQUE_TRIANGLES),
                     ("any-hit, nearer child first", M.ANYHIT_OPAQUE_SHADOW | M.CULL_BEHIND),
                     ("any-hit + own triangle first + longer span first", M.GPU_MIRROR)):
    st = oracle_lib.render(O, p, rp, threads=8, rows=(y, y + rows), want_ids=False, flags=flags)[3]
    res[label] = st
    print(f"{label:55s} rays {st['rays']:>10d} closest {st['rays_closest']:>9d} walked {st['rays_shadow_walked']:>9d} shortcut {st['rays_shadow_shortcut']:>8d} "
          f"node visits {st['node_visits']:>11d} tri tests {st['tri_tests']:>10d}")
a = res["closest-hit shadow walks (as shipped on cfg3)"]["node_visits"]
for k, st in res.items():
    print(f"{k:55s} node visits {st['node_visits'] / a:.3f} of shipped")
