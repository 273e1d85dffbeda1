"""Diagnostic: GPU vs oracle differences on a small scene (run on the GPU box)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib
from pathtracer_amd import scenes
from pathtracer_amd.renderer import Renderer

name = sys.argv[1] if len(sys.argv) > 1 else "cube"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 1
w, h = 128, 128
scene = {"cube": scenes.cube_scene, "cornell": scenes.cornell_scene, "textured": scenes.textured_scene}[name]()
r = Renderer(0)
packed = r.pack(scene); r.upload(packed)
rp = packed.render_params(r.lib, w, h, spp, seed=5)
rgba, accum, ids, st = r.render_params(rp)
o_rgba, o_accum, o_ids, o_st = oracle_lib.render(oracle_lib.load(), packed, rp)
print("ids agree", (ids == o_ids).mean())
d = np.abs(accum - o_accum)
bad = (accum.view(np.uint32) != o_accum.view(np.uint32)).any(axis=-1)
print("pixels differing", bad.sum(), "of", bad.size, "max abs", d.max(), "max rel", (d / np.maximum(np.abs(o_accum), 1e-20)).max())
ys, xs = np.nonzero(bad)
for y, x in list(zip(ys, xs))[:6]:
    print((y, x), accum[y, x], o_accum[y, x], [hex(v) for v in accum[y, x].view(np.uint32)], [hex(v) for v in o_accum[y, x].view(np.uint32)])
for k in ["paths", "rays", "node_visits", "tri_tests", "hits_shaded", "shaded_vertices", "nan_samples"]:
    print(k, st[k], o_st[k])
