"""Latency of one wave's path vertex when the GPU is otherwise empty: cfg2's scene on an 8x8-pixel image (one tile = one
wave), 121 strata, everything in ONE k_paths launch (PT_PATHS_VERTICES large). Usage: python scripts/lone_wave_latency.py"""
import os, sys
os.environ.setdefault("PT_PATHS_VERTICES", "100000")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracer_amd import scenes
from pathtracer_amd.renderer import Renderer
r = Renderer(0)
p = r.pack(scenes.cornell_scene())
r.upload(p)
for (w, h) in ((8, 8), (64, 64), (256, 256)):
    rp = p.render_params(r.lib, w, h, 128, seed=3)
    r.render_params(rp)
    _, _, _, st = r.render_params(rp)
    waves = (w // 8) * (h // 8)
    vertices = st["shaded_vertices"] + st["paths"]  # upper bound on lane-vertices: every path ends with one more traversal or the step cap
    per_lane = vertices / (w * h)
    print(f"{w}x{h}: {waves} waves, render {st['render_ms']:.3f} ms, {st['iterations']} launches, ~{per_lane:.0f} vertices per lane "
          f"-> {1e3 * st['render_ms'] / per_lane:.2f} us per wave-vertex (mean lane), rays {st['rays']}")
