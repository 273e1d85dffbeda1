#!/usr/bin/env python3
"""Digest of a frame's accumulators + primary ids under the current PT_* environment (pipeline A/B debugging).
Usage: pipeline_digest.py [width height spp]"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracer_amd import scenes
from pathtracer_amd.renderer import Renderer
w, h, spp = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (1920, 1080, 16)
r = Renderer(0)
packed = r.pack(scenes.cornell_scene())
r.upload(packed)
rp = packed.render_params(r.lib, w, h, spp, seed=0x5EED0000)
rgba, accum, ids, st = r.render_params(rp)
print(hashlib.sha256(accum.tobytes()).hexdigest()[:16], hashlib.sha256(ids.tobytes()).hexdigest()[:16], st["paths"], st["rays"], st["node_visits"], st["nan_samples"], st["iterations"],
      {k: os.environ[k] for k in os.environ if k.startswith("PT_")})
