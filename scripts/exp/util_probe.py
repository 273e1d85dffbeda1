"""Experiment: read the instrumentation counters of an instrumented variant library after one cfg2 frame."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from pathtracer_amd import scenes, lib
from pathtracer_amd.renderer import Renderer
r = Renderer(0)
p = r.pack(scenes.cornell_scene())
r.upload(p)
rp = p.render_params(r.lib, 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 16, (0.0, 0.0, 0.0), 7)
_, _, _, st = r.render_params(rp)
out = (C.c_ulonglong * 32)()
r.lib.pt_exp_util(out)
print([int(x) for x in out[:16]], st["rays"])
