"""EXPERIMENT: the frame as 1, 2, 3, 4 concurrent bands (PT_CONCURRENT_BANDS). Same bits, launches of one band fill the tails of
the other's. Usage: python scripts/exp_concurrent_bands.py [workload-factory] [width height spp]"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
from pathtracer_amd import scenes  # noqa: E402
from pathtracer_amd.renderer import Renderer  # noqa: E402

factory = sys.argv[1] if len(sys.argv) > 1 else "cornell_scene"
w, h, spp = (int(a) for a in sys.argv[2:5]) if len(sys.argv) >= 5 else (1920, 1080, 128)
kw = {"atrium_scene": {"detail": 16, "tex_size": 1024}, "rocks_scene": {"n_tris": 5_000_000}}.get(factory, {})
scene = getattr(scenes, factory)(**kw)
bg = (0.239, 0.239, 0.239) if factory in ("cornell_scene", "cube_scene") else (0.5, 0.7, 1.0)
ref = None
for conc in [int(x) for x in os.environ.get("PT_EXP_BANDS", "1,2,3,4,1").split(",")]:
    os.environ["PT_CONCURRENT_BANDS"] = str(conc)
    r = Renderer(0)
    packed = r.pack(scene)
    r.upload(packed)
    rp = packed.render_params(r.lib, w, h, spp, bg, 0x5EED0000)
    r.render_params(rp, want_accum=False, want_ids=False)  # warm-up
    best = 1e30
    for _ in range(3):
        _, accum, ids, st = r.render_params(rp)
        best = min(best, st["render_ms"])
    d = hashlib.sha256(accum.tobytes() + ids.tobytes()).hexdigest()[:16]
    key = (d, st["rays"], st["paths"], st["node_visits"], st["tri_tests"], st["nan_samples"])
    if ref is None:
        ref = key
    print(f"{factory} {w}x{h} spp {spp}  bands {conc}: {best:9.3f} ms  {st['rays'] / best / 1e3:9.1f} Mrays/s  launches {st['iterations']}  "
          + ("same bits and counters" if key == ref else f"DIFFERENT {key} vs {ref}"), flush=True)
    r.close()
