"""EXPERIMENT: sweep of the launch-shape knobs on one workload, host side only — every setting runs the shipped
kernels and must give the same bits and counters. Usage:
  python scripts/exp_knobs.py <workload-factory> <width> <height> <spp> "K1=v1,K2=v2;K3=v3;..."   (';' separates settings)"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
from pathtracer_amd import scenes  # noqa: E402
from pathtracer_amd.renderer import Renderer  # noqa: E402

factory = sys.argv[1]
w, h, spp = (int(a) for a in sys.argv[2:5])
settings = [dict(kv.split("=") for kv in s.split(",") if kv) for s in sys.argv[5].split(";")]
kw = {"atrium_scene": {"detail": 16, "tex_size": 1024}, "rocks_scene": {"n_tris": 5_000_000}}.get(factory, {})
scene = getattr(scenes, factory)(**kw)
bg = (0.239, 0.239, 0.239) if factory in ("cornell_scene", "cube_scene") else (0.5, 0.7, 1.0)
packed = None
ref = None
for env in settings:
    for k, v in env.items():
        os.environ[k] = v
    r = Renderer(0)
    if packed is None:
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
    print(f"{factory} {w}x{h} spp {spp}  {env or 'defaults'}: {best:9.3f} ms  {st['rays'] / best / 1e3:9.1f} Mrays/s  launches {st['iterations']}  "
          + ("same bits and counters" if key == ref else f"DIFFERENT {key} vs {ref}"), flush=True)
    r.close()
    for k in env:
        del os.environ[k]
