"""EXPERIMENT: path kernel against the wavefront pipeline with lane refill, on all-opaque scenes of several sizes read from
global memory (which scenes should take the second automatically?). Usage: python scripts/exp_pipeline_choice.py [spp]"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402  (the HIP runtime torch bundles must load first)
from pathtracer_amd import scenes  # noqa: E402
from pathtracer_amd.renderer import Renderer  # noqa: E402

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
CASES = [("rocks 30k", lambda: scenes.rocks_scene(30_000)), ("rocks 250k", lambda: scenes.rocks_scene(250_000)),
         ("rocks 1M", lambda: scenes.rocks_scene(1_000_000)), ("soup 300k + emitters", lambda: scenes.soup_scene(300_000, emissive_every=200)),
         ("soup 2M", lambda: scenes.soup_scene(2_000_000, extent=40.0))]
ENVS = [("paths", {}), ("wavefront+refill", {"PT_PIPELINE": "wavefront", "PT_EXTEND_REFILL": "16", "PT_REFILL_CHUNK": "64"})]
for name, make in CASES:
    scene = make()
    row = []
    digest = None
    for tag, env in ENVS:
        for k in ("PT_PIPELINE", "PT_EXTEND_REFILL", "PT_REFILL_CHUNK"):
            os.environ.pop(k, None)
        os.environ.update(env)
        r = Renderer(0)
        packed = r.pack(scene)
        r.upload(packed)
        rp = packed.render_params(r.lib, 1920, 1080, spp, (0.5, 0.7, 1.0), 7)
        r.render_params(rp)  # warm-up
        _, accum, ids, st = r.render_params(rp)
        d = hashlib.sha256(accum.tobytes() + ids.tobytes()).hexdigest()[:12]
        same = digest is None or d == digest
        digest = d
        row.append(f"{tag}: {st['render_ms']:8.1f} ms {st['rays'] / st['render_ms'] / 1e3:8.1f} Mrays/s" + ("" if same else "  DIFFERENT BITS"))
        r.close()
    print(f"{name:24s} {len(scene.triangles):8d} tris  " + "   ".join(row), flush=True)
