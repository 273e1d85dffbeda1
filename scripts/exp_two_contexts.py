"""EXPERIMENT: would the wavefront pipeline gain from concurrent bands as the path pipeline did? Emulated without new code: two
contexts on one GPU, each rendering half of every pixel's strata (rank 0 / 1 of 2) from its own host thread, against one
context rendering all of them. Usage: python scripts/exp_two_contexts.py [scene-factory] [spp]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
from pathtracer_amd import scenes  # noqa: E402
from pathtracer_amd.renderer import Renderer  # noqa: E402

factory = sys.argv[1] if len(sys.argv) > 1 else "rocks_scene"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
kw = {"atrium_scene": {"detail": 16, "tex_size": 1024}, "rocks_scene": {"n_tris": 5_000_000}}.get(factory, {})
scene = getattr(scenes, factory)(**kw)
bg = (0.5, 0.7, 1.0)
MODES = [("paths, 2 bands", {}), ("paths, 1 band", {"PT_CONCURRENT_BANDS": "1"}),
         ("wavefront + refill", {"PT_PIPELINE": "wavefront", "PT_EXTEND_REFILL": "16", "PT_REFILL_CHUNK": "64"}),
         ("wavefront", {"PT_PIPELINE": "wavefront"})]
for name, env in MODES:
    for k in ("PT_CONCURRENT_BANDS", "PT_PIPELINE", "PT_EXTEND_REFILL", "PT_REFILL_CHUNK"):
        os.environ.pop(k, None)
    os.environ.update(env)
    for n_ctx in (1, 2):
        rs = [Renderer(0) for _ in range(n_ctx)]
        packed = rs[0].pack(scene)
        rps = []
        for i, r in enumerate(rs):
            r.upload(packed)
            rps.append(packed.render_params(r.lib, 1920, 1080, spp, bg, 7, i, n_ctx))
        rays = [0] * n_ctx

        def work(i):
            st = rs[i].render_params(rps[i], want_accum=False, want_ids=False)[3]
            rays[i] = st["rays"]

        best = 1e30
        for rep in range(3):
            ts = [threading.Thread(target=work, args=(i,)) for i in range(n_ctx)]
            t0 = time.perf_counter()
            for t in ts:
                t.start()
            for t in ts:
                t.join()
            dt = time.perf_counter() - t0
            if rep:
                best = min(best, dt)
        print(f"{factory} spp {spp}  {name:22s} {n_ctx} context(s): {1e3 * best:9.1f} ms  {sum(rays) / best / 1e6:8.1f} Mrays/s", flush=True)
        for r in rs:
            r.close()
