"""PT_BVH=wide4: the opt-in walk of 64-byte records of up to four children with 8-bit boxes (half the dependent node fetches per ray).
Not bit-exact by design — the boxes are supersets of the reference's and children are entered nearest first, so the closest hit can
differ from the reference's where two candidates tie exactly or a ray grazes a box the reference's slab test misses by a rounding —
and therefore validated like PT_ARITH=fast, by the tolerance BASELINE.json's north_star states: primary-hit ids agree on >= 99.99 %
of pixels, the converged image is within relMSE 1e-3 of the oracle's. The default stays the reference's tree, bit for bit."""
import numpy as np
import pytest

import oracle_lib
from pathtracer_amd import scenes
from pathtracer_amd.renderer import Renderer
from test_gpu_fast_arith import REL_MSE_TOLERANCE, block_means, rel_mse

pytestmark = pytest.mark.gpu


def render(monkeypatch, packed_or_scene, w, h, spp, bg, seed, bvh):
    if bvh:
        monkeypatch.setenv("PT_BVH", bvh)
    else:
        monkeypatch.delenv("PT_BVH", raising=False)
    r = Renderer(0)
    try:
        packed = packed_or_scene if hasattr(packed_or_scene, "desc") else r.pack(packed_or_scene)
        r.upload(packed)
        rp = packed.render_params(r.lib, w, h, spp, bg, seed=seed)
        _, accum, ids, st = r.render_params(rp)
    finally:
        r.close()
    return packed, rp, accum, ids, st


@pytest.mark.parametrize("scene_name,kwargs,bg", [
    ("rocks_scene", {"n_tris": 60000}, (0.5, 0.7, 1.0)),                     # all opaque, one directional light, deep tree
    ("soup_scene", {"n_tris": 20000, "emissive_every": 50}, (0.5, 0.7, 1.0)),  # all opaque, area light
    ("atrium_scene", {"detail": 2, "tex_size": 64}, (0.5, 0.7, 1.0)),        # textures, alpha cloth, sun + emissive globes
])
def test_wide4_is_within_the_stated_tolerance(oracle, monkeypatch, scene_name, kwargs, bg):
    w, h, spp = 160, 96, 256
    packed, rp, accum, ids, st = render(monkeypatch, getattr(scenes, scene_name)(**kwargs), w, h, spp, bg, 77, "wide4")
    assert st["bvh_wide4"] == 1 and st["merged_walk"] == 1 and st["scene_in_lds"] == 0
    _, _, e_accum, e_ids, e_st = render(monkeypatch, packed, w, h, spp, bg, 77, None)  # the default: the reference's tree
    assert e_st["bvh_wide4"] == 0
    _, o_accum, o_ids, _ = oracle_lib.render(oracle, packed, rp, threads=8)
    assert (e_ids == o_ids).all() and (e_accum.view(np.uint32) == o_accum.view(np.uint32)).all()  # the default is still exact
    assert (ids == o_ids).mean() >= 0.9999
    assert st["paths"] == e_st["paths"] and np.isfinite(accum).all()
    assert st["node_visits"] < 0.75 * e_st["node_visits"]  # the point of it: fewer dependent fetches per ray
    assert abs(st["rays"] - e_st["rays"]) < 2e-3 * e_st["rays"]
    assert rel_mse(block_means(accum, 8), block_means(o_accum, 8)) <= REL_MSE_TOLERANCE
    assert rel_mse(accum, o_accum) <= max(REL_MSE_TOLERANCE, 0.02)  # per pixel: identical paths nearly everywhere


def test_wide4_falls_back_where_it_does_not_apply(monkeypatch):
    """LDS-resident scenes and scenes with more than two light samples per vertex keep walking the reference's tree."""
    _, _, _, _, st = render(monkeypatch, scenes.cornell_scene(), 64, 40, 9, (0.239,) * 3, 3, "wide4")
    assert st["bvh_wide4"] == 0 and st["scene_in_lds"] == 1
    _, _, _, _, st = render(monkeypatch, scenes.textured_scene(grid=48), 64, 40, 9, (0.5, 0.7, 1.0), 3, "wide4")  # three light samples per vertex
    assert st["bvh_wide4"] == 0 and st["merged_walk"] == 0
