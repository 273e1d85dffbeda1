"""PT_ARITH=fast: the opt-in build of the path kernel whose shading half runs on the hardware's approximate divide /
sqrt / rsq / exp2 / log2 / sin / cos with FMA contraction. Not bit-exact by design: validated by the tolerance
BASELINE.json's north_star states — primary-hit ids agree on >= 99.99 % of pixels (here: all, because ray generation
and the intersection tests stay contraction-free) and the converged image is within relMSE 1e-3 of the oracle's."""
import numpy as np
import pytest

import oracle_lib
from pathtracer_amd import scenes
from pathtracer_amd.renderer import Renderer

pytestmark = pytest.mark.gpu

REL_MSE_TOLERANCE = 1e-3  # north_star: converged-image relMSE <= 1e-3
EPS = 1e-2                # relMSE = mean((a-b)^2 / (b^2 + eps)) on display-domain floats


def display(accum):
    rgb = accum[..., :3].astype(np.float64) / np.maximum(accum[..., 3:4].astype(np.float64), 1.0)
    return np.minimum(np.clip(rgb, 0.0, None) ** (1.0 / 2.2), 1.0)


def rel_mse(a, b):
    a, b = display(a), display(b)
    return float(np.mean((a - b) ** 2 / (b * b + EPS)))


@pytest.mark.parametrize("scene_name,kwargs,w,h,spp,bg", [
    ("cornell_scene", {}, 320, 180, 256, (0.239, 0.239, 0.239)),
    ("cornell_scene", {"with_point_light": True, "with_directional_light": True}, 160, 90, 256, (0.239, 0.239, 0.239)),
    ("textured_scene", {}, 160, 90, 256, (0.5, 0.7, 1.0)),
])
def test_fast_arithmetic_is_within_the_stated_tolerance(oracle, monkeypatch, scene_name, kwargs, w, h, spp, bg):
    monkeypatch.setenv("PT_ARITH", "fast")
    r = Renderer(0)
    try:
        packed = r.pack(getattr(scenes, scene_name)(**kwargs))
        r.upload(packed)
        rp = packed.render_params(r.lib, w, h, spp, bg, seed=123)
        _, accum, ids, st = r.render_params(rp)
    finally:
        r.close()
    _, o_accum, o_ids, o_st = oracle_lib.render(oracle, packed, rp, threads=8)
    assert (ids == o_ids).mean() >= 0.9999          # north_star; in fact identical:
    assert (ids == o_ids).all()
    assert st["paths"] == o_st["paths"]
    assert st["scene_in_lds"] == 1 and st["fast_arith"] == 1
    assert np.isfinite(accum).all()
    e = rel_mse(accum, o_accum)
    assert e <= REL_MSE_TOLERANCE, e
    # and it really is the other build: not the exact bits
    assert (accum.view(np.uint32) != o_accum.view(np.uint32)).any()


def block_means(accum, n=16):
    """Accumulator frame [H, W, 4] summed over n x n pixel blocks (ragged edges dropped): the same estimator at 1 / n^2 of the variance."""
    h, w = accum.shape[0] // n * n, accum.shape[1] // n * n
    return accum[:h, :w].reshape(h // n, n, w // n, n, 4).sum(axis=(1, 3))


def test_fast_arithmetic_leaves_scenes_read_from_global_memory_exact(monkeypatch):
    """PT_ARITH=fast is for scenes staged in LDS, which VALU issue bounds. A scene read from global memory waits on memory: the fast
    compilation never gained there (round 2) and its register allocation made it slower in round 3 (atrium: 1256 against 2052
    Mrays/s), so pt_render runs the exact kernels on such a scene whatever PT_ARITH says. The headline scene (253 k triangles, 44
    1024^2 textures; a 480 x 270 frame to keep the test short) must therefore come out bit-identical, and pt_stats must say so."""
    import bench
    factory, kw, _, _, _, _ = bench.WORKLOADS["atrium1080"]
    packed, frames = None, {}
    for arith in ("exact", "fast"):
        monkeypatch.setenv("PT_ARITH", arith)
        r = Renderer(0)
        try:
            if packed is None:
                packed = r.pack(getattr(scenes, factory)(**kw))
            r.upload(packed)
            rp = packed.render_params(r.lib, 480, 270, 64, bench.BG["atrium1080"], seed=bench.SEED)
            _, accum, ids, st = r.render_params(rp)
            frames[arith] = (accum, ids, st)
        finally:
            r.close()
    (a, ia, sa), (b, ib, sb) = frames["exact"], frames["fast"]
    assert sa["scene_in_lds"] == 0 and sa["fast_arith"] == 0 and sb["fast_arith"] == 0
    assert (ia == ib).all() and (a.view(np.uint32) == b.view(np.uint32)).all()
    assert {k: v for k, v in sa.items() if not k.endswith("_ms")} == {k: v for k, v in sb.items() if not k.endswith("_ms")}
