"""BASELINE.json configs[2..4] at their FULL scene sizes and resolutions on the GPU: the 253 k-triangle
textured atrium (44 1024^2 textures, BVH depth 24), the 4.7 M-triangle rocks (depth 33, the 5-wave k_paths variant) and the
4K / 4096-spp shape (64 x 64 strata per pixel). Sample counts are reduced where the config's own would take minutes on
the oracle; everything else — geometry, texel pool, resolution, tree depth, kernel variants — is the config's.
Checks: size-independent properties of the whole frame, a band of rows against the oracle bit for bit, and the
wavefront pipeline's frame equal to the path pipeline's."""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle_lib
from pathtracer_amd import scenes
from pathtracer_amd.renderer import Renderer

pytestmark = pytest.mark.gpu

SKY = (0.5, 0.7, 1.0)
COUNTERS = ["paths", "rays", "node_visits", "tri_tests", "hits_shaded", "texture_taps", "shaded_vertices", "nan_samples"]


def properties(accum, rgba, st, w, h, eff):
    assert st["paths"] == w * h * eff
    assert np.isfinite(accum).all() and (accum[..., :3] >= 0).all() and (accum[..., :3] <= 10.0 * eff).all()  # clamp 10 per sample
    assert (accum[..., 3] <= eff).all() and int((eff - accum[..., 3]).sum()) == st["nan_samples"]
    assert st["rays"] == st["rays_closest"] + st["rays_shadow_walked"] + st["rays_shadow_shortcut"]
    assert st["rays_closest"] >= st["paths"] and st["shaded_vertices"] <= 12 * st["paths"]
    assert (rgba[..., 3] == 255).all()


def band_equals_oracle(oracle, packed, rp, accum, ids, rgba, rows):
    o_rgba, o_accum, o_ids, _ = oracle_lib.render(oracle, packed, rp, threads=16, rows=rows)
    sl = slice(*rows)
    assert (ids[sl] == o_ids[sl]).all()
    assert (accum[sl].view(np.uint32) == o_accum[sl].view(np.uint32)).all()
    assert (rgba[sl] == o_rgba[sl]).all()


def other_pipeline_digest(packed, rp, monkeypatch, env):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    r2 = Renderer(0)
    try:
        r2.upload(packed)
        _, accum, ids, st = r2.render_params(rp)
    finally:
        r2.close()
        for k in env:
            monkeypatch.delenv(k)
    return hashlib.sha256(accum.tobytes() + ids.tobytes()).hexdigest(), st


@pytest.mark.parametrize("factory,kwargs,spp,depth", [
    ("atrium_scene", {"detail": 16, "tex_size": 1024}, 9, 24),   # cfg3's scene as bench.py renders it
    ("rocks_scene", {"n_tris": 5_000_000}, 4, 33),               # cfg4's scene
])
def test_full_size_scene_at_1080p(oracle, monkeypatch, factory, kwargs, spp, depth):
    r = Renderer(0)
    try:
        packed = r.pack(getattr(scenes, factory)(**kwargs))
        assert packed.bvh_max_depth == depth
        r.upload(packed)
        rp = packed.render_params(r.lib, 1920, 1080, spp, SKY, seed=0x5EED0000)
        rgba, accum, ids, st = r.render_params(rp)
        properties(accum, rgba, st, 1920, 1080, spp)
        assert st["scene_in_lds"] == 0
        if factory == "atrium_scene":
            assert st["texture_taps"] > st["paths"]  # textured PBR: taps on (nearly) every vertex
        band_equals_oracle(oracle, packed, rp, accum, ids, rgba, (536, 544))
        digest = hashlib.sha256(accum.tobytes() + ids.tobytes()).hexdigest()
        rgba2, accum2, ids2, _ = r.render_params(rp)  # idempotent
        assert hashlib.sha256(accum2.tobytes() + ids2.tobytes()).hexdigest() == digest
    finally:
        r.close()
    d2, st2 = other_pipeline_digest(packed, rp, monkeypatch, {"PT_PIPELINE": "wavefront"})
    assert d2 == digest
    for k in COUNTERS:
        assert st[k] == st2[k], k


@pytest.mark.parametrize("name", ["atrium1080", "rocks1080"])
def test_cfg3_cfg4_at_their_own_sample_counts_equal_the_oracle_band_digest(monkeypatch, name):
    """cfg3 / cfg4 exactly as bench.py renders them — full scene, 1080p, 256 / 484 strata, the bench seed — against the committed
    oracle digest of a 32-row band (tests/golden/make_band_digest.py), with the frame rendered as one, two (the default) and three
    concurrent bands: hundreds of regenerations per slot, the long drain and the bands' interleaved launches, which 9 / 4 spp
    do not exercise."""
    import bench
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"{name}_full_digest.json")))
    factory, kw, w, h, spp, _ = bench.WORKLOADS[name]
    assert (gold["width"], gold["height"], gold["sample_count"], gold["seed"]) == (w, h, spp, bench.SEED)
    y0, y1 = gold["rows"]
    assert y1 - y0 >= 32
    eff = int(np.sqrt(np.float32(spp))) ** 2
    packed = None
    for bands in ("2", "1", "3"):
        monkeypatch.setenv("PT_CONCURRENT_BANDS", bands)
        r = Renderer(0)  # the context reads the knob at creation
        try:
            if packed is None:
                packed = r.pack(getattr(scenes, factory)(**kw))
            r.upload(packed)
            rp = packed.render_params(r.lib, w, h, spp, bench.BG[name], seed=bench.SEED)
            rgba, accum, ids, st = r.render_params(rp)
        finally:
            r.close()
        if bands == "2":
            properties(accum, rgba, st, w, h, eff)
        assert hashlib.sha256(np.ascontiguousarray(accum[y0:y1]).tobytes() + np.ascontiguousarray(ids[y0:y1]).tobytes()).hexdigest() == \
            gold["sha256_accum_then_ids"], bands
        assert hashlib.sha256(np.ascontiguousarray(rgba[y0:y1]).tobytes()).hexdigest() == gold["sha256_rgba8"], bands


def test_cfg5_shape_4k_4096_strata(oracle):
    """3840x2160 with sample_count 4096 (64 x 64 strata per pixel, ~1500 k_paths launches, 34 G paths): the whole frame
    on the GPU, two rows on the oracle (31 M paths)."""
    r = Renderer(0)
    try:
        packed = r.pack(scenes.cornell_scene())
        r.upload(packed)
        rp = packed.render_params(r.lib, 3840, 2160, 4096, seed=0x5EED0000)
        rgba, accum, ids, st = r.render_params(rp)
        properties(accum, rgba, st, 3840, 2160, 4096)
        assert st["scene_in_lds"] == 1
        band_equals_oracle(oracle, packed, rp, accum, ids, rgba, (1079, 1081))
    finally:
        r.close()
