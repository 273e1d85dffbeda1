"""GPU parity tests proper: the HIP path, called through the C ABI, against the CPU oracle
on the same seeded inputs.  The bar (DESIGN.md "Arithmetic contract"): bit-exact — the
float accumulators (sum of the clamped per-sample radiance + valid-sample count), the
primary-hit triangle ids, the work counters and the rgba8 image are all compared for
equality, not tolerance, because both sides evaluate the same FMA-free expressions with
the same transcendental definitions (include/pt_glsl_math.h)."""
import os

import numpy as np
import pytest

import oracle_lib
from pathtracer_amd import scenes
from pathtracer_amd.renderer import Renderer, list_devices
from pathtracer_amd.scene import PackedScene

pytestmark = pytest.mark.gpu

COUNTERS = ["paths", "rays", "node_visits", "tri_tests", "hits_shaded", "texture_taps", "shaded_vertices", "nan_samples"]


@pytest.fixture(scope="module")
def renderer():
    r = Renderer(0)
    yield r
    r.close()


def compare(renderer, oracle, scene, w, h, spp, seed, rank=0, num_ranks=1, rows=None, bg=(0.239, 0.239, 0.239)):
    packed = renderer.pack(scene)
    renderer.upload(packed)
    rp = packed.render_params(renderer.lib, w, h, spp, bg, seed, rank, num_ranks)
    rgba, accum, ids, st = renderer.render_params(rp)
    # the oracle in the reference's own form (full closest-hit shadow walks) pins the images;
    # its checker modes (any-hit shadow rays, behind-origin cull: identical images, CPU-tested) pins the GPU's work counters
    o_rgba, o_accum, o_ids, _ = oracle_lib.render(oracle, packed, rp, threads=8, rows=rows)
    sl = slice(*rows) if rows else slice(None)
    id_agree = float((ids[sl] == o_ids[sl]).mean())
    assert id_agree == 1.0, f"primary-hit id agreement {id_agree}"
    mism = accum[sl].view(np.uint32) != o_accum[sl].view(np.uint32)
    assert not mism.any(), f"{mism.any(axis=-1).sum()} pixels differ; first {np.argwhere(mism)[:3]}"
    assert (rgba[sl] == o_rgba[sl]).all()
    if rows is None:
        o_st = oracle_lib.render(oracle, packed, rp, threads=8, flags=oracle_lib.mirror_flags(st))[3]
        for k in COUNTERS:
            assert st[k] == o_st[k], (k, st[k], o_st[k])
    return st


def test_device_is_listed():
    names = list_devices()
    assert names and "gfx950" in names[0]


def test_cube_cfg1_small(renderer, oracle):
    st = compare(renderer, oracle, scenes.cube_scene(), 128, 128, 16, seed=5)
    assert st["paths"] == 128 * 128 * 16


def test_cube_cfg1_full_size(renderer, oracle):
    # BASELINE configs[0] at its full size: 512x512, 16 spp
    compare(renderer, oracle, scenes.cube_scene(), 512, 512, 16, seed=0xC0FFEE)


def test_cornell_area_light(renderer, oracle):
    compare(renderer, oracle, scenes.cornell_scene(), 160, 90, 16, seed=11)


def test_cornell_effective_spp_is_floor_sqrt_squared(renderer, oracle):
    st = compare(renderer, oracle, scenes.cornell_scene(), 64, 36, 10, seed=3)  # 10 -> 3x3 = 9 (Q1)
    assert st["paths"] == 64 * 36 * 9


def test_cornell_all_light_kinds_glass_and_metal(renderer, oracle):
    compare(renderer, oracle, scenes.cornell_scene(True, True, True, True), 160, 90, 16, seed=12)


def test_textured_pbr(renderer, oracle):
    compare(renderer, oracle, scenes.textured_scene(), 160, 90, 16, seed=13)


def test_soup_deep_bvh(renderer, oracle):
    compare(renderer, oracle, scenes.soup_scene(20000, emissive_every=50), 160, 90, 9, seed=14)


def test_atrium_cfg3_class_textured(renderer, oracle):
    # the cfg3 generator at low tessellation: relief walls, columns, arches, alpha cloth, emissive globes, sun
    compare(renderer, oracle, scenes.atrium_scene(detail=2, tex_size=64), 160, 90, 9, seed=21, bg=(0.5, 0.7, 1.0))


def test_rocks_cfg4_class_deep_bvh(renderer, oracle):
    compare(renderer, oracle, scenes.rocks_scene(60000), 160, 90, 9, seed=22, bg=(0.5, 0.7, 1.0))


@pytest.mark.parametrize("seed", range(12))
def test_random_scenes(renderer, oracle, seed):
    # fuzz: random materials (metal / rough / alpha / emissive / textured), 0-2 lights of each punctual kind, random camera
    compare(renderer, oracle, scenes.random_scene(1000 + seed), 64, 40, 9, seed=100 + seed)


@pytest.mark.parametrize("textures,punctual,alpha", [(t, p, a) for t in (False, True) for p in (False, True) for a in (False, True)])
def test_every_kernel_specialisation(renderer, oracle, textures, punctual, alpha):
    # k_paths is instantiated per scene class (opaque or not) x (textured | punctual lights only | area light only):
    # two seeded scenes of each of the eight switch settings
    for seed in (7, 8):
        s = scenes.random_scene(2000 + seed, with_textures=textures, with_punctual=punctual, with_alpha=alpha)
        compare(renderer, oracle, s, 48, 32, 9, seed=seed)


def test_ragged_resolution_and_single_sample(renderer, oracle):
    # not a multiple of the 8x8 slot tile, portrait, 1 spp
    compare(renderer, oracle, scenes.cornell_scene(with_point_light=True), 37, 53, 1, seed=15)


def test_single_triangle_and_one_pixel(renderer, oracle):
    compare(renderer, oracle, scenes.soup_scene(1, seed=9, extent=0.0, edge=3.0), 1, 1, 4, seed=16)


def test_rank_split_matches_oracle_per_rank(renderer, oracle):
    s = scenes.cornell_scene(with_directional_light=True)
    for r in range(3):
        compare(renderer, oracle, s, 96, 54, 16, seed=17, rank=r, num_ranks=3)


@pytest.mark.parametrize("env", [{"PT_PIPELINE": "wavefront"}, {"PT_PIPELINE": "wavefront", "PT_SPLIT_SHADOW": "0"},
                                 {"PT_PIPELINE": "wavefront", "PT_COMPACT_EVERY": "1", "PT_TAIL_ENTRIES": "0"},
                                 {"PT_PATHS_VERTICES": "1", "PT_PATHS_BLOCK_THREADS": "256"}, {"PT_PATHS_VERTICES": "1000"},
                                 {"PT_BAND_SLOTS": "4096"}, {"PT_BAND_SLOTS": "4096", "PT_PIPELINE": "wavefront"},
                                 {"PT_XCD_CHUNK": "5"}, {"PT_XCD_CHUNK": "100000", "PT_PATHS_BLOCKS": "24"},
                                 {"PT_LDS_STACK": "2"}, {"PT_LDS_STACK": "3", "PT_PIPELINE": "wavefront"}, {"PT_LDS_STACK": "0"},
                                 {"PT_CONCURRENT_BANDS": "3", "PT_BAND_MIN_TILES": "1"},
                                 {"PT_CONCURRENT_BANDS": "7", "PT_BAND_MIN_TILES": "1", "PT_PATHS_VERTICES": "5"},
                                 {"PT_CONCURRENT_BANDS": "1"}, {"PT_CHILD_ENTRIES": "0"},
                                 {"PT_MERGED": "0"}, {"PT_MERGED_LDS_STACK": "2"}, {"PT_MERGED_LDS_STACK": "0", "PT_PATHS_VERTICES": "3"},
                                 {"PT_CHILD_ENTRIES": "0", "PT_MERGED_LDS_STACK": "3", "PT_PATHS_BLOCK_THREADS": "256"},
                                 {"PT_PAIR_ORDER": "1"}, {"PT_PAIR_ORDER": "2", "PT_PAIR_ORDER_LEVELS": "3"}, {"PT_PAIR_ORDER": "4", "PT_MERGED": "0"},
                                 {"PT_PAIR_ORDER": "3", "PT_CHILD_ENTRIES": "0"},
                                 {"PT_CHILD_ENTRIES": "0", "PT_PIPELINE": "wavefront", "PT_LDS_STACK": "3"}])
def test_alternative_pipelines_give_the_same_bits(oracle, monkeypatch, env):
    """The wavefront pipeline (k_extend + k_shade, fused or with the split k_shadow, lazy or per-iteration
    compaction, with or without the k_paths tail) and other k_paths launch shapes: same results as the default."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    r = Renderer(0)  # the context reads its tuning knobs at creation
    try:
        compare(r, oracle, scenes.cornell_scene(True, True, True, True), 160, 90, 16, seed=31)       # LDS-resident, alpha
        compare(r, oracle, scenes.cornell_scene(), 96, 54, 9, seed=32)                                # LDS-resident, all opaque
        compare(r, oracle, scenes.textured_scene(), 160, 90, 9, seed=33)                              # global memory, alpha, 2 lights
        compare(r, oracle, scenes.soup_scene(20000, emissive_every=50), 96, 54, 9, seed=34)           # global memory, all opaque
        compare(r, oracle, scenes.soup_scene(12000, seed=9), 96, 54, 9, seed=35)                      # ... with ONE light sample per vertex (its own merged kernel)
    finally:
        r.close()


def test_leaf_too_large_for_packed_stack_entries(renderer, oracle):
    """A leaf of 128 triangles or more (here 150 copies of one triangle: the builder cannot split them) does not fit the
    packed (ref << 7 | cnt) stack entry, so the upload selects the kernels with (record, side) entries (pt_kernels_rs.hip)."""
    s = scenes.soup_scene(3000, seed=5, emissive_every=40)
    v, t = s.vertices, s.triangles
    one = v[t[7]]
    s.vertices = np.concatenate([v] + [one] * 150)
    extra = len(v) + np.arange(3 * 150, dtype=np.uint32).reshape(150, 3)
    s.triangles = np.concatenate([t, extra])
    packed = renderer.pack(s)
    assert int(packed.packed_nodes["triangle_count"].max()) >= 128
    st = compare(renderer, oracle, s, 96, 54, 9, seed=35, bg=(0.5, 0.7, 1.0))
    assert st["scene_in_lds"] == 0


def test_zero_samples_gives_black_not_a_hang(renderer, oracle):
    packed = renderer.pack(scenes.cube_scene())
    renderer.upload(packed)
    rp = packed.render_params(renderer.lib, 16, 16, 0)
    with pytest.raises(RuntimeError):
        renderer.render_params(rp)


def test_render_before_upload_fails_loudly():
    r = Renderer(0)
    try:
        packed = PackedScene(scenes.cube_scene(), r.lib)
        rp = packed.render_params(r.lib, 8, 8, 4)
        with pytest.raises(RuntimeError, match="pt_upload_scene"):
            r.render_params(rp)
    finally:
        r.close()


def test_cfg2_full_size_properties(renderer, oracle, monkeypatch):
    """BASELINE configs[1] at its full size (1920x1080, sample_count 128 -> 121 strata): size-independent
    properties on the whole frame, a band of rows against the oracle at the full sample count, and the same
    frame from the wavefront pipeline and from other band counts."""
    import hashlib
    packed = renderer.pack(scenes.cornell_scene())
    renderer.upload(packed)
    rp = packed.render_params(renderer.lib, 1920, 1080, 128, seed=0x5EED0000)
    rgba, accum, ids, st = renderer.render_params(rp)
    assert st["paths"] == 1920 * 1080 * 121
    assert (accum[..., 3] <= 121).all() and int((121 - accum[..., 3]).sum()) == st["nan_samples"]
    assert st["rays"] >= st["paths"] and st["shaded_vertices"] <= 8 * st["paths"] + 4 * st["paths"]
    assert np.isfinite(accum).all() and (accum[..., :3] >= 0).all() and (accum[..., :3] <= 10.0 * 121).all()  # clamp 10 per sample
    assert (rgba[..., 3] == 255).all()
    o_rgba, o_accum, o_ids, _ = oracle_lib.render(oracle, packed, rp, threads=16, rows=(536, 548))
    assert (ids[536:548] == o_ids[536:548]).all()
    assert (accum[536:548].view(np.uint32) == o_accum[536:548].view(np.uint32)).all()
    assert (rgba[536:548] == o_rgba[536:548]).all()
    digest = hashlib.sha256(accum.tobytes() + ids.tobytes()).hexdigest()
    # the WHOLE frame against the oracle: its digest of this frame was computed on the CPU (250.9 M paths, minutes)
    # and committed (tests/golden/make_cfg2_digest.py)
    import json
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cfg2_full_digest.json")))
    assert (gold["width"], gold["height"], gold["sample_count"], gold["seed"]) == (1920, 1080, 128, 0x5EED0000)
    assert digest == gold["sha256_accum_then_ids"]
    assert hashlib.sha256(rgba.tobytes()).hexdigest() == gold["sha256_rgba8"]
    for k in COUNTERS:
        assert st[k] == gold["counters"][k], k
    rgba2, accum2, ids2, st2 = renderer.render_params(rp)  # idempotent
    assert hashlib.sha256(accum2.tobytes() + ids2.tobytes()).hexdigest() == digest
    # the wavefront pipeline at full size: queue striding (more entries than the shade grid), compaction on
    # every iteration (the shape that once lost slots to a race in the block-level append) and the default
    # ... and the path pipeline with the frame as one band and as three (the default above: two concurrent bands)
    for env in ({"PT_PIPELINE": "wavefront", "PT_COMPACT_EVERY": "1", "PT_TAIL_ENTRIES": "0"}, {"PT_PIPELINE": "wavefront"},
                {"PT_CONCURRENT_BANDS": "1"}, {"PT_CONCURRENT_BANDS": "3"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        r2 = Renderer(0)
        try:
            r2.upload(packed)
            _, accum3, ids3, st3 = r2.render_params(rp)
        finally:
            r2.close()
        for k in env:
            monkeypatch.delenv(k)
        for k in COUNTERS:
            assert st[k] == st3[k], (env, k)
        assert hashlib.sha256(accum3.tobytes() + ids3.tobytes()).hexdigest() == digest, env


def test_4k_frame_fits_and_matches_a_band(renderer, oracle):
    # BASELINE configs[4] resolution (3840x2160) at 4 spp: the state of 8.3 M slots, a band against the oracle
    st = compare(renderer, oracle, scenes.cornell_scene(), 3840, 2160, 4, seed=19, rows=(1076, 1084))
    assert st["paths"] == 3840 * 2160 * 4


def test_cornell_1080p_band_against_oracle(renderer, oracle):
    # BASELINE configs[1] resolution; full image on the GPU at 9 spp, a 24-row band on the oracle
    compare(renderer, oracle, scenes.cornell_scene(), 1920, 1080, 9, seed=18, rows=(528, 552))
