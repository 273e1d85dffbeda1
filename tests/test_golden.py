"""Committed fixtures (tests/golden/, provenance in make_golden.py): the oracle must keep
reproducing them on the CPU, and the HIP path must reproduce them on the GPU — bit for bit."""
import importlib.util
import json
import os

import numpy as np
import pytest

from pathtracer_amd import scenes
from pathtracer_amd.renderer import Renderer
from pathtracer_amd.scene import PackedScene

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLDEN, "make_golden.py"))
make_golden = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(make_golden)
CASES = sorted(make_golden.CASES)


def load(name):
    return np.load(os.path.join(GOLDEN, f"shader_{name}.npz"))


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_golden(oracle, name):
    g = load(name)
    rgba, accum, ids, counters = make_golden.render_case(oracle, name)
    assert np.array_equal(ids, g["ids"])
    assert np.array_equal(accum.view(np.uint32), g["accum"].view(np.uint32))
    assert np.array_equal(rgba, g["rgba"])
    assert counters.tolist() == g["counters"].tolist()


def test_bvh_kat_fixture_matches_the_inline_table():
    from test_oracle_bvh import CUBE_NODES, CUBE_ORDER
    kat = json.load(open(os.path.join(GOLDEN, "bvh_kat_reference.json")))
    nodes = kat["cube"]["nodes_min_max_left_begin_count"]
    assert len(nodes) == len(CUBE_NODES) == kat["cube"]["node_count"]
    for (mn, mx, left, begin, count), row in zip(CUBE_NODES, nodes):
        assert [float(x) for x in row[0]] == list(map(float, mn)) and [float(x) for x in row[1]] == list(map(float, mx))
        assert row[2:] == [left, begin, count]
    assert [tuple(t) for t in kat["cube"]["triangles_out"]] == CUBE_ORDER
    assert [tuple(t) for t in kat["cube"]["triangles_in"]] == [tuple(int(x) for x in t) for t in scenes.kat_cube().triangles]


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_reproduces_golden(name):
    g = load(name)
    factory, kw, w, h, spp, seed = make_golden.CASES[name]
    r = Renderer(0)
    try:
        packed = r.pack(getattr(scenes, factory)(**kw))
        r.upload(packed)
        rp = packed.render_params(r.lib, w, h, spp, seed=seed)
        rgba, accum, ids, st = r.render_params(rp)
    finally:
        r.close()
    assert np.array_equal(ids, g["ids"])
    assert np.array_equal(accum.view(np.uint32), g["accum"].view(np.uint32))
    assert np.array_equal(rgba, g["rgba"])
    # paths / rays / shaded vertices do not depend on the kernels' result-preserving shortcuts; visits, tests, shaded hits and
    # texture taps are counted after them and are checked against the oracle's checker modes elsewhere
    for k in ("paths", "rays", "shaded_vertices", "nan_samples"):
        assert st[k] == int(g["counters"][make_golden.COUNTERS.index(k)]), k


def test_cfg2_full_digest_fixture_is_self_consistent():
    """The committed full-size oracle digest of BASELINE configs[1] (make_cfg2_digest.py): shape, sample count and
    counters hang together; the GPU test compares the frame itself (tests/test_gpu_parity.py)."""
    g = json.load(open(os.path.join(GOLDEN, "cfg2_full_digest.json")))
    c = g["counters"]
    assert c["paths"] == g["width"] * g["height"] * 121 and g["sample_count"] == 128
    assert c["rays"] > c["paths"] and c["hits_shaded"] == 2 * c["shaded_vertices"] and c["texture_taps"] == 0
    assert len(g["sha256_accum_then_ids"]) == 64 and len(g["sha256_rgba8"]) == 64
