"""CPU tests of the product's host side through the C ABI: the library loads and exports
every symbol include/pathtracer_amd.h declares, the BVH builder reproduces the reference's
known-answer vectors and equals the oracle bit for bit, packing and push
constants equal the oracle's."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from pathtracer_amd import abi, scenes
from pathtracer_amd.scene import PackedScene
from test_oracle_bvh import check_cube, check_soup

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_exports_every_declared_symbol(ptlib):
    header = open(os.path.join(ROOT, "include", "pathtracer_amd.h")).read()
    names = re.findall(r"PT_API\s+[\w\s\*]+?\b(pt_\w+)\s*\(", header)
    assert len(names) >= 13
    for n in names:
        assert hasattr(ptlib, n), n


def test_struct_sizes_match_header():
    assert C.sizeof(abi.PushConstants) == 96
    assert C.sizeof(abi.RenderParams) == 108
    assert C.sizeof(abi.Camera) == 28
    assert abi.VERTEX.itemsize == 60 and abi.BVH_NODE.itemsize == 36


def test_builder_cube_kat(ptlib):
    check_cube(ptlib, "pt_")


def test_builder_soup_kat(ptlib):
    check_soup(ptlib, "pt_")


@pytest.mark.parametrize("name", ["cube", "cornell", "textured", "soup", "coincident", "single", "planar"])
def test_builder_and_packing_equal_oracle(ptlib, oracle, name):
    if name == "cube":
        s = scenes.cube_scene()
    elif name == "cornell":
        s = scenes.cornell_scene(with_point_light=True, glass_pane=True)
    elif name == "textured":
        s = scenes.textured_scene()
    elif name == "soup":
        s = scenes.soup_scene(5000, seed=3, emissive_every=97)
    elif name == "coincident":  # all centroids identical: no split possible (Q23), one big leaf
        s = scenes.soup_scene(40, seed=5, extent=0.0, edge=0.5)
        c = s.vertices["position"].reshape(-1, 3, 3)
        c -= c.mean(axis=1, keepdims=True)
        s.vertices["position"] = np.round(c * 64).reshape(-1, 3) / 64  # exactly representable, exact zero centroids
    elif name == "single":
        s = scenes.soup_scene(1, seed=9)
    else:  # planar grid: one degenerate axis everywhere
        s = scenes.textured_scene(grid=9)
        keep = np.arange(2 * 81)
        s.triangles = s.triangles[keep]
    a = PackedScene(s, ptlib, "pt_")
    b = PackedScene(s, oracle, "orc_")
    assert a.bvh_max_depth == b.bvh_max_depth
    assert a.nodes.tobytes() == b.nodes.tobytes()
    assert a.triangles.tobytes() == b.triangles.tobytes()
    assert a.packed_nodes.tobytes() == b.packed_nodes.tobytes()
    assert a.vertex_positions.tobytes() == b.vertex_positions.tobytes()
    assert a.packed_vertex_data.tobytes() == b.packed_vertex_data.tobytes()
    assert a.emissive.tobytes() == b.emissive.tobytes()
    for (w, h) in [(512, 512), (1920, 1080), (333, 777)]:
        pa = a.render_params(ptlib, w, h, 16, seed=7)
        pb = b.render_params(oracle, w, h, 16, seed=7, prefix="orc_")
        assert bytes(pa) == bytes(pb)


def test_builder_rejects_bad_input(ptlib):
    s = scenes.cube_scene()
    tris = np.ascontiguousarray(s.triangles).copy()
    tris[0, 0] = 10_000
    nodes = np.zeros(2 * len(tris), dtype=abi.BVH_NODE)
    n, d = C.c_uint32(), C.c_uint32()
    rc = ptlib.pt_build_bvh(abi.ptr(s.vertices), C.c_uint32(len(s.vertices)), abi.ptr(tris), C.c_uint32(len(tris)),
                            C.c_uint32(16), abi.ptr(nodes), C.byref(n), C.byref(d))
    assert rc == -1
    rc = ptlib.pt_build_bvh(abi.ptr(s.vertices), C.c_uint32(len(s.vertices)), abi.ptr(tris), C.c_uint32(0),
                            C.c_uint32(16), abi.ptr(nodes), C.byref(n), C.byref(d))
    assert rc == -1


def test_reference_work_group_size(ptlib):
    x, y = C.c_uint32(), C.c_uint32()
    for (w, h, ex, ey) in [(512, 512, 32, 32), (1920, 1080, 32, 16), (1080, 1920, 16, 32)]:
        ptlib.pt_reference_work_group_size(C.c_uint32(w), C.c_uint32(h), C.byref(x), C.byref(y))
        assert (x.value, y.value) == (ex, ey)
