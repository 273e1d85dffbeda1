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


def _build(lib, fn, verts, tris, *extra):
    t = np.ascontiguousarray(tris).copy()
    nodes = np.zeros(2 * len(t), dtype=abi.BVH_NODE)
    n, d = C.c_uint32(), C.c_uint32()
    rc = getattr(lib, fn)(abi.ptr(verts), C.c_uint32(len(verts)), abi.ptr(t), C.c_uint32(len(t)), C.c_uint32(16),
                          abi.ptr(nodes), C.byref(n), C.byref(d), *extra)
    assert rc == 0
    return nodes.tobytes(), t.tobytes(), n.value, d.value


def _soup(n, seed, quant=None, signed_zero_axis=None):
    rng = np.random.RandomState(seed)
    p = (rng.uniform(-10, 10, (n, 1, 3)) + rng.uniform(-0.3, 0.3, (n, 3, 3))).astype(np.float32)
    if quant:  # many equal centroids and box faces: ties in every min/max and in the partition
        p = (np.round(p * quant) / quant).astype(np.float32)
    if signed_zero_axis is not None:  # a degenerate axis (Q23) made of +0 and -0: min/max must keep the leftmost
        p[..., signed_zero_axis] = np.float32(0.0) * np.sign(rng.uniform(-1, 1, (n, 3))).astype(np.float32)
    v = np.zeros(3 * n, dtype=abi.VERTEX)
    v["position"] = p.reshape(-1, 3)
    return v, np.arange(3 * n, dtype=np.uint32).reshape(n, 3)


@pytest.mark.parametrize("case", ["soup", "ties", "signed_zero_plane", "just_above_threshold"])
def test_threaded_builder_is_bit_identical(ptlib, oracle, case):
    """nodes, numbering, triangle permutation and depth from 2, 3 and 8 threads equal the
    sequential builder's and the oracle's recursive restatement."""
    v, t = {"soup": lambda: _soup(60_000, 1), "ties": lambda: _soup(80_000, 2, quant=2.0),
            "signed_zero_plane": lambda: _soup(50_000, 3, signed_zero_axis=1), "just_above_threshold": lambda: _soup(4100, 4)}[case]()
    ref = _build(ptlib, "pt_build_bvh_mt", v, t, C.c_uint32(1))
    assert ref == _build(oracle, "orc_build_bvh", v, t)
    for threads in (2, 3, 8):
        assert _build(ptlib, "pt_build_bvh_mt", v, t, C.c_uint32(threads)) == ref
    if case == "ties":  # 80 000 triangles: pt_build_bvh itself takes the threaded route
        assert _build(ptlib, "pt_build_bvh", v, t) == ref


def test_threaded_builder_soup_kat(ptlib, monkeypatch):
    """The reference's own 20 000-triangle known answer through the threaded route."""
    class Threaded:  # PackedScene calls <prefix>build_bvh: route it through pt_build_bvh_mt with 4 threads
        def __getattr__(self, name):
            if name == "pt_build_bvh":
                return lambda *a: ptlib.pt_build_bvh_mt(*a, C.c_uint32(4))
            return getattr(ptlib, name)
    check_soup(Threaded(), "pt_")


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


def test_pt_create_multi_without_rccl_fails_cleanly():
    """on a machine without librccl, pt_create_multi must return NULL with a message, not crash (the dlopen failure
    path called dlerror() twice and built a std::string from NULL). PT_RCCL_LIB points the loader at a file that does not exist;
    a fresh process, because the library caches the loaded API. Also: ids are normalised before the duplicate check."""
    import subprocess
    import sys
    code = """
import ctypes as C, sys
sys.path.insert(0, %r)
from pathtracer_amd import lib
L = lib.load()
L.pt_create_multi.restype = C.c_void_p
L.pt_last_error.restype = C.c_char_p
L.pt_last_error.argtypes = [C.c_void_p]
ids = (C.c_int * 2)(-1, 0)
assert L.pt_create_multi(ids, 2) is None
assert b"listed twice" in L.pt_last_error(None), L.pt_last_error(None)
ids = (C.c_int * 2)(0, 1)
assert L.pt_create_multi(ids, 2) is None
msg = L.pt_last_error(None)
assert b"librccl.so not found" in msg and b"no_such_rccl" in msg, msg
print("ok")
""" % ROOT
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PT_RCCL_LIB="/nonexistent/no_such_rccl.so", PT_NO_TORCH_PRELOAD="1"),
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "ok" in p.stdout, p.stderr[-2000:]


@pytest.mark.parametrize("make", [lambda: scenes.rocks_scene(n_tris=6000), lambda: scenes.soup_scene(3000, seed=4), lambda: scenes.cornell_scene(),
                                  lambda: scenes.atrium_scene(detail=2, tex_size=16), lambda: scenes.cube_scene()])
def test_wide4_records_are_a_conservative_cover_of_the_reference_tree(ptlib, make):
    """pt_build_wide4 (the records PT_BVH=wide4 walks): decoded with the arithmetic of the header, every child box CONTAINS the
    reference's box of the node it stands for; every leaf of the binary tree is reached exactly once with its triangle range intact;
    a record holds 2 to 4 children; the stated stack need bounds a depth-first walk."""
    packed = PackedScene(make(), ptlib)
    N = packed.packed_nodes
    count = C.c_uint32(0)
    need = C.c_uint32(0)
    assert ptlib.pt_build_wide4(N.ctypes.data_as(C.c_void_p), C.c_uint32(len(N)), None, C.c_uint32(0), C.byref(count), C.byref(need)) == 0
    rec = np.zeros((count.value, 16), dtype=np.uint32)
    assert ptlib.pt_build_wide4(N.ctypes.data_as(C.c_void_p), C.c_uint32(len(N)), rec.ctypes.data_as(C.c_void_p), C.c_uint32(len(rec)), C.byref(count),
                                C.byref(need)) == 0
    if N[0]["triangle_count"] > 0:
        assert count.value == 0
        return
    # binary leaves by first triangle -> (count, box)
    leaves = {int(n["left_or_begin"]): n for n in N if n["triangle_count"] > 0 and np.isfinite(n["aabb_min"]).all()}
    # walk the binary tree to know which leaves are reachable from the root
    reach, todo = set(), [0]
    while todo:
        n = N[todo.pop()]
        if n["triangle_count"] > 0:
            reach.add(int(n["left_or_begin"]))
        else:
            todo += [int(n["left_or_begin"]), int(n["left_or_begin"]) + 1]
    seen = []
    origin = rec[:, 0:3].view(np.float32)

    def boxes(w):
        ex = int(rec[w, 3])
        cell = np.array([np.ldexp(np.float32(1.0), ((ex >> (8 * a)) & 255) - 127) for a in range(3)], dtype=np.float32)
        lo = np.array([[(int(rec[w, 8 + a]) >> (8 * c)) & 255 for a in range(3)] for c in range(4)], dtype=np.float32)
        hi = np.array([[(int(rec[w, 11 + a]) >> (8 * c)) & 255 for a in range(3)] for c in range(4)], dtype=np.float32)
        return origin[w] + lo * cell, origin[w] + hi * cell  # float32: the decoder's arithmetic

    def union_of_subtree(w, c, lo, hi):
        """Check child c of record w against everything below it; returns the deepest stack a walk entering it needs."""
        e = int(rec[w, 4 + c])
        if e & 127:
            first, cnt = e >> 7, e & 127
            leaf = leaves[first]
            assert int(leaf["triangle_count"]) == cnt
            assert (lo[c] <= leaf["aabb_min"]).all() and (hi[c] >= leaf["aabb_max"]).all(), (w, c)
            seen.append(first)
            return 0, leaf["aabb_min"], leaf["aabb_max"]
        return check(e >> 7, lo[c], hi[c])

    def check(w, plo=None, phi=None):
        lo, hi = boxes(w)
        kids = [c for c in range(4) if int(rec[w, 4 + c]) != 0xFFFFFFFF]
        assert 2 <= len(kids) <= 4 and kids == list(range(len(kids)))
        deepest, ulo, uhi = 0, np.full(3, np.inf, dtype=np.float32), np.full(3, -np.inf, dtype=np.float32)
        for c in kids:
            d, blo, bhi = union_of_subtree(w, c, lo, hi)
            deepest = max(deepest, d)
            ulo, uhi = np.minimum(ulo, blo), np.maximum(uhi, bhi)
        if plo is not None:  # the parent's byte box of this record covers everything in it
            assert (plo <= ulo).all() and (phi >= uhi).all(), w
        return len(kids) - 1 + deepest, ulo, uhi

    import sys
    sys.setrecursionlimit(10000)
    deepest, _, _ = check(0)
    assert sorted(seen) == sorted(reach)  # every reachable leaf exactly once
    assert deepest <= need.value
