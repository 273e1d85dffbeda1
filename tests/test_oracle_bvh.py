"""Pins the oracle's BVH builder (oracle/pt_oracle.c: orc_build_bvh) against the two
known-answer vectors SURVEY.md §8c recorded from the reference's own src/bvh/bvh.cc."""
import numpy as np

from pathtracer_amd import abi, scenes
from pathtracer_amd.scene import PackedScene, Scene

INF = float("inf")

# idx: (min, max, left, begin, count) — SURVEY.md §8c first KAT
CUBE_NODES = [
    ((-1, -1, -1), (1, 1, 1), 2, 0, 0),
    ((INF,) * 3, (-INF,) * 3, 0, 0, 0),
    ((-1, -1, -1), (-1, 1, 1), 0, 0, 2),
    ((-1, -1, -1), (1, 1, 1), 4, 2, 0),
    ((-1, -1, -1), (1, 1, 1), 6, 2, 0),
    ((1, -1, -1), (1, 1, 1), 0, 10, 2),
    ((-1, -1, -1), (1, -1, 1), 0, 2, 2),
    ((-1, -1, -1), (1, 1, 1), 8, 4, 0),
    ((-1, -1, -1), (1, 1, 1), 10, 4, 0),
    ((-1, 1, -1), (1, 1, 1), 0, 8, 2),
    ((-1, -1, -1), (1, 1, -1), 0, 4, 2),
    ((-1, -1, 1), (1, 1, 1), 0, 6, 2),
]
CUBE_ORDER = [(0, 6, 4), (0, 2, 6), (0, 5, 1), (0, 4, 5), (0, 3, 2), (0, 1, 3),
              (4, 6, 7), (4, 7, 5), (2, 7, 6), (2, 3, 7), (1, 5, 7), (1, 7, 3)]


def check_cube(lib, prefix):
    p = PackedScene(scenes.kat_cube(), lib, prefix=prefix)
    assert len(p.nodes) == 12 and p.bvh_max_depth == 6
    for i, (mn, mx, left, begin, count) in enumerate(CUBE_NODES):
        n = p.nodes[i]
        assert tuple(n["aabb_min"]) == mn and tuple(n["aabb_max"]) == mx, i
        assert (n["left_child_index"], n["begin_index"], n["triangle_count"]) == (left, begin, count), i
    assert [tuple(t) for t in p.triangles] == CUBE_ORDER
    # packing rule render_job.cc:113
    for i, (_, _, left, begin, count) in enumerate(CUBE_NODES):
        assert p.packed_nodes[i]["left_or_begin"] == (begin if count > 0 else left)


def mt19937_soup(n_tris=20000, seed=1234):
    """std::mt19937 g(1234) + libstdc++ uniform_real_distribution<float>, drawn in the
    sequenced order SURVEY.md §8c states."""
    bg = np.random.MT19937()
    bg._legacy_seeding(seed)
    raw = bg.random_raw(12 * n_tris).astype(np.uint32)
    canon = raw.astype(np.float32) / np.float32(4294967296.0)  # generate_canonical<float,24>
    canon = np.where(canon >= np.float32(1.0), np.nextafter(np.float32(1.0), np.float32(0.0)), canon).astype(np.float32)
    canon = canon.reshape(n_tris, 12)
    a_u, b_u = np.float32(-10.0), np.float32(10.0)
    a_e, b_e = np.float32(-0.3), np.float32(0.3)
    c = (canon[:, 0:3] * (b_u - a_u) + a_u).astype(np.float32)
    e = (canon[:, 3:12] * (b_e - a_e) + a_e).astype(np.float32).reshape(n_tris, 3, 3)
    pos = (c[:, None, :] + e).astype(np.float32).reshape(-1, 3)
    v = np.zeros(3 * n_tris, dtype=abi.VERTEX)
    v["position"] = pos
    tris = np.arange(3 * n_tris, dtype=np.uint32).reshape(n_tris, 3)
    return Scene(vertices=v, triangles=tris)


def fnv1a64(data: bytes) -> int:
    h = 1469598103934665603
    for chunk_start in range(0, len(data), 1 << 16):
        for b in data[chunk_start:chunk_start + (1 << 16)]:
            h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def check_soup(lib, prefix):
    p = PackedScene(mt19937_soup(), lib, prefix=prefix)
    assert len(p.nodes) == 40000 and p.bvh_max_depth == 18
    leaves = p.nodes[p.nodes["triangle_count"] > 0]
    assert len(leaves) == 20000 and leaves["triangle_count"].max() == 1
    rec = np.zeros(len(p.nodes), dtype=[("mn", "<f4", 3), ("mx", "<f4", 3), ("lob", "<u4"), ("cnt", "<u4")])
    rec["mn"], rec["mx"] = p.nodes["aabb_min"], p.nodes["aabb_max"]
    rec["lob"], rec["cnt"] = p.packed_nodes["left_or_begin"], p.nodes["triangle_count"]
    assert "%016x" % fnv1a64(rec.tobytes()) == "80ec3dca092a6d04"
    assert "%016x" % fnv1a64(p.triangles.tobytes()) == "9f1ebe0dde198907"


def test_oracle_bvh_cube_kat(oracle):
    check_cube(oracle, "orc_")


def test_oracle_bvh_soup_kat(oracle):
    check_soup(oracle, "orc_")
