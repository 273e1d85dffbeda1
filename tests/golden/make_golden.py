#!/usr/bin/env python3
"""Generates the fixtures in this directory. Run from the repo root: python tests/golden/make_golden.py

Provenance — read before trusting a file here:
  bvh_kat_reference.json   REFERENCE OUTPUT. The two known-answer vectors SURVEY.md §8c recorded from the
                           reference's own src/bvh/bvh.cc (12-triangle cube: node table + triangle order;
                           20 000-triangle mt19937(1234) soup: FNV-1a-64 hashes). Transcribed, not regenerated:
                           the reference cannot be rebuilt in this image (glm absent).
  shader_*.npz             ORACLE OUTPUT, not reference output: main.comp cannot be run anywhere we have
                           access to ("parity unpinned"). They freeze oracle/pt_oracle.c so that
                           a change to the restatement (or to the kernels) cannot go unnoticed: float
                           accumulators, primary-hit ids, rgba8 and work counters of small seeded renders.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CASES = {
    # name: (scene factory, kwargs, width, height, sample_count, seed)
    "cube": ("cube_scene", {}, 64, 64, 16, 101),
    "cornell": ("cornell_scene", {}, 64, 36, 9, 102),
    "cornell_all": ("cornell_scene", dict(with_point_light=True, with_directional_light=True, glass_pane=True, metal_box=True), 64, 36, 9, 103),
    "textured": ("textured_scene", {}, 64, 36, 9, 104),
    "soup": ("soup_scene", dict(n_tris=4000, seed=6, emissive_every=50), 48, 27, 4, 105),
}
COUNTERS = ["paths", "rays", "node_visits", "tri_tests", "hits_shaded", "texture_taps", "shaded_vertices", "nan_samples"]


def render_case(oracle, name):
    import oracle_lib
    from pathtracer_amd import scenes
    from pathtracer_amd.scene import PackedScene
    factory, kw, w, h, spp, seed = CASES[name]
    packed = PackedScene(getattr(scenes, factory)(**kw), oracle, prefix="orc_")
    rp = packed.render_params(oracle, w, h, spp, seed=seed, prefix="orc_")
    rgba, accum, ids, st = oracle_lib.render(oracle, packed, rp, threads=8)
    return rgba, accum, ids, np.array([st[k] for k in COUNTERS], dtype=np.uint64)


def main():
    import oracle_lib
    oracle = oracle_lib.load()
    for name in CASES:
        rgba, accum, ids, counters = render_case(oracle, name)
        np.savez_compressed(os.path.join(HERE, f"shader_{name}.npz"), rgba=rgba, accum=accum, ids=ids, counters=counters)
        print(name, rgba.shape, counters.tolist())
    inf = "inf"
    kat = {
        "provenance": "SURVEY.md §8c: produced by the reference's own src/bvh/bvh.cc (g++ 11.4 and ROCm clang 22, -O2 -ffp-contract=off) during the survey",
        "cube": {
            "vertices": "v_i = (i&1 ? +1 : -1, i&2 ? +1 : -1, i&4 ? +1 : -1), i = 0..7",
            "triangles_in": [[0, 1, 3], [0, 3, 2], [4, 6, 7], [4, 7, 5], [0, 4, 5], [0, 5, 1], [2, 3, 7], [2, 7, 6], [0, 2, 6], [0, 6, 4], [1, 5, 7], [1, 7, 3]],
            "bin_count": 16, "node_count": 12, "depth": 6,
            "nodes_min_max_left_begin_count": [
                [[-1, -1, -1], [1, 1, 1], 2, 0, 0], [[inf, inf, inf], ["-inf", "-inf", "-inf"], 0, 0, 0],
                [[-1, -1, -1], [-1, 1, 1], 0, 0, 2], [[-1, -1, -1], [1, 1, 1], 4, 2, 0], [[-1, -1, -1], [1, 1, 1], 6, 2, 0],
                [[1, -1, -1], [1, 1, 1], 0, 10, 2], [[-1, -1, -1], [1, -1, 1], 0, 2, 2], [[-1, -1, -1], [1, 1, 1], 8, 4, 0],
                [[-1, -1, -1], [1, 1, 1], 10, 4, 0], [[-1, 1, -1], [1, 1, 1], 0, 8, 2], [[-1, -1, -1], [1, 1, -1], 0, 4, 2],
                [[-1, -1, 1], [1, 1, 1], 0, 6, 2]],
            "triangles_out": [[0, 6, 4], [0, 2, 6], [0, 5, 1], [0, 4, 5], [0, 3, 2], [0, 1, 3], [4, 6, 7], [4, 7, 5], [2, 7, 6], [2, 3, 7], [1, 5, 7], [1, 7, 3]],
        },
        "soup": {
            "generator": "std::mt19937 g(1234); U(-10,10) centre x,y,z then E(-0.3,0.3) offsets per vertex, sequenced; indices (3i,3i+1,3i+2)",
            "triangle_count": 20000, "node_count": 40000, "depth": 18, "leaves": 20000, "max_leaf": 1,
            "fnv1a64_nodes_min_max_leftorbegin_count": "80ec3dca092a6d04",
            "fnv1a64_triangles": "9f1ebe0dde198907",
        },
    }
    with open(os.path.join(HERE, "bvh_kat_reference.json"), "w") as f:
        json.dump(kat, f, indent=1)


if __name__ == "__main__":
    main()
