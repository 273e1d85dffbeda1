#!/usr/bin/env python3
"""Diagnostic (CPU only): lane occupancy of a wave's traversal loop as a function of how rays are dealt to lanes. The oracle
logs the trips (inner visits + leaves entered) of every traversal of a few rows of a workload; rays are then dealt at
random to waves of 64 lanes, one ray per lane (what k_paths does) or R rays per lane walked one after the other, and the
wave's trip count is the largest lane total. Usage: trip_model.py <workload> [rows]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, oracle_lib
from pathtracer_amd import scenes, lib
from pathtracer_amd.scene import PackedScene

name = sys.argv[1]; rows = int(sys.argv[2]) if len(sys.argv) > 2 else 2
factory, kw, w, h, spp, _ = bench.WORKLOADS[name]
spp = int(os.environ.get("SPP", 16))
L = lib.load(); O = oracle_lib.load()
O.orc_set_trip_log.argtypes = [C.c_void_p, C.c_size_t]
O.orc_trip_log_length.restype = C.c_size_t
packed = PackedScene(getattr(scenes, factory)(**kw), L)
rp = packed.render_params(L, w, h, spp, bench.BG.get(name, (0.239,) * 3), bench.SEED)
log = np.zeros(200_000_000 // 4, dtype=np.uint32)
O.orc_set_trip_log(log.ctypes.data_as(C.c_void_p), log.size)
y = h // 2
oracle_lib.render(O, packed, rp, threads=1, rows=(y, y + rows), want_ids=False, flags=oracle_lib.GPU_MIRROR)
n = O.orc_trip_log_length()
O.orc_set_trip_log(None, 0)
t = log[:n]
closest = (t[(t >> 31) == 0] & 0x1FFFFFFF).astype(np.int64)
anyhit = (t[(t >> 31) == 1] & 0x1FFFFFFF).astype(np.int64)
print(f"{name}: {closest.size} closest-hit traversals (mean {closest.mean():.1f} trips, max {closest.max()}), "
      f"{anyhit.size} any-hit walks (mean {anyhit.mean() if anyhit.size else 0:.1f})")
rng = np.random.RandomState(1)
for label, trips in (("closest-hit", closest), ("any-hit", anyhit)):
    if trips.size < 64 * 64:
        continue
    base = None
    for R in (1, 2, 3, 4, 8):
        m = trips.size // (64 * R) * (64 * R)
        x = rng.permutation(trips)[:m].reshape(-1, 64, R).sum(axis=2)  # lane totals
        wave_trips = x.max(axis=1).sum()
        occ = x.sum() / (64.0 * wave_trips)
        per_ray = wave_trips / (m / 64.0)
        if base is None:
            base = per_ray
        print(f"  {label:11s} {R} ray(s) per lane: lane occupancy {100 * occ:5.1f} %, wave trips per 64 rays {per_ray:7.2f} ({100 * per_ray / base:5.1f} % of one ray per lane)")
