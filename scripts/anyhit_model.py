#!/usr/bin/env python3
"""Diagnostic (CPU only): how much of the any-hit shadow walks' work could a different visiting ORDER save? In an all-opaque
scene a shadow query is a yes/no question, so any order gives the same answer; but a walk that finds nothing visits every
node whose box the ray crosses whatever the order — only walks that END on an occluder can get shorter. The oracle's trip log
(checker mode = the HIP path's any-hit walks) says how the trips divide. Usage: anyhit_model.py <workload> [rows]  (SPP=<n>)"""
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
trips = (t & 0x1FFFFFFF).astype(np.int64)
closest, walk, ended = (t >> 31) == 0, (t >> 31) == 1, ((t >> 30) & 3) == 3
total = trips.sum()
print(f"{name}: {n} traversals, {total} trips")
print(f"  closest-hit            : {closest.sum():9d} traversals, {100 * trips[closest].sum() / total:5.1f} % of all trips, mean {trips[closest].mean():6.1f}")
if walk.any():
    free = walk & ~ended
    print(f"  any-hit, nothing found : {free.sum():9d} walks,      {100 * trips[free].sum() / total:5.1f} % of all trips, mean {trips[free].mean() if free.any() else 0:6.1f}  (order cannot shorten these)")
    print(f"  any-hit, occluder found: {ended.sum():9d} walks,      {100 * trips[ended].sum() / total:5.1f} % of all trips, mean {trips[ended].mean() if ended.any() else 0:6.1f}  (the most any order could save)")
else:
    print("  no any-hit walks (the scene is not all-opaque: shadow queries are closest-hit walks, main.comp:306-332)")
