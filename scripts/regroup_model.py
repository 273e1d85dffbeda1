#!/usr/bin/env python3
"""Diagnostic (CPU only): what would moving whole RAYS between the waves of a workgroup buy the closest-hit traversal of
k_paths? The oracle logs, for every closest-hit traversal of a few tile rows, its trips, its pixel and how many light steps
the path had done. k_paths is then replayed as it schedules: a wave is an 8x8 tile, a lane owns a pixel and walks that
pixel's traversals in order, the lanes of a wave traverse in lock step (phase k pairs the k-th traversal of every lane;
a wave's phase lasts as long as its longest lane). Against that:
  * 'by depth': the 256 rays of a 4-tile workgroup are re-dealt to its 4 waves at every phase, sorted by the number of
    light steps done (camera rays with camera rays: something a kernel knows before it traverses);
  * 'by trips': the same, sorted by the trips the traversal will turn out to need — an oracle no kernel has, the bound of
    every regrouping scheme;
  * 'by last trips': sorted by the trips of the lane's PREVIOUS traversal (known, and correlated if a path stays in a region).
Usage: regroup_model.py <workload> [tile rows]   (SPP=<n> overrides the samples per pixel, default 16)"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, oracle_lib
from pathtracer_amd import scenes, lib
from pathtracer_amd.scene import PackedScene

name = sys.argv[1]; tile_rows = int(sys.argv[2]) if len(sys.argv) > 2 else 1
factory, kw, w, h, spp, _ = bench.WORKLOADS[name]
spp = int(os.environ.get("SPP", 16))
L = lib.load(); O = oracle_lib.load()
O.orc_set_trip_log_tagged.argtypes = [C.c_void_p, C.c_size_t]
O.orc_trip_log_length.restype = C.c_size_t
packed = PackedScene(getattr(scenes, factory)(**kw), L)
rp = packed.render_params(L, w, h, spp, bench.BG.get(name, (0.239,) * 3), bench.SEED)
log = np.zeros(400_000_000 // 4, dtype=np.uint32)
O.orc_set_trip_log_tagged(log.ctypes.data_as(C.c_void_p), log.size)
y0 = (h // 2) // 8 * 8
oracle_lib.render(O, packed, rp, threads=1, rows=(y0, y0 + 8 * tile_rows), want_ids=False, flags=oracle_lib.GPU_MIRROR)
n = O.orc_trip_log_length()
O.orc_set_trip_log(None, 0)
t = log[:n].reshape(-1, 2)
t = t[(t[:, 0] >> 31) == 0]  # closest-hit traversals only
trips, pixel, depth = (t[:, 0] & 0x1FFFFFFF).astype(np.int64), (t[:, 1] >> 4).astype(np.int64), (t[:, 1] & 15).astype(np.int64)
print(f"{name}: {trips.size} closest-hit traversals of {8 * tile_rows} rows at {spp} spp, mean {trips.mean():.1f} trips, max {trips.max()}")

# per pixel: its traversals in order -> (pixels, phases) arrays padded with -1 (the pixel has finished all its samples)
px, py = pixel % w, pixel // w - y0
tiles_x = w // 8
lane = (py % 8) * 8 + px % 8
tile = (py // 8) * tiles_x + px // 8
keep = px < tiles_x * 8
trips, depth, lane, tile = trips[keep], depth[keep], lane[keep], tile[keep]
slot = tile * 64 + lane
order = np.argsort(slot, kind="stable")
slot_s, trips_s, depth_s = slot[order], trips[order], depth[order]
counts = np.bincount(slot_s, minlength=tile.max() * 64 + 64)
phases = counts.max()
start = np.concatenate([[0], np.cumsum(counts)[:-1]])
k = np.arange(slot_s.size) - start[slot_s]
T = np.full((counts.size, phases), -1, dtype=np.int64)
D = np.zeros((counts.size, phases), dtype=np.int64)
T[slot_s, k], D[slot_s, k] = trips_s, depth_s
n_wg = counts.size // 256
T, D = T[:n_wg * 256].reshape(n_wg, 256, phases), D[:n_wg * 256].reshape(n_wg, 256, phases)
live = T >= 0
work = np.where(live, T, 0)
total = work.sum()


def wave_trips(key):
    """Sum over workgroups, phases and the 4 waves of the longest lane, with the workgroup's rays dealt to waves in `key` order
    (None: lanes stay where they are). Finished lanes sort last."""
    if key is None:
        x = work.reshape(n_wg, 4, 64, phases)
    else:
        idx = np.argsort(np.where(live, key, 1 << 40), axis=1, kind="stable")
        x = np.take_along_axis(work, idx, axis=1).reshape(n_wg, 4, 64, phases)
    return x.max(axis=2).sum()


prev = np.concatenate([np.zeros((n_wg, 256, 1), dtype=np.int64), work[:, :, :-1]], axis=2)
base = wave_trips(None)
for label, key in (("as scheduled (lane = pixel)", None), ("re-dealt by depth", D), ("re-dealt by last trips", prev),
                   ("re-dealt by depth, then last trips", D * 1000 + np.minimum(prev, 999)), ("re-dealt by trips (oracle)", work)):
    wt = wave_trips(key)
    print(f"  {label:36s}: {total / (64.0 * wt) * 64:5.1f} of 64 lanes busy per trip, wave trips {100.0 * wt / base:5.1f} % of as scheduled")
