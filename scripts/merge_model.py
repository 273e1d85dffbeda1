#!/usr/bin/env python3
"""Diagnostic (CPU only): what would walking a vertex's LAST shadow ray and the next vertex's closest-hit ray in ONE traversal loop
buy k_paths? Today the lanes of a wave walk in lock step per phase — closest hit of vertex k, then the shadow walk(s) of vertex k —
and a phase lasts as long as its longest lane: sum over phases of max over lanes. Merged, a lane that has finished its shadow walk
goes straight on with its next ray, and the wave pays max over lanes of (shadow + closest). The oracle logs every traversal's trips,
pixel and kind; k_paths' schedule is replayed (a wave = an 8x8 tile, a lane = a pixel, lock step per vertex).
Usage: merge_model.py <workload> [tile rows]   (SPP=<n>, default 16)"""
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
log = np.zeros(800_000_000 // 4, dtype=np.uint32)
O.orc_set_trip_log_tagged(log.ctypes.data_as(C.c_void_p), log.size)
y0 = (h // 2) // 8 * 8
oracle_lib.render(O, packed, rp, threads=1, rows=(y0, y0 + 8 * tile_rows), want_ids=False, flags=oracle_lib.GPU_MIRROR)
n = O.orc_trip_log_length()
O.orc_set_trip_log(None, 0)
t = log[:n].reshape(-1, 2)
is_shadow = ((t[:, 0] >> 29) & 1).astype(bool)  # a step of a shadow walk (any-hit or closest-hit)
trips, pixel = (t[:, 0] & 0x1FFFFFFF).astype(np.int64), (t[:, 1] >> 4).astype(np.int64)
# Per pixel: vertex index = number of closest-hit traversals so far - 1; a vertex's shadow walks follow its closest-hit traversal.
# (The log is written pixel by pixel, in time order.)
px, py = pixel % w, pixel // w - y0
tiles_x = w // 8
keep = px < tiles_x * 8
trips, is_shadow, px, py = trips[keep], is_shadow[keep], px[keep], py[keep]
slot = ((py // 8) * tiles_x + px // 8) * 64 + (py % 8) * 8 + px % 8
order = np.argsort(slot, kind="stable")
slot, trips, is_shadow = slot[order], trips[order], is_shadow[order]
first = np.concatenate([[True], slot[1:] != slot[:-1]])
closest_no = np.cumsum(~is_shadow)
base_no = np.maximum.accumulate(np.where(first, closest_no - (~is_shadow), 0))
vertex = closest_no - base_no - 1  # index of the vertex this traversal belongs to, per pixel
n_slots, n_vert = slot.max() // 64 * 64 + 64, vertex.max() + 2
Cl = np.zeros((n_slots, n_vert), dtype=np.int64)
Sh_last = np.zeros((n_slots, n_vert), dtype=np.int64)   # the vertex's last shadow walk
Sh_other = np.zeros((n_slots, n_vert), dtype=np.int64)  # the earlier ones (walked one after the other, each its own phase: summed per lane as a bound)
c = ~is_shadow
Cl[slot[c], vertex[c]] = trips[c]
s_idx = np.flatnonzero(is_shadow)
last_of_vertex = np.ones(s_idx.size, dtype=bool)
nxt = s_idx + 1
ok = nxt < slot.size
last_of_vertex[ok] = ~(is_shadow[nxt[ok]] & (slot[nxt[ok]] == slot[s_idx[ok]]))
np.add.at(Sh_last, (slot[s_idx[last_of_vertex]], vertex[s_idx[last_of_vertex]]), trips[s_idx[last_of_vertex]])
np.add.at(Sh_other, (slot[s_idx[~last_of_vertex]], vertex[s_idx[~last_of_vertex]]), trips[s_idx[~last_of_vertex]])
W = lambda a: a.reshape(-1, 64, n_vert)
cl, sl, so = W(Cl), W(Sh_last), W(Sh_other)
total = cl.sum() + sl.sum() + so.sum()
now = cl.max(axis=1).sum() + sl.max(axis=1).sum() + so.max(axis=1).sum()
sl_shift = np.concatenate([np.zeros_like(sl[:, :, :1]), sl[:, :, :-1]], axis=2)  # vertex k-1's last shadow walk rides with vertex k's closest hit
merged = (cl + sl_shift).max(axis=1).sum() + so.max(axis=1).sum()
all_merged = (cl + sl_shift + np.concatenate([np.zeros_like(so[:, :, :1]), so[:, :, :-1]], axis=2)).max(axis=1).sum()
print(f"{name}: {8 * tile_rows} rows, {spp} spp: {c.sum()} closest-hit traversals (mean {trips[c].mean():.1f} trips), {is_shadow.sum()} shadow walks (mean {trips[is_shadow].mean():.1f})")
print(f"  lock step per phase (k_paths now)            : wave trips 100.0 %, {total / now:5.1f} of 64 lanes busy per trip")
print(f"  last shadow walk + next closest hit merged   : wave trips {100.0 * merged / now:5.1f} %, {total / merged:5.1f} of 64 lanes busy per trip")
print(f"  every shadow walk + next closest hit merged  : wave trips {100.0 * all_merged / now:5.1f} %, {total / all_merged:5.1f} of 64 lanes busy per trip")
