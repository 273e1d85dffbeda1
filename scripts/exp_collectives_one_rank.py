#!/usr/bin/env python3
"""De-risking N > 1 on ONE GPU. With N > 1 a rank owns torch's stream, the context's stream, two band streams and
RCCL's own, and HIP deals streams to GPU_MAX_HW_QUEUES = 4 hardware queues: two bands that share a queue are slower than one band
(DESIGN.md §5). So: bench.py's ms_per_step with a live one-rank RCCL communicator (PT_BENCH_FORCE_COLLECTIVES=1: process group bound to
the device, the accumulator reduce on the render stream, barrier, all-reduces) against the plain run, for cfg2 and cfg3, two bands and
one; then pt_create_multi with R = 1 (ncclCommInitAll, grouped ncclReduce to the first GPU) at 1080p and 4K: reduce_ms of the 33 / 133 MB
accumulator frame. Usage (GPU box): python scripts/exp_collectives_one_rank.py > gpurun_out/r03_collectives_one_rank.txt"""
import ctypes as C, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def bench(workload, steps, env):
    e = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29677", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", **env)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--steps", str(steps), "--warmup", "1", "--configs", "none",
                        "--no-cpu-baseline", "--no-fast-arith", "--no-kernel-timing"], env=e, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    return json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])


print("bench.py, one rank, with and without a live RCCL communicator (ms per frame; reduce = events around dist.reduce on the render stream)")
for wl, steps in (("cornell1080", 10), ("atrium1080", 3)):
    for bands in ("2", "1"):
        plain = bench(wl, steps, {"PT_CONCURRENT_BANDS": bands})
        coll = bench(wl, steps, {"PT_CONCURRENT_BANDS": bands, "PT_BENCH_FORCE_COLLECTIVES": "1"})
        print(f"  {wl:12s} bands {bands}: plain {plain['ms_per_step']:9.2f} ms   with communicator {coll['ms_per_step']:9.2f} ms "
              f"({100.0 * (coll['ms_per_step'] / plain['ms_per_step'] - 1.0):+5.1f} %), reduce {coll['reduce_ms_per_step']:.3f} ms, frame verified {coll.get('frame_verified')}",
              flush=True)

import numpy as np
from pathtracer_amd import abi, lib, scenes
from pathtracer_amd.scene import PackedScene
L = lib.load()
ids = (C.c_int * 1)(0)
ctx = L.pt_create_multi(ids, 1)
assert ctx, L.pt_last_error(None)
plain = L.pt_create(0)
packed = PackedScene(scenes.cornell_scene(), L)
assert L.pt_upload_scene(ctx, C.byref(packed.desc)) == 0 and L.pt_upload_scene(plain, C.byref(packed.desc)) == 0
print("pt_create_multi, R = 1 (ncclCommInitAll + grouped ncclReduce in one process) against the plain context, Cornell box, 16 spp")
for w, h in ((1920, 1080), (3840, 2160)):
    rp = packed.render_params(L, w, h, 16, seed=5)
    rgba = np.zeros((h, w, 4), dtype=np.uint8)
    res = {}
    for name, c in (("plain", plain), ("group", ctx)):
        st = abi.Stats()
        best = 1e9
        for _ in range(4):
            t0 = time.perf_counter()
            assert L.pt_render(c, C.byref(rp), abi.ptr(rgba), None, None, C.byref(st)) == 0, L.pt_last_error(c)
            best = min(best, time.perf_counter() - t0)
        res[name] = (1e3 * best, st.render_ms, st.reduce_ms)
    print(f"  {w}x{h}: plain pt_render {res['plain'][0]:8.2f} ms (device {res['plain'][1]:.2f});  group {res['group'][0]:8.2f} ms (device {res['group'][1]:.2f}, "
          f"reduce of {16 * w * h / 1e6:.0f} MB {res['group'][2]:.3f} ms)", flush=True)
L.pt_destroy(ctx)
L.pt_destroy(plain)
