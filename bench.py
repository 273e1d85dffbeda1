#!/usr/bin/env python3
"""bench.py — the hot path (main.comp's work) on N MI355X, one process per GPU.

A "step" is one full render of the workload (one pass of the hot path over one frame of
synthetic input, scene already resident in HBM).  Default workload = BASELINE.json
configs[1]: the generated Cornell box with an area light, 1920x1080, sample_count 128
(121 effective strata, main.comp:596) on one GPU.

N > 1 (launched by torch.distributed.run, one rank per GPU, RCCL): weak scaling — every
rank renders the same frame with all 121 strata from its own decorrelated seed, the float4
accumulators (sum RGB + valid-sample count) are sum-reduced to rank 0 over xGMI (the
path's one real exchange step) and rank 0 resolves the N x 121 spp image.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_TFLOPS = 157.3       # packed-FMA fp32 vector peak; plain v_fma_f32 is half (78.6)

WORKLOADS = {
    # name: (scene factory name, kwargs, width, height, sample_count, description)
    "cornell1080": ("cornell_scene", {}, 1920, 1080, 128, "cfg2: Cornell box + area light, 1920x1080, 128 spp (121 effective)"),
    "cube512": ("cube_scene", {}, 512, 512, 16, "cfg1: cube, 512x512, 16 spp"),
    "textured1080": ("textured_scene", {"grid": 48}, 1920, 1080, 64, "textured PBR test scene, 1920x1080, 64 spp"),
    "soup1080": ("soup_scene", {"n_tris": 1_000_000}, 1920, 1080, 16, "1M-triangle soup, 1920x1080, 16 spp"),
    "atrium1080": ("atrium_scene", {"detail": 16, "tex_size": 1024}, 1920, 1080, 256,
                   "cfg3: generated Sponza-class textured PBR atrium (~253k triangles, 44 1024^2 textures), 1920x1080, 256 spp"),
    "rocks1080": ("rocks_scene", {"n_tris": 5_000_000}, 1920, 1080, 512,
                  "cfg4: generated ~5M-triangle rock clusters (deep BVH), 1920x1080, 512 spp (484 effective)"),
}
WORKLOADS["cornell4k"] = ("cornell_scene", {}, 3840, 2160, 4096,
                          "cfg5: Cornell box + area light, 3840x2160, 4096 spp (64x64 strata), one GPU's share when N > 1 is the full frame (weak scaling)")
BG = {"atrium1080": (0.5, 0.7, 1.0), "rocks1080": (0.5, 0.7, 1.0), "soup1080": (0.5, 0.7, 1.0)}  # sky; others: README default


def kernel_roofline(name: str, ks: dict, verts: int) -> dict:
    """Algorithmic bytes/flops of one kernel over its measured device time."""
    V, T, H, X = ks["node_visits"], ks["tri_tests"], ks["hits_shaded"], ks["texture_taps"]
    bytes_total = 64 * V + 36 * T + 164 * H + 16 * X
    flops_total = 44 * V + 55 * T + (400 * verts if name in ("shade", "paths") else 0)
    ms = ks["ms"]
    launches = max(ks["launches"], 1)
    gbs = bytes_total / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    tfs = flops_total / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    return {
        "kernel": "k_" + name, "launches": ks["launches"], "avg_launch_us": 1e3 * ms / launches,
        "bytes_per_launch": bytes_total / launches, "flops_per_launch": flops_total / launches,
        "hbm": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None},
        "valu": {"bound": "valu", "achieved": tfs, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tfs / VALU_PEAK_TFLOPS},
    }


def cpu_baseline(packed, rp, seconds_hint: float = 15.0) -> dict:
    """The CPU oracle (a port of main.comp, oracle/pt_oracle.c) timed on the host cores on a
    bounded band of rows of the same workload. Reported beside the GPU number; never the target."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    lib = oracle_lib.load()
    cores = min(16, os.cpu_count() or 1)
    h = rp.pc.resolution_y
    # warm up (library load, thread start), calibrate on 4 rows, then size the band for ~seconds_hint
    mid = h // 2
    t0 = time.perf_counter()
    oracle_lib.render(lib, packed, rp, threads=cores, rows=(mid, min(mid + 1, h)), want_ids=False)
    first = time.perf_counter() - t0
    cal = 1 if first > 2.0 else min(4, h)  # heavy workloads (hundreds of spp on millions of triangles): one row is sample enough
    t0 = time.perf_counter()
    oracle_lib.render(lib, packed, rp, threads=cores, rows=(mid, min(mid + cal, h)), want_ids=False)
    dt = max(time.perf_counter() - t0, 1e-6)
    progress(f"oracle: {cal} row(s) in {dt:.1f} s")
    rows = int(max(cal, min(h, cal * seconds_hint / dt)))
    y0 = max(0, mid - rows // 2)
    y1 = min(h, y0 + rows)
    t0 = time.perf_counter()
    _, _, _, st = oracle_lib.render(lib, packed, rp, threads=cores, rows=(y0, y1), want_ids=False)
    dt = time.perf_counter() - t0
    return {
        "value": st["rays"] / dt / 1e6, "unit": "Mrays/s", "mpaths_per_s": st["paths"] / dt / 1e6, "cores": cores,
        "kind": "port", "seconds": dt,
        "sample": f"rows {y0}..{y1} of {h} of the same frame ({st['paths']} paths, {st['rays']} rays), oracle/pt_oracle.c, {cores} threads",
    }


T_START = time.perf_counter()


def progress(msg: str) -> None:
    """One line on stderr per phase: long workloads (5 M triangles) must not look hung."""
    print(f"[bench +{time.perf_counter() - T_START:6.1f}s] {msg}", file=sys.stderr, flush=True)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cornell1080", choices=sorted(WORKLOADS))
    ap.add_argument("--spp", type=int, default=0, help="override sample_count (dev only; invalidates the headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true", help="skip the extra profiled step that feeds `roofline`")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    from pathtracer_amd import abi, scenes
    from pathtracer_amd.renderer import Renderer

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py needs a GPU (the HIP path has no CPU fallback)", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    factory, kwargs, width, height, spp, desc = WORKLOADS[args.workload]
    if args.spp:
        spp = args.spp
    progress(f"generating {args.workload}")
    scene = getattr(scenes, factory)(**kwargs)
    progress(f"{len(scene.triangles)} triangles; building the BVH")
    r = Renderer(local_rank)
    t0 = time.perf_counter()
    packed = r.pack(scene)
    build_s = time.perf_counter() - t0
    progress(f"{len(packed.nodes)} nodes, depth {packed.bvh_max_depth}; uploading")
    r.upload(packed)
    progress("rendering")
    from pathtracer_amd.distributed import rank_seed
    seed = rank_seed(0x5EED0000, rank)  # "replicate" split: one decorrelated stream family per rank
    rp = packed.render_params(r.lib, width, height, spp, BG.get(args.workload, (0.239, 0.239, 0.239)), seed)
    pixels = width * height

    accum = torch.empty(pixels * 4, dtype=torch.float32, device=dev)
    rgba = torch.empty(pixels, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev)
    stats = abi.Stats()

    def step(flags: int = 0):
        rp.flags = flags
        rc = r.lib.pt_render_accum_device(r.ctx, C.byref(rp), C.c_void_p(accum.data_ptr()), None,
                                          C.c_void_p(stream.cuda_stream), C.byref(stats))
        r._check(rc, "pt_render_accum_device")
        if world > 1:
            dist.reduce(accum, dst=0, op=dist.ReduceOp.SUM)  # RCCL over xGMI: 16 B/pixel
        if rank == 0:
            rc = r.lib.pt_resolve_device(r.ctx, C.c_void_p(accum.data_ptr()), C.c_uint32(pixels),
                                         C.c_void_p(rgba.data_ptr()), C.c_void_p(stream.cuda_stream))
            r._check(rc, "pt_resolve_device")

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    rays = paths = 0
    render_ms = 0.0
    for _ in range(args.steps):
        step()
        rays += stats.rays
        paths += stats.paths
        render_ms += stats.render_ms
    fence()
    elapsed = time.perf_counter() - t0
    progress(f"{args.steps} timed steps in {elapsed:.2f} s")

    tot = torch.tensor([elapsed, float(rays), float(paths)], dtype=torch.float64, device=dev)
    if world > 1:
        mx = tot.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        elapsed = float(mx[0])
    total_rays, total_paths = float(tot[1]), float(tot[2])

    roof = None
    kernels = None
    if not args.no_kernel_timing:
        # per-kernel device time: HIP events around every launch on the launch stream
        step(abi.FLAG_TIME_KERNELS)
        torch.cuda.synchronize(dev)
        sd = stats.as_dict()
        work = ("rays", "node_visits", "tri_tests", "hits_shaded", "texture_taps")
        if sd["path_kernel"]["launches"] and not sd["extend"]["launches"]:
            # path pipeline: k_paths does the extend, shade and shadow-walk work of every vertex in one launch
            pk = dict(sd["path_kernel"])
            for k in work:
                pk[k] = sd["extend"][k] + sd["shade"][k] + sd["shadow"][k]
            sd["paths"] = pk
            names = ("paths",)
        else:
            if sd["shadow"]["launches"] == 0:  # the shadow walks run inside k_shade: their traversal work belongs to it
                for k in work:
                    sd["shade"][k] += sd["shadow"][k]
            names = ("extend", "shade", "shadow")
        kernels = {k: kernel_roofline(k, sd[k], sd["shaded_vertices"]) for k in names if sd[k]["launches"]}
        dominant = max(kernels, key=lambda k: sd[k]["ms"])
        roof = dict(kernels[dominant]["hbm"])
        roof["kernel"] = kernels[dominant]["kernel"]
        roof["avg_launch_us"] = kernels[dominant]["avg_launch_us"]
        roof["bytes_per_launch"] = kernels[dominant]["bytes_per_launch"]
        roof["valu"] = kernels[dominant]["valu"]
        roof["timed_step_ms"] = sd["render_ms"]
        roof["kernel_ms"] = {k: sd[k]["ms"] for k in kernels}
        if sd["path_kernel"]["launches"] and "paths" not in kernels:
            roof["kernel_ms"]["paths"] = sd["path_kernel"]["ms"]
        scene_bytes = 64 * (len(packed.nodes) // 2) + 48 * int(scene.triangles.shape[0])
        if scene_bytes > 16384:
            # scenes read from global memory: the measured ceiling of dependent random 64-byte gathers on this GPU
            # (scripts/micro/gather64.hip: 11.1 TB/s within one XCD's L2, 4.4 TB/s at 32 MiB, 3.55 TB/s from 128 MiB up)
            ceiling = 11100.0 if scene_bytes <= (4 << 20) else (4400.0 if scene_bytes <= (48 << 20) else 3550.0)
            roof["random_gather_ceiling_gbs"] = ceiling
            roof["frac_of_gather_ceiling"] = roof["achieved"] / ceiling
        else:
            # LDS-resident scene: the algorithmic bytes are read from each workgroup's LDS copy, not from HBM, so this
            # fraction of the HBM peak can pass 1; what reaches memory is `traffic`, what binds is VALU issue (`valu`)
            roof["served_from"] = "lds"
        traffic_file = os.path.join(ROOT, "profiles", f"traffic_{args.workload}.json")
        if os.path.exists(traffic_file) and not args.spp:
            # HBM-side bytes per launch from the committed rocprofv3 --pmc passes of this same command (scripts/pmc_passes.sh)
            tk = json.load(open(traffic_file)).get("kernels", {}).get(roof["kernel"])
            if tk:
                roof["traffic"] = tk["bytes_per_launch"]
                roof["traffic_source"] = os.path.relpath(traffic_file, ROOT)

    if rank == 0:
        last = stats.as_dict()
        out = {
            "metric": "Mrays/s at 1080p (rays = BVH traversals: closest-hit + shadow-walk steps)",
            "value": total_rays / elapsed / 1e6,
            "unit": "Mrays/s",
            "mpaths_per_s": total_paths / elapsed / 1e6,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / max(args.steps, 1),
            "device_ms_per_step": render_ms / max(args.steps, 1),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": desc, "name": args.workload, "resolution": [width, height], "sample_count": spp,
                       "effective_spp": int(np.sqrt(np.float32(spp))) ** 2, "triangles": int(scene.triangles.shape[0]),
                       "bvh_nodes": int(len(packed.nodes)), "bvh_depth": int(packed.bvh_max_depth),
                       "parallelism": f"dp{world}: full frame per GPU, own seed, RCCL sum-reduce of float4 accumulators",
                       "rays_per_step_per_gpu": int(last["rays"]), "paths_per_step_per_gpu": int(last["paths"]),
                       "iterations": int(last["iterations"]), "host_bvh_build_s": build_s},
            "counters": {k: int(last[k]) for k in ("paths", "rays", "node_visits", "tri_tests", "hits_shaded", "texture_taps",
                                                   "shaded_vertices", "nan_samples")},
        }
        if roof:
            out["roofline"] = roof
            out["kernels"] = kernels
        if not args.no_cpu_baseline and world == 1:
            progress("timing the CPU oracle on a band of rows")
            out["cpu_baseline"] = cpu_baseline(packed, rp)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    r.close()


if __name__ == "__main__":
    main()
