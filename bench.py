#!/usr/bin/env python3
"""bench.py — the hot path (main.comp's work) on N MI355X, one process per GPU.

A "step" is one full render of the workload (one pass of the hot path over one frame of
synthetic input, scene already resident in HBM).  Headline workload = BASELINE.json
configs[2], the largest configuration defined on ONE GPU: the generated Sponza-class
textured PBR atrium (~253 k triangles, 44 1024^2 textures), 1920x1080, 256 spp.

N > 1 (launched by torch.distributed.run, one rank per GPU, RCCL):
  --split strata (default: north_star's partitioning, the split BASELINE configs[3]/[4] are
      defined on; strong scaling of one frame): rank r renders strata [r n^2/N, (r+1) n^2/N)
      of every pixel;
  --split replicate (weak scaling): every rank renders the same frame with all its strata
      from its own decorrelated seed.
Either way the float4 accumulators (sum RGB + valid-sample count) are sum-reduced to rank 0
over xGMI — the path's one exchange step, SURVEY.md §8e — and rank 0 resolves the image.
Render, reduce and resolve are ordered on ONE explicit (non-default) stream per rank.

At N = 1 the same command also measures the other single-GPU configs of BASELINE.json
(`--configs`, default cfg2 cornell1080, cfg1 cube512, cfg4's one-GPU frame rocks1080, cfg5 cornell4k) and
reports them in the `configs` array of the same JSON line, each with counters, roofline and
a bounded CPU-oracle sample that is ALSO compared bit for bit with the GPU frame.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes as C
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_TFLOPS = 157.3       # packed-FMA fp32 vector peak (256 CUs x 128 lanes x 2 flop x 2.4 GHz)
VALU_NO_FMA_TFLOPS = 78.6      # what an FMA-free instruction stream (the bit-exact arithmetic contract) can reach at best
LDS_PEAK_GBS = 150000.0        # ds_read_b128 aggregate, every CU streaming (MI355X_MICROARCH.md §LDS)

WORKLOADS = {
    # name: (scene factory name, kwargs, width, height, sample_count, description)
    "cornell1080": ("cornell_scene", {}, 1920, 1080, 128, "cfg2: Cornell box + area light, 1920x1080, 128 spp (121 effective)"),
    "cube512": ("cube_scene", {}, 512, 512, 16, "cfg1: cube, 512x512, 16 spp"),
    "textured1080": ("textured_scene", {"grid": 48}, 1920, 1080, 64, "textured PBR test scene, 1920x1080, 64 spp"),
    "soup1080": ("soup_scene", {"n_tris": 1_000_000}, 1920, 1080, 16, "1M-triangle soup, 1920x1080, 16 spp"),
    "atrium1080": ("atrium_scene", {"detail": 16, "tex_size": 1024}, 1920, 1080, 256,
                   "cfg3: generated Sponza-class textured PBR atrium (~253k triangles, 44 1024^2 textures), 1920x1080, 256 spp"),
    "rocks1080": ("rocks_scene", {"n_tris": 5_000_000}, 1920, 1080, 512,
                  "cfg4: generated ~5M-triangle rock clusters (deep BVH), 1920x1080, 512 spp (484 effective); one GPU's frame"),
    "cornell4k": ("cornell_scene", {}, 3840, 2160, 4096,
                  "cfg5: Cornell box + area light, 3840x2160, 4096 spp (64x64 strata)"),
    "cornell_tiny": ("cornell_scene", {}, 96, 54, 16, "test-sized Cornell box (tests/test_bench_script.py); not a bench line"),
}
BG = {"atrium1080": (0.5, 0.7, 1.0), "rocks1080": (0.5, 0.7, 1.0), "soup1080": (0.5, 0.7, 1.0)}  # sky; others: README default
HEADLINE = "atrium1080"
FAST_ARITH_WORKLOAD = "cornell1080"  # cfg2: the LDS-resident scene PT_ARITH=fast is for
EXTRA_DEFAULT = "cornell1080,cube512,atrium1080,rocks1080,cornell4k"  # (the headline workload is skipped in this list)
EXTRA_STEPS = {"cornell1080": 20, "cube512": 20, "atrium1080": 2, "rocks1080": 2, "cornell4k": 1}
SEED = 0x5EED0000
COUNTERS = ("paths", "rays", "rays_closest", "rays_shadow_walked", "rays_shadow_shortcut", "node_visits", "tri_tests",
            "hits_shaded", "texture_taps", "shaded_vertices", "nan_samples")

T_START = time.perf_counter()


def progress(msg: str) -> None:
    """One line on stderr per phase: long workloads (5 M triangles) must not look hung."""
    print(f"[bench +{time.perf_counter() - T_START:6.1f}s] {msg}", file=sys.stderr, flush=True)


# --------------------------------------------------------------------------- backends

class HipBackend:
    """The product path: libpathtracer_amd.so through the C ABI, device buffers and the collective from torch.
    Render, reduce and resolve all run on `self.stream`, an explicit non-default stream (the library's NULL stream
    is its own non-blocking stream, unordered against a collective: include/pathtracer_amd.h)."""
    dist_backend = "nccl"

    def __init__(self, local_rank: int):
        import torch
        from pathtracer_amd.renderer import Renderer
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
        self.torch = torch
        if local_rank >= torch.cuda.device_count():  # a launcher that masks the devices per rank leaves each rank its own as device 0
            local_rank %= torch.cuda.device_count()
        torch.cuda.set_device(local_rank)
        self.dev = torch.device("cuda", local_rank)
        self.r = Renderer(local_rank)
        self.lib = self.r.lib
        self.stream = torch.cuda.Stream(self.dev)
        self.tensor_device = self.dev
        self.reduce_events = []

    def pack(self, scene):
        return self.r.pack(scene)

    def upload(self, packed):
        self.r.upload(packed)

    def alloc(self, pixels: int):
        t = self.torch
        self.pixels = pixels
        self.accum = t.empty(pixels * 4, dtype=t.float32, device=self.dev)
        self.ids = t.empty(pixels, dtype=t.int32, device=self.dev)
        self.rgba = t.empty(pixels, dtype=t.int32, device=self.dev)

    def render(self, rp, stats, want_ids: bool = False):
        from pathtracer_amd import abi  # noqa: F401
        with self.torch.cuda.stream(self.stream):
            rc = self.lib.pt_render_accum_device(self.r.ctx, C.byref(rp), C.c_void_p(self.accum.data_ptr()),
                                                 C.c_void_p(self.ids.data_ptr()) if want_ids else None,
                                                 C.c_void_p(self.stream.cuda_stream), C.byref(stats) if stats is not None else None)
        self.r._check(rc, "pt_render_accum_device")

    def reduce(self, dist):
        t = self.torch
        with t.cuda.stream(self.stream):
            e0, e1 = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
            e0.record(self.stream)
            dist.reduce(self.accum, dst=0, op=dist.ReduceOp.SUM)  # RCCL over xGMI: 16 B/pixel
            e1.record(self.stream)
            self.reduce_events.append((e0, e1))

    def reduce_ms(self) -> float:
        ms = sum(a.elapsed_time(b) for a, b in self.reduce_events)
        self.reduce_events = []
        return ms

    def resolve(self):
        with self.torch.cuda.stream(self.stream):
            rc = self.lib.pt_resolve_device(self.r.ctx, C.c_void_p(self.accum.data_ptr()), C.c_uint32(self.pixels),
                                            C.c_void_p(self.rgba.data_ptr()), C.c_void_p(self.stream.cuda_stream))
        self.r._check(rc, "pt_resolve_device")

    def sync(self):
        self.stream.synchronize()
        self.torch.cuda.synchronize(self.dev)

    def host_frame(self):
        """(accum float32 [H*W,4], ids uint32 [H*W], rgba uint8 [H*W,4]) of the last render, on the host."""
        import numpy as np
        self.sync()
        return (self.accum.cpu().numpy().reshape(-1, 4), self.ids.cpu().numpy().view(np.uint32),
                self.rgba.cpu().numpy().view(np.uint8).reshape(-1, 4))

    def close(self):
        self.r.close()


# --------------------------------------------------------------------------- rooflines

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
        "kernel": "k_" + name, "launches": ks["launches"], "avg_launch_us": 1e3 * ms / launches, "ms": ms,
        "bytes_per_launch": bytes_total / launches, "flops_per_launch": flops_total / launches,
        "algorithmic_gbs": gbs, "algorithmic_tflops": tfs,
    }


def roofline_of(sd: dict, scene_in_lds: bool, workload: str, scene_bytes: int, modified: bool) -> tuple[dict, dict]:
    """`roofline` object of the dominant kernel of one profiled step (stats dict `sd`), plus the per-kernel table."""
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
    if "paths" in kernels and sd.get("merged_walk"):
        kernels["paths"]["kernel"] = "k_paths_merged"  # scenes in global memory with <= 2 light samples per vertex (pt_stats.merged_walk)
    dom = kernels[max(kernels, key=lambda k: sd[k]["ms"])]
    traffic = traffic_source = None
    traffic_file = os.path.join(ROOT, "profiles", f"traffic_{workload}.json")
    if os.path.exists(traffic_file) and not modified:
        # HBM-side bytes per launch from the committed rocprofv3 --pmc passes of this same workload (scripts/pmc_passes.sh)
        tk = json.load(open(traffic_file)).get("kernels", {}).get(dom["kernel"])
        if tk:
            traffic, traffic_source = tk["bytes_per_launch"], os.path.relpath(traffic_file, ROOT)
    common = {"kernel": dom["kernel"], "launches": dom["launches"], "avg_launch_us": dom["avg_launch_us"],
              "bytes_per_launch": dom["bytes_per_launch"], "flops_per_launch": dom["flops_per_launch"],
              "traffic": traffic, "traffic_source": traffic_source, "timed_step_ms": sd["render_ms"]}
    valu = {"achieved": dom["algorithmic_tflops"], "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": dom["algorithmic_tflops"] / VALU_PEAK_TFLOPS,
            "frac_of_no_fma_peak": dom["algorithmic_tflops"] / VALU_NO_FMA_TFLOPS,
            "note": "algorithmic flops 44/node visit + 55/triangle test + 400/shaded vertex (SURVEY.md 8d) over the packed-FMA fp32 "
                    "vector peak; the bit-exact contract forms no FMA, so 78.6 TFLOP/s is its ceiling"}
    if scene_in_lds:
        # The scene is staged in every workgroup's LDS: the algorithmic bytes are LDS reads, HBM sees only `traffic`
        # (path state in and out per launch). What bounds the kernel is VALU issue, so that is the roofline reported.
        roof = {"bound": "valu", **valu, **common, "served_from": "lds",
                "lds_algorithmic_gbs": dom["algorithmic_gbs"], "lds_frac_of_read_peak": dom["algorithmic_gbs"] / LDS_PEAK_GBS}
        if traffic is not None:
            roof["hbm_traffic_gbs"] = traffic / (dom["avg_launch_us"] * 1e-6) / 1e9
            roof["hbm_traffic_frac"] = roof["hbm_traffic_gbs"] / HBM_PEAK_GBS
    else:
        frac = dom["algorithmic_gbs"] / HBM_PEAK_GBS
        roof = {"bound": "hbm", "achieved": dom["algorithmic_gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": frac, **common,
                "served_from": "l2/infinity-cache/hbm", "valu": valu}
        # the measured ceiling of dependent random 64-byte gathers on this GPU (scripts/micro/gather64.hip:
        # 11.1 TB/s within one XCD's L2, 4.4 TB/s at 32 MiB, 3.55 TB/s from 128 MiB up)
        ceiling = 11100.0 if scene_bytes <= (4 << 20) else (4400.0 if scene_bytes <= (48 << 20) else 3550.0)
        roof["random_gather_ceiling_gbs"] = ceiling
        roof["frac_of_gather_ceiling"] = dom["algorithmic_gbs"] / ceiling
    return roof, kernels


# --------------------------------------------------------------------------- CPU baseline (the oracle, as checker and yardstick)

def lavapipe_probe() -> str:
    """SURVEY.md §8d / BASELINE.md: the preferred CPU baseline is the reference's own Vulkan path on Mesa lavapipe. Probe what
    that needs on this machine and name the first thing that is missing; the number is never made up."""
    import ctypes.util
    import glob
    import shutil
    if not ctypes.util.find_library("vulkan") and not glob.glob("/usr/lib/*/libvulkan.so*") and not glob.glob("/usr/lib64/libvulkan.so*"):
        return "unavailable (no Vulkan loader: libvulkan.so.1 not found)"
    icds = [f for d in ("/usr/share/vulkan/icd.d", "/etc/vulkan/icd.d", "/usr/local/share/vulkan/icd.d") for f in glob.glob(d + "/lvp_icd*.json")]
    if not icds:
        return "unavailable (no lavapipe ICD: lvp_icd*.json not found in the Vulkan icd.d directories)"
    if not shutil.which("glslc"):
        return "unavailable (no glslc to compile main.comp to SPIR-V)"
    return "unavailable (loader, lavapipe ICD and glslc found, but the reference's sources and its five submodules do not travel with this repository)"


def host_cpu_facts() -> dict:
    """What the host gives this process: the machine's CPUs, the ones it may run on, and the cgroup quota if one is set."""
    facts = {"os_cpu_count": os.cpu_count() or 1}
    try:
        facts["affinity_cores"] = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        facts["affinity_cores"] = facts["os_cpu_count"]
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        facts["cgroup_quota_cores"] = None if quota == "max" else float(quota) / float(period)
    except (OSError, ValueError):
        facts["cgroup_quota_cores"] = None
    return facts


def cpu_baseline(packed, rp, gpu_accum, gpu_ids, seconds_hint: float) -> dict:
    """The CPU oracle (a port of main.comp, oracle/pt_oracle.c) timed on the host cores on a bounded band of rows of
    the same frame, and that band compared bit for bit with the GPU's frame. Reported beside the GPU number; never
    the target. This leg is the only place bench.py touches oracle/.
    `cores` = the threads used: min(16, CPUs this process may run on, cgroup quota). A one-thread calibration on one row of the
    same frame at 4 spp is reported beside it, so that two boxes' numbers can be told apart (clock and load against thread count)."""
    import copy
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    lib = oracle_lib.load()
    facts = host_cpu_facts()
    cores = min(16, facts["affinity_cores"])
    if facts["cgroup_quota_cores"]:
        cores = max(1, min(cores, int(facts["cgroup_quota_cores"])))
    w, h = rp.pc.resolution_x, rp.pc.resolution_y
    mid = h // 2
    small = copy.copy(rp)
    small.pc = copy.copy(rp.pc)
    small.pc.sample_count = 4
    t0 = time.perf_counter()
    st1 = oracle_lib.render(lib, packed, small, threads=1, rows=(mid, min(mid + 1, h)), want_ids=False)[3]  # one-thread calibration
    dt1 = max(time.perf_counter() - t0, 1e-6)
    t0 = time.perf_counter()
    oracle_lib.render(lib, packed, rp, threads=cores, rows=(mid, min(mid + 1, h)), want_ids=False)  # warm-up + sizing of the band
    dt = max(time.perf_counter() - t0, 1e-6)
    rows = int(max(1, min(h, seconds_hint / dt)))
    y0 = max(0, mid - rows // 2)
    y1 = min(h, y0 + rows)
    t0 = time.perf_counter()
    _, o_accum, o_ids, st = oracle_lib.render(lib, packed, rp, threads=cores, rows=(y0, y1), want_ids=True)
    dt = time.perf_counter() - t0
    value, single = st["rays"] / dt / 1e6, st1["rays"] / dt1 / 1e6
    out = {
        "value": value, "unit": "Mrays/s", "mpaths_per_s": st["paths"] / dt / 1e6, "cores": cores,
        "kind": "port", "seconds": dt,
        "sample": f"rows {y0}..{y1} of {h} of the same frame ({st['paths']} paths, {st['rays']} rays), oracle/pt_oracle.c, {cores} threads",
        "host": facts, "threads": cores,
        "single_thread": {"value": single, "unit": "Mrays/s", "seconds": dt1,
                          "sample": f"row {mid} of the same frame at 4 spp ({st1['rays']} rays), 1 thread"},
        "parallel_speedup": value / single if single > 0 else None,
        "lavapipe": lavapipe_probe(),
    }
    if gpu_accum is not None:
        ga = gpu_accum.reshape(h, w, 4)[y0:y1].view(np.uint32)
        gi = gpu_ids.reshape(h, w)[y0:y1]
        out["gpu_band_bit_exact"] = bool((ga == o_accum[y0:y1].view(np.uint32)).all() and (gi == o_ids[y0:y1]).all())
    return out


# --------------------------------------------------------------------------- one workload

def run_workload(be, name: str, steps: int, warmup: int, world: int, rank: int, split: str, spp_override: int,
                 kernel_timing: bool, baseline_seconds: float, dist, collectives: bool | None = None, want_noise: bool = False) -> dict | None:
    """Render `name` steps times on every rank; returns rank 0's result dict (None elsewhere).  `collectives`: run the
    reduce / barrier / all-reduce legs (default: whenever world > 1; a one-GPU box forces them on a one-rank group to
    exercise the RCCL path, PT_BENCH_FORCE_COLLECTIVES=1)."""
    import numpy as np
    import torch
    from pathtracer_amd import abi, scenes
    from pathtracer_amd.distributed import rank_seed

    factory, kwargs, width, height, spp, desc = WORKLOADS[name]
    if spp_override:
        spp = spp_override
    progress(f"{name}: generating the scene")
    scene = getattr(scenes, factory)(**kwargs)
    progress(f"{name}: {len(scene.triangles)} triangles; building the BVH")
    t0 = time.perf_counter()
    packed = be.pack(scene)
    build_s = time.perf_counter() - t0
    progress(f"{name}: {len(packed.nodes)} nodes, depth {packed.bvh_max_depth}; uploading")
    t0 = time.perf_counter()
    be.upload(packed)
    upload_s = time.perf_counter() - t0  # validation + re-layout into pair records + H2D, once per scene (never per frame)
    bg = BG.get(name, (0.239, 0.239, 0.239))
    if split == "strata":
        rp = packed.render_params(be.lib, width, height, spp, bg, SEED, rank, world)
    else:  # replicate: one decorrelated stream family per rank, all strata
        rp = packed.render_params(be.lib, width, height, spp, bg, rank_seed(SEED, rank))
    pixels = width * height
    be.alloc(pixels)
    stats = abi.Stats()
    if collectives is None:
        collectives = world > 1

    def step(flags: int = 0, want_ids: bool = False, collective: bool = True):
        rp.flags = flags
        be.render(rp, stats, want_ids)
        if collectives and collective:
            be.reduce(dist)
        if rank == 0:
            be.resolve()

    def fence():
        be.sync()
        if collectives:
            dist.barrier()
        be.sync()

    # The profiled step (HIP events around every launch on the launch stream) runs first: it is also a warm-up.
    sd = None
    if kernel_timing:
        step(abi.FLAG_TIME_KERNELS)
        be.sync()
        sd = stats.as_dict()
    for _ in range(warmup):
        step()
    fence()
    if hasattr(be, "reduce_events"):
        be.reduce_events = []
    progress(f"{name}: timing {steps} step(s)")
    t0 = time.perf_counter()
    rays = paths = 0
    render_ms = 0.0
    for _ in range(steps):
        step()
        rays += stats.rays
        paths += stats.paths
        render_ms += stats.render_ms
    fence()
    elapsed = time.perf_counter() - t0
    reduce_ms = be.reduce_ms() if collectives else 0.0
    progress(f"{name}: {steps} timed step(s) in {elapsed:.2f} s")

    tot = torch.tensor([elapsed, float(rays), float(paths)], dtype=torch.float64, device=be.tensor_device)
    if collectives:
        mx = tot.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        elapsed = float(mx[0])
    total_rays, total_paths = float(tot[1]), float(tot[2])

    # Rank 0's own frame once more, un-reduced and with primary-hit ids: the frame the checks below look at.
    verify = rank == 0 and (split == "replicate" or world == 1)
    frame = None
    if verify:
        step(want_ids=True, collective=False)
        frame = be.host_frame()
    last = stats.as_dict()
    reduced = None
    if collectives and split == "strata" and world > 1:
        # strata split: no rank holds the frame on its own, and rank r > 0 draws from its own stream family, so the sum cannot be
        # compared with a one-GPU digest. What must hold for the REDUCED frame is checked instead (below): every stratum of every
        # pixel arrived exactly once (valid + dropped samples == n^2), radiance within the clamp, counters summed over the ranks.
        step(want_ids=False, collective=True)
        nan_total = torch.tensor([float(stats.nan_samples), float(stats.paths)], dtype=torch.float64, device=be.tensor_device)
        dist.all_reduce(nan_total, op=dist.ReduceOp.SUM)
        if rank == 0:
            reduced = (be.host_frame(), int(nan_total[0]), int(nan_total[1]))
    if collectives:
        dist.barrier()
    if rank != 0:
        return None

    eff = int(np.sqrt(np.float32(spp))) ** 2
    out = {
        "name": name, "workload": desc,
        "value": total_rays / elapsed / 1e6, "unit": "Mrays/s", "mpaths_per_s": total_paths / elapsed / 1e6,
        "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * elapsed / max(steps, 1),
        "device_ms_per_step": render_ms / max(steps, 1),
        "config": {"workload": desc, "name": name, "resolution": [width, height], "sample_count": spp, "effective_spp": eff,
                   "triangles": int(scene.triangles.shape[0]), "bvh_nodes": int(len(packed.nodes)),
                   "bvh_depth": int(packed.bvh_max_depth), "host_bvh_build_s": build_s, "scene_upload_s": upload_s, "iterations": int(last["iterations"]),
                   "rays_per_step_per_gpu": int(stats.rays), "paths_per_step_per_gpu": int(stats.paths)},
        "counters": {k: int(last[k]) for k in COUNTERS},
    }
    if collectives:
        out["reduce_ms_per_step"] = reduce_ms / max(steps, 1)
    if last["rays"]:
        out["ray_mix"] = {"closest_hit": last["rays_closest"] / last["rays"], "shadow_walked": last["rays_shadow_walked"] / last["rays"],
                          "shadow_shortcut_one_triangle_test": last["rays_shadow_shortcut"] / last["rays"]}
    scene_bytes = 64 * (len(packed.nodes) // 2) + 48 * int(scene.triangles.shape[0])
    if sd is not None:
        out["roofline"], out["kernels"] = roofline_of(sd, bool(sd["scene_in_lds"]), name, scene_bytes, bool(spp_override))
    if "roofline" in out and out["device_ms_per_step"] > 0:
        # The kernel figures above come from the profiled step, which renders the frame's bands one after another (a launch
        # that shares the GPU with another band's launch cannot be timed on its own). The timed steps overlap the bands'
        # launches (2 streams), so the whole frame runs above the isolated kernel's rate: same work over the frame's device time.
        c, dev_s = out["counters"], out["device_ms_per_step"] * 1e-3
        fl = 44 * c["node_visits"] + 55 * c["tri_tests"] + 400 * c["shaded_vertices"]
        by = 64 * c["node_visits"] + 36 * c["tri_tests"] + 164 * c["hits_shaded"] + 16 * c["texture_taps"]
        wf = {"device_ms": out["device_ms_per_step"], "algorithmic_tflops": fl / dev_s / 1e12, "valu_frac": fl / dev_s / 1e12 / VALU_PEAK_TFLOPS,
              "note": "timed steps: the frame's bands overlap on 2 streams (PT_CONCURRENT_BANDS); init / finalize / resolve included"}
        if not sd["scene_in_lds"]:
            wf["algorithmic_gbs"] = by / dev_s / 1e9
            wf["hbm_frac"] = by / dev_s / 1e9 / HBM_PEAK_GBS
        out["roofline"]["whole_frame"] = wf
    if reduced is not None:
        (r_accum, _, r_rgba), nan_sum, path_sum = reduced
        valid = r_accum[:, 3]
        out["reduced_frame_properties_ok"] = bool(np.isfinite(r_accum).all() and (r_accum[:, :3] >= 0).all() and (r_accum[:, :3] <= 10.0 * eff).all()
                                                  and (valid <= eff).all() and int((eff - valid).sum()) == nan_sum
                                                  and path_sum == pixels * eff and (r_rgba[:, 3] == 255).all())
    if frame is not None:
        accum, ids, rgba = frame
        out["_frame"] = frame
        if want_noise:
            # The same frame from ANOTHER seed: the Monte-Carlo noise between two sample sets at this sample count, i.e. what a per-pixel
            # relMSE between the exact build and an opt-in build (whose paths decorrelate at the first grazing ray, Q2) can be held against.
            rp2 = packed.render_params(be.lib, width, height, spp, bg, SEED + 1)
            be.render(rp2, None, False)
            out["_noise"] = rel_mse(be.host_frame()[0], accum)
        gold_file = os.path.join(ROOT, "tests", "golden", f"{name}_full_digest.json")
        if name == "cornell1080":
            gold_file = os.path.join(ROOT, "tests", "golden", "cfg2_full_digest.json")
        if os.path.exists(gold_file) and not spp_override:
            # the oracle's SHA-256 of this frame at its own sample count, computed on the CPU and committed (tests/golden/make_*_digest.py):
            # the whole frame with its work counters (cfg2), or a band of rows of it (cfg3 / cfg4: the oracle needs hours for the frame)
            gold = json.load(open(gold_file))
            same_job = (gold["width"], gold["height"], gold["sample_count"], gold["seed"]) == (width, height, spp, rp.pc.seed)
            y0, y1 = gold.get("rows", (0, height))
            band = lambda a: np.ascontiguousarray(a.reshape(height, width, -1)[y0:y1])
            ok = (same_job and hashlib.sha256(band(accum).tobytes() + band(ids).tobytes()).hexdigest() == gold["sha256_accum_then_ids"]
                  and hashlib.sha256(band(rgba).tobytes()).hexdigest() == gold["sha256_rgba8"]
                  and all(int(last[k]) == gold["counters"][k] for k in gold.get("counters", {})))
            out["frame_verified"] = bool(ok)
            out["frame_verified_against"] = os.path.relpath(gold_file, ROOT)
            out["frame_verified_rows"] = [int(y0), int(y1)]
        valid = accum[:, 3]
        out["frame_properties_ok"] = bool(np.isfinite(accum).all() and (accum[:, :3] >= 0).all() and (accum[:, :3] <= 10.0 * eff).all()
                                          and (valid <= eff).all() and int((eff - valid).sum()) == last["nan_samples"]
                                          and last["paths"] == pixels * eff and (rgba[:, 3] == 255).all())
        if baseline_seconds > 0 and world == 1:
            progress(f"{name}: timing the CPU oracle on a band of rows (and checking the GPU frame against it)")
            rp.flags = 0
            out["cpu_baseline"] = cpu_baseline(packed, rp, accum, ids, baseline_seconds)
            progress(f"{name}: lavapipe baseline: {out['cpu_baseline']['lavapipe']}; CPU oracle on {out['cpu_baseline']['threads']} threads instead")
    return out


def display(accum):
    """Display-domain floats of an accumulator frame [N,4]: mean of the valid samples, gamma 1/2.2, clipped to [0,1]."""
    import numpy as np
    rgb = accum[:, :3].astype(np.float64) / np.maximum(accum[:, 3:4].astype(np.float64), 1.0)
    return np.clip(rgb, 0.0, None) ** (1.0 / 2.2)


def rel_mse(a, b, eps: float = 1e-2) -> float:
    """relMSE = mean((a - b)^2 / (b^2 + eps)) on display-domain floats (clipped to [0,1])."""
    import numpy as np
    a, b = np.minimum(display(a), 1.0), np.minimum(display(b), 1.0)
    return float(np.mean((a - b) ** 2 / (b * b + eps)))


def block_sums(accum, width: int, height: int, n: int = 16):
    """Accumulator frame [H*W, 4] summed over n x n pixel blocks (ragged edges dropped): the same estimator at 1 / n^2 of the variance."""
    h, w = height // n * n, width // n * n
    return accum.reshape(height, width, 4)[:h, :w].reshape(h // n, n, w // n, n, 4).sum(axis=(1, 3)).reshape(-1, 4)


def optin_leg(name: str, steps: int, exact_frame, exact_value: float, env: dict, note: str, noise: float | None) -> dict:
    """The same workload on an OPT-IN build that is not bit-exact by design, timed the same way and compared with the exact frame by
    north_star's tolerance (relMSE <= 1e-3, primary-hit ids >= 99.99 %) — what the bit-exact contract costs:
      PT_ARITH=fast  the path kernel with hardware rcp / sqrt / rsq / exp2 / log2 / sin / cos and FMA contraction in the shading half;
      PT_BVH=wide4   64-byte records of up to four children with 8-bit boxes instead of the reference's binary tree.
    In an open scene one grazing ray that hits in one build and escapes in the other shifts the pixel's RNG stream (Q2) and decorrelates
    the rest of its samples, so per pixel the frames differ by Monte-Carlo noise (`relmse_noise_other_seed`: the exact build against
    itself from another seed); the tolerance applies where that noise is averaged out (`relmse_block16_vs_exact`)."""
    import numpy as np
    from pathtracer_amd import abi, scenes
    os.environ.update(env)
    try:
        be = HipBackend(int(os.environ.get("LOCAL_RANK", "0")))  # the context reads its knobs at creation
    finally:
        for k in env:
            del os.environ[k]
    factory, kwargs, width, height, spp, _ = WORKLOADS[name]
    packed = be.pack(getattr(scenes, factory)(**kwargs))
    be.upload(packed)
    rp = packed.render_params(be.lib, width, height, spp, BG.get(name, (0.239, 0.239, 0.239)), SEED)
    be.alloc(width * height)
    stats = abi.Stats()
    t0 = time.perf_counter()
    be.render(rp, stats)  # warm-up
    be.sync()
    steps = max(1, min(steps, int(20.0 / max(time.perf_counter() - t0, 1e-3))))  # a side leg: about 20 s of frames at most
    t0 = time.perf_counter()
    rays = 0
    for _ in range(steps):
        be.render(rp, stats)
        be.resolve()
        rays += stats.rays
    be.sync()
    elapsed = time.perf_counter() - t0
    be.render(rp, stats, want_ids=True)
    accum, ids, _ = be.host_frame()
    be.close()
    e_accum, e_ids, _ = exact_frame
    value = rays / elapsed / 1e6
    return {"workload": name, "value": value, "unit": "Mrays/s", "ms_per_step": 1e3 * elapsed / steps, "steps": steps,
            "speedup_over_exact": value / exact_value, "relmse_vs_exact": rel_mse(accum, e_accum),
            "relmse_block16_vs_exact": rel_mse(block_sums(accum, width, height), block_sums(e_accum, width, height)),
            "relmse_noise_other_seed": noise, "relmse_tolerance": 1e-3,
            "primary_id_agreement": float((ids == e_ids).mean()), "nan_samples": int(stats.nan_samples),
            "node_visits": int(stats.node_visits), "rays": int(stats.rays),
            "in_effect": {"fast_arith": int(stats.fast_arith), "bvh_wide4": int(stats.bvh_wide4), "merged_walk": int(stats.merged_walk)},
            "note": note + "; the headline value above is the exact build, which is bit-identical to the oracle"}


def main(argv=None, backend_factory=None) -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=HEADLINE, choices=sorted(WORKLOADS))
    ap.add_argument("--split", default="strata", choices=("strata", "replicate"),
                    help="N > 1: the strata of every pixel divided over the ranks (north_star's partitioning; strong scaling of one frame) "
                         "or every rank all strata from its own seed (weak scaling)")
    ap.add_argument("--configs", default=EXTRA_DEFAULT,
                    help="N = 1: further single-GPU configs measured by the same command, comma separated ('none' to skip)")
    ap.add_argument("--spp", type=int, default=0, help="override sample_count (dev only; invalidates the headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true", help="skip the extra profiled step that feeds `roofline`")
    ap.add_argument("--no-fast-arith", action="store_true", help="skip the opt-in comparison legs (PT_ARITH=fast, PT_BVH=wide4; N = 1 only)")
    ap.add_argument("--time-budget-s", type=float, default=420.0, help="extra configs are skipped once the run is this old")
    args = ap.parse_args(argv)

    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        sys.exit(2)
    be = backend_factory(local_rank) if backend_factory else HipBackend(local_rank)
    # A one-GPU box cannot run N > 1; PT_BENCH_FORCE_COLLECTIVES=1 runs the same reduce / barrier / all-reduce calls on a
    # one-rank process group, so the RCCL leg of this script is executed by the GPU suite at least once.
    collectives = world > 1 or os.environ.get("PT_BENCH_FORCE_COLLECTIVES") == "1"
    if collectives:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        kw = {"device_id": be.dev} if be.dist_backend == "nccl" else {}
        dist.init_process_group(be.dist_backend, rank=rank, world_size=world, **kw)

    optin = (world == 1 and not args.no_fast_arith and backend_factory is None and not args.spp
             and os.environ.get("PT_ARITH") != "fast" and os.environ.get("PT_BVH") != "wide4")
    head = run_workload(be, args.workload, args.steps, args.warmup, world, rank, args.split, args.spp,
                        not args.no_kernel_timing, 0.0 if args.no_cpu_baseline else 12.0, dist, collectives, want_noise=optin)
    fast = wide = None
    if optin and head is not None and "_frame" in head:
        progress(f"{args.workload}: the opt-in tree (PT_BVH=wide4)")
        wide = optin_leg(args.workload, args.steps, head["_frame"], head["value"], {"PT_BVH": "wide4"},
                         "opt-in tree (PT_BVH=wide4: scenes read from global memory with at most two light samples per vertex; elsewhere it is the default walk)",
                         head.get("_noise"))
    extras = []
    if world == 1 and args.configs != "none":
        for name in [c for c in args.configs.split(",") if c and c != args.workload]:
            if time.perf_counter() - T_START > args.time_budget_s:
                extras.append({"name": name, "skipped": f"run older than --time-budget-s {args.time_budget_s:.0f}"})
                continue
            extras.append(run_workload(be, name, EXTRA_STEPS.get(name, 2), 0, 1, 0, "replicate", 0, not args.no_kernel_timing,
                                       0.0 if args.no_cpu_baseline else 4.0, dist))

    # PT_ARITH=fast applies to scenes staged in LDS (pt_render runs the exact kernels on scenes read from global memory, where the
    # fast compilation measured slower): its leg is timed on cfg2, against the exact cfg2 frame of the list above.
    cfg2 = next((e for e in extras + [head] if e and e.get("name") == FAST_ARITH_WORKLOAD and "_frame" in e), None)
    if optin and cfg2 is not None:
        progress(f"{FAST_ARITH_WORKLOAD}: the opt-in build (PT_ARITH=fast)")
        fast = optin_leg(FAST_ARITH_WORKLOAD, 8, cfg2["_frame"], cfg2["value"], {"PT_ARITH": "fast"},
                         "opt-in build (PT_ARITH=fast: scenes staged in LDS only; scenes read from global memory, the headline among them, run the exact kernels whatever PT_ARITH says)",
                         None)

    if head is not None:
        head.pop("_frame", None)
        head.pop("_noise", None)
    if rank == 0:
        cfg = dict(head["config"])
        cfg["parallelism"] = (f"dp{world}, {args.split}: " +
                              ("strata of every pixel divided over the GPUs" if args.split == "strata" else "full frame per GPU, own seed") +
                              ", RCCL sum-reduce of float4 accumulators")
        out = {
            "metric": "Mrays/s at 1080p (rays = ray_scene_intersect calls of the reference: closest-hit + shadow-walk steps)",
            "value": head["value"], "unit": "Mrays/s", "mpaths_per_s": head["mpaths_per_s"],
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"],
            "device_ms_per_step": head["device_ms_per_step"],
            "higher_is_better": True, "scaling": "strong" if (args.split == "strata" and world > 1) else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "config": cfg,
        }
        for k in ("counters", "ray_mix", "reduce_ms_per_step", "roofline", "kernels", "frame_verified", "frame_verified_against",
                  "frame_verified_rows", "frame_properties_ok", "reduced_frame_properties_ok", "cpu_baseline"):
            if k in head:
                out[k] = head[k]
        if fast:
            out["fast_arith"] = fast
        if wide:
            out["bvh_wide4"] = wide
        for e in extras:
            if e:
                e.pop("_frame", None)
        if extras:
            out["configs"] = extras
        print(json.dumps(out))
    if collectives:
        dist.barrier()
        dist.destroy_process_group()
    be.close()


if __name__ == "__main__":
    main()
