"""Multi-GPU driver: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" on CPU for tests).

The path shards by SAMPLES: the scene is replicated in every GPU's HBM,
every rank renders all pixels, and the only exchange step is one sum-reduce of the float4
accumulators (sum R, sum G, sum B, valid-sample count; 16 B per pixel) to rank 0, which then
runs main.comp's resolve (divide, gamma, rgba8).  Two ways to split:

  "strata"    rank r renders strata [r*n*n/R, (r+1)*n*n/R) of the n x n grid (strong
              scaling of one frame; cfg5: 4096 spp over 8 GPUs), from a per-rank stream
              derived from (pixel, seed, rank).
  "replicate" every rank renders all n x n strata from its own seed (weak scaling: R times
              the samples per pixel); this is what bench.py measures.

`render_accum` is injected: the HIP path on a GPU (`hip_backend`), anything producing the
same float4 accumulator elsewhere (the tests drive it with the CPU oracle over gloo).
"""
from __future__ import annotations

import ctypes as C
from typing import Callable

import numpy as np
import torch
import torch.distributed as dist

from . import abi

RANK_SEED_STRIDE = 0x9E3779B9


def rank_seed(seed: int, rank: int) -> int:
    return (seed + RANK_SEED_STRIDE * rank) & 0xFFFFFFFF


def strata_range(total: int, rank: int, world: int) -> tuple[int, int]:
    """The strata a rank renders in "strata" mode (same formula as the kernels)."""
    return (rank * total) // world, ((rank + 1) * total) // world


_STREAMS: dict = {}


def render_stream(device):
    """The ONE explicit, non-default stream per device on which render, reduce and resolve are ordered.
    The library's NULL stream is its own non-blocking stream (include/pathtracer_amd.h), unordered against
    torch's streams and against the collective; torch's default stream has handle 0 == NULL, so it must
    never be the one passed."""
    key = (device.type, device.index)
    if key not in _STREAMS:
        _STREAMS[key] = torch.cuda.Stream(device)
    return _STREAMS[key]


def hip_backend(renderer) -> Callable:
    """render_accum(rp) -> float32 CUDA tensor [H*W*4] using the HIP path on renderer's GPU."""
    def render(rp: abi.RenderParams) -> torch.Tensor:
        dev = torch.device("cuda", torch.cuda.current_device())
        pixels = rp.pc.resolution_x * rp.pc.resolution_y
        stream = render_stream(dev)
        with torch.cuda.stream(stream):  # allocation, render (and the caller's reduce, see hip_reduce) on this stream
            accum = torch.empty(pixels * 4, dtype=torch.float32, device=dev)
            assert stream.cuda_stream != 0
            rc = renderer.lib.pt_render_accum_device(renderer.ctx, C.byref(rp), C.c_void_p(accum.data_ptr()), None,
                                                     C.c_void_p(stream.cuda_stream), None)
        renderer._check(rc, "pt_render_accum_device")
        return accum
    return render


def hip_resolve(renderer) -> Callable:
    def resolve(accum: torch.Tensor) -> np.ndarray:
        pixels = accum.numel() // 4
        stream = render_stream(accum.device)
        with torch.cuda.stream(stream):
            rgba = torch.empty(pixels, dtype=torch.int32, device=accum.device)
            rc = renderer.lib.pt_resolve_device(renderer.ctx, C.c_void_p(accum.data_ptr()), C.c_uint32(pixels),
                                                C.c_void_p(rgba.data_ptr()), C.c_void_p(stream.cuda_stream))
        renderer._check(rc, "pt_resolve_device")
        stream.synchronize()
        return rgba.cpu().numpy().view(np.uint8).reshape(-1, 4)
    return resolve


def reduce_to_rank0(accum: torch.Tensor) -> None:
    """The path's one exchange step. For a CUDA tensor the collective is issued with the render stream current, so
    RCCL orders it after the render and the resolve (same stream) after it."""
    if accum.is_cuda:
        with torch.cuda.stream(render_stream(accum.device)):
            dist.reduce(accum, dst=0, op=dist.ReduceOp.SUM)
    else:
        dist.reduce(accum, dst=0, op=dist.ReduceOp.SUM)


def render_distributed(make_params: Callable[[int, int, int], abi.RenderParams], render_accum: Callable,
                       resolve: Callable, mode: str = "strata", seed: int = 1):
    """Run this rank's share, sum-reduce to rank 0, resolve there.

    make_params(seed, rank, num_ranks) builds the pt_render_params of a rank (push constants
    through the C ABI).  Returns (rgba8 [H*W,4] uint8, accum tensor) on rank 0, (None, None)
    elsewhere.  With world size 1 it degenerates to a plain render (no collective)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if mode == "strata":
        rp = make_params(seed, rank, world)
    elif mode == "replicate":
        rp = make_params(rank_seed(seed, rank), 0, 1)
    else:
        raise ValueError(mode)
    accum = render_accum(rp)
    if world > 1:
        reduce_to_rank0(accum)
    if rank != 0:
        return None, None
    return resolve(accum), accum
