"""World-size-2 (and 3) CPU test of the multi-GPU driver over gloo: the sample split covers
every stratum exactly once, the accumulator sum-reduce + resolve reproduces what a single
process computes from the same per-rank parts, and both split modes converge to the same
image statistically.  The per-rank renderer here is the CPU oracle (test infrastructure);
on GPUs the same driver runs the HIP path over RCCL (bench.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, SPP, SEED = 48, 32, 16, 77


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _setup():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    from pathtracer_amd import scenes
    from pathtracer_amd.scene import PackedScene
    lib = oracle_lib.load()
    packed = PackedScene(scenes.cornell_scene(with_point_light=True), lib, "orc_")

    def make_params(seed, rank, num_ranks):
        return packed.render_params(lib, W, H, SPP, seed=seed, rank=rank, num_ranks=num_ranks, prefix="orc_")

    def render_accum(rp):
        _, accum, _, _ = oracle_lib.render(lib, packed, rp, threads=2, want_ids=False)
        return torch.from_numpy(accum.reshape(-1).copy())

    def resolve(accum):
        import ctypes as C
        a = accum.numpy().reshape(-1, 4)
        out = np.zeros((a.shape[0], 4), dtype=np.uint8)
        lib.orc_resolve_pixel.argtypes = [C.c_float] * 4 + [C.c_void_p]
        for i in range(a.shape[0]):
            lib.orc_resolve_pixel(float(a[i, 0]), float(a[i, 1]), float(a[i, 2]), float(a[i, 3]), out[i].ctypes.data)
        return out

    return make_params, render_accum, resolve


def _worker(rank, world, port, mode, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    make_params, render_accum, resolve = _setup()
    from pathtracer_amd.distributed import render_distributed
    rgba, accum = render_distributed(make_params, render_accum, resolve, mode=mode, seed=SEED)
    if rank == 0:
        np.save(os.path.join(outdir, f"rgba_{mode}_{world}.npy"), rgba)
        np.save(os.path.join(outdir, f"accum_{mode}_{world}.npy"), accum.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _run(world, mode, outdir):
    mp.spawn(_worker, args=(world, _free_port(), mode, str(outdir)), nprocs=world, join=True)
    return (np.load(os.path.join(outdir, f"rgba_{mode}_{world}.npy")),
            np.load(os.path.join(outdir, f"accum_{mode}_{world}.npy")).reshape(-1, 4))


@pytest.mark.parametrize("world", [2, 3])
def test_strata_split_reduce_equals_sum_of_rank_parts(tmp_path, world):
    rgba, accum = _run(world, "strata", tmp_path)
    make_params, render_accum, resolve = _setup()
    parts = [render_accum(make_params(SEED, r, world)).numpy().reshape(-1, 4) for r in range(world)]
    expect = parts[0].copy()
    for p in parts[1:]:
        expect = expect + p  # gloo reduces rank 0 + 1 + ...; float32 addition order matters only in the last ulp
    assert np.allclose(accum, expect, rtol=1e-6, atol=0)
    assert (accum[:, 3] == 16).all()  # every stratum rendered exactly once, none dropped
    assert np.abs(rgba.astype(int) - resolve(torch.from_numpy(expect)).astype(int)).max() <= 1


def test_replicate_mode_accumulates_world_times_the_samples(tmp_path):
    rgba, accum = _run(2, "replicate", tmp_path)
    assert (accum[:, 3] == 32).all()
    # statistically the same image as one rank's 16 spp: compare means over the frame
    make_params, render_accum, _ = _setup()
    one = render_accum(make_params(SEED, 0, 1)).numpy().reshape(-1, 4)
    m2 = accum[:, :3].sum(0) / accum[:, 3].sum()
    m1 = one[:, :3].sum(0) / one[:, 3].sum()
    assert np.allclose(m1, m2, rtol=0.08)


def test_strata_ranges_partition():
    from pathtracer_amd.distributed import strata_range
    for total in (1, 9, 16, 121, 484, 4096):
        for world in (1, 2, 3, 4, 8):
            got = [strata_range(total, r, world) for r in range(world)]
            assert got[0][0] == 0 and got[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(got, got[1:]))
