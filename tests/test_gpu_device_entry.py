"""The multi-GPU halves of the C ABI — pt_render_accum_device / pt_resolve_device, the entry points bench.py times —
and the Python driver over them (pathtracer_amd/distributed.py), on one GPU: against pt_render and against the oracle."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib
from pathtracer_amd import abi, distributed, scenes
from pathtracer_amd.renderer import Renderer

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def renderer():
    r = Renderer(0)
    yield r
    r.close()


def test_render_distributed_world1_equals_pt_render_and_the_oracle(renderer, oracle):
    import torch
    torch.cuda.set_device(0)
    scene = scenes.cornell_scene(True, True, True, True)
    packed = renderer.pack(scene)
    renderer.upload(packed)
    w, h, spp = 160, 90, 16

    def make_params(seed, rank, num_ranks):
        return packed.render_params(renderer.lib, w, h, spp, seed=seed, rank=rank, num_ranks=num_ranks)

    for mode in ("strata", "replicate"):
        rgba_d, accum_d = distributed.render_distributed(make_params, distributed.hip_backend(renderer),
                                                         distributed.hip_resolve(renderer), mode=mode, seed=77)
        rp = make_params(77, 0, 1)  # world 1: both modes are the plain frame of seed 77
        rgba, accum, ids, _ = renderer.render_params(rp)
        accum_d = accum_d.cpu().numpy().reshape(h, w, 4)
        assert (accum_d.view(np.uint32) == accum.view(np.uint32)).all()
        assert (rgba_d.reshape(h, w, 4) == rgba).all()
        o_rgba, o_accum, _, _ = oracle_lib.render(oracle, packed, rp, threads=8)
        assert (accum_d.view(np.uint32) == o_accum.view(np.uint32)).all() and (rgba_d.reshape(h, w, 4) == o_rgba).all()
    assert distributed.render_stream(torch.device("cuda", 0)).cuda_stream != 0  # never the NULL stream (see the header)


def test_three_way_strata_split_summed_on_the_device_equals_the_oracle(renderer, oracle):
    """What an R-rank job computes, on one GPU: every rank's accumulator by pt_render_accum_device into caller-owned
    device memory on a caller-owned stream, summed on the device in rank order, resolved by pt_resolve_device."""
    import torch
    dev = torch.device("cuda", 0)
    scene = scenes.cornell_scene(with_directional_light=True)
    packed = renderer.pack(scene)
    renderer.upload(packed)
    w, h, spp, R = 96, 54, 16, 3
    stream = torch.cuda.Stream(dev)
    total = torch.zeros(w * h * 4, dtype=torch.float32, device=dev)
    o_total = np.zeros((h, w, 4), dtype=np.float32)
    with torch.cuda.stream(stream):
        for r in range(R):
            rp = packed.render_params(renderer.lib, w, h, spp, seed=91, rank=r, num_ranks=R)
            part = torch.full((w * h * 4,), float("nan"), dtype=torch.float32, device=dev)
            ids = torch.zeros(w * h, dtype=torch.int32, device=dev)
            st = abi.Stats()
            rc = renderer.lib.pt_render_accum_device(renderer.ctx, C.byref(rp), C.c_void_p(part.data_ptr()), C.c_void_p(ids.data_ptr()),
                                                     C.c_void_p(stream.cuda_stream), C.byref(st))
            renderer._check(rc, "pt_render_accum_device")
            total += part
            stream.synchronize()
            _, o_accum, o_ids, o_st = oracle_lib.render(oracle, packed, rp, threads=8, flags=oracle_lib.mirror_flags(st.as_dict()))
            assert (part.cpu().numpy().reshape(h, w, 4).view(np.uint32) == o_accum.view(np.uint32)).all(), r
            assert (ids.cpu().numpy().view(np.uint32).reshape(h, w) == o_ids).all()
            assert st.paths == o_st["paths"] and st.rays == o_st["rays"] and st.node_visits == o_st["node_visits"]
            assert st.rays_shadow_shortcut == o_st["rays_shadow_shortcut"] and st.rays_closest == o_st["rays_closest"]
            o_total += o_accum
        rgba = torch.zeros(w * h, dtype=torch.int32, device=dev)
        rc = renderer.lib.pt_resolve_device(renderer.ctx, C.c_void_p(total.data_ptr()), C.c_uint32(w * h), C.c_void_p(rgba.data_ptr()),
                                            C.c_void_p(stream.cuda_stream))
        renderer._check(rc, "pt_resolve_device")
    stream.synchronize()
    assert (total.cpu().numpy().reshape(h, w, 4).view(np.uint32) == o_total.view(np.uint32)).all()
    assert (o_total[..., 3] == 16).mean() > 0.99  # all 16 strata of (almost) every pixel: 5 + 5 + 6
    got = rgba.cpu().numpy().view(np.uint8).reshape(h, w, 4)
    px = (C.c_uint8 * 4)()
    for y in range(0, h, 3):
        for x in range(w):
            a = o_total[y, x]
            oracle.orc_resolve_pixel(C.c_float(a[0]), C.c_float(a[1]), C.c_float(a[2]), C.c_float(a[3]), px)
            assert tuple(got[y, x]) == tuple(px), (x, y)


def test_device_entry_points_reject_bad_arguments(renderer):
    packed = renderer.pack(scenes.cube_scene())
    renderer.upload(packed)
    rp = packed.render_params(renderer.lib, 16, 16, 4)
    assert renderer.lib.pt_render_accum_device(renderer.ctx, C.byref(rp), None, None, None, None) == -1
    assert renderer.lib.pt_resolve_device(renderer.ctx, None, C.c_uint32(16), None, None) == -1
