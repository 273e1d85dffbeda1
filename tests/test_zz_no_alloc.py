"""pt_render_accum_device_noalloc (the entry point that allocates nothing: its working set is a file-scope
device array of the library) against pt_render_accum_device, through the C ABI, bit for bit — on two calls
in a row, the second with a smaller frame, so stale contents of the shared array would show."""
import ctypes as C

import numpy as np
import pytest

from pathtracer_amd import abi, scenes
from pathtracer_amd.renderer import Renderer

pytestmark = pytest.mark.gpu

COUNTERS = ["paths", "rays", "node_visits", "tri_tests", "hits_shaded", "texture_taps", "shaded_vertices", "nan_samples",
            "rays_closest", "rays_shadow_walked", "rays_shadow_shortcut"]


def _render(r, entry, rp):
    import torch
    dev = torch.device("cuda", 0)
    pixels = rp.pc.resolution_x * rp.pc.resolution_y
    accum = torch.full((pixels * 4,), float("nan"), dtype=torch.float32, device=dev)
    ids = torch.full((pixels,), 12345, dtype=torch.int32, device=dev)
    st = abi.Stats()
    stream = torch.cuda.Stream(dev)
    torch.cuda.synchronize(dev)
    rc = getattr(r.lib, entry)(r.ctx, C.byref(rp), C.c_void_p(accum.data_ptr()), C.c_void_p(ids.data_ptr()),
                               C.c_void_p(stream.cuda_stream), C.byref(st))
    r._check(rc, entry)
    stream.synchronize()
    return accum.cpu().numpy().view(np.uint32), ids.cpu().numpy().view(np.uint32), st.as_dict()


def test_noalloc_entry_point_matches_the_allocating_one_twice_in_a_row():
    r = Renderer(0)
    try:
        # first an LDS-resident scene with every light kind and alpha, then (smaller frame) a scene read from global memory
        cases = [(scenes.cornell_scene(True, True, True, True), 200, 120, 16, 41),
                 (scenes.textured_scene(), 64, 40, 9, 42)]
        for scene, w, h, spp, seed in cases:
            packed = r.pack(scene)
            r.upload(packed)
            rp = packed.render_params(r.lib, w, h, spp, seed=seed)
            a_ref, i_ref, st_ref = _render(r, "pt_render_accum_device", rp)
            a, i, st = _render(r, "pt_render_accum_device_noalloc", rp)
            assert (a == a_ref).all() and (i == i_ref).all()
            # "allocates nothing", looked at from outside: with the outputs already allocated, the device's free memory is the same
            # before and after a call (hipMemGetInfo through torch; torch's own caching allocator is not involved in the call)
            import torch
            pixels = w * h
            accum = torch.empty(pixels * 4, dtype=torch.float32, device="cuda:0")
            torch.cuda.synchronize()
            free_before = torch.cuda.mem_get_info(0)[0]
            rc = r.lib.pt_render_accum_device_noalloc(r.ctx, C.byref(rp), C.c_void_p(accum.data_ptr()), None, None, None)
            r._check(rc, "pt_render_accum_device_noalloc")
            torch.cuda.synchronize()
            assert torch.cuda.mem_get_info(0)[0] == free_before
            assert (accum.cpu().numpy().view(np.uint32) == a_ref).all()
            for k in COUNTERS:
                assert st[k] == st_ref[k], k
            assert st["rays"] == st["rays_closest"] + st["rays_shadow_walked"] + st["rays_shadow_shortcut"]
    finally:
        r.close()


def test_noalloc_entry_point_rejects_what_it_cannot_hold():
    r = Renderer(0)
    try:
        packed = r.pack(scenes.cube_scene())
        r.upload(packed)
        import torch
        accum = torch.empty(520 * 512 * 4, dtype=torch.float32, device="cuda:0")
        rp = packed.render_params(r.lib, 520, 512, 4)  # 65 x 64 tiles = 266 240 slots > 262 144
        rc = r.lib.pt_render_accum_device_noalloc(r.ctx, C.byref(rp), C.c_void_p(accum.data_ptr()), None, None, None)
        assert rc == -4 and b"262144" in r.lib.pt_last_error(r.ctx)
        rp = packed.render_params(r.lib, 512, 512, 4, flags=abi.FLAG_TIME_KERNELS)
        rc = r.lib.pt_render_accum_device_noalloc(r.ctx, C.byref(rp), C.c_void_p(accum.data_ptr()), None, None, None)
        assert rc == -1
        rp = packed.render_params(r.lib, 512, 512, 4)  # exactly the limit: BASELINE configs[0]'s frame
        st = abi.Stats()
        rc = r.lib.pt_render_accum_device_noalloc(r.ctx, C.byref(rp), C.c_void_p(accum.data_ptr()), None, None, C.byref(st))
        r._check(rc, "noalloc")
        assert st.paths == 512 * 512 * 4
    finally:
        r.close()
