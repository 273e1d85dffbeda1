"""pt_create_multi: one context over several GPUs of one process, strata split + one RCCL sum-reduce.
A one-GPU box can only run the one-rank group — the same code path (threads, ncclCommInitAll, ncclReduce, resolve on
rank 0) with R = 1, where the frame must equal the plain context's bit for bit; the R-rank arithmetic (per-rank strata and
streams, sum, resolve) is covered on one GPU by tests/test_gpu_device_entry.py and on CPU by tests/test_distributed_gloo.py."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

from pathtracer_amd import abi, gltf_export, frontend, lib, scenes
from pathtracer_amd.renderer import Renderer

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "pathtracer_amd", "pathtracer")


def test_one_rank_group_equals_the_plain_context():
    L = lib.load()
    ids = (C.c_int * 1)(0)
    ctx = L.pt_create_multi(ids, 1)
    assert ctx, L.pt_last_error(None)
    r = Renderer(0)
    try:
        packed = r.pack(scenes.cornell_scene(True, True, True, True))
        r.upload(packed)
        assert L.pt_upload_scene(ctx, C.byref(packed.desc)) == 0, L.pt_last_error(ctx)
        rp = packed.render_params(L, 160, 90, 16, seed=5)
        rgba, accum, ids_ref, st = r.render_params(rp)
        g_rgba = np.zeros_like(rgba); g_accum = np.zeros_like(accum); g_ids = np.zeros_like(ids_ref)
        gst = abi.Stats()
        rc = L.pt_render(ctx, C.byref(rp), abi.ptr(g_rgba), abi.ptr(g_accum), abi.ptr(g_ids), C.byref(gst))
        assert rc == 0, L.pt_last_error(ctx)
        assert (g_accum.view(np.uint32) == accum.view(np.uint32)).all() and (g_ids == ids_ref).all() and (g_rgba == rgba).all()
        assert gst.paths == st["paths"] and gst.rays == st["rays"] and gst.reduce_ms >= 0.0
        # a group splits the strata itself
        rp2 = packed.render_params(L, 160, 90, 16, seed=5, rank=1, num_ranks=2)
        assert L.pt_render(ctx, C.byref(rp2), abi.ptr(g_rgba), None, None, None) == -1
    finally:
        r.close()
        L.pt_destroy(ctx)


def test_create_multi_rejects_bad_device_lists():
    L = lib.load()
    assert not L.pt_create_multi(None, 1)
    ids = (C.c_int * 2)(0, 0)
    assert not L.pt_create_multi(ids, 2) and b"twice" in L.pt_last_error(None)
    ids = (C.c_int * 1)(99)
    assert not L.pt_create_multi(ids, 1)


def test_cli_num_gpus_key(tmp_path):
    glb = tmp_path / "cornell.glb"
    gltf_export.write_glb(scenes.cornell_scene(with_point_light=True), str(glb))
    images = {}
    for name, extra in (("single", {}), ("group", {"num_gpus": 1})):
        d = tmp_path / name
        d.mkdir()
        cfg = {"resolution_x": 96, "resolution_y": 54, "sample_count": 16, "bg_color": "#3D3D3D", "gltf_path": str(glb), "seed": 77,
               "stats": True, **extra}
        (d / "config.json").write_text(json.dumps(cfg))
        out = subprocess.run([CLI], capture_output=True, text=True, cwd=d, timeout=120)
        assert out.returncode == 0 and "Exception" not in out.stdout, out.stdout + out.stderr
        images[name] = frontend.decode_image((d / "cornell.png").read_bytes())
        assert "reduce_ms" in out.stdout
    assert np.array_equal(images["single"], images["group"])
