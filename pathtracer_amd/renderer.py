"""Python mirror of the reference's `Renderer` (src/renderer/renderer.h:22-55) on top of
the C ABI: `Renderer(device_index)`, `Renderer.render(scene, config)`, `list_devices()`.
The compute runs in libpathtracer_amd.so (hand-written HIP); nothing here falls back to
a CPU path."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import abi, lib as _lib
from .scene import PackedScene, Scene


def list_devices() -> list[str]:
    """list_devices(), renderer.cc:196-213."""
    L = _lib.load()
    out = []
    buf = C.create_string_buffer(256)
    for i in range(L.pt_device_count()):
        if L.pt_device_name(i, buf, 256) == 0:
            out.append(f"{i}: {buf.value.decode()}")
    return out


class Renderer:
    """Renderer(device_index = -1): binds one GPU (renderer.h:24)."""

    def __init__(self, device_index: int = -1):
        self.lib = _lib.load()
        self.ctx = self.lib.pt_create(int(device_index))
        if not self.ctx:
            raise RuntimeError("pt_create failed: " + self.lib.pt_last_error(None).decode())
        self._uploaded = None

    def close(self) -> None:
        if getattr(self, "ctx", None):
            self.lib.pt_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, what: str) -> None:
        if rc != 0:
            raise RuntimeError(f"{what} failed ({rc}): {self.lib.pt_last_error(self.ctx).decode()}")

    def pack(self, scene: Scene) -> PackedScene:
        """RenderJob ctor, host half: build_bvh + process_scene (render_job.cc:25-31)."""
        return PackedScene(scene, self.lib)

    def upload(self, packed: PackedScene) -> None:
        self._check(self.lib.pt_upload_scene(self.ctx, C.byref(packed.desc)), "pt_upload_scene")
        self._uploaded = packed

    def render_params(self, rp: abi.RenderParams, want_accum=True, want_ids=True):
        """One dispatch: returns (rgba8 HxWx4, accum HxWx4 float32 | None, ids HxW | None, stats dict)."""
        w, h = rp.pc.resolution_x, rp.pc.resolution_y
        rgba = np.zeros((h, w, 4), dtype=np.uint8)
        accum = np.zeros((h, w, 4), dtype=np.float32) if want_accum else None
        ids = np.zeros((h, w), dtype=np.uint32) if want_ids else None
        st = abi.Stats()
        rc = self.lib.pt_render(self.ctx, C.byref(rp), abi.ptr(rgba), abi.ptr(accum), abi.ptr(ids), C.byref(st))
        self._check(rc, "pt_render")
        return rgba, accum, ids, st.as_dict()

    def render(self, scene: Scene | PackedScene, width: int, height: int, sample_count: int,
               bg_color=(0.239, 0.239, 0.239), seed: int = 1) -> np.ndarray:
        """Renderer::render(Scene&, RenderConfig) (renderer.cc:183-194): RGBA8 image."""
        packed = scene if isinstance(scene, PackedScene) else self.pack(scene)
        if self._uploaded is not packed:
            self.upload(packed)
        rp = packed.render_params(self.lib, width, height, sample_count, bg_color, seed)
        return self.render_params(rp, want_accum=False, want_ids=False)[0]
