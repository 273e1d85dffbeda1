"""Python view of the C++ front end (csrc/host/gltf_loader.cpp, image_io.cpp): the
reference's `load_gltf(path) -> Scene` (src/gltf_loader/gltf_loader.h:5) and
`stbi_write_png` (src/main.cc:78-83) reached through the C ABI. The parsing is native;
this module only copies the borrowed arrays into a `Scene`."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import abi, lib as _lib
from .scene import Scene


class HostSceneArrays(C.Structure):
    _fields_ = [
        ("vertices", C.c_void_p), ("vertex_count", C.c_uint32),
        ("triangles", C.c_void_p), ("triangle_count", C.c_uint32),
        ("materials", C.c_void_p), ("material_count", C.c_uint32),
        ("images", C.c_void_p), ("image_count", C.c_uint32),
        ("samplers", C.c_void_p), ("sampler_count", C.c_uint32),
        ("textures", C.c_void_p), ("texture_count", C.c_uint32),
        ("directional_lights", C.c_void_p), ("directional_light_array_size", C.c_uint32),
        ("point_lights", C.c_void_p), ("point_light_array_size", C.c_uint32),
        ("camera", abi.Camera),
    ]


def _copy(ptr, count: int, dtype) -> np.ndarray:
    dtype = np.dtype(dtype)
    if not count:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (count * dtype.itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=count).copy()


def load_gltf(path: str) -> Scene:
    """load_gltf(path): .glb by extension, anything else as JSON .gltf (gltf_loader.cc:126-130)."""
    L = _lib.load()
    L.pt_host_scene_load.restype = C.c_void_p
    L.pt_host_scene_load.argtypes = [C.c_char_p]
    L.pt_host_scene_view.argtypes = [C.c_void_p, C.c_void_p]
    L.pt_host_scene_free.argtypes = [C.c_void_p]
    L.pt_host_last_error.restype = C.c_char_p
    hs = L.pt_host_scene_load(str(path).encode())
    if not hs:
        raise RuntimeError(L.pt_host_last_error().decode())
    try:
        a = HostSceneArrays()
        if L.pt_host_scene_view(hs, C.byref(a)) != 0:
            raise RuntimeError("pt_host_scene_view failed")
        images = []
        for im in _copy(a.images, a.image_count, np.dtype([("w", "<u4"), ("h", "<u4"), ("p", "<u8")])):
            images.append(_copy(int(im["p"]), int(im["w"]) * int(im["h"]) * 4, np.uint8).reshape(int(im["h"]), int(im["w"]), 4))
        s = Scene(
            vertices=_copy(a.vertices, a.vertex_count, abi.VERTEX),
            triangles=_copy(a.triangles, 3 * a.triangle_count, np.uint32).reshape(-1, 3),
            materials=_copy(a.materials, a.material_count, abi.MATERIAL),
            images=images,
            samplers=_copy(a.samplers, a.sampler_count, abi.SAMPLER),
            textures=_copy(a.textures, a.texture_count, abi.TEXTURE),
            directional_lights=_copy(a.directional_lights, a.directional_light_array_size, abi.DIRECTIONAL_LIGHT),
            point_lights=_copy(a.point_lights, a.point_light_array_size, abi.POINT_LIGHT),
        )
        s.camera_position = tuple(a.camera.position)
        s.camera_lookat = tuple(a.camera.lookat)
        s.camera_vfov = float(a.camera.vertical_fov)
        return s
    finally:
        L.pt_host_scene_free(hs)


def write_png(path: str, rgba: np.ndarray) -> None:
    L = _lib.load()
    img = np.ascontiguousarray(rgba, dtype=np.uint8)
    h, w = img.shape[:2]
    L.pt_write_png.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32]
    if L.pt_write_png(str(path).encode(), w, h, abi.ptr(img), w * 4) != 0:
        raise RuntimeError(f"pt_write_png({path}) failed")


def decode_image(data: bytes) -> np.ndarray:
    L = _lib.load()
    L.pt_decode_image.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    w, h = C.c_uint32(), C.c_uint32()
    if L.pt_decode_image(data, len(data), C.byref(w), C.byref(h), None, 0) != 0:
        raise RuntimeError("pt_decode_image failed")
    out = np.zeros((h.value, w.value, 4), dtype=np.uint8)
    if L.pt_decode_image(data, len(data), C.byref(w), C.byref(h), abi.ptr(out), out.nbytes) != 0:
        raise RuntimeError("pt_decode_image failed")
    return out
