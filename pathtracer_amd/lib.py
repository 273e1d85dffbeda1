"""Loader for the C-ABI library. Fails loudly if the HIP extension has not been built —
there is no CPU fallback in the product path."""
from __future__ import annotations

import ctypes as C
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PT_LIB_PATH", os.path.join(_HERE, "libpathtracer_amd.so"))  # PT_LIB_PATH: experiment builds
_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m pathtracer_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        # PyTorch's ROCm wheels bundle their own libamdhip64; this library links the system one. Whichever is loaded
        # first serves both (same SONAME), but with the system runtime first torch then finds "No HIP GPUs". Callers
        # that use torch for device buffers / torch.distributed next to this library (bench.py, distributed.py, the
        # tests) therefore get torch's runtime loaded first, in whatever order they import things. Without torch
        # (the C++ CLI, plain ctypes users) nothing changes.
        if "torch" not in sys.modules and not os.environ.get("PT_NO_TORCH_PRELOAD"):
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        lib = C.CDLL(LIB_PATH)
        lib.pt_create.restype = C.c_void_p
        lib.pt_create.argtypes = [C.c_int]
        lib.pt_create_multi.restype = C.c_void_p
        lib.pt_create_multi.argtypes = [C.c_void_p, C.c_int]
        lib.pt_destroy.argtypes = [C.c_void_p]
        lib.pt_last_error.restype = C.c_char_p
        lib.pt_last_error.argtypes = [C.c_void_p]
        lib.pt_device_count.restype = C.c_int
        lib.pt_device_name.argtypes = [C.c_int, C.c_char_p, C.c_size_t]
        lib.pt_upload_scene.argtypes = [C.c_void_p, C.c_void_p]
        lib.pt_render.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.pt_render_accum_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.pt_render_accum_device_noalloc.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.pt_resolve_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        for name in ("pt_upload_scene", "pt_render", "pt_render_accum_device", "pt_render_accum_device_noalloc", "pt_resolve_device",
                     "pt_build_bvh", "pt_build_bvh_mt", "pt_process_scene", "pt_make_push_constants", "pt_build_wide4"):
            getattr(lib, name).restype = C.c_int
        _lib = lib
    return _lib
