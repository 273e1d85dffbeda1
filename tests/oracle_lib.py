"""Loader for the CPU oracle (oracle/_build/libpt_oracle.so). Test infrastructure only."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(ROOT, "oracle", "_build", "libpt_oracle.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        # (every file the library is compiled from: the oracle shares pt_stats and the arithmetic header with the product, and a
        # library built against another pt_stats writes past the caller's struct)
        deps = [os.path.join(ROOT, "oracle", "pt_oracle.c"), os.path.join(ROOT, "oracle", "pt_oracle.h"), os.path.join(ROOT, "oracle", "Makefile"),
                os.path.join(ROOT, "include", "pathtracer_amd.h"), os.path.join(ROOT, "include", "pt_glsl_math.h")]
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(d) for d in deps if os.path.exists(d)):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
        _lib = C.CDLL(_SO)
        _lib.orc_triple32.restype = C.c_uint32
        _lib.orc_pow.restype = C.c_float
        _lib.orc_pow.argtypes = [C.c_float, C.c_float]
        _lib.orc_exp2.restype = C.c_float
        _lib.orc_exp2.argtypes = [C.c_float]
        _lib.orc_log2.restype = C.c_float
        _lib.orc_log2.argtypes = [C.c_float]
        _lib.orc_sincos.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        _lib.orc_render.restype = C.c_int
    return _lib


ANYHIT_OPAQUE_SHADOW = 1
CULL_BEHIND = 2
SELF_FIRST = 4
ANYHIT_SPAN_ORDER = 8
OPAQUE_TRIANGLES = 16
TRANSPARENT_PROBE = 64  # model / mirror of the round-3 experiment kernels that cleared shadow queries of a not-all-opaque scene by a probe of its maybe-transparent triangles (exact; not shipped)
STALE_SKIP = 32  # model / mirror of the round-3 experiment kernel that skipped stale inner stack entries (exact; node_visits shrinks; not shipped)
# the checker modes that reproduce the HIP path's work counters (identical images by construction, tests/test_oracle_shader.py):
GPU_MIRROR = ANYHIT_OPAQUE_SHADOW | CULL_BEHIND | SELF_FIRST | ANYHIT_SPAN_ORDER | OPAQUE_TRIANGLES  # scenes read from global memory
GPU_MIRROR_LDS = ANYHIT_OPAQUE_SHADOW | CULL_BEHIND | OPAQUE_TRIANGLES  # scenes staged in LDS: no own-triangle-first test, any-hit walks nearer-first


def mirror_flags(st):
    """Checker flags for a render whose pt_stats the HIP path reported (the stats dict; a bare scene_in_lds value is accepted too)."""
    if not isinstance(st, dict):
        return GPU_MIRROR_LDS if st else GPU_MIRROR
    return GPU_MIRROR_LDS if st["scene_in_lds"] else GPU_MIRROR


def render(lib, packed, rp, threads=8, rows=None, want_ids=True, flags=0):
    """orc_render over full image buffers; returns (rgba8, accum, ids, stats dict)."""
    import numpy as np
    from pathtracer_amd import abi
    w, h = rp.pc.resolution_x, rp.pc.resolution_y
    rgba = np.zeros((h, w, 4), dtype=np.uint8)
    accum = np.zeros((h, w, 4), dtype=np.float32)
    ids = np.full((h, w), 0xFFFFFFFF, dtype=np.uint32)
    st = abi.Stats()
    y0, y1 = rows if rows else (0, h)
    rc = lib.orc_render_ex(C.byref(packed.desc), C.byref(rp), abi.ptr(rgba), abi.ptr(accum),
                           abi.ptr(ids) if want_ids else None, C.byref(st), C.c_int(threads),
                           C.c_uint32(y0), C.c_uint32(y1), C.c_uint32(flags))
    assert rc == 0
    return rgba, accum, ids, st.as_dict()
