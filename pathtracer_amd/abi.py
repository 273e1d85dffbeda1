"""ctypes / numpy mirror of include/pathtracer_amd.h.

Plumbing only: the byte layouts of the C ABI (which are the reference's own host/shader
layouts, src/scene.h and src/renderer/render_job.h) as numpy dtypes and ctypes
structures, so Python callers (tests, bench.py, the multi-GPU driver) can hand data to
`libpathtracer_amd.so` without copies.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

# ---------------------------------------------------------------- numpy dtypes

VERTEX = np.dtype(
    [
        ("position", "<f4", 3),
        ("normal", "<f4", 3),
        ("tangent", "<f4", 3),
        ("bitangent", "<f4", 3),
        ("texcoord", "<f4", 2),
        ("material_index", "<i4"),
    ]
)  # Scene::Vertex, scene.h:9-16
assert VERTEX.itemsize == 60

MATERIAL = np.dtype(
    [
        ("base_color_factor", "<f4", 4),
        ("base_color_texture_index", "<i4"),
        ("metallic_factor", "<f4"),
        ("roughness_factor", "<f4"),
        ("metallic_roughness_texture_index", "<i4"),
        ("normal_scale", "<f4"),
        ("normal_texture_index", "<i4"),
        ("occlusion_strength", "<f4"),
        ("occlusion_texture_index", "<i4"),
        ("emissive_factor", "<f4", 3),
        ("emissive_texture_index", "<i4"),
    ]
)  # Scene::Material, scene.h:20-32
assert MATERIAL.itemsize == 64

DIRECTIONAL_LIGHT = np.dtype(
    [("color", "<f4", 3), ("intensity", "<f4"), ("direction", "<f4", 3), ("_pad", "<f4")]
)  # scene.h:57-61
POINT_LIGHT = np.dtype(
    [("color", "<f4", 3), ("intensity", "<f4"), ("position", "<f4", 3), ("_pad", "<f4")]
)  # scene.h:63-67
assert DIRECTIONAL_LIGHT.itemsize == 32 and POINT_LIGHT.itemsize == 32

SAMPLER = np.dtype(
    [("mag_filter", "<i4"), ("min_filter", "<i4"), ("wrap_s", "<i4"), ("wrap_t", "<i4")]
)  # scene.h:42-45
TEXTURE = np.dtype([("sampler_index", "<i4"), ("image_index", "<i4")])  # scene.h:52-55

BVH_NODE = np.dtype(
    [
        ("aabb_min", "<f4", 3),
        ("aabb_max", "<f4", 3),
        ("left_child_index", "<u4"),
        ("begin_index", "<u4"),
        ("triangle_count", "<u4"),
    ]
)  # bvh.h:36-41
assert BVH_NODE.itemsize == 36

PACKED_VERTEX = np.dtype(
    [
        ("position", "<f4", 4),
        ("normal_and_texcoord_u", "<f4", 4),
        ("tangent_and_texcoord_v", "<f4", 4),
        ("bitangent", "<f4", 3),
        ("material_index", "<i4"),
    ]
)  # render_job.h:7-13
assert PACKED_VERTEX.itemsize == 64

PACKED_BVH_NODE = np.dtype(
    [
        ("aabb_min", "<f4", 3),
        ("left_or_begin", "<u4"),
        ("aabb_max", "<f4", 3),
        ("triangle_count", "<u4"),
    ]
)  # render_job.h:15-20
assert PACKED_BVH_NODE.itemsize == 32

FILTER_NEAREST, FILTER_LINEAR = 0, 1
WRAP_CLAMP_TO_EDGE, WRAP_MIRRORED_REPEAT, WRAP_REPEAT = 0, 1, 2
NO_HIT = 0xFFFFFFFF

# ------------------------------------------------------------ ctypes structures


class Image(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("rgba8", C.c_void_p)]


class Camera(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("lookat", C.c_float * 3), ("vertical_fov", C.c_float)]


class PushConstants(C.Structure):
    _fields_ = [
        ("camera_position", C.c_float * 3), ("_p0", C.c_float),
        ("pixel_delta_u", C.c_float * 3), ("_p1", C.c_float),
        ("pixel_delta_v", C.c_float * 3), ("_p2", C.c_float),
        ("corner_pixel_pos", C.c_float * 3), ("_p3", C.c_float),
        ("resolution_x", C.c_uint32), ("resolution_y", C.c_uint32),
        ("seed", C.c_uint32), ("sample_count", C.c_uint32),
        ("bg_color", C.c_float * 3), ("emissive_triangle_count", C.c_uint32),
    ]


assert C.sizeof(PushConstants) == 96


class RenderConfig(C.Structure):
    _fields_ = [
        ("resolution_x", C.c_uint32), ("resolution_y", C.c_uint32),
        ("sample_count", C.c_uint32), ("bg_color", C.c_float * 3),
    ]


class SceneDesc(C.Structure):
    _fields_ = [
        ("triangle_count", C.c_uint32),
        ("vertex_positions", C.c_void_p),
        ("packed_vertex_data", C.c_void_p),
        ("bvh_nodes", C.c_void_p),
        ("bvh_node_count", C.c_uint32),
        ("bvh_max_depth", C.c_uint32),
        ("emissive_triangles", C.c_void_p),
        ("emissive_triangle_count", C.c_uint32),
        ("materials", C.c_void_p),
        ("material_count", C.c_uint32),
        ("directional_lights", C.c_void_p),
        ("directional_light_count", C.c_uint32),
        ("point_lights", C.c_void_p),
        ("point_light_count", C.c_uint32),
        ("textures", C.c_void_p),
        ("texture_count", C.c_uint32),
        ("images", C.c_void_p),
        ("image_count", C.c_uint32),
        ("samplers", C.c_void_p),
        ("sampler_count", C.c_uint32),
    ]


class RenderParams(C.Structure):
    _fields_ = [
        ("pc", PushConstants),
        ("rank", C.c_uint32),
        ("num_ranks", C.c_uint32),
        ("flags", C.c_uint32),
    ]


class KernelStats(C.Structure):
    _fields_ = [
        ("launches", C.c_uint64), ("rays", C.c_uint64), ("node_visits", C.c_uint64), ("tri_tests", C.c_uint64),
        ("hits_shaded", C.c_uint64), ("texture_taps", C.c_uint64), ("ms", C.c_double),
    ]

    def as_dict(self) -> dict:
        return {name: getattr(self, name) for name, _ in self._fields_}


class Stats(C.Structure):
    _fields_ = [
        ("paths", C.c_uint64), ("rays", C.c_uint64), ("node_visits", C.c_uint64),
        ("tri_tests", C.c_uint64), ("hits_shaded", C.c_uint64), ("texture_taps", C.c_uint64),
        ("shaded_vertices", C.c_uint64), ("nan_samples", C.c_uint64), ("iterations", C.c_uint64),
        ("render_ms", C.c_double), ("upload_ms", C.c_double), ("download_ms", C.c_double),
        ("extend", KernelStats), ("shade", KernelStats), ("shadow", KernelStats), ("path_kernel", KernelStats),
        ("rays_closest", C.c_uint64), ("rays_shadow_walked", C.c_uint64), ("rays_shadow_shortcut", C.c_uint64),
        ("reduce_ms", C.c_double), ("scene_in_lds", C.c_uint64), ("merged_walk", C.c_uint64), ("bvh_wide4", C.c_uint64), ("fast_arith", C.c_uint64),
    ]

    def as_dict(self) -> dict:
        d = {}
        for name, _ in self._fields_:
            v = getattr(self, name)
            d[name] = v.as_dict() if isinstance(v, KernelStats) else v
        return d


FLAG_TIME_KERNELS = 1


def ptr(a: np.ndarray | None) -> C.c_void_p:
    """Raw pointer of a C-contiguous numpy array (None -> NULL)."""
    if a is None:
        return C.c_void_p(None)
    assert a.flags["C_CONTIGUOUS"]
    return C.c_void_p(a.ctypes.data)
