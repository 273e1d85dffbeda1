"""Host-side scene container (the reference's `struct Scene`, src/scene.h:8-85) and the
pre-render pipeline of `RenderJob`'s constructor (src/renderer/render_job.cc:20-61):
build_bvh -> process_scene -> descriptor arrays, all through the C ABI.

The heavy lifting is native (pt_build_bvh / pt_process_scene in libpathtracer_amd.so);
this module only owns the numpy buffers and fills `pt_scene_desc`.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass, field

import numpy as np

from . import abi


def default_material() -> np.ndarray:
    """gltf_loader.cc:12-22 — material 0 of every scene (Q17)."""
    m = np.zeros(1, dtype=abi.MATERIAL)
    m["base_color_factor"] = (0.8, 0.8, 0.8, 1.0)
    m["base_color_texture_index"] = -1
    m["metallic_factor"] = 0.0
    m["roughness_factor"] = 1.0
    m["metallic_roughness_texture_index"] = -1
    m["normal_texture_index"] = 0  # zero-initialised by the designated initialiser
    m["occlusion_strength"] = 0.0
    m["occlusion_texture_index"] = -1
    m["emissive_texture_index"] = -1
    return m


def make_material(base_color=(0.8, 0.8, 0.8, 1.0), metallic=0.0, roughness=1.0, emissive=(0, 0, 0),
                  base_color_texture=-1, metallic_roughness_texture=-1, occlusion_texture=-1,
                  occlusion_strength=1.0, emissive_texture=-1) -> np.ndarray:
    """A material as process_material would produce it (gltf_loader.cc:234-277)."""
    m = np.zeros(1, dtype=abi.MATERIAL)
    m["base_color_factor"] = base_color
    m["base_color_texture_index"] = base_color_texture
    m["metallic_factor"] = metallic
    m["roughness_factor"] = roughness
    m["metallic_roughness_texture_index"] = metallic_roughness_texture
    m["normal_scale"] = 1.0
    m["normal_texture_index"] = -1
    m["occlusion_strength"] = occlusion_strength
    m["occlusion_texture_index"] = occlusion_texture
    m["emissive_factor"] = emissive
    m["emissive_texture_index"] = emissive_texture
    return m


def default_image() -> np.ndarray:
    """gltf_loader.cc:31-53 — image 0: 64x64 black/white checker, alpha 255."""
    size = 64
    img = np.full((size, size, 4), 255, dtype=np.uint8)
    i, j = np.meshgrid(np.arange(size), np.arange(size), indexing="ij")
    img[(i + j) % 2 == 0, :3] = 0
    return img


def default_sampler() -> np.ndarray:
    """gltf_loader.cc:24-29 — sampler 0: linear / repeat."""
    s = np.zeros(1, dtype=abi.SAMPLER)
    s["mag_filter"] = abi.FILTER_LINEAR
    s["min_filter"] = abi.FILTER_LINEAR
    s["wrap_s"] = abi.WRAP_REPEAT
    s["wrap_t"] = abi.WRAP_REPEAT
    return s


@dataclass
class Scene:
    """Mirror of `struct Scene` (scene.h:75-84) with the loader's index conventions (Q17)
    already applied: material 0 / image 0 / sampler 0 are the defaults, one dummy light
    is appended at the end of each light array, at least one texture exists."""

    vertices: np.ndarray  # abi.VERTEX
    triangles: np.ndarray  # (N, 3) uint32
    materials: np.ndarray = field(default_factory=default_material)
    images: list = field(default_factory=lambda: [default_image()])
    samplers: np.ndarray = field(default_factory=default_sampler)
    textures: np.ndarray = field(default_factory=lambda: np.zeros(1, dtype=abi.TEXTURE))
    directional_lights: np.ndarray = field(default_factory=lambda: np.zeros(1, dtype=abi.DIRECTIONAL_LIGHT))
    point_lights: np.ndarray = field(default_factory=lambda: np.zeros(1, dtype=abi.POINT_LIGHT))
    camera_position: tuple = (0.0, 0.0, 0.0)
    camera_lookat: tuple = (0.0, 0.0, -1.0)
    camera_vfov: float = math.radians(45.0)

    def auto_camera(self) -> None:
        """gltf_loader.cc:457-471: no glTF camera -> 1.75 x the max corner, looking at the origin."""
        p = self.vertices["position"].max(axis=0).astype(np.float32) * np.float32(1.75)
        neg = (-p).astype(np.float32)
        inv = np.float32(1.0) / np.sqrt(np.float32(neg[0] * neg[0] + neg[1] * neg[1]) + np.float32(neg[2] * neg[2]))
        self.camera_position = tuple(float(x) for x in p)
        self.camera_lookat = tuple(float(np.float32(x * inv)) for x in neg)
        self.camera_vfov = float(np.float32(math.radians(45.0)))

    def camera(self) -> abi.Camera:
        cam = abi.Camera()
        cam.position[:] = self.camera_position
        cam.lookat[:] = self.camera_lookat
        cam.vertical_fov = self.camera_vfov
        return cam


class PackedScene:
    """What RenderJob holds after its constructor: BVH nodes, the four storage arrays,
    the three uniform arrays and the texture tables, plus the filled `pt_scene_desc`."""

    def __init__(self, scene: Scene, lib, prefix: str = "pt_", bin_count: int = 16):
        self.scene = scene
        n_tri = int(scene.triangles.shape[0])
        self.triangles = np.ascontiguousarray(scene.triangles, dtype=np.uint32).copy()  # permuted by the builder
        vertices = np.ascontiguousarray(scene.vertices)
        self.vertices = vertices
        nodes = np.zeros(2 * n_tri, dtype=abi.BVH_NODE)
        node_count = C.c_uint32(0)
        depth = C.c_uint32(0)
        build = getattr(lib, prefix + "build_bvh")
        build.restype = C.c_int
        rc = build(abi.ptr(vertices), C.c_uint32(len(vertices)), abi.ptr(self.triangles), C.c_uint32(n_tri),
                   C.c_uint32(bin_count), abi.ptr(nodes), C.byref(node_count), C.byref(depth))
        if rc != 0:
            raise RuntimeError(f"{prefix}build_bvh failed: {rc}")
        self.nodes = nodes[: node_count.value].copy()
        self.bvh_max_depth = depth.value

        self.vertex_positions = np.zeros((n_tri * 3, 4), dtype=np.float32)
        self.packed_vertex_data = np.zeros(n_tri * 3, dtype=abi.PACKED_VERTEX)
        emissive = np.zeros(max(n_tri, 1), dtype=np.uint32)
        emissive_count = C.c_uint32(0)
        self.packed_nodes = np.zeros(len(self.nodes), dtype=abi.PACKED_BVH_NODE)
        self.materials = np.ascontiguousarray(scene.materials)
        process = getattr(lib, prefix + "process_scene")
        process.restype = C.c_int
        rc = process(abi.ptr(vertices), C.c_uint32(len(vertices)), abi.ptr(self.triangles), C.c_uint32(n_tri),
                     abi.ptr(self.materials), C.c_uint32(len(self.materials)), abi.ptr(self.nodes),
                     C.c_uint32(len(self.nodes)), abi.ptr(self.vertex_positions), abi.ptr(self.packed_vertex_data),
                     abi.ptr(emissive), C.byref(emissive_count), abi.ptr(self.packed_nodes))
        if rc != 0:
            raise RuntimeError(f"{prefix}process_scene failed: {rc}")
        self.emissive = emissive[: emissive_count.value].copy()

        self.directional_lights = np.ascontiguousarray(scene.directional_lights)
        self.point_lights = np.ascontiguousarray(scene.point_lights)
        self.textures = np.ascontiguousarray(scene.textures)
        self.samplers = np.ascontiguousarray(scene.samplers)
        self._image_arrays = [np.ascontiguousarray(img, dtype=np.uint8) for img in scene.images]
        self._images = (abi.Image * len(self._image_arrays))()
        for k, img in enumerate(self._image_arrays):
            self._images[k].height, self._images[k].width = img.shape[0], img.shape[1]
            self._images[k].rgba8 = img.ctypes.data

        d = abi.SceneDesc()
        d.triangle_count = n_tri
        d.vertex_positions = self.vertex_positions.ctypes.data
        d.packed_vertex_data = self.packed_vertex_data.ctypes.data
        d.bvh_nodes = self.packed_nodes.ctypes.data
        d.bvh_node_count = len(self.packed_nodes)
        d.bvh_max_depth = self.bvh_max_depth
        d.emissive_triangles = self.emissive.ctypes.data if len(self.emissive) else None
        d.emissive_triangle_count = len(self.emissive)
        d.materials = self.materials.ctypes.data
        d.material_count = len(self.materials)
        d.directional_lights = self.directional_lights.ctypes.data
        d.directional_light_count = len(self.directional_lights) - 1  # render_job.cc:567
        d.point_lights = self.point_lights.ctypes.data
        d.point_light_count = len(self.point_lights) - 1  # render_job.cc:568
        d.textures = self.textures.ctypes.data
        d.texture_count = len(self.textures)
        d.images = C.cast(self._images, C.c_void_p)
        d.image_count = len(self._image_arrays)
        d.samplers = self.samplers.ctypes.data
        d.sampler_count = len(self.samplers)
        self.desc = d

    def render_params(self, lib, width: int, height: int, sample_count: int, bg_color=(0.239, 0.239, 0.239),
                      seed: int = 1, rank: int = 0, num_ranks: int = 1, flags: int = 0,
                      prefix: str = "pt_") -> abi.RenderParams:
        """PushConstants ctor (render_job.cc:836-873) through the C ABI, seed explicit."""
        cfg = abi.RenderConfig()
        cfg.resolution_x, cfg.resolution_y, cfg.sample_count = width, height, sample_count
        cfg.bg_color[:] = bg_color
        cam = self.scene.camera()
        rp = abi.RenderParams()
        fn = getattr(lib, prefix + "make_push_constants")
        fn.restype = C.c_int
        rc = fn(C.byref(cfg), C.byref(cam), C.c_uint32(len(self.emissive)), C.c_uint32(seed), C.byref(rp.pc))
        if rc != 0:
            raise RuntimeError(f"{prefix}make_push_constants failed: {rc}")
        rp.rank, rp.num_ranks, rp.flags = rank, num_ranks, flags
        return rp
