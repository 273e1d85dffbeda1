"""Seeded synthetic scenes for the BASELINE.json configurations.

The reference ships no assets (no cube.glb, no Cornell box, no Sponza), so every
configuration's scene is generated here, deterministically, in the `Scene` layout the
reference's glTF loader would have produced (index conventions Q17, auto-camera Q18).
"""
from __future__ import annotations

import math

import numpy as np

from . import abi
from .scene import Scene, default_image, default_material, default_sampler, make_material


def _verts(pos, nrm, uv, mat) -> np.ndarray:
    pos = np.asarray(pos, dtype=np.float32)
    v = np.zeros(len(pos), dtype=abi.VERTEX)
    v["position"] = pos
    v["normal"] = np.asarray(nrm, dtype=np.float32)
    v["texcoord"] = np.asarray(uv, dtype=np.float32)
    v["material_index"] = np.asarray(mat, dtype=np.int32)
    return v


def _quad(p0, p1, p2, p3, normal, mat, uv_scale=1.0):
    """Two triangles (0,1,2)(0,2,3) over a planar quad; per-face normal."""
    pos = [p0, p1, p2, p3]
    nrm = [normal] * 4
    uv = [(0, 0), (uv_scale, 0), (uv_scale, uv_scale), (0, uv_scale)]
    return pos, nrm, uv, [mat] * 4, [(0, 1, 2), (0, 2, 3)]


class _Mesh:
    def __init__(self):
        self.pos, self.nrm, self.uv, self.mat, self.tri = [], [], [], [], []

    def add(self, pos, nrm, uv, mat, tri):
        base = len(self.pos)
        self.pos += list(pos)
        self.nrm += list(nrm)
        self.uv += list(uv)
        self.mat += list(mat)
        self.tri += [(a + base, b + base, c + base) for a, b, c in tri]

    def add_box(self, lo, hi, mat, rot_y=0.0, inward=False):
        lo = np.asarray(lo, dtype=np.float64)
        hi = np.asarray(hi, dtype=np.float64)
        c = (lo + hi) / 2
        h = (hi - lo) / 2
        cs, sn = math.cos(rot_y), math.sin(rot_y)

        def xf(p):
            x, y, z = p
            return (c[0] + cs * x + sn * z, c[1] + y, c[2] - sn * x + cs * z)

        def xn(n):
            x, y, z = n
            return (cs * x + sn * z, y, -sn * x + cs * z)

        faces = [
            ((+1, 0, 0), [(h[0], -h[1], -h[2]), (h[0], h[1], -h[2]), (h[0], h[1], h[2]), (h[0], -h[1], h[2])]),
            ((-1, 0, 0), [(-h[0], -h[1], h[2]), (-h[0], h[1], h[2]), (-h[0], h[1], -h[2]), (-h[0], -h[1], -h[2])]),
            ((0, +1, 0), [(-h[0], h[1], -h[2]), (-h[0], h[1], h[2]), (h[0], h[1], h[2]), (h[0], h[1], -h[2])]),
            ((0, -1, 0), [(-h[0], -h[1], h[2]), (-h[0], -h[1], -h[2]), (h[0], -h[1], -h[2]), (h[0], -h[1], h[2])]),
            ((0, 0, +1), [(h[0], -h[1], h[2]), (h[0], h[1], h[2]), (-h[0], h[1], h[2]), (-h[0], -h[1], h[2])]),
            ((0, 0, -1), [(-h[0], -h[1], -h[2]), (-h[0], h[1], -h[2]), (h[0], h[1], -h[2]), (h[0], -h[1], -h[2])]),
        ]
        for n, corners in faces:
            nn = xn(n)
            if inward:
                nn = tuple(-x for x in nn)
            self.add(*_quad(*[xf(p) for p in corners], nn, mat))

    def scene(self, **kw) -> Scene:
        return Scene(vertices=_verts(self.pos, self.nrm, self.uv, self.mat),
                     triangles=np.asarray(self.tri, dtype=np.uint32), **kw)


def cube_scene() -> Scene:
    """cfg1: the README's cube.glb stand-in — 24 vertices / 12 triangles, per-face normals,
    default material (index 0), no lights, no camera (auto-camera, Q18)."""
    m = _Mesh()
    m.add_box((-1, -1, -1), (1, 1, 1), 0)
    s = m.scene()
    s.auto_camera()
    return s


def kat_cube() -> Scene:
    """SURVEY.md §8c first known-answer vector: 8 shared vertices, 12 triangles in the
    stated order."""
    pos = [((1 if i & 1 else -1), (1 if i & 2 else -1), (1 if i & 4 else -1)) for i in range(8)]
    tris = [(0, 1, 3), (0, 3, 2), (4, 6, 7), (4, 7, 5), (0, 4, 5), (0, 5, 1),
            (2, 3, 7), (2, 7, 6), (0, 2, 6), (0, 6, 4), (1, 5, 7), (1, 7, 3)]
    v = _verts(pos, [(0, 0, 1)] * 8, [(0, 0)] * 8, [0] * 8)
    return Scene(vertices=v, triangles=np.asarray(tris, dtype=np.uint32))


def cornell_scene(with_point_light: bool = False, with_directional_light: bool = False,
                  glass_pane: bool = False, metal_box: bool = False) -> Scene:
    """cfg2: Cornell box — floor, ceiling, back wall, red/green side walls (open front),
    a short and a tall box, and an emissive ceiling quad: 5*2 + 2*12 + 2 = 36 triangles.
    Materials: 0 default, 1 white, 2 red, 3 green, 4 light (emissive 17,12,4), 5 metal,
    6 glass (alpha 0.4, only with glass_pane)."""
    mats = np.concatenate([
        default_material(),
        make_material((0.73, 0.73, 0.73, 1.0)),
        make_material((0.65, 0.05, 0.05, 1.0)),
        make_material((0.12, 0.45, 0.15, 1.0)),
        make_material((0.0, 0.0, 0.0, 1.0), emissive=(17.0, 12.0, 4.0)),
        make_material((0.9, 0.85, 0.6, 1.0), metallic=1.0, roughness=0.25),
    ] + ([make_material((0.6, 0.8, 0.9, 0.4), metallic=0.0, roughness=0.1)] if glass_pane else []))
    m = _Mesh()
    W, WHITE, RED, GREEN, LIGHT, METAL, GLASS = 1.0, 1, 2, 3, 4, 5, 6
    # room spans [-1,1]^3, open toward +z (camera side)
    m.add(*_quad((-W, -W, W), (W, -W, W), (W, -W, -W), (-W, -W, -W), (0, 1, 0), WHITE))      # floor
    m.add(*_quad((-W, W, -W), (W, W, -W), (W, W, W), (-W, W, W), (0, -1, 0), WHITE))         # ceiling
    m.add(*_quad((-W, -W, -W), (W, -W, -W), (W, W, -W), (-W, W, -W), (0, 0, 1), WHITE))      # back
    m.add(*_quad((-W, -W, W), (-W, -W, -W), (-W, W, -W), (-W, W, W), (1, 0, 0), RED))        # left
    m.add(*_quad((W, -W, -W), (W, -W, W), (W, W, W), (W, W, -W), (-1, 0, 0), GREEN))         # right
    m.add_box((0.10, -1.0, -0.05), (0.70, -0.40, 0.55), METAL if metal_box else WHITE, rot_y=-0.3)
    m.add_box((-0.70, -1.0, -0.70), (-0.10, 0.20, -0.10), WHITE, rot_y=0.3)
    m.add(*_quad((-0.25, 0.999, -0.25), (0.25, 0.999, -0.25), (0.25, 0.999, 0.25), (-0.25, 0.999, 0.25),
                 (0, -1, 0), LIGHT))
    if glass_pane:
        m.add(*_quad((-0.6, -0.8, 0.6), (0.2, -0.8, 0.6), (0.2, 0.3, 0.6), (-0.6, 0.3, 0.6), (0, 0, 1), GLASS))
    kw = {}
    if with_directional_light:
        dl = np.zeros(2, dtype=abi.DIRECTIONAL_LIGHT)
        d = np.array([0.3, -1.0, -0.6], dtype=np.float64)
        d /= np.linalg.norm(d)
        dl[0]["color"] = (1.0, 0.95, 0.9)
        dl[0]["intensity"] = 1.5
        dl[0]["direction"] = d
        kw["directional_lights"] = dl
    if with_point_light:
        pl = np.zeros(2, dtype=abi.POINT_LIGHT)
        pl[0]["color"] = (0.9, 0.9, 1.0)
        pl[0]["intensity"] = 40.0
        pl[0]["position"] = (0.0, 0.3, 0.4)
        kw["point_lights"] = pl
    s = m.scene(materials=mats, **kw)
    s.camera_position = (0.0, 0.0, 3.4)
    s.camera_lookat = (0.0, 0.0, -1.0)
    s.camera_vfov = float(np.float32(math.radians(40.0)))
    return s


def procedural_texture(size: int, seed: int, kind: str = "color") -> np.ndarray:
    """Seeded RGBA8 texture: smooth colour noise ('color'), a metal/roughness map
    ('mr': G = roughness, B = metallic), or an alpha-cutout pattern ('alpha')."""
    rng = np.random.RandomState(seed)
    coarse = rng.rand(8, 8, 4)
    reps = size // 8
    up = np.kron(coarse, np.ones((reps, reps, 1)))
    y, x = np.mgrid[0:size, 0:size]
    stripes = 0.5 + 0.5 * np.sin((x * 7 + y * 3) * (2 * math.pi / size))
    img = np.empty((size, size, 4), dtype=np.float64)
    if kind == "color":
        img[..., :3] = 0.25 + 0.7 * up[..., :3] * (0.6 + 0.4 * stripes[..., None])
        img[..., 3] = 1.0
    elif kind == "mr":
        img[..., 0] = 1.0
        img[..., 1] = 0.15 + 0.8 * up[..., 1]
        img[..., 2] = (up[..., 2] > 0.6).astype(np.float64)
        img[..., 3] = 1.0
    elif kind == "alpha":
        img[..., :3] = 0.3 + 0.6 * up[..., :3]
        img[..., 3] = np.where(stripes > 0.5, 1.0, 0.35)
    else:
        raise ValueError(kind)
    return np.clip(np.rint(img * 255.0), 0, 255).astype(np.uint8)


def textured_scene(seed: int = 7, grid: int = 6) -> Scene:
    """Small textured PBR test scene: a tessellated floor, a few boxes, textured
    base-colour / metal-rough / occlusion / emissive maps with the three wrap modes and
    both filters, an alpha-textured pane, one directional and one point light."""
    imgs = [default_image(), procedural_texture(64, seed, "color"), procedural_texture(32, seed + 1, "mr"),
            procedural_texture(64, seed + 2, "alpha"), procedural_texture(16, seed + 3, "color")]
    samplers = np.zeros(4, dtype=abi.SAMPLER)
    samplers[0] = default_sampler()[0]
    samplers[1] = (abi.FILTER_LINEAR, abi.FILTER_LINEAR, abi.WRAP_MIRRORED_REPEAT, abi.WRAP_CLAMP_TO_EDGE)
    samplers[2] = (abi.FILTER_NEAREST, abi.FILTER_NEAREST, abi.WRAP_REPEAT, abi.WRAP_MIRRORED_REPEAT)
    samplers[3] = (abi.FILTER_LINEAR, abi.FILTER_NEAREST, abi.WRAP_CLAMP_TO_EDGE, abi.WRAP_REPEAT)
    textures = np.zeros(6, dtype=abi.TEXTURE)
    textures[0] = (0, 1)  # colour, linear/repeat
    textures[1] = (1, 2)  # metal-rough, mirrored/clamp
    textures[2] = (2, 3)  # alpha pattern, nearest
    textures[3] = (3, 4)  # small colour, clamp/repeat
    textures[4] = (0, 0)  # default checker
    textures[5] = (1, 1)
    mats = np.concatenate([
        default_material(),
        make_material((1, 1, 1, 1), 1.0, 1.0, base_color_texture=0, metallic_roughness_texture=1,
                      occlusion_texture=4, occlusion_strength=0.5),
        make_material((0.9, 0.9, 0.9, 1.0), 0.0, 0.6, base_color_texture=3),
        make_material((1, 1, 1, 1), 0.0, 0.3, base_color_texture=2),           # alpha cut-out
        make_material((0.2, 0.2, 0.2, 1.0), 0.0, 1.0, emissive=(6.0, 5.0, 3.0), emissive_texture=5),
        make_material((0.95, 0.9, 0.8, 1.0), 0.8, 0.05),
    ])
    m = _Mesh()
    n = grid
    for i in range(n):
        for j in range(n):
            x0, x1 = -3 + 6 * i / n, -3 + 6 * (i + 1) / n
            z0, z1 = -3 + 6 * j / n, -3 + 6 * (j + 1) / n
            pos = [(x0, 0, z1), (x1, 0, z1), (x1, 0, z0), (x0, 0, z0)]
            uv = [(i * 0.75 - 0.6, j * 0.75 + 0.2), ((i + 1) * 0.75 - 0.6, j * 0.75 + 0.2),
                  ((i + 1) * 0.75 - 0.6, (j + 1) * 0.75 + 0.2), (i * 0.75 - 0.6, (j + 1) * 0.75 + 0.2)]
            m.add(pos, [(0, 1, 0)] * 4, uv, [1] * 4, [(0, 1, 2), (0, 2, 3)])
    m.add_box((-1.6, 0, -1.2), (-0.4, 1.3, -0.2), 2, rot_y=0.5)
    m.add_box((0.5, 0, -0.9), (1.5, 0.8, 0.1), 5, rot_y=-0.2)
    m.add(*_quad((-1.2, 0.05, 1.2), (1.2, 0.05, 1.2), (1.2, 1.6, 1.2), (-1.2, 1.6, 1.2), (0, 0, 1), 3, uv_scale=2.0))
    m.add(*_quad((-0.5, 2.5, -0.5), (0.5, 2.5, -0.5), (0.5, 2.5, 0.5), (-0.5, 2.5, 0.5), (0, -1, 0), 4))
    dl = np.zeros(2, dtype=abi.DIRECTIONAL_LIGHT)
    d = np.array([-0.4, -1.0, -0.3])
    d /= np.linalg.norm(d)
    dl[0]["color"], dl[0]["intensity"], dl[0]["direction"] = (1.0, 0.97, 0.9), 2.0, d
    pl = np.zeros(2, dtype=abi.POINT_LIGHT)
    pl[0]["color"], pl[0]["intensity"], pl[0]["position"] = (0.6, 0.7, 1.0), 60.0, (1.8, 1.5, 1.6)
    s = m.scene(materials=mats, images=imgs, samplers=samplers, textures=textures,
                directional_lights=dl, point_lights=pl)
    s.camera_position = (2.2, 2.0, 4.2)
    look = -np.array(s.camera_position) + np.array([0.0, 0.6, 0.0])
    look /= np.linalg.norm(look)
    s.camera_lookat = tuple(float(np.float32(x)) for x in look)
    s.camera_vfov = float(np.float32(math.radians(45.0)))
    return s


def soup_scene(n_tris: int, seed: int = 20261111, extent: float = 10.0, edge: float = 0.3,
               n_materials: int = 4, emissive_every: int = 0) -> Scene:
    """cfg4-class procedural triangle soup: n_tris random triangles (centres uniform in
    [-extent, extent]^3, vertex offsets uniform in [-edge, edge]^3 so areas stay far above
    the 1e-6 determinant cut-off, Q11), a diffuse/metal material mix, sky background and
    one directional light."""
    rng = np.random.RandomState(seed)
    c = rng.uniform(-extent, extent, size=(n_tris, 1, 3))
    e = rng.uniform(-edge, edge, size=(n_tris, 3, 3))
    pos = (c + e).astype(np.float32).reshape(-1, 3)
    p = pos.reshape(n_tris, 3, 3).astype(np.float64)
    nrm = np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0])
    nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-30)
    nrm = np.repeat(nrm[:, None, :], 3, axis=1).reshape(-1, 3)
    uv = np.tile(np.array([[0, 0], [1, 0], [0, 1]], dtype=np.float32), (n_tris, 1))
    mat_ids = 1 + (np.arange(n_tris) % n_materials)
    mats = [default_material()]
    for k in range(n_materials):
        col = tuple(0.3 + 0.6 * rng.rand(3)) + (1.0,)
        mats.append(make_material(col, metallic=float(k % 2), roughness=0.2 + 0.2 * k))
    if emissive_every:
        mats.append(make_material((0, 0, 0, 1), emissive=(8.0, 8.0, 8.0)))
        mat_ids = np.where(np.arange(n_tris) % emissive_every == 0, n_materials + 1, mat_ids)
    v = _verts(pos, nrm, uv, np.repeat(mat_ids, 3))
    tris = np.arange(3 * n_tris, dtype=np.uint32).reshape(n_tris, 3)
    dl = np.zeros(2, dtype=abi.DIRECTIONAL_LIGHT)
    d = np.array([0.2, -1.0, 0.3])
    d /= np.linalg.norm(d)
    dl[0]["color"], dl[0]["intensity"], dl[0]["direction"] = (1.0, 1.0, 1.0), 2.0, d
    s = Scene(vertices=v, triangles=tris, materials=np.concatenate(mats), directional_lights=dl)
    s.camera_position = (0.0, 0.0, 2.6 * extent)
    s.camera_lookat = (0.0, 0.0, -1.0)
    s.camera_vfov = float(np.float32(math.radians(45.0)))
    return s
