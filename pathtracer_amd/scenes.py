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


# ------------------------------------------------------------------ parametric surfaces
def _surface(f, nu: int, nv: int, mat: int, uv_scale=(1.0, 1.0), flip: bool = False):
    """Tessellate pos = f(u, v), (u, v) in [0,1]^2, into 2*nu*nv triangles with smooth
    normals (cross product of central differences) and uv = (u, v) * uv_scale."""
    u = np.linspace(0.0, 1.0, nu + 1)
    v = np.linspace(0.0, 1.0, nv + 1)
    U, V = np.meshgrid(u, v, indexing="ij")
    P = f(U, V)
    h = 1e-4
    du = (f(np.clip(U + h, 0, 1), V) - f(np.clip(U - h, 0, 1), V))
    dv = (f(U, np.clip(V + h, 0, 1)) - f(U, np.clip(V - h, 0, 1)))
    N = np.cross(du, dv)
    if flip:
        N = -N
    ln = np.linalg.norm(N, axis=-1, keepdims=True)
    N = np.where(ln > 1e-20, N / np.maximum(ln, 1e-20), np.array([0.0, 1.0, 0.0]))
    idx = np.arange((nu + 1) * (nv + 1)).reshape(nu + 1, nv + 1)
    a, b, c, d = idx[:-1, :-1], idx[1:, :-1], idx[1:, 1:], idx[:-1, 1:]
    if flip:
        tris = np.stack([np.stack([a, c, b], -1), np.stack([a, d, c], -1)], axis=2)
    else:
        tris = np.stack([np.stack([a, b, c], -1), np.stack([a, c, d], -1)], axis=2)
    uv = np.stack([U * uv_scale[0], V * uv_scale[1]], -1)
    n = (nu + 1) * (nv + 1)
    return (P.reshape(n, 3), N.reshape(n, 3), uv.reshape(n, 2), np.full(n, mat, dtype=np.int32), tris.reshape(-1, 3))


class _BigMesh:
    """Array-based mesh accumulator for the large generated scenes."""

    def __init__(self):
        self.parts = []
        self.count = 0

    def add(self, pos, nrm, uv, mat, tris):
        self.parts.append((np.asarray(pos, np.float32), np.asarray(nrm, np.float32), np.asarray(uv, np.float32),
                           np.asarray(mat, np.int32), np.asarray(tris, np.uint32) + np.uint32(self.count)))
        self.count += len(pos)

    def scene(self, **kw) -> Scene:
        pos, nrm, uv, mat, tris = (np.concatenate([p[k] for p in self.parts]) for k in range(5))
        # drop zero-area triangles (sphere poles): the shader could never hit them (Q11)
        P = pos[tris.reshape(-1)].reshape(-1, 3, 3).astype(np.float64)
        area2 = np.linalg.norm(np.cross(P[:, 1] - P[:, 0], P[:, 2] - P[:, 0]), axis=1)
        tris = tris[area2 > 1e-9]
        return Scene(vertices=_verts(pos, nrm, uv, mat), triangles=tris, **kw)


def atrium_scene(detail: int = 16, tex_size: int = 1024, n_materials: int = 24, seed: int = 3) -> Scene:
    """cfg3: a generated "Sponza-class" textured PBR atrium: a long hall with a
    relief floor, two storeys of colonnades with arches, relief walls, a vaulted ceiling open to
    the sky along its ridge, cloth banners with alpha cut-outs, emissive lamps, one directional
    (sun) light. detail=16 gives ~262 k triangles; every material has a tex_size^2 PNG-able
    base-colour map and a metal-rough map (linear / repeat)."""
    rng = np.random.RandomState(seed)
    d = max(1, int(detail))
    m = _BigMesh()
    L, W, H = 24.0, 8.0, 9.0  # hall length (x), half-width (z), height (y)

    n_tex_mats = max(4, n_materials - 2)
    MAT = lambda k: 1 + (k % n_tex_mats)  # textured PBR materials 1..n_tex_mats
    LAMP = n_tex_mats + 1
    CLOTH = n_tex_mats + 2

    def bump(U, V, fu, fv, amp):
        return amp * (np.sin(U * fu * 2 * math.pi) * np.sin(V * fv * 2 * math.pi))

    # floor and upper gallery floors (relief tiles)
    m.add(*_surface(lambda U, V: np.stack([-L + 2 * L * U, bump(U, V, 24, 8, 0.015), -W + 2 * W * V], -1),
                    10 * d, 4 * d, MAT(0), (12, 4), flip=True))
    for side in (-1, 1):
        z0, z1 = (side * W, side * (W - 2.5)) if side < 0 else (side * (W - 2.5), side * W)
        m.add(*_surface(lambda U, V, z0=z0, z1=z1: np.stack([-L + 2 * L * U, 4.2 + bump(U, V, 30, 3, 0.01), z0 + (z1 - z0) * V], -1),
                        8 * d, d, MAT(1), (16, 2), flip=True))
        m.add(*_surface(lambda U, V, z0=z0, z1=z1: np.stack([-L + 2 * L * U, 4.0 + 0 * U, z0 + (z1 - z0) * V], -1),
                        4 * d, max(1, d // 2), MAT(2), (16, 2)))
    # side walls and end walls with brick relief
    for side in (-1, 1):
        m.add(*_surface(lambda U, V, s=side: np.stack([-L + 2 * L * U, H * V, s * (W + bump(U, V, 40, 18, 0.02))], -1),
                        10 * d, 4 * d, MAT(3 + (side > 0)), (10, 4), flip=(side > 0)))
        m.add(*_surface(lambda U, V, s=side: np.stack([s * (L + bump(U, V, 14, 18, 0.02)), H * V, -W + 2 * W * U], -1),
                        4 * d, 4 * d, MAT(5 + (side > 0)), (4, 4), flip=(side < 0)))
    # vaulted ceiling, open slot along the ridge (sky light enters)
    for side in (-1, 1):
        m.add(*_surface(lambda U, V, s=side: np.stack([-L + 2 * L * U,
                                                        H + 2.0 * np.sin(V * 0.45 * math.pi),
                                                        s * W * (1.0 - 0.62 * V) + 0 * U], -1),
                        8 * d, 2 * d, MAT(7), (12, 2), flip=(side < 0)))
    # colonnades: two storeys, both sides; columns + arches
    n_col = 10
    xs = np.linspace(-L + 2.4, L - 2.4, n_col)
    for storey, (y0, y1, r) in enumerate([(0.0, 3.2, 0.32), (4.2, 7.0, 0.24)]):
        for side in (-1, 1):
            zc = side * (W - 2.5)
            for ci, xc in enumerate(xs):
                mat = MAT(8 + storey * 2 + (ci % 2))
                flute = 12
                m.add(*_surface(lambda U, V, xc=xc, zc=zc, y0=y0, y1=y1, r=r: np.stack([
                    xc + (r * (1 + 0.06 * np.cos(U * flute * 2 * math.pi)) * (1 + 0.25 * (np.exp(-V * 14) + np.exp(-(1 - V) * 14)))) * np.cos(U * 2 * math.pi),
                    y0 + (y1 - y0) * V,
                    zc + (r * (1 + 0.06 * np.cos(U * flute * 2 * math.pi)) * (1 + 0.25 * (np.exp(-V * 14) + np.exp(-(1 - V) * 14)))) * np.sin(U * 2 * math.pi)], -1),
                    3 * d, 2 * d, mat, (3, 4), flip=True))
                if ci + 1 < n_col:  # arch to the next column: a half-ring band
                    xn = xs[ci + 1]
                    span = (xn - xc) / 2
                    m.add(*_surface(lambda U, V, xm=(xc + xn) / 2, span=span, zc=zc, y1=y1, r=r: np.stack([
                        xm - (span - r * (0.2 + 0.6 * V) * 0) * np.cos(U * math.pi) * (1 - 0.08 * V),
                        y1 + 0.8 * span * np.sin(U * math.pi) * (1 - 0.08 * V) * 0.55,
                        zc + (V - 0.5) * 2 * r + 0 * U], -1),
                        2 * d, max(1, d // 2), MAT(12 + storey), (4, 1)))
    # cloth banners with alpha cut-outs hanging from the upper gallery
    for k, xc in enumerate(np.linspace(-L + 5, L - 5, 6)):
        side = -1 if k % 2 else 1
        m.add(*_surface(lambda U, V, xc=xc, s=side, ph=k: np.stack([
            xc + 1.6 * (U - 0.5), 7.6 - 3.0 * V, s * (W - 2.9) + 0.12 * np.sin(U * 3 * math.pi + ph) * (0.3 + V)], -1),
            d, 2 * d, CLOTH, (1, 2)))
    # emissive lamps: glowing globes under the gallery (NEE towards an opaque emitter returns 0
    # in the reference's estimator, Q5: their light arrives through BSDF-sampled hits only)
    for k, xc in enumerate(np.linspace(-L + 3.6, L - 3.6, 8)):
        for side in (-1, 1):
            m.add(*_surface(lambda U, V, xc=xc, s=side: np.stack([
                xc + 0.45 * np.sin(V * math.pi) * np.cos(U * 2 * math.pi),
                3.3 + 0.45 * np.cos(V * math.pi),
                s * (W - 1.25) + 0.45 * np.sin(V * math.pi) * np.sin(U * 2 * math.pi)], -1),
                max(4, d // 2), max(3, d // 3), LAMP, (1, 1)))

    # materials + textures: image i+1 / texture 2k, 2k+1 per textured material k
    imgs = [default_image()]
    textures = []
    mats = [default_material()]
    samplers = np.zeros(2, dtype=abi.SAMPLER)
    samplers[0] = default_sampler()[0]
    samplers[1] = (abi.FILTER_LINEAR, abi.FILTER_LINEAR, abi.WRAP_REPEAT, abi.WRAP_REPEAT)
    for k in range(n_tex_mats):
        imgs.append(procedural_texture(tex_size, seed * 100 + 2 * k, "color"))
        imgs.append(procedural_texture(tex_size, seed * 100 + 2 * k + 1, "mr"))
        textures += [(1, len(imgs) - 2), (1, len(imgs) - 1)]
        mats.append(make_material((1, 1, 1, 1), 1.0, 1.0, base_color_texture=2 * k, metallic_roughness_texture=2 * k + 1))
    mats.append(make_material((0, 0, 0, 1), emissive=(30.0, 24.0, 14.0)))  # LAMP
    imgs.append(procedural_texture(min(tex_size, 256), seed * 100 + 99, "alpha"))
    textures.append((1, len(imgs) - 1))
    mats.append(make_material((0.9, 0.9, 0.9, 1.0), 0.0, 0.8, base_color_texture=len(textures) - 1))  # CLOTH
    tex = np.zeros(len(textures), dtype=abi.TEXTURE)
    for i, (s_, im_) in enumerate(textures):
        tex[i] = (s_, im_)
    dl = np.zeros(2, dtype=abi.DIRECTIONAL_LIGHT)
    sun = np.array([0.25, -1.0, 0.35])
    sun /= np.linalg.norm(sun)
    dl[0]["color"], dl[0]["intensity"], dl[0]["direction"] = (1.0, 0.96, 0.88), 6.0, sun
    s = m.scene(materials=np.concatenate(mats), images=imgs, samplers=samplers, textures=tex, directional_lights=dl)
    s.camera_position = (-L + 1.5, 2.2, 0.6)
    look = np.array([1.0, 0.12, -0.04])
    look /= np.linalg.norm(look)
    s.camera_lookat = tuple(float(np.float32(x)) for x in look)
    s.camera_vfov = float(np.float32(math.radians(55.0)))
    return s


def rocks_scene(n_tris: int = 5_000_000, seed: int = 20261111, extent: float = 50.0, res: int = 16) -> Scene:
    """cfg4: procedural ~n_tris-triangle scene: seeded clusters of displaced
    spheres ("rocks", 2*res*2res triangles each, radius 0.4..2.5, edge lengths ~0.05..0.5 so
    areas stay far above the 1e-6 cut-off, Q11) scattered through a (2*extent)^3 volume over
    a tessellated ground, diffuse + metal material mix, sky background, one directional light."""
    rng = np.random.RandomState(seed)
    nu, nv = 2 * res, res
    per_rock = 2 * nu * nv
    ground_n = max(8, int(math.sqrt(n_tris * 0.04 / 2)))
    n_rocks = max(1, (n_tris - 2 * ground_n * ground_n) // per_rock)
    m = _BigMesh()
    n_mats = 8
    # ground
    m.add(*_surface(lambda U, V: np.stack([-extent + 2 * extent * U,
                                           -extent + 0.6 * np.sin(U * 37.0) * np.cos(V * 29.0),
                                           -extent + 2 * extent * V], -1), ground_n, ground_n, 1, (40, 40), flip=True))
    # rocks in clusters
    n_clusters = max(1, n_rocks // 6)
    centers = rng.uniform(-extent * 0.9, extent * 0.9, size=(n_clusters, 3))
    centers[:, 1] = rng.uniform(-extent * 0.95, extent * 0.4, size=n_clusters)
    u = np.linspace(0.0, 1.0, nu + 1)
    v = np.linspace(0.0, 1.0, nv + 1)
    U, V = np.meshgrid(u, v, indexing="ij")
    theta, phi = U * 2 * math.pi, V * math.pi
    sph = np.stack([np.sin(phi) * np.cos(theta), np.cos(phi), np.sin(phi) * np.sin(theta)], -1)  # unit sphere grid
    idx = np.arange((nu + 1) * (nv + 1)).reshape(nu + 1, nv + 1)
    a, b, c, d_ = idx[:-1, :-1], idx[1:, :-1], idx[1:, 1:], idx[:-1, 1:]
    tris = np.stack([np.stack([a, c, b], -1), np.stack([a, d_, c], -1)], axis=2).reshape(-1, 3)
    uv = np.stack([U * 2, V], -1).reshape(-1, 2)
    nvert = (nu + 1) * (nv + 1)
    for k in range(n_rocks):
        ctr = centers[k % n_clusters] + rng.normal(0.0, 2.5, size=3)
        radius = rng.uniform(0.4, 2.5)
        f1, f2, f3 = rng.randint(2, 6, size=3)
        p1, p2 = rng.uniform(0, 6.28, size=2)
        disp = 1.0 + 0.18 * np.sin(f1 * theta + p1) * np.sin(f2 * phi) + 0.08 * np.cos(f3 * theta + p2) * np.sin(3 * phi)
        P = ctr + radius * disp[..., None] * sph
        N = sph * np.array(rng.uniform(0.9, 1.1, size=3))  # smooth-ish shading normals
        N /= np.linalg.norm(N, axis=-1, keepdims=True)
        m.add(P.reshape(nvert, 3), N.reshape(nvert, 3), uv, np.full(nvert, 1 + (k % n_mats), np.int32), tris)
    mats = [default_material()]
    for k in range(n_mats):
        col = tuple(0.25 + 0.7 * rng.rand(3)) + (1.0,)
        mats.append(make_material(col, metallic=float(k % 3 == 0), roughness=0.15 + 0.1 * k))
    dl = np.zeros(2, dtype=abi.DIRECTIONAL_LIGHT)
    sun = np.array([0.3, -1.0, 0.25])
    sun /= np.linalg.norm(sun)
    dl[0]["color"], dl[0]["intensity"], dl[0]["direction"] = (1.0, 0.98, 0.92), 3.0, sun
    s = m.scene(materials=np.concatenate(mats), directional_lights=dl)
    s.camera_position = (0.0, -extent * 0.3, extent * 1.25)
    look = np.array([0.0, -0.05, -1.0])
    look /= np.linalg.norm(look)
    s.camera_lookat = tuple(float(np.float32(x)) for x in look)
    s.camera_vfov = float(np.float32(math.radians(45.0)))
    return s


def random_scene(seed: int, with_textures: bool = True, with_punctual: bool = True, with_alpha: bool = True) -> Scene:
    """Seeded random small scene for parity fuzzing: a room of quads and boxes with random PBR
    materials (metal / rough / alpha / emissive, some textured with random filters and wrap
    modes), 0-2 directional and 0-2 point lights, random camera. The three switches take
    textures, punctual lights or non-opaque materials out, so that every kernel
    specialisation gets its own scenes."""
    rng = np.random.RandomState(seed)
    n_tex = int(rng.randint(0, 4)) if with_textures else 0
    imgs = [default_image()] + [procedural_texture(int(rng.choice([8, 16, 32])), seed * 10 + k, str(rng.choice(["color", "mr", "alpha"])))
                                for k in range(n_tex)]
    samplers = np.zeros(1 + n_tex, dtype=abi.SAMPLER)
    samplers[0] = default_sampler()[0]
    for k in range(n_tex):
        samplers[1 + k] = (int(rng.randint(0, 2)), int(rng.randint(0, 2)), int(rng.randint(0, 3)), int(rng.randint(0, 3)))
    textures = np.zeros(max(1, n_tex + 1), dtype=abi.TEXTURE)
    for k in range(n_tex):
        textures[k] = (1 + k, 1 + k)
    textures[-1] = (0, 0)
    n_mat = int(rng.randint(2, 7))
    mats = [default_material()]

    def tex():
        pick = int(rng.randint(0, len(textures))) if rng.rand() < 0.4 else -1
        return pick if n_tex else -1

    for k in range(n_mat):
        a = 1.0 if rng.rand() < 0.7 else float(np.float32(rng.uniform(0.2, 0.9)))
        a = a if with_alpha else 1.0
        emissive = tuple(float(x) for x in rng.uniform(0.5, 12.0, 3)) if rng.rand() < 0.3 else (0.0, 0.0, 0.0)
        mats.append(make_material(tuple(float(x) for x in rng.uniform(0.1, 0.95, 3)) + (a,),
                                  metallic=float(rng.choice([0.0, 1.0, rng.uniform(0, 1)])), roughness=float(rng.uniform(0.02, 1.0)),
                                  emissive=emissive, base_color_texture=tex(), metallic_roughness_texture=tex(),
                                  occlusion_texture=tex(), occlusion_strength=float(rng.uniform(0, 1)), emissive_texture=tex()))
    m = _Mesh()
    m.add_box((-2, -1.5, -2), (2, 1.5, 2), 1 + int(rng.randint(0, n_mat)), inward=True)
    for _ in range(int(rng.randint(1, 6))):
        c = rng.uniform(-1.4, 1.4, 3)
        h = rng.uniform(0.15, 0.6, 3)
        m.add_box(tuple(c - h), tuple(c + h), 1 + int(rng.randint(0, n_mat)), rot_y=float(rng.uniform(0, 3)))
    for _ in range(int(rng.randint(0, 4))):
        c = rng.uniform(-1.5, 1.5, 3)
        a, b = rng.uniform(-0.8, 0.8, 3), rng.uniform(-0.8, 0.8, 3)
        n = np.cross(a, b)
        n /= max(np.linalg.norm(n), 1e-9)
        m.add(*_quad(tuple(c), tuple(c + a), tuple(c + a + b), tuple(c + b), tuple(n), 1 + int(rng.randint(0, n_mat)),
                     uv_scale=float(rng.uniform(0.5, 3.0))))
    kw = {}
    nd, npnt = int(rng.randint(0, 3)), int(rng.randint(0, 3))
    if not with_punctual:
        nd = npnt = 0
    dl = np.zeros(nd + 1, dtype=abi.DIRECTIONAL_LIGHT)
    for k in range(nd):
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        dl[k]["color"], dl[k]["intensity"], dl[k]["direction"] = rng.uniform(0.3, 1, 3), float(rng.uniform(0.5, 3)), d
    pl = np.zeros(npnt + 1, dtype=abi.POINT_LIGHT)
    for k in range(npnt):
        pl[k]["color"], pl[k]["intensity"], pl[k]["position"] = rng.uniform(0.3, 1, 3), float(rng.uniform(5, 60)), rng.uniform(-1.5, 1.5, 3)
    s = m.scene(materials=np.concatenate(mats), images=imgs, samplers=samplers, textures=textures, directional_lights=dl, point_lights=pl)
    pos = rng.uniform(-1.6, 1.6, 3)
    look = rng.normal(size=3)
    look /= np.linalg.norm(look)
    s.camera_position = tuple(float(np.float32(x)) for x in pos)
    s.camera_lookat = tuple(float(np.float32(x)) for x in look)
    s.camera_vfov = float(np.float32(math.radians(rng.uniform(30, 80))))
    return s
