"""More independent pins of the shader restatement, through the whole renderer (oracle on the CPU, HIP path on the GPU):
float64 EXPECTATIONS of what main.comp's estimators return, written in this file from the GLSL text and the Vulkan rules,
for three scenes whose light transport closes after one or two vertices.

  * a textured, purely emissive plane — uv interpolation (main.comp:226-240), texel addressing / filtering / the three wrap
    modes (Vulkan spec "Texel coordinate systems": unnormalised = u * size - 0.5, repeat / mirrored repeat / clamp to edge,
    nearest = floor), R8G8B8A8_UNORM, `pow(rgb, 2.2)` after filtering and the emissive factor (main.comp:256);
  * a Lambert plane under one point light — sample_punctual_light's sphere-cap sampling and its pdf (main.comp:431-456);
  * a Lambert plane under an emissive quad — sample_area_light (main.comp:464-501), the emission picked up again by the
    bounce ray (trace() adds `emission` at every vertex: light sources are counted by both estimators, main.comp:536-538)
    and trace()'s lobe weight `c * (diffuse_pdf + microfacet_pdf)` (main.comp:571), whose second term is the pdf of the
    OTHER, unused sample.

  * (exact, not statistical) the primary hit of every pixel at one sample per pixel against a float32 brute force over ALL
    triangles, with the camera ray re-derived here from main():588-605 and triple32: the BVH traversal (main.comp:263-304) and
    the builder's boxes lose no hit, invent none, and report a closest one — t equal bit for bit.

  * (exact) the camera block against the PushConstants constructor's formulas and their geometric meaning, the packed arrays
    against process_scene written out in numpy, and the invariants any correct tree over the triangles satisfies.

The first three are statistical checks (jittered samples), with tolerances a few times the measured noise and far below what a
misread factor (a missing 1/count, cos, 1/pi, the 2.2) would cause. Like tests/test_oracle_functions.py they do not pin
parity with the reference (nothing can here); they remove the single-author risk for a1, a6-a10 and a13-a17."""
import numpy as np
import pytest

from pathtracer_amd import abi, scenes
from pathtracer_amd.scene import PackedScene, Scene, make_material

import oracle_lib

PI = 3.1415926535          # main.comp:13
GAMMA = 2.2                # :14
POINT_LIGHT_RADIUS = 0.2   # :17


def render_on(kind, scene, w, h, spp, seed, oracle=None, ptlib=None):
    if kind == "oracle":
        packed = PackedScene(scene, ptlib)
        rp = packed.render_params(ptlib, w, h, spp, (0.0, 0.0, 0.0), seed=seed)
        _, accum, ids, st = oracle_lib.render(oracle, packed, rp, threads=8)
        return packed, rp, accum, ids, st
    from pathtracer_amd.renderer import Renderer
    r = Renderer(0)
    try:
        packed = r.pack(scene)
        r.upload(packed)
        rp = packed.render_params(r.lib, w, h, spp, (0.0, 0.0, 0.0), seed=seed)
        _, accum, ids, st = r.render_params(rp)
    finally:
        r.close()
    return packed, rp, accum, ids, st


def plane_points(rp, sub):
    """Float64 hit points on the plane z = 0 of a sub x sub grid of positions in every pixel: (h, w, sub*sub, 3), and the
    unit ray directions. Camera rays as main():588-605 forms them: corner + x * du + y * dv - camera."""
    pc = rp.pc
    w, h = pc.resolution_x, pc.resolution_y
    cam = np.array(pc.camera_position[:], dtype=np.float64)
    du, dv = np.array(pc.pixel_delta_u[:], dtype=np.float64), np.array(pc.pixel_delta_v[:], dtype=np.float64)
    corner = np.array(pc.corner_pixel_pos[:], dtype=np.float64)
    o = (np.arange(sub) + 0.5) / sub
    xs = (np.arange(w)[None, :, None, None] + o[None, None, None, :]) + np.zeros((h, 1, sub, 1))
    ys = (np.arange(h)[:, None, None, None] + o[None, None, :, None]) + np.zeros((1, w, 1, sub))
    xs, ys = xs.reshape(h, w, -1), ys.reshape(h, w, -1)
    d = corner + xs[..., None] * du + ys[..., None] * dv - cam
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    t = -cam[2] / d[..., 2]
    return cam + t[..., None] * d, d


def quad(z, half, normal, material, uv_of=None):
    verts = np.zeros(4, dtype=abi.VERTEX)
    for i, (x, y) in enumerate(((-half, -half), (half, -half), (half, half), (-half, half))):
        verts[i]["position"] = (x, y, z)
        verts[i]["normal"] = normal
        verts[i]["material_index"] = material
        if uv_of is not None:
            verts[i]["texcoord"] = uv_of(x, y)
    return verts


# ------------------------------------------------------------------ a8 + a9: textured emissive plane

U_OF = (0.45, 0.30)   # u = 0.45 x + 0.30: -0.39 .. 0.99 across the view, so every wrap mode acts
V_OF = (0.40, 0.50)   # v = 0.40 y + 0.50
FACTOR = np.array([1.5, 1.0, 0.5])
SAMPLERS = {"linear_repeat_mirror": (abi.FILTER_LINEAR, abi.WRAP_REPEAT, abi.WRAP_MIRRORED_REPEAT),
            "linear_clamp_repeat": (abi.FILTER_LINEAR, abi.WRAP_CLAMP_TO_EDGE, abi.WRAP_REPEAT),
            "nearest_mirror_clamp": (abi.FILTER_NEAREST, abi.WRAP_MIRRORED_REPEAT, abi.WRAP_CLAMP_TO_EDGE)}


def texture_image():
    return np.random.RandomState(11).randint(20, 256, size=(5, 6, 4)).astype(np.uint8)  # 6 wide, 5 high: W and H differ


def textured_plane_scene(which):
    filt, wrap_s, wrap_t = SAMPLERS[which]
    base = scenes.cube_scene()
    samplers = np.zeros(2, dtype=abi.SAMPLER)
    samplers[0] = base.samplers[0]
    samplers[1] = (filt, filt, wrap_s, wrap_t)
    textures = np.zeros(2, dtype=abi.TEXTURE)
    textures[0] = (0, 0)
    textures[1] = (1, 1)
    mats = np.concatenate([base.materials[:1], make_material((0.0, 0.0, 0.0, 1.0), 0.0, 1.0, emissive=tuple(FACTOR), emissive_texture=1)])
    verts = quad(0.0, 2.0, (0, 0, 1), 1, lambda x, y: (U_OF[0] * x + U_OF[1], V_OF[0] * y + V_OF[1]))
    return Scene(vertices=verts, triangles=np.array([[0, 1, 2], [0, 2, 3]], dtype=np.uint32), materials=mats,
                 images=[base.images[0], texture_image()], samplers=samplers, textures=textures,
                 camera_position=(0.0, 0.0, 6.0), camera_lookat=(0.0, 0.0, -1.0), camera_vfov=0.5)


def wrap_index(i, size, mode):
    """Vulkan spec, "Wrapping operation"."""
    if mode == abi.WRAP_REPEAT:
        return np.mod(i, size)
    if mode == abi.WRAP_MIRRORED_REPEAT:
        n = np.mod(i, 2 * size) - size
        return (size - 1) - np.where(n >= 0, n, -(1 + n))
    return np.clip(i, 0, size - 1)


def vulkan_texture(img, u, v, filt, wrap_s, wrap_t):
    """texture() of an R8G8B8A8_UNORM image without mipmaps, float64, from the Vulkan spec's texel coordinate rules."""
    hgt, wid = img.shape[:2]
    tex = img.astype(np.float64) / 255.0
    if filt == abi.FILTER_NEAREST:
        i = wrap_index(np.floor(u * wid).astype(np.int64), wid, wrap_s)
        j = wrap_index(np.floor(v * hgt).astype(np.int64), hgt, wrap_t)
        return tex[j, i]
    x, y = u * wid - 0.5, v * hgt - 0.5
    i0, j0 = np.floor(x).astype(np.int64), np.floor(y).astype(np.int64)
    a, b = (x - i0)[..., None], (y - j0)[..., None]
    i1, j1 = wrap_index(i0 + 1, wid, wrap_s), wrap_index(j0 + 1, hgt, wrap_t)
    i0, j0 = wrap_index(i0, wid, wrap_s), wrap_index(j0, hgt, wrap_t)
    return (1 - a) * (1 - b) * tex[j0, i0] + a * (1 - b) * tex[j0, i1] + (1 - a) * b * tex[j1, i0] + a * b * tex[j1, i1]


def check_textured_plane(which, rp, accum, ids):
    filt, wrap_s, wrap_t = SAMPLERS[which]
    sub = 12
    p, _ = plane_points(rp, sub)
    u, v = U_OF[0] * p[..., 0] + U_OF[1], V_OF[0] * p[..., 1] + V_OF[1]
    assert u.min() < -0.2 and u.max() > 0.9 and v.min() < -0.05 and v.max() > 1.05  # the view leaves [0, 1] on both axes
    per_sub = FACTOR * vulkan_texture(texture_image(), u, v, filt, wrap_s, wrap_t)[..., :3] ** GAMMA
    want = per_sub.mean(axis=2)
    spread = (per_sub.max(axis=2) - per_sub.min(axis=2)).max(axis=-1)  # how much the pixel's own footprint varies
    assert (ids != 0xFFFFFFFF).all() and (accum[..., 3] == 64).all()
    got = accum[..., :3].astype(np.float64) / accum[..., 3:4]
    err = np.abs(got - want).max(axis=-1)
    if filt == abi.FILTER_LINEAR:
        # 8 x 8 jittered strata against a 12 x 12 midpoint rule over a piecewise-bilinear function
        assert err.max() < 0.012 and err.mean() < 0.003, (err.max(), err.mean())
    else:
        flat = spread == 0.0  # pixels inside one texel: the value is exact up to pow()'s rounding
        assert flat.mean() > 0.35
        assert err[flat].max() < 2e-5, err[flat].max()
        assert (err[~flat] <= 0.2 * spread[~flat] + 1e-5).all()  # straddling pixels: area fractions seen by 64 strata


@pytest.mark.parametrize("which", sorted(SAMPLERS))
def test_textured_emissive_plane_against_the_vulkan_texel_rules_on_the_cpu(which, oracle, ptlib):
    _, rp, accum, ids, _ = render_on("oracle", textured_plane_scene(which), 32, 32, 64, 21, oracle, ptlib)
    check_textured_plane(which, rp, accum, ids)


@pytest.mark.gpu
@pytest.mark.parametrize("which", sorted(SAMPLERS))
def test_textured_emissive_plane_against_the_vulkan_texel_rules_through_the_hip_path(which):
    _, rp, accum, ids, _ = render_on("hip", textured_plane_scene(which), 32, 32, 64, 21)
    check_textured_plane(which, rp, accum, ids)


# ------------------------------------------------------------------ shared: the BRDF of main.comp:356-366 in float64 (a2 = 1, metal = 0)

BASE = np.array([0.8, 0.5, 0.2])


def brdf_rough_dielectric(n, v, l):
    """microfacet_brdf with metal = 0, a2 = 1: D = 1/pi, G1(x) = 1 / (|n.x| + 1) where h.x > 0. Broadcasts over leading axes."""
    hv = l + v
    hv = hv / np.linalg.norm(hv, axis=-1, keepdims=True)
    fres = 0.04 + 0.96 * (1.0 - np.abs(np.sum(v * hv, axis=-1))) ** 5
    g1l = np.where(np.sum(hv * l, axis=-1) > 0, 1.0 / (np.abs(np.sum(n * l, axis=-1)) + 1.0), 0.0)
    g1v = np.where(np.sum(hv * v, axis=-1) > 0, 1.0 / (np.abs(np.sum(n * v, axis=-1)) + 1.0), 0.0)
    return (1.0 - fres)[..., None] * BASE / PI + (fres * g1l * g1v / PI)[..., None]


def lambert_plane(extra_vertices=None, extra_triangles=None, extra_materials=None, point_light=None):
    base = scenes.cube_scene()
    mats = [base.materials[:1], make_material(tuple(BASE) + (1.0,), 0.0, 1.0)]
    verts = [quad(0.0, 60.0, (0, 0, 1), 1)]
    tris = [np.array([[0, 1, 2], [0, 2, 3]], dtype=np.uint32)]
    if extra_vertices is not None:
        tris.append(extra_triangles + 4)
        verts.append(extra_vertices)
        mats.append(extra_materials)
    pl = np.zeros(2 if point_light else 1, dtype=abi.POINT_LIGHT)
    if point_light:
        pl[0]["color"], pl[0]["intensity"], pl[0]["position"] = point_light
    return Scene(vertices=np.concatenate(verts), triangles=np.concatenate(tris), materials=np.concatenate(mats), images=base.images,
                 samplers=base.samplers, textures=base.textures, point_lights=pl,
                 camera_position=(0.6, -3.0, 2.5), camera_lookat=(-0.1, 0.75, -0.65), camera_vfov=0.35)


def block_means(a, by, bx):
    h, w = a.shape[:2]
    return a.reshape(h // by, by, w // bx, bx, -1).mean(axis=(1, 3))


# ------------------------------------------------------------------ a13: point light

LIGHT = ((1.0, 0.8, 0.6), 40.0, (0.3, 0.4, 1.6))


def point_light_expectation(rp):
    """E[ radiance * brdf * |n.l| / pdf ] of main.comp:431-456 at every pixel centre: sample_dir uniform on the hemisphere
    around normalize(pos - light.pos) (density 1 / 2 pi), sample_pos = light.pos + r * sample_dir, pdf = dist2 /
    (cos_theta0 * 4 pi r^2), radiance = color * intensity / (4 pi r^2) / (4 pi). Nothing in the scene can shadow the plane
    (the light has no geometry), so every sample is let through — also those past the sphere's silhouette, whose cos_theta0,
    pdf and contribution are negative: trace() clamps each path's colour to [0, 10] (main.comp:582), which zeroes them."""
    p, d = plane_points(rp, 1)
    p, d = p[:, :, 0], d[:, :, 0]
    color, intensity, lpos = (np.array(LIGHT[0]), LIGHT[1], np.array(LIGHT[2]))
    r = POINT_LIGHT_RADIUS
    area = 4.0 * PI * r * r
    n = np.array([0.0, 0.0, 1.0])
    axis = p - lpos
    axis /= np.linalg.norm(axis, axis=-1, keepdims=True)
    # an orthonormal frame around `axis` (any: the integrand does not depend on the frame), midpoint rule in (phi, cos_theta)
    helper = np.where(np.abs(axis[..., :1]) > 0.9, np.array([0.0, 1.0, 0.0]), np.array([1.0, 0.0, 0.0]))
    t1 = np.cross(axis, helper)
    t1 /= np.linalg.norm(t1, axis=-1, keepdims=True)
    t2 = np.cross(axis, t1)
    nphi, ncos = 48, 48
    total = np.zeros(p.shape)
    for ct in (np.arange(ncos) + 0.5) / ncos:
        st = np.sqrt(1.0 - ct * ct)
        for phi in 2.0 * PI * (np.arange(nphi) + 0.5) / nphi:
            sdir = np.cos(phi) * st * t1 + np.sin(phi) * st * t2 + ct * axis
            l = lpos + r * sdir - p
            dist2 = np.sum(l * l, axis=-1)
            l = l / np.sqrt(dist2)[..., None]
            cos0 = np.sum(sdir * -l, axis=-1)
            pdf = dist2 / (cos0 * area)
            ndl = np.abs(np.sum(n * l, axis=-1))
            radiance = color * intensity / area / (4.0 * PI)
            total += np.clip(radiance * brdf_rough_dielectric(n, -d, l) * (ndl / pdf)[..., None], 0.0, 10.0)
    return total / (nphi * ncos)


def without_the_zero_state_pixel(got, want, rp, seed):
    """main():590 seeds a pixel's stream with `index ^ seed`, and 0 is a fixed point of triple32: the pixel whose index equals
    the seed draws 0.0 for ever (every sample on one spot). Reproduced, and left out of statistical comparisons."""
    got = got.copy()
    w = rp.pc.resolution_x
    got[seed // w, seed % w] = want[seed // w, seed % w]
    return got


def check_point_light(rp, accum, seed):
    want = point_light_expectation(rp)
    assert (accum[..., 3] == 256).all()
    got = without_the_zero_state_pixel(accum[..., :3].astype(np.float64) / accum[..., 3:4], want, rp, seed)
    assert want.min() > 0.005
    rel_px = np.abs(got - want) / want
    rel_block = np.abs(block_means(got, 4, 4) - block_means(want, 4, 4)) / block_means(want, 4, 4)
    ratio = got.sum() / want.sum()
    # per sample the contribution varies with cos_theta0 over the cap: 256 samples -> ~4.5 % per pixel at one sigma, 4 x 4
    # blocks ~1.1 %, the frame ~0.2 % (measured on the oracle: 0.149 / 0.033 / 1.006 — the last is the quadrature's own error)
    assert rel_px.max() < 0.25 and rel_block.max() < 0.06 and 0.99 < ratio < 1.02, (rel_px.max(), rel_block.max(), ratio)


def test_point_lit_lambert_plane_has_its_expected_radiance_on_the_cpu(oracle, ptlib):
    _, rp, accum, _, st = render_on("oracle", lambert_plane(point_light=LIGHT), 32, 24, 256, 5, oracle, ptlib)
    check_point_light(rp, accum, 5)
    assert st["rays_shadow_walked"] > 0


@pytest.mark.gpu
def test_point_lit_lambert_plane_has_its_expected_radiance_through_the_hip_path():
    _, rp, accum, _, _ = render_on("hip", lambert_plane(point_light=LIGHT), 32, 24, 256, 5)
    check_point_light(rp, accum, 5)


# ------------------------------------------------------------------ a14 + a15: emissive quad above the plane

EMISSION = np.array([4.0, 3.0, 2.0])
EMITTER_Z, EMITTER_HALF, EMITTER_CENTRE = 1.4, 0.7, np.array([0.2, 1.2])


def emitter_scene(alpha):
    verts = quad(EMITTER_Z, EMITTER_HALF, (0, 0, -1), 2)
    verts["position"][:, :2] += EMITTER_CENTRE.astype(np.float32)
    # black, rough, non-metallic: what an opaque emitter reflects back is the 4 % Fresnel term of a two-bounce path (< 0.5 % here)
    mat = make_material((0.0, 0.0, 0.0, alpha), 0.0, 1.0, emissive=tuple(EMISSION))
    return lambert_plane(verts, np.array([[0, 2, 1], [0, 3, 2]], dtype=np.uint32), mat)


def microfacet_pdf_values(n, v, count=32):
    """The values `microfacet_pdf` takes (main.comp:392-405, a2 = 1) over a count x count midpoint grid of its two random
    numbers, for view directions v (..., 3): cos2_theta = 1 - rand, h around n, l = reflect(-v, h),
    pdf = D(cos_theta) * cos_theta / (4 |h.l|) with D = 1/pi."""
    helper = np.array([1.0, 0.0, 0.0])
    t1 = np.cross(n, helper)
    t1 /= np.linalg.norm(t1)
    t2 = np.cross(n, t1)
    out = []
    for rand in (np.arange(count) + 0.5) / count:
        ct, st = np.sqrt(1.0 - rand), np.sqrt(rand)
        for phi in 2.0 * PI * (np.arange(count) + 0.5) / count:
            hvec = np.cos(phi) * st * t1 + np.sin(phi) * st * t2 + ct * n
            hv = np.sum(hvec * v, axis=-1)
            l = 2.0 * hv[..., None] * hvec - v  # reflect(-v, h)
            out.append((1.0 / PI) * ct / (4.0 * np.abs(np.sum(hvec * l, axis=-1))))
    return np.stack(out, axis=-1)


def emitter_expectation(rp):
    """Per pixel centre, float64: A = E[sample_area_light] at the plane vertex = integral over the quad of
    emission * brdf * |n.l| * cos_theta0 / dist2 dA (the pdf is the exact area density, 1 / (area * count) per triangle);
    B = the emission found by the bounce ray: diffuse_sample draws l with density n.l / pi, trace() weights the vertex by
    brdf * n.l / (diffuse_pdf + microfacet_pdf) where microfacet_pdf belongs to the independent, discarded microfacet sample
    -> integral over the quad of emission * brdf * n.l * (n.l / pi) * E_m[1 / (n.l / pi + m)] * cos_theta0 / dist2 dA."""
    p, d = plane_points(rp, 1)
    p, d = p[:, :, 0], d[:, :, 0]
    n = np.array([0.0, 0.0, 1.0])
    v = -d
    m = microfacet_pdf_values(n, v)  # (h, w, count^2)
    k = 32
    g = (np.arange(k) + 0.5) / k * 2.0 - 1.0
    cell = (2.0 * EMITTER_HALF / k) ** 2
    a_term, b_term = np.zeros(p.shape), np.zeros(p.shape)
    for gx in g:
        for gy in g:
            q = np.array([EMITTER_CENTRE[0] + gx * EMITTER_HALF, EMITTER_CENTRE[1] + gy * EMITTER_HALF, EMITTER_Z])
            l = q - p
            dist2 = np.sum(l * l, axis=-1)
            l = l / np.sqrt(dist2)[..., None]
            ndl = l[..., 2]
            cos0 = l[..., 2]  # the emitter's normal is turned against l: dot((0, 0, -1), -l)
            f = brdf_rough_dielectric(n, v, l) * (ndl * cos0 / dist2 * cell)[..., None]
            a_term += EMISSION * f
            dpdf = ndl / PI
            weight = dpdf * np.mean(1.0 / (dpdf[..., None] + m), axis=-1)
            b_term += EMISSION * f * weight[..., None]
    return a_term, b_term


def check_emitter(alpha, packed, rp, accum, ids, seed):
    """Opaque emitter (alpha 1): the shadow walk of sample_area_light ends on the emitter itself, an opaque closest hit, and
    returns 0 before any distance test (main.comp:324) — the emitter lights the plane through the bounce ray alone.
    Half-transparent emitter (alpha 0.5): the walk passes it once (transparency 0.5, main.comp:325) and finds nothing behind,
    the bounce ray collects alpha * emission and goes on into the black sky (main.comp:542-547): half of both estimators."""
    a_term, b_term = emitter_expectation(rp)
    want = b_term if alpha == 1.0 else alpha * (a_term + b_term)
    plane_ids = [i for i, t in enumerate(packed.triangles) if t.max() < 4]  # (the builder permutes the triangles)
    assert np.isin(ids, plane_ids).all()  # every camera ray lands on the plane, none on the emitter
    assert (accum[..., 3] == 1024).all()
    got = without_the_zero_state_pixel(accum[..., :3].astype(np.float64) / accum[..., 3:4], want, rp, seed)
    assert 0.5 < (b_term.sum() / a_term.sum()) < 0.8  # the lobe weight matters: with the textbook weight the two terms would be equal
    rel_block = np.abs(block_means(got, 8, 8) - block_means(want, 8, 8)) / block_means(want, 8, 8)
    ratio = got.sum() / want.sum()
    # measured on the oracle (three seeds): opaque 0.061-0.068 / 1.014-1.016 (the excess is what `want` leaves out: light the
    # black emitter reflects at grazing angles, Fresnel -> 1), alpha 0.5: 0.021-0.026 / 1.006-1.007 (quadrature)
    if alpha == 1.0:
        assert rel_block.max() < 0.12 and 0.995 < ratio < 1.035, (rel_block.max(), ratio)
    else:
        assert rel_block.max() < 0.05 and 0.99 < ratio < 1.025, (rel_block.max(), ratio)


@pytest.mark.parametrize("alpha", [1.0, 0.5])
def test_plane_under_an_emissive_quad_has_its_expected_radiance_on_the_cpu(alpha, oracle, ptlib):
    packed, rp, accum, ids, _ = render_on("oracle", emitter_scene(alpha), 32, 24, 1024, 9, oracle, ptlib)
    check_emitter(alpha, packed, rp, accum, ids, 9)


@pytest.mark.gpu
@pytest.mark.parametrize("alpha", [1.0, 0.5])
def test_plane_under_an_emissive_quad_has_its_expected_radiance_through_the_hip_path(alpha):
    packed, rp, accum, ids, _ = render_on("hip", emitter_scene(alpha), 32, 24, 1024, 9)
    check_emitter(alpha, packed, rp, accum, ids, 9)


# ------------------------------------------------------------------ a1 + a7 (+ a16): BVH traversal against brute force

F32 = np.float32


def triple32_v(x):
    """main.comp:135-143 on a uint32 array (https://nullprogram.com/blog/2018/07/31/)"""
    x = x.astype(np.uint64)
    for shift, mul in ((17, 0xED5AD4BB), (11, 0xAC4C1B51), (15, 0x31848BAB)):
        x ^= x >> np.uint64(shift)
        x = (x * np.uint64(mul)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(14)
    return x.astype(np.uint32)


def dot_v(a, b):  # dot(): x*x + y*y + z*z, left to right, one float32 rounding per operation
    return (a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]) + a[..., 2] * b[..., 2]


def cross_v(a, b):
    return np.stack([a[..., 1] * b[..., 2] - b[..., 1] * a[..., 2], a[..., 2] * b[..., 0] - b[..., 2] * a[..., 0],
                     a[..., 0] * b[..., 1] - b[..., 0] * a[..., 1]], axis=-1)


def first_camera_rays(rp, seed):
    """The one ray per pixel of main():588-605 at sample_count 1, float32 operation by operation: state = index ^ seed, two
    draws, pixel_center = corner + (x + (0 + r1) / 1) * du + (y + (0 + r2) / 1) * dv, dir = normalize(pixel_center - camera)."""
    pc = rp.pc
    w, h = pc.resolution_x, pc.resolution_y
    cam = np.array(pc.camera_position[:], dtype=F32)
    du, dv = np.array(pc.pixel_delta_u[:], dtype=F32), np.array(pc.pixel_delta_v[:], dtype=F32)
    corner = np.array(pc.corner_pixel_pos[:], dtype=F32)
    ys, xs = np.mgrid[0:h, 0:w]
    state = triple32_v((ys * w + xs).astype(np.uint32) ^ np.uint32(seed))
    r1 = state.astype(F32) / F32(0xFFFFFFFF)
    state = triple32_v(state)
    r2 = state.astype(F32) / F32(0xFFFFFFFF)
    ax = xs.astype(F32) + (F32(0.0) + r1) / F32(1.0)
    ay = ys.astype(F32) + (F32(0.0) + r2) / F32(1.0)
    centre = (corner + ax[..., None] * du) + ay[..., None] * dv
    d = centre - cam
    inv = F32(1.0) / np.sqrt(dot_v(d, d))
    return cam, d * inv[..., None]


def brute_force_hits(origin, dirs, tri_pos):
    """ray_triangle_intersect (main.comp:189-217) of every ray against EVERY triangle, float32 operation by operation; returns
    t (rays, tris) with +inf where the triangle is not accepted."""
    eps = F32(1e-6)
    p0, p1, p2 = tri_pos[:, 0, :3], tri_pos[:, 1, :3], tri_pos[:, 2, :3]
    edge1, edge2 = p1 - p0, p2 - p0
    s = origin - p0
    sce1 = cross_v(s, edge1)
    t_all = np.full((dirs.shape[0], tri_pos.shape[0]), np.inf, dtype=F32)
    for i in range(0, dirs.shape[0], 512):
        d = dirs[i:i + 512, None, :]
        rce2 = cross_v(d, edge2[None])
        det = dot_v(edge1[None], rce2)
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            inv_det = F32(1.0) / det
            u = inv_det * dot_v(s[None], rce2)
            v = inv_det * dot_v(d, sce1[None])
            t = inv_det * dot_v(edge2, sce1)[None]
            ok = ~(np.abs(det) < eps) & ~((u < 0) | (u > 1)) & ~((v < 0) | (u + v > 1)) & (t > eps)
        t_all[i:i + 512] = np.where(ok, t, np.inf)
    return t_all


def check_against_brute_force(packed, rp, ids, seed):
    origin, dirs = first_camera_rays(rp, seed)
    h, w = ids.shape
    t = brute_force_hits(origin, dirs.reshape(-1, 3), packed.vertex_positions.reshape(-1, 3, 4).astype(F32))
    tmin = t.min(axis=1)
    ids = ids.reshape(-1)
    hit = ids != 0xFFFFFFFF
    assert 0.1 < hit.mean() < 0.999  # both outcomes occur
    assert (np.isinf(tmin) == ~hit).all()                        # the tree loses no hit and invents none
    assert (t[hit, ids[hit]] == tmin[hit]).all()                  # and the hit it reports is a closest one
    unique = (t == tmin[:, None]).sum(axis=1) == 1
    assert (np.argmin(t, axis=1)[hit & unique] == ids[hit & unique]).all()
    assert (hit & unique).mean() > 0.1


def brute_force_scene(which):
    return scenes.rocks_scene(n_tris=6000) if which == "rocks" else scenes.atrium_scene(detail=2, tex_size=16)


@pytest.mark.parametrize("which", ["rocks", "atrium"])
def test_primary_hits_equal_a_brute_force_over_all_triangles_on_the_cpu(which, oracle, ptlib):
    packed = PackedScene(brute_force_scene(which), ptlib)
    rp = packed.render_params(ptlib, 96, 64, 1, (0.0, 0.0, 0.0), seed=77)
    _, _, ids, _ = oracle_lib.render(oracle, packed, rp, threads=8)
    check_against_brute_force(packed, rp, ids, 77)


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["rocks", "atrium"])
def test_primary_hits_equal_a_brute_force_over_all_triangles_through_the_hip_path(which):
    from pathtracer_amd.renderer import Renderer
    r = Renderer(0)
    try:
        packed = r.pack(brute_force_scene(which))
        r.upload(packed)
        rp = packed.render_params(r.lib, 96, 64, 1, (0.0, 0.0, 0.0), seed=77)
        _, _, ids, _ = r.render_params(rp)
    finally:
        r.close()
    check_against_brute_force(packed, rp, ids, 77)


# ------------------------------------------------------------------ a17: the camera block of the push constants

def test_camera_block_against_the_formulas_of_the_push_constants_constructor(ptlib):
    """render_job.cc:846-872 in float64: w = normalize(-lookat), u = normalize(cross(up, w)), v = cross(w, u), viewport 2 high and
    2 * rx / ry wide at focal length 1 / tan(fov / 2); and what they mean: the ray through the middle of the frame runs along
    lookat, the top and bottom edges are fov apart, pixels are square."""
    rs = np.random.RandomState(4)
    verts = np.zeros(3, dtype=abi.VERTEX)
    verts["position"] = ((0, 0, 0), (1, 0, 0), (0, 1, 0))
    for case in range(12):
        look = rs.normal(size=3)
        look /= np.linalg.norm(look)
        if abs(look[1]) > 0.95:
            continue
        pos, fov = rs.uniform(-5, 5, size=3), rs.uniform(0.2, 1.8)
        rx, ry = int(rs.randint(8, 400)), int(rs.randint(8, 300))
        scene = Scene(vertices=verts, triangles=np.array([[0, 1, 2]], dtype=np.uint32), camera_position=tuple(pos),
                      camera_lookat=tuple(look), camera_vfov=float(fov))
        pc = PackedScene(scene, ptlib).render_params(ptlib, rx, ry, 4, (0.0, 0.0, 0.0), seed=1).pc
        pos32, look32, fov32 = np.float32(pos).astype(np.float64), np.float32(look).astype(np.float64), float(np.float32(fov))
        w = -look32 / np.linalg.norm(look32)
        u = np.cross([0.0, 1.0, 0.0], w)
        u /= np.linalg.norm(u)
        v = np.cross(w, u)
        vu, vv = (2.0 * rx / ry) * u, -2.0 * v
        want = {"camera_position": pos32, "pixel_delta_u": vu / rx, "pixel_delta_v": vv / ry,
                "corner_pixel_pos": pos32 - w / np.tan(fov32 / 2.0) - 0.5 * (vu + vv)}
        for name, val in want.items():
            got = np.array(getattr(pc, name)[:3], dtype=np.float64)
            assert np.allclose(got, val, rtol=0, atol=4e-6 * max(1.0, np.abs(val).max())), (case, name, got, val)
        du, dv, corner = (np.array(getattr(pc, k)[:3], dtype=np.float64) for k in ("pixel_delta_u", "pixel_delta_v", "corner_pixel_pos"))
        centre = corner + 0.5 * rx * du + 0.5 * ry * dv - pos32
        assert np.allclose(centre / np.linalg.norm(centre), look32 / np.linalg.norm(look32), atol=1e-5)
        top, bottom = corner + 0.5 * rx * du - pos32, corner + 0.5 * rx * du + ry * dv - pos32
        angle = np.arccos(np.dot(top, bottom) / np.linalg.norm(top) / np.linalg.norm(bottom))
        assert abs(angle - fov32) < 1e-5
        assert abs(np.linalg.norm(du) - np.linalg.norm(dv)) < 1e-6 * np.linalg.norm(du) and abs(np.dot(du, dv)) < 1e-9
        assert (pc.resolution_x, pc.resolution_y) == (rx, ry)


# ------------------------------------------------------------------ a6 + a16 + a17: the packed arrays and the tree's invariants

@pytest.mark.parametrize("which", ["textured", "atrium", "rocks"])
def test_packed_arrays_against_process_scene_and_tree_invariants(which, ptlib):
    """render_job.cc:76-121 written out in numpy (positions as vec4(p, 0), the 64-byte vertex record, the emissive list by the
    FIRST vertex's material: factor != 0 or an emissive texture, left_or_begin) and what any correct tree over these triangles
    satisfies whatever split heuristic built it: leaves partition the (permuted) triangle list, a node's box is exactly the
    bounds of its triangles' vertices, children sit side by side at (left, left + 1), the reported depth is the deepest leaf's."""
    scene = {"textured": lambda: scenes.textured_scene(), "atrium": lambda: scenes.atrium_scene(detail=2, tex_size=16),
             "rocks": lambda: scenes.rocks_scene(n_tris=6000)}[which]()
    packed = PackedScene(scene, ptlib)
    tris, verts = packed.triangles, scene.vertices
    assert sorted(map(tuple, tris.tolist())) == sorted(map(tuple, scene.triangles.tolist()))  # a permutation of the input
    flat = verts[tris.reshape(-1)]
    assert np.array_equal(packed.vertex_positions[:, :3], flat["position"]) and (packed.vertex_positions[:, 3] == 0).all()
    pv = packed.packed_vertex_data
    assert np.array_equal(pv["position"][:, :3], flat["position"]) and (pv["position"][:, 3] == 0).all()
    assert np.array_equal(pv["normal_and_texcoord_u"][:, :3], flat["normal"]) and np.array_equal(pv["normal_and_texcoord_u"][:, 3], flat["texcoord"][:, 0])
    assert np.array_equal(pv["tangent_and_texcoord_v"][:, :3], flat["tangent"]) and np.array_equal(pv["tangent_and_texcoord_v"][:, 3], flat["texcoord"][:, 1])
    assert np.array_equal(pv["bitangent"], flat["bitangent"]) and np.array_equal(pv["material_index"], flat["material_index"])
    mats = scene.materials[verts["material_index"][tris[:, 0]]]
    emissive = np.nonzero((mats["emissive_factor"] != 0).any(axis=1) | (mats["emissive_texture_index"] != -1))[0]
    assert np.array_equal(packed.emissive, emissive.astype(np.uint32))
    nodes, pn = packed.nodes, packed.packed_nodes
    assert np.array_equal(pn["aabb_min"], nodes["aabb_min"]) and np.array_equal(pn["aabb_max"], nodes["aabb_max"])
    assert np.array_equal(pn["triangle_count"], nodes["triangle_count"])
    assert np.array_equal(pn["left_or_begin"], np.where(nodes["triangle_count"] > 0, nodes["begin_index"], nodes["left_child_index"]))
    # the tree, walked from the root over the PACKED nodes (what the traversal reads)
    tri_pos = flat["position"].reshape(-1, 3, 3)
    seen = np.zeros(len(tris), dtype=np.int32)
    deepest, stack = 0, [(0, 1)]
    visited = 0

    def bounds_of(node):  # triangle range under a node, and the exact bounds of its vertices
        if pn["triangle_count"][node] > 0:
            b = int(pn["left_or_begin"][node])
            return b, b + int(pn["triangle_count"][node])
        lo0, hi0 = bounds_of(int(pn["left_or_begin"][node]))
        lo1, hi1 = bounds_of(int(pn["left_or_begin"][node]) + 1)
        assert hi0 == lo1  # the children split a contiguous range
        return lo0, hi1

    import sys
    sys.setrecursionlimit(10000)
    while stack:
        node, depth = stack.pop()
        visited += 1
        lo, hi = bounds_of(node) if visited <= 64 else (None, None)  # full range check near the root, leaves below
        if pn["triangle_count"][node] > 0:
            b, c = int(pn["left_or_begin"][node]), int(pn["triangle_count"][node])
            seen[b:b + c] += 1
            pts = tri_pos[b:b + c].reshape(-1, 3)
            assert np.array_equal(pn["aabb_min"][node], pts.min(axis=0)) and np.array_equal(pn["aabb_max"][node], pts.max(axis=0))
            deepest = max(deepest, depth)
        else:
            left = int(pn["left_or_begin"][node])
            assert left % 2 == 0 and left + 1 < len(pn)  # an even-aligned pair: one 64-byte line
            if lo is not None:
                pts = tri_pos[lo:hi].reshape(-1, 3)
                assert np.array_equal(pn["aabb_min"][node], pts.min(axis=0)) and np.array_equal(pn["aabb_max"][node], pts.max(axis=0))
            assert (np.minimum(pn["aabb_min"][left], pn["aabb_min"][left + 1]) == pn["aabb_min"][node]).all()
            assert (np.maximum(pn["aabb_max"][left], pn["aabb_max"][left + 1]) == pn["aabb_max"][node]).all()
            stack += [(left, depth + 1), (left + 1, depth + 1)]
    assert (seen == 1).all()
    assert deepest == packed.bvh_max_depth
