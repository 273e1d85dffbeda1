"""Independent pins for the shader restatement.

The oracle (oracle/pt_oracle.c) and the HIP kernels were written by one author and share include/pt_glsl_math.h, so
`kernel == oracle` cannot catch a shared misreading of main.comp. The checks here do not go through either:

  * every function below is RE-WRITTEN IN THIS FILE from the GLSL text (main.comp lines cited per function) in numpy
    float32, one rounding per operation, and compared with the oracle's export — bit for bit where the function is
    + - * / sqrt min max only, within a few ulp where it calls pow / sin / cos (the Vulkan driver's, unpinned);
  * quadrature: each pdf integrates to 1 and the sampled directions follow their pdf;
  * closed forms: barycentric hit points and hit / miss masks of single triangles, and the radiance of a
    directionally lit Lambert plane, evaluated in float64 from the formulas of main.comp:356-366, 412-428.

None of this makes parity "pinned" (the reference still cannot run here), it removes the single-author risk."""
import ctypes as C

import numpy as np
import pytest

from pathtracer_amd import scenes
from pathtracer_amd.scene import PackedScene, Scene

import oracle_lib

F = np.float32
EPSILON = F(1e-6)        # main.comp:21
PI = F(3.1415926535)     # main.comp:13


def fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def vec(*xs):
    return np.array(xs, dtype=np.float32)


@pytest.fixture(scope="module")
def orc(oracle):
    o = oracle
    o.orc_ray_aabb.restype = C.c_int
    o.orc_ray_triangle.restype = C.c_int
    o.orc_ray_triangle.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
    o.orc_distribution_ggx.restype = C.c_float
    o.orc_distribution_ggx.argtypes = [C.c_float, C.c_float]
    o.orc_geometry_smith1.restype = C.c_float
    o.orc_geometry_smith1.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float]
    o.orc_microfacet_brdf.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
    o.orc_rand_float.restype = C.c_float
    o.orc_microfacet_sample.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
    return o


# ------------------------------------------------------------------ GLSL, re-written here in float32 (NOT from the oracle)

def g_dot(a, b):  # dot(): x*x + y*y + z*z, left to right
    return F(F(F(a[0] * b[0]) + F(a[1] * b[1])) + F(a[2] * b[2]))


def g_cross(a, b):  # cross(): (a.y b.z - b.y a.z, a.z b.x - b.z a.x, a.x b.y - b.x a.y)
    return vec(F(a[1] * b[2]) - F(b[1] * a[2]), F(a[2] * b[0]) - F(b[2] * a[0]), F(a[0] * b[1]) - F(b[0] * a[1]))


def g_normalize(a):  # normalize(): v * inversesqrt(dot(v, v))
    inv = F(1.0) / np.sqrt(g_dot(a, a))
    return vec(a[0] * inv, a[1] * inv, a[2] * inv)


def glsl_ray_aabb(o, d, bmin, bmax):
    """main.comp:169-185"""
    inv_dir = F(1.0) / d
    t_min = inv_dir * (bmin - o)
    t_max = inv_dir * (bmax - o)
    t1, t2 = np.minimum(t_min, t_max), np.maximum(t_min, t_max)
    t_near = max(max(t1[0], t1[1]), t1[2])
    t_far = min(min(t2[0], t2[1]), t2[2])
    return bool(t_near <= t_far), F(t_near)


def glsl_ray_triangle(o, d, p0, p1, p2, t_in):
    """main.comp:189-217; returns None when the triangle is not accepted"""
    edge1, edge2 = p1 - p0, p2 - p0
    rce2 = g_cross(d, edge2)
    det = g_dot(edge1, rce2)
    if abs(det) < EPSILON:
        return None
    inv_det = F(1.0) / det
    s = o - p0
    u = F(inv_det * g_dot(s, rce2))
    if u < 0.0 or u > 1.0:
        return None
    sce1 = g_cross(s, edge1)
    v = F(inv_det * g_dot(d, sce1))
    if v < 0.0 or F(u + v) > 1.0:
        return None
    t = F(inv_det * g_dot(edge2, sce1))
    if t > EPSILON and t < t_in:
        return t, u, v
    return None


def glsl_distribution_ggx(n_dot_h, a2):
    """main.comp:336-342"""
    if a2 < EPSILON and n_dot_h > F(F(1.0) - EPSILON):
        return F(1.0)
    denom = F(F(F(n_dot_h * n_dot_h) * F(a2 - F(1.0))) + F(1.0))
    denom = F(F(PI * denom) * denom)
    return F(a2 / denom)


def glsl_geometry_smith1(n, h, x, a2):
    """main.comp:344-350"""
    if g_dot(h, x) <= 0.0:
        return F(0.0)
    ndx = g_dot(n, x)
    denom = F(abs(ndx) + np.sqrt(F(a2 + F(F(F(F(1.0) - a2) * ndx) * ndx))))
    return F(F(1.0) / denom)


def glsl_microfacet_brdf(n, v, l, metal, a2, base):
    """main.comp:356-366 (pow is the driver's: numpy's float32 pow stands in, so compare within a few ulp)"""
    h = g_normalize(l + v)
    n_dot_h = max(g_dot(n, h), EPSILON)
    f0 = vec(0.04, 0.04, 0.04) * F(F(1.0) - metal) + base * metal  # mix(x, y, a) = x*(1-a) + y*a
    p5 = np.power(F(F(1.0) - abs(g_dot(v, h))), F(5.0), dtype=np.float32)
    f = f0 + (F(1.0) - f0) * p5
    diffuse = (F(1.0) - f) * F(F(1.0) - metal) * base / PI
    g2 = F(glsl_geometry_smith1(n, h, l, a2) * glsl_geometry_smith1(n, h, v, a2))  # geometry_smith2(n, h, l, v, a2)
    specular = f * glsl_distribution_ggx(n_dot_h, a2) * g2
    return (diffuse + specular).astype(np.float32)


def glsl_orthogonal_matrix(w):
    """main.comp:151-156: columns (u, v, w)"""
    a = vec(0, 1, 0) if abs(w[0]) > F(0.9) else vec(1, 0, 0)
    v = g_normalize(g_cross(w, a))
    u = g_cross(w, v)
    return u, v, w


def g_mat3_mul(cols, p):  # mat3 * vec3, column-major: c0*p.x + c1*p.y + c2*p.z per component
    u, v, w = cols
    return vec(*[F(F(F(u[i] * p[0]) + F(v[i] * p[1])) + F(w[i] * p[2])) for i in range(3)])


def triple32(x):
    """main.comp:135-143 (https://nullprogram.com/blog/2018/07/31/)"""
    x &= 0xFFFFFFFF
    x ^= x >> 17; x = (x * 0xED5AD4BB) & 0xFFFFFFFF
    x ^= x >> 11; x = (x * 0xAC4C1B51) & 0xFFFFFFFF
    x ^= x >> 15; x = (x * 0x31848BAB) & 0xFFFFFFFF
    x ^= x >> 14
    return x


class Rng:
    def __init__(self, state):
        self.state = state

    def rand_float(self):
        """main.comp:146-149: float(state) / float(~0u); float(0xFFFFFFFF) rounds to 2^32"""
        self.state = triple32(self.state)
        return F(F(self.state) / F(0xFFFFFFFF))


def glsl_uniform_hemisphere_sampling(n, rng):
    """main.comp:370-376"""
    phi = F(F(F(2.0) * PI) * rng.rand_float())
    cos_theta = F(F(1.0) - rng.rand_float())
    sin_theta = np.sqrt(F(F(1.0) - F(cos_theta * cos_theta)))
    p_local = vec(np.cos(phi) * sin_theta, np.sin(phi) * sin_theta, cos_theta)
    return g_mat3_mul(glsl_orthogonal_matrix(n), p_local)


def glsl_diffuse_sample(n, rng):
    """main.comp:378-389"""
    phi = F(F(F(2.0) * PI) * rng.rand_float())
    r = rng.rand_float()
    l_local = vec(np.cos(phi) * np.sqrt(r), np.sin(phi) * np.sqrt(r), np.sqrt(F(F(1.0) - r)))
    l = g_mat3_mul(glsl_orthogonal_matrix(n), l_local)
    return l, F(max(g_dot(n, l), F(0.0)) / PI)


def glsl_microfacet_sample(n, v, a2, rng):
    """main.comp:391-405"""
    phi = F(F(F(2.0) * PI) * rng.rand_float())
    r = rng.rand_float()
    cos2 = F(min(max(F(F(F(1.0) - r) / F(F(r * F(a2 - F(1.0))) + F(1.0))), F(0.0)), F(1.0)))
    cos_t, sin_t = np.sqrt(cos2), np.sqrt(F(F(1.0) - cos2))
    h = g_normalize(g_mat3_mul(glsl_orthogonal_matrix(n), vec(np.cos(phi) * sin_t, np.sin(phi) * sin_t, cos_t)))
    i = -v
    l = i - h * F(F(2.0) * g_dot(h, i))  # reflect(I, N) = I - 2 dot(N, I) N
    pdf = F(F(glsl_distribution_ggx(cos_t, a2) * cos_t) / F(F(4.0) * abs(g_dot(h, l))))
    return l.astype(np.float32), pdf


def unit(rs, n=1):
    v = rs.normal(size=(n, 3))
    return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)


# ------------------------------------------------------------------ per-function vectors

def test_rand_float_stream_and_the_inclusive_upper_end(orc):
    rng = Rng(12345)
    st = C.c_uint32(12345)
    for _ in range(1000):
        assert orc.orc_rand_float(C.byref(st)) == rng.rand_float() and st.value == rng.state
    # Q3: float(~0u) is 2^32, and states from 0xFFFFFF80 on round up to it: rand_float() can be exactly 1.0
    assert F(F(0xFFFFFF80) / F(0xFFFFFFFF)) == 1.0 and F(F(0xFFFFFF7F) / F(0xFFFFFFFF)) < 1.0


def test_ray_aabb_bit_exact_against_the_glsl_text(orc):
    rs = np.random.RandomState(1)
    hits = 0
    for _ in range(3000):
        o = (rs.uniform(-3, 3, 3)).astype(np.float32)
        d = unit(rs)[0]
        c = rs.uniform(-2, 2, 3)
        e = rs.uniform(0.01, 1.5, 3)
        bmin, bmax = (c - e).astype(np.float32), (c + e).astype(np.float32)
        t = C.c_float()
        got = orc.orc_ray_aabb(fptr(o), fptr(d), fptr(bmin), fptr(bmax), C.byref(t))
        want, t_near = glsl_ray_aabb(o, d, bmin, bmax)
        assert bool(got) == want and F(t.value) == t_near
        hits += want
    assert 300 < hits < 2700  # both outcomes exercised
    # Q13: a box entirely BEHIND the origin still "hits" (no t_far >= 0 test), with a negative t_near
    o, d = vec(0, 0, 0), vec(0, 0, 1)
    t = C.c_float()
    assert orc.orc_ray_aabb(fptr(o), fptr(d), fptr(vec(-1, -1, -5)), fptr(vec(1, 1, -4)), C.byref(t)) == 1 and t.value == -5.0


def test_ray_triangle_bit_exact_and_barycentric_closed_form(orc):
    rs = np.random.RandomState(2)
    accepted = 0
    for k in range(4000):
        p = rs.uniform(-1, 1, (3, 3)).astype(np.float32)
        o = (rs.uniform(-1, 1, 3) + np.array([0, 0, 4.0])).astype(np.float32)
        target = p.mean(axis=0) + rs.uniform(-0.8, 0.8, 3)
        d = g_normalize((target - o).astype(np.float32))
        t_in = F(3.4e38) if k % 3 else F(rs.uniform(2.0, 6.0))
        t, u, v = C.c_float(), C.c_float(), C.c_float()
        got = orc.orc_ray_triangle(fptr(o), fptr(d), fptr(p.reshape(-1)), t_in, C.byref(t), C.byref(u), C.byref(v))
        want = glsl_ray_triangle(o, d, p[0], p[1], p[2], t_in)
        assert bool(got) == (want is not None)
        if want is None:
            continue
        accepted += 1
        assert (F(t.value), F(u.value), F(v.value)) == want
        # closed form (float64): the hit point is the same whether reached along the ray or by the barycentrics
        P = p.astype(np.float64)
        along = o.astype(np.float64) + float(t.value) * d.astype(np.float64)
        bary = (1 - u.value - v.value) * P[0] + u.value * P[1] + v.value * P[2]
        assert np.abs(along - bary).max() < 2e-5
        # and it solves o + t d = p0 + u e1 + v e2 in float64
        A = np.stack([-d.astype(np.float64), P[1] - P[0], P[2] - P[0]], axis=1)
        if abs(np.linalg.det(A)) > 1e-3:
            sol = np.linalg.solve(A, o.astype(np.float64) - P[0])
            assert np.abs(sol - np.array([t.value, u.value, v.value])).max() < 1e-3
    assert accepted > 300
    # Q11: abs(det) < 1e-6 rejects (a ray in the triangle's plane), t must exceed epsilon, ties keep the FIRST tested
    tri = vec(0, 0, 0, 1, 0, 0, 0, 1, 0)
    t, u, v = C.c_float(), C.c_float(), C.c_float()
    assert orc.orc_ray_triangle(fptr(vec(-1, 0.2, 0)), fptr(vec(1, 0, 0)), fptr(tri), F(3.4e38), C.byref(t), C.byref(u), C.byref(v)) == 0
    assert orc.orc_ray_triangle(fptr(vec(0.2, 0.2, 1)), fptr(vec(0, 0, -1)), fptr(tri), F(3.4e38), C.byref(t), C.byref(u), C.byref(v)) == 1
    assert t.value == 1.0 and abs(u.value - 0.2) < 1e-7 and abs(v.value - 0.2) < 1e-7
    assert orc.orc_ray_triangle(fptr(vec(0.2, 0.2, 1)), fptr(vec(0, 0, -1)), fptr(tri), F(1.0), C.byref(t), C.byref(u), C.byref(v)) == 0  # t < info.t is strict
    assert orc.orc_ray_triangle(fptr(vec(0.2, 0.2, 5e-7)), fptr(vec(0, 0, -1)), fptr(tri), F(3.4e38), C.byref(t), C.byref(u), C.byref(v)) == 0
    # inclusive edges: u == 0 and u + v == 1 are accepted
    assert orc.orc_ray_triangle(fptr(vec(0, 0.5, 1)), fptr(vec(0, 0, -1)), fptr(tri), F(3.4e38), C.byref(t), C.byref(u), C.byref(v)) == 1
    assert orc.orc_ray_triangle(fptr(vec(0.5, 0.5, 1)), fptr(vec(0, 0, -1)), fptr(tri), F(3.4e38), C.byref(t), C.byref(u), C.byref(v)) == 1


def test_ggx_and_smith_bit_exact_against_the_glsl_text(orc):
    rs = np.random.RandomState(3)
    for _ in range(3000):
        ndh = F(rs.uniform(1e-6, 1.0))
        a2 = F(rs.choice([rs.uniform(0, 1) ** 4, 1.0, 0.0, 1e-7, 0.25]))
        assert F(orc.orc_distribution_ggx(ndh, a2)) == glsl_distribution_ggx(ndh, a2)
        n, h, x = unit(rs, 3)
        assert F(orc.orc_geometry_smith1(fptr(n), fptr(h), fptr(x), a2)) == glsl_geometry_smith1(n, h, x, a2)
    # the mirror special case (main.comp:337) and a back-facing x (main.comp:345)
    assert orc.orc_distribution_ggx(F(1.0), F(0.0)) == 1.0
    n = vec(0, 0, 1)
    assert orc.orc_geometry_smith1(fptr(n), fptr(n), fptr(vec(0, 0, -1)), F(0.5)) == 0.0


def test_ggx_is_a_normalised_distribution_of_normals():
    """Integral over the hemisphere of D(h) cos(theta_h) d omega = 1 (float64 quadrature of the GLSL formula)."""
    for a2 in (0.9, 0.25, 0.01, 1e-3):
        mu = np.linspace(0.0, 1.0, 2_000_001)
        d = a2 / (np.pi * (mu * mu * (a2 - 1.0) + 1.0) ** 2)
        assert abs(2 * np.pi * np.trapezoid(d * mu, mu) - 1.0) < 2e-3


def test_microfacet_brdf_against_the_glsl_text_and_its_symmetry(orc):
    rs = np.random.RandomState(4)
    out = np.zeros(3, dtype=np.float32)
    out2 = np.zeros(3, dtype=np.float32)
    for _ in range(1500):
        n = unit(rs)[0]
        v, l = unit(rs, 2)
        if g_dot(n, v) < 0:
            v = -v
        if g_dot(n, l) < 0:
            l = -l
        metal, a2 = F(rs.choice([0.0, 1.0, rs.uniform()])), F(rs.uniform(0.001, 1.0) ** 2)
        base = rs.uniform(0, 1, 3).astype(np.float32)
        orc.orc_microfacet_brdf(fptr(n), fptr(v), fptr(l), metal, a2, fptr(base), fptr(out))
        want = glsl_microfacet_brdf(n, v, l, metal, a2, base)
        assert np.allclose(out, want, rtol=2e-5, atol=1e-7), (out, want)  # pow(x, 5): driver-defined, a few ulp
        # Helmholtz reciprocity of the formula: h, F(|v.h|) and G1(l) G1(v) are symmetric in l <-> v
        orc.orc_microfacet_brdf(fptr(n), fptr(l), fptr(v), metal, a2, fptr(base), fptr(out2))
        assert np.allclose(out, out2, rtol=2e-5, atol=1e-7)
    # a2 = 1, metal = 0 (the Lambert-plane test below relies on this closed form): D = 1/pi, G1(x) = 1 / (|n.x| + 1)
    n, v, l, base = vec(0, 0, 1), g_normalize(vec(0.3, 0.1, 1)), g_normalize(vec(-0.2, 0.4, 1)), vec(0.8, 0.5, 0.2)
    orc.orc_microfacet_brdf(fptr(n), fptr(v), fptr(l), F(0), F(1), fptr(base), fptr(out))
    h = (l + v) / np.linalg.norm((l + v).astype(np.float64))
    fres = 0.04 + 0.96 * (1 - abs(float(np.dot(v, h)))) ** 5
    closed = (1 - fres) * base.astype(np.float64) / np.pi + fres / np.pi / ((abs(float(l[2])) + 1) * (abs(float(v[2])) + 1))
    assert np.allclose(out, closed, rtol=1e-5)


def test_orthogonal_matrix_is_an_orthonormal_basis(orc):
    rs = np.random.RandomState(5)
    m = np.zeros(9, dtype=np.float32)
    for w in list(unit(rs, 200)) + [vec(1, 0, 0), vec(-1, 0, 0), vec(0.95, 0.3122499, 0)]:
        orc.orc_orthogonal_matrix(fptr(w), fptr(m))
        u, v, w2 = glsl_orthogonal_matrix(w)
        assert (m[:3] == u).all() and (m[3:6] == v).all() and (m[6:] == w2).all()
        M = m.reshape(3, 3).astype(np.float64)
        assert np.abs(M @ M.T - np.eye(3)).max() < 1e-6


def test_samplers_against_the_glsl_text(orc):
    rs = np.random.RandomState(6)
    out = np.zeros(3, dtype=np.float32)
    pdf = C.c_float()
    for k in range(1500):
        n, v = unit(rs, 2)
        if g_dot(n, v) < 0:
            v = -v
        seed = int(rs.randint(0, 2 ** 31))
        a2 = F(rs.uniform(0.01, 1.0) ** 2)
        # sin / cos are the driver's (numpy's here): directions agree to ~1e-6, the RNG state exactly
        st = C.c_uint32(seed); rng = Rng(seed)
        orc.orc_uniform_hemisphere_sampling(fptr(n), C.byref(st), fptr(out))
        want = glsl_uniform_hemisphere_sampling(n, rng)
        assert st.value == rng.state and np.abs(out - want).max() < 3e-6
        assert abs(np.linalg.norm(out.astype(np.float64)) - 1) < 1e-5 and np.dot(out, n) > -1e-6

        st = C.c_uint32(seed); rng = Rng(seed)
        orc.orc_diffuse_sample(fptr(n), C.byref(st), fptr(out), C.byref(pdf))
        want, want_pdf = glsl_diffuse_sample(n, rng)
        assert st.value == rng.state and np.abs(out - want).max() < 3e-6 and abs(pdf.value - want_pdf) < 2e-6
        assert abs(pdf.value - max(float(np.dot(out.astype(np.float64), n)), 0.0) / np.pi) < 1e-6  # cosine pdf of ITS OWN direction

        st = C.c_uint32(seed); rng = Rng(seed)
        orc.orc_microfacet_sample(fptr(n), fptr(v), a2, C.byref(st), fptr(out), C.byref(pdf))
        want, want_pdf = glsl_microfacet_sample(n, v, a2, rng)
        assert st.value == rng.state and np.abs(out - want).max() < 2e-5
        assert abs(pdf.value - want_pdf) <= 2e-4 * max(1.0, abs(float(want_pdf)))
        # l is v mirrored about a unit half vector: |l| = |v| = 1 and h = normalize(l + v) gives back pdf's n.h
        assert abs(np.linalg.norm(out.astype(np.float64)) - 1) < 1e-5


def test_sampled_directions_follow_their_pdfs(orc):
    """Histograms over 100 k draws: uniform hemisphere -> cos(theta) uniform on [0,1]; cosine lobe -> cos^2 uniform;
    GGX half vectors -> the CDF that inverts main.comp:394; and azimuths uniform."""
    n = g_normalize(vec(0.3, -0.5, 0.8))
    N = 100_000
    out = np.zeros(3, dtype=np.float32)
    pdf = C.c_float()
    cols = glsl_orthogonal_matrix(n)
    st = C.c_uint32(2024)
    cu, cd, ch, az = np.zeros(N), np.zeros(N), np.zeros(N), np.zeros(N)
    a2 = F(0.09)
    for i in range(N):
        orc.orc_uniform_hemisphere_sampling(fptr(n), C.byref(st), fptr(out))
        cu[i] = np.dot(out, n)
        orc.orc_diffuse_sample(fptr(n), C.byref(st), fptr(out), C.byref(pdf))
        cd[i] = np.dot(out, n)
        az[i] = np.arctan2(np.dot(out, cols[1]), np.dot(out, cols[0]))
        orc.orc_microfacet_sample(fptr(n), fptr(n), a2, C.byref(st), fptr(out), C.byref(pdf))  # v = n: h = normalize(l + n)
        h = out.astype(np.float64) + n
        ch[i] = np.dot(h / np.linalg.norm(h), n)
    tol = 4 * np.sqrt(0.1 * 0.9 / N)  # 4 sigma of a 10-bin histogram share
    assert np.abs(np.histogram(cu, bins=10, range=(0, 1))[0] / N - 0.1).max() < tol
    assert np.abs(np.histogram(cd ** 2, bins=10, range=(0, 1))[0] / N - 0.1).max() < tol
    assert np.abs(np.histogram(az, bins=10, range=(-np.pi, np.pi))[0] / N - 0.1).max() < tol
    # GGX: cos^2 = (1 - r) / (r (a2 - 1) + 1)  <=>  r = (1 - cos^2) / (1 + cos^2 (a2 - 1)), r uniform
    r = (1 - ch ** 2) / (1 + ch ** 2 * (float(a2) - 1))
    assert np.abs(np.histogram(r, bins=10, range=(0, 1))[0] / N - 0.1).max() < tol


def test_microfacet_pdf_integrates_to_one_over_directions():
    """pdf(l) = D(n.h) (n.h) / (4 |h.l|), h = normalize(l + v) (main.comp:402), integrated over the sphere of l."""
    n = np.array([0.0, 0.0, 1.0])
    for a2, v in ((0.25, np.array([0.0, 0.0, 1.0])), (0.09, np.array([0.6, 0.0, 0.8]))):
        th = (np.arange(1500) + 0.5) * np.pi / 1500
        ph = (np.arange(3000) + 0.5) * 2 * np.pi / 3000
        T, P = np.meshgrid(th, ph, indexing="ij")
        l = np.stack([np.sin(T) * np.cos(P), np.sin(T) * np.sin(P), np.cos(T)], axis=-1)
        h = l + v
        h /= np.linalg.norm(h, axis=-1, keepdims=True)
        ndh = np.clip(h @ n, 0, None)
        d = a2 / (np.pi * (ndh * ndh * (a2 - 1.0) + 1.0) ** 2)
        pdf = np.where(ndh > 0, d * ndh / (4 * np.abs(np.sum(h * l, axis=-1)) + 1e-300), 0.0)
        total = np.sum(pdf * np.sin(T)) * (np.pi / 1500) * (2 * np.pi / 3000)
        # (for oblique v the half vectors with h.v < 0 — 1.2 % of D's mass here — reflect to directions whose own
        # normalize(l + v) is -h, below the horizon: that share is missing from the integral, not from the sampler)
        assert abs(total - 1.0) < (1e-2 if v[2] == 1.0 else 2e-2), (a2, total)


# ------------------------------------------------------------------ closed forms through the whole renderer

def single_triangle_scene():
    """One tilted triangle in front of an explicit camera (covers about a quarter of the frame)."""
    from pathtracer_amd import abi
    verts = np.zeros(3, dtype=abi.VERTEX)
    for i, p in enumerate(((-0.9, -0.6, 0.2), (1.1, -0.4, -0.3), (0.1, 0.8, 0.4))):
        verts[i]["position"] = p
        verts[i]["normal"] = (0, 0, 1)
    return Scene(vertices=verts, triangles=np.array([[0, 1, 2]], dtype=np.uint32),
                 camera_position=(0.1, -0.05, 3.0), camera_lookat=(0.0, 0.05, -1.0), camera_vfov=0.9)


def expected_mask(packed, rp, margin_px=0.75):
    """Float64 hit / miss of the pixel CENTRES against the scene's single triangle, with a don't-care band near edges."""
    pc = rp.pc
    w, h = pc.resolution_x, pc.resolution_y
    cam = np.array(pc.camera_position[:], dtype=np.float64)
    du, dv = np.array(pc.pixel_delta_u[:], dtype=np.float64), np.array(pc.pixel_delta_v[:], dtype=np.float64)
    corner = np.array(pc.corner_pixel_pos[:], dtype=np.float64)
    p = packed.vertex_positions[:3, :3].astype(np.float64)
    xs, ys = np.meshgrid(np.arange(w) + 0.5, np.arange(h) + 0.5)
    inside, near_edge = np.zeros((h, w), bool), np.zeros((h, w), bool)
    e1, e2 = p[1] - p[0], p[2] - p[0]
    for y in range(h):
        for x in range(w):
            def bary(px, py):
                d = corner + px * du + py * dv - cam
                A = np.stack([-d, e1, e2], axis=1)
                t, u, v = np.linalg.solve(A, cam - p[0])
                return t, u, v
            t, u, v = bary(xs[y, x], ys[y, x])
            inside[y, x] = t > 0 and u >= 0 and v >= 0 and u + v <= 1
            flips = 0
            for ox, oy in ((-margin_px, 0), (margin_px, 0), (0, -margin_px), (0, margin_px)):
                t2, u2, v2 = bary(xs[y, x] + ox, ys[y, x] + oy)
                flips += (t2 > 0 and u2 >= 0 and v2 >= 0 and u2 + v2 <= 1) != inside[y, x]
            near_edge[y, x] = flips > 0
    return inside, near_edge


def check_single_triangle_mask(ids, packed, rp):
    inside, near_edge = expected_mask(packed, rp)
    hit = ids == 0
    assert ((ids == 0) | (ids == 0xFFFFFFFF)).all()
    sure = ~near_edge
    assert (hit[sure] == inside[sure]).all()
    assert 0.02 < inside.mean() < 0.98 and sure.mean() > 0.7  # the test sees both outcomes and decides most pixels


def test_single_triangle_hit_mask_closed_form_on_the_cpu(oracle, ptlib):
    packed = PackedScene(single_triangle_scene(), ptlib)
    rp = packed.render_params(ptlib, 48, 32, 1, seed=3)
    _, _, ids, _ = oracle_lib.render(oracle, packed, rp, threads=2)
    check_single_triangle_mask(ids, packed, rp)


@pytest.mark.gpu
def test_single_triangle_hit_mask_closed_form_through_the_hip_path():
    from pathtracer_amd.renderer import Renderer
    r = Renderer(0)
    try:
        packed = r.pack(single_triangle_scene())
        r.upload(packed)
        rp = packed.render_params(r.lib, 48, 32, 1, seed=3)
        _, _, ids, _ = r.render_params(rp)
    finally:
        r.close()
    check_single_triangle_mask(ids, packed, rp)


def lambert_plane_scene():
    """A large plane z = 0 (two triangles, base colour c, metal 0, roughness 1 -> a2 = 1), one directional light, camera above."""
    from pathtracer_amd import abi
    verts = np.zeros(4, dtype=abi.VERTEX)
    for i, (x, y) in enumerate(((-50, -50), (50, -50), (50, 50), (-50, 50))):
        verts[i]["position"] = (x, y, 0)
        verts[i]["normal"] = (0, 0, 1)
        verts[i]["material_index"] = 1
    tris = np.array([[0, 1, 2], [0, 2, 3]], dtype=np.uint32)
    s = scenes.cube_scene()  # borrow the default tables (materials[0], sampler / image / texture conventions, Q17)
    mats = np.zeros(2, dtype=abi.MATERIAL)
    mats[0] = s.materials[0]
    mats[1] = s.materials[0]
    mats[1]["base_color_factor"] = (0.8, 0.5, 0.2, 1.0)
    mats[1]["metallic_factor"] = 0.0
    mats[1]["roughness_factor"] = 1.0
    mats[1]["emissive_factor"] = (0, 0, 0)
    for k in ("base_color_texture_index", "metallic_roughness_texture_index", "occlusion_texture_index", "emissive_texture_index",
              "normal_texture_index"):
        mats[1][k] = -1
    dl = np.zeros(2, dtype=abi.DIRECTIONAL_LIGHT)  # one light + the trailing dummy (Q17)
    ldir = np.array([0.3, -0.2, -1.0]) / np.linalg.norm([0.3, -0.2, -1.0])
    dl[0]["color"] = (1.0, 0.9, 0.7)
    dl[0]["intensity"] = 2.5
    dl[0]["direction"] = ldir
    return Scene(vertices=verts, triangles=tris, materials=mats, images=s.images, samplers=s.samplers, textures=s.textures,
                 directional_lights=dl, point_lights=np.zeros(1, dtype=abi.POINT_LIGHT),
                 camera_position=(1.0, -2.0, 6.0), camera_lookat=(-0.1, 0.3, -1.0), camera_vfov=0.6), ldir


def lambert_closed_form(packed, rp, ldir):
    """Per-pixel radiance at the pixel centre, float64, from main.comp:356-366 and 412-428 with a2 = 1, metal = 0:
    color * intensity * [ (1 - F) base / pi + F * (1/pi) * G1(l) G1(v) ] * max(n.l, 0), F = 0.04 + 0.96 (1 - |v.h|)^5,
    G1(x) = 1 / (|n.x| + 1). Paths go on after the first hit but every later ray leaves the plane into a black sky."""
    pc = rp.pc
    w, h = pc.resolution_x, pc.resolution_y
    cam = np.array(pc.camera_position[:], dtype=np.float64)
    du, dv = np.array(pc.pixel_delta_u[:], dtype=np.float64), np.array(pc.pixel_delta_v[:], dtype=np.float64)
    corner = np.array(pc.corner_pixel_pos[:], dtype=np.float64)
    xs, ys = np.meshgrid(np.arange(w) + 0.5, np.arange(h) + 0.5)
    d = corner + xs[..., None] * du + ys[..., None] * dv - cam
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    n = np.array([0.0, 0.0, 1.0])
    v = -d
    l = -ldir
    hv = l + v
    hv /= np.linalg.norm(hv, axis=-1, keepdims=True)
    fres = 0.04 + 0.96 * (1 - np.abs(np.sum(v * hv, axis=-1))) ** 5
    base = np.array([0.8, 0.5, 0.2])
    g2 = 1.0 / ((abs(l @ n) + 1.0) * (np.abs(v @ n) + 1.0))
    brdf = (1 - fres)[..., None] * base / np.pi + (fres * g2 / np.pi)[..., None]
    light = np.array([1.0, 0.9, 0.7]) * 2.5
    radiance = light * brdf * max(l @ n, 0.0)
    return radiance, d[..., 2] < 0  # rays that reach the plane


def check_lambert(accum, packed, rp, ldir):
    want, on_plane = lambert_closed_form(packed, rp, ldir)
    assert on_plane.all()
    got = accum[..., :3].astype(np.float64) / accum[..., 3:4]
    rel = np.abs(got - want) / want
    assert rel.max() < 2e-3, rel.max()  # 64 jittered samples per pixel around the centre: the integrand is smooth
    assert (accum[..., 3] == 64).all()


def test_directionally_lit_lambert_plane_has_its_closed_form_radiance_on_the_cpu(oracle, ptlib):
    scene, ldir = lambert_plane_scene()
    packed = PackedScene(scene, ptlib)
    rp = packed.render_params(ptlib, 40, 24, 64, (0.0, 0.0, 0.0), seed=8)
    _, accum, ids, st = oracle_lib.render(oracle, packed, rp, threads=4)
    check_lambert(accum, packed, rp, ldir)
    assert st["rays_shadow_walked"] > 0


@pytest.mark.gpu
def test_directionally_lit_lambert_plane_has_its_closed_form_radiance_through_the_hip_path():
    from pathtracer_amd.renderer import Renderer
    scene, ldir = lambert_plane_scene()
    r = Renderer(0)
    try:
        packed = r.pack(scene)
        r.upload(packed)
        rp = packed.render_params(r.lib, 40, 24, 64, (0.0, 0.0, 0.0), seed=8)
        _, accum, _, _ = r.render_params(rp)
    finally:
        r.close()
    check_lambert(accum, packed, rp, ldir)
