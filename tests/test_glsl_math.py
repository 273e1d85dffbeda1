"""include/pt_glsl_math.h (the build's definition of sin/cos/pow) against libm, through the
oracle's exports: accuracy on the ranges main.comp uses, and the exact special values the
shading code relies on."""
import ctypes as C
import math

import numpy as np


def ulp_err(got, want):
    want32 = np.float32(want)
    return abs(float(got) - float(want)) / max(float(np.spacing(np.abs(want32))), 1e-45)


def test_sincos_accuracy_on_0_2pi(oracle):
    s, c = C.c_float(), C.c_float()
    worst = 0.0
    for x in np.linspace(0.0, 2 * math.pi * 1.0001, 20001, dtype=np.float32):
        oracle.orc_sincos(C.c_float(float(x)), C.byref(s), C.byref(c))
        worst = max(worst, abs(s.value - math.sin(float(x))), abs(c.value - math.cos(float(x))))
    assert worst < 2.5e-7


def test_sincos_negative_and_large_arguments(oracle):
    s, c = C.c_float(), C.c_float()
    for x in [-0.5, -3.0, 100.0, -1000.0, 7000.0]:
        oracle.orc_sincos(C.c_float(x), C.byref(s), C.byref(c))
        xf = float(np.float32(x))
        assert abs(s.value - math.sin(xf)) < 2e-6 and abs(c.value - math.cos(xf)) < 2e-6


def test_pow_accuracy_for_the_exponents_in_use(oracle):
    xs = np.concatenate([np.linspace(1e-4, 1.0, 4001), np.linspace(1.0, 10.0, 1001)]).astype(np.float32)
    for y in (2.2, 5.0, 1.0 / 2.2):
        y32 = float(np.float32(y))
        worst = max(ulp_err(oracle.orc_pow(float(x), y32), float(x) ** y32) for x in xs)
        assert worst < 64, (y, worst)  # exp2(y*log2 x) in fp32: ~|y log2 x| * 2^-24 relative, like a GPU pow


def test_pow_special_values(oracle):
    assert oracle.orc_pow(1.0, 2.2) == 1.0          # untextured lookups: pow(vec3(1), 2.2) must stay exactly 1
    assert oracle.orc_pow(0.0, 2.2) == 0.0
    assert oracle.orc_pow(0.0, 5.0) == 0.0
    assert math.isnan(oracle.orc_pow(-1e-7, 5.0))   # 1 - |v.h| slightly below 0 -> NaN sample (Q12)
    assert math.isnan(oracle.orc_pow(float("nan"), 2.2))
    assert oracle.orc_pow(float("inf"), 0.4545) == float("inf")


def test_exp2_log2(oracle):
    for x in [-130.0, -126.5]:
        assert oracle.orc_exp2(x) == 0.0
    assert oracle.orc_exp2(0.0) == 1.0 and oracle.orc_exp2(3.0) == 8.0 and oracle.orc_exp2(-2.0) == 0.25
    assert oracle.orc_exp2(128.0) == float("inf")
    assert abs(oracle.orc_exp2(127.75) / 2.0 ** 127.75 - 1) < 1e-6
    assert oracle.orc_log2(1.0) == 0.0 and oracle.orc_log2(8.0) == 3.0
    for x in [0.3, 0.7071, 1.5, 1e-5, 12345.0]:
        assert abs(oracle.orc_log2(x) - math.log2(float(np.float32(x)))) < 4e-7 * max(1.0, abs(math.log2(x)))
