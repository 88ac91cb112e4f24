"""The transcendental-light GELU of the INT8 FFN (csrc/encoder_int8_fast.h: gelu_i8x2), restated in numpy float32 with the same coefficients and the
same operation order: its stated bound -- |gelu error| <= 3e-7 absolute, <= 1e-7 |x| -- is checked here against the exact function (math.erf), so that
a change of the coefficients in the kernel without a change here (or the other way round) is caught by the source comparison below."""
import math
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f32 = np.float32
COEF = [-2.753492845e-05, 3.102343180e-04, -1.085499767e-03, -1.382132061e-03, 2.874283120e-02, -1.486884505e-01, -9.183745980e-01, -1.627911806e+00, 4.870492631e-08]


def gelu_i8(x):
    x = x.astype(f32)
    z = (np.abs(x) * f32(0.70710678118654752440)).astype(f32)           # no clamp: past the fitted interval the polynomial falls faster than log2 erfc
    p = np.full_like(z, f32(COEF[0]))
    for c in COEF[1:]:
        p = (p * z + f32(c)).astype(f32)                      # (the kernel fuses each step; the unfused form differs by < 1 ulp per step)
    with np.errstate(under="ignore"):
        e = np.exp2(p.astype(np.float64)).astype(f32)
    return (np.maximum(x, f32(0)) - (np.abs(x) * f32(0.5)) * e).astype(f32)


def test_bound_against_the_exact_function():
    x = np.linspace(-9, 9, 900001).astype(f32)
    xe = x.astype(np.float64)
    exact = xe * 0.5 * (1.0 + np.vectorize(math.erf)(xe / math.sqrt(2.0)))
    err = np.abs(gelu_i8(x).astype(np.float64) - exact)
    assert err.max() <= 3.5e-7, err.max()
    assert (err / np.maximum(np.abs(xe), 1e-2)).max() <= 1.2e-7
    # the properties the range pass relies on: one minimum near x* = -0.7518, non-positive left of 0, equal to x far right
    i = int(np.argmin(gelu_i8(x)))
    assert abs(float(x[i]) + 0.7517916) < 1e-3 and gelu_i8(np.array([-0.7517916], f32))[0] < -0.1699
    assert gelu_i8(np.array([7.5, 20.0, 1e6], f32)).tolist() == [7.5, 20.0, 1e6] and abs(float(gelu_i8(np.array([-20.0], f32))[0])) < 1e-7
    # beyond the fitted interval (z > 4.4) the polynomial only falls: the correction term stays below 5e-10 |x| and never turns into inf / nan
    with np.errstate(over="ignore"):
        far = np.array([-6.3, -7.0, -9.0, -30.0, -1e3, -1e10, -3e38, 6.3, 1e3, 3e38], f32)
        g = gelu_i8(far)
    assert np.isfinite(g).all() and (np.abs(g - np.maximum(far, 0)) <= 5e-10 * np.abs(far)).all(), g


def test_kernel_source_carries_these_coefficients():
    src = open(os.path.join(ROOT, "shodh_memory_amd", "csrc", "encoder_int8_fast.h")).read()
    body = src[src.index("f32x2q gelu_i8x2(f32x2q x) {"):src.index("__device__ __forceinline__ float gelu_i8(float x)")]
    got = [float(v) for v in re.findall(r"\(f32x2q\)(-?\d\.\d+e[+-]\d+)f", body)]
    assert got == COEF, got
    assert "fminf" not in body and body.count("0.70710678118654752440f") == 2
