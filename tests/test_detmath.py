"""bhrt_detmath.h (shared by the HIP kernels and the oracle's device-math mode) against libm.
The deterministic functions are single-precision Cephes-class implementations (trig) and a double-precision
log/exp (pow): they must stay within a few ulp of libm — the spread libm float functions show across platforms."""
import numpy as np
import pytest


def ulp_diff(a, b):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia)
    ib = np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    return np.abs(ia - ib)


CASES = [
    ("sin", 0, lambda r: r.uniform(-8, 8, 200000), None),
    ("cos", 1, lambda r: r.uniform(-8, 8, 200000), None),
    ("tan", 2, lambda r: r.uniform(0, 1.55, 200000), None),
    ("acos", 3, lambda r: r.uniform(-1, 1, 200000), None),
    ("asin", 4, lambda r: r.uniform(-1, 1, 200000), None),
    ("atan2", 5, lambda r: r.normal(size=200000), lambda r: r.normal(size=200000)),
    ("pow_gloss", 6, lambda r: r.uniform(0, 1, 200000), lambda r: r.choice([1 / 21.0, 1 / 11.0, 1 / 20001.0, 20, 10, 200000, 1 / 2.2, 1.5], 200000)),
    ("pow_exp", 6, lambda r: np.full(200000, 2.7182818), lambda r: -r.uniform(0, 30, 200000)),
]


@pytest.mark.parametrize("name,fn,ga,gb", CASES, ids=[c[0] for c in CASES])
def test_close_to_libm(name, fn, ga, gb, O):
    rng = np.random.RandomState(7)
    a = ga(rng).astype(np.float32)
    b = gb(rng).astype(np.float32) if gb else None
    dev = O.math_eval(fn, O.MATH_DEVICE, a, b)
    ref = O.math_eval(fn, O.MATH_LIBM, a, b)
    finite = np.isfinite(ref) & (np.abs(ref) > 1e-30)
    assert np.array_equal(np.isnan(dev), np.isnan(ref))
    d = ulp_diff(dev[finite], ref[finite])
    if name in ("sin", "cos"):  # near zeros of sin/cos the absolute error is what matters
        assert np.max(np.abs(dev[finite] - ref[finite])) < 1.5e-7
    else:
        assert d.max() <= 4, f"{name}: max ulp diff {d.max()}"
    assert (d <= 1).mean() > 0.97, f"{name}: only {(d <= 1).mean():.4f} within 1 ulp"
    assert (d <= 2).mean() > 0.999


def test_special_values(O):
    f = np.float32
    # powf special cases the samplers can produce (negative base with integral glossiness, zeros, ones)
    a = np.array([0, 0, 1, -0.5, -0.5, -0.5, 2, 0.5, np.nan], f)
    b = np.array([0, 2, 7, 2, 3, 2.5, 0, 0, 1], f)
    dev = O.math_eval(6, O.MATH_DEVICE, a, b)
    ref = O.math_eval(6, O.MATH_LIBM, a, b)
    assert np.array_equal(np.isnan(dev), np.isnan(ref))
    assert np.allclose(dev[~np.isnan(ref)], ref[~np.isnan(ref)], rtol=1e-6)
    for fn, x in ((3, [1, -1, 0, 1.5]), (4, [1, -1, 0, -1.5])):
        x = np.array(x, f)
        dev, ref = O.math_eval(fn, O.MATH_DEVICE, x), O.math_eval(fn, O.MATH_LIBM, x)
        assert np.array_equal(np.isnan(dev), np.isnan(ref))
        assert np.allclose(dev[:3], ref[:3], atol=2e-7)
