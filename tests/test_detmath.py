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


@pytest.mark.gpu
def test_device_bits_equal_host_bits(B, O):
    """The whole point of bhrt_detmath.h: gfx950 and x86-64 produce IDENTICAL bits (plus IEEE / and sqrt)."""
    if B.device_count() < 1:
        pytest.fail("no HIP device")
    rng = np.random.RandomState(11)
    n = 1 << 20
    wide = (rng.standard_normal(n) * np.exp(rng.uniform(-20, 20, n))).astype(np.float32)
    cases = [
        (0, rng.uniform(-50, 50, n), None), (1, rng.uniform(-50, 50, n), None), (2, rng.uniform(-1.6, 1.6, n), None),
        (0, rng.uniform(0, 6.2832, n), None), (1, rng.uniform(0, 6.2832, n), None), (2, rng.uniform(0, 1.5708, n), None),
        (3, rng.uniform(-1.001, 1.001, n), None), (4, rng.uniform(-1.001, 1.001, n), None),
        (3, 1 - np.exp(rng.uniform(-30, 0, n)), None),
        (5, rng.standard_normal(n), rng.standard_normal(n)), (5, wide, wide[::-1].copy()),
        (6, rng.uniform(0, 1, n), rng.choice([1 / 21.0, 1 / 11.0, 1 / 20001.0, 1 / 200001.0, 20, 10, 20000, 200000, 1 / 2.2, 1.5, 5, 2], n)),
        (6, np.full(n, 2.7182818), -rng.uniform(0, 100, n)), (6, rng.uniform(-2, 2, n), rng.randint(-4, 40, n).astype(np.float32)),
        (6, np.abs(wide), rng.uniform(-3, 3, n)),
        (7, rng.randint(0, 2**31 - 1, n).astype(np.uint32).view(np.float32), None),
        (8, wide, wide[::-1].copy()), (8, rng.uniform(-30, 30, n), rng.uniform(-1, 1, n)), (9, np.abs(wide), None),
    ]
    for fn, a, b in cases:
        a = np.asarray(a, np.float32)
        b = None if b is None else np.asarray(b, np.float32)
        dev = B.math_eval_dev(fn, a, b)
        host = O.math_eval(fn, O.MATH_DEVICE, a, b)
        same = (dev.view(np.uint32) == host.view(np.uint32)) | (np.isnan(dev) & np.isnan(host))
        assert same.all(), f"fn {fn}: {np.count_nonzero(~same)} of {n} differ, e.g. a={a[~same][:3]} b={None if b is None else b[~same][:3]} dev={dev[~same][:3]} host={host[~same][:3]}"
