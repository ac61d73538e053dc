"""The f64-multiply division of the f32 kernels (csrc/fs_device.h f64div): (float)((double)x * (1.0 / d)) == x / d for every f32 x whose
quotient is not a denormal; denormal results are repaired (FS_F64DIV_FIX: recomputed by the IEEE division, or made exact by one f64
Newton step) because x / d can then be EXACTLY a tie between two denormals, which the rounded f64 product misses by one ulp.
CPU: the identity in numpy (IEEE arithmetic) over all 2^23 significands of several binades and random bit patterns, for the
loop-invariant divisors of several resolutions, and both repairs on the tie dividends.  GPU: the same check on the device
(fs_selftest_f64div, which runs the library's own f64div)."""
import ctypes

import numpy as np
import pytest


def divisors(res, re=1.0e6):
    dx = 1.0 / res
    f = np.float32
    return {"dx": f(dx), "2dx": f(2.0 * dx), "dx**2": f(dx ** 2), "dx**3": f(dx ** 3), "dx*dx": f(dx) * f(dx), "6dx": f(6) * f(dx),
            "8dt": f(8) * f(0.05 / res), "re": f(re)}


def guarded(x, d, r):
    """fs_device.h f64div, FS_F64DIV_FIX = 2: the f64 product, and the IEEE quotient where the result came out denormal."""
    with np.errstate(all="ignore"):
        q = (x.astype(np.float64) * r).astype(np.float32)
        den = (np.abs(q) < np.float32(2.0 ** -126)) & (q != 0)
        return np.where(den, (x / d).astype(np.float32), q)


def tie_dividends(d, count=1 << 23):
    """f32 dividends x = d (m + 1/2) 2^-149 that are EXACT: x / d is a tie between two denormals."""
    t = np.ldexp(np.arange(count, dtype=np.float64) + 0.5, -149)
    xd = t * np.float64(abs(d))                  # exact: 21 + 24 bits
    x = xd.astype(np.float32)
    return x[x.astype(np.float64) == xd]


@pytest.mark.parametrize("d", [1.0e5, 9600.0, 1.0e6, 1.0 / 1600, 0.05 * 8 / 400, 3.0])
def test_denormal_ties_need_the_repair_and_get_it(d):
    from fractions import Fraction
    d = np.float32(d)
    r = np.float64(1.0) / np.float64(d)
    x = tie_dividends(d)
    # x / d is exactly a denormal tie for some f32 x iff d is an even integer (csrc/fs_device.h; csrc/fs_host.h tie_free)
    even_integer = float(d) == int(float(d)) and int(float(d)) % 2 == 0
    assert (x.size > 0) == even_integer, (float(d), x.size)
    if x.size == 0:
        with np.errstate(all="ignore"):       # a tie-free divisor: the bare product is right on every denormal quotient of a sample
            q = np.ldexp(np.arange(1, 1 << 20, dtype=np.float64), -149)
            xs = np.unique((q * np.float64(d)).astype(np.float32))
            assert np.array_equal((xs.astype(np.float64) * r).astype(np.float32).view(np.uint32), (xs / d).astype(np.float32).view(np.uint32))
        return
    for s in (x, -x):
        with np.errstate(all="ignore"):
            exact = (s / d).astype(np.float32)
            bare = (s.astype(np.float64) * r).astype(np.float32)
        assert np.array_equal(guarded(s, d, r).view(np.uint32), exact.view(np.uint32))
        if float(d) in (1.0e5, 9600.0):          # the advisor's divisors: the unguarded product IS wrong on some ties
            assert (bare.view(np.uint32) != exact.view(np.uint32)).any()
    # FS_F64DIV_FIX = 1 (one Newton step in f64, two fused multiply-adds) on a sample, the fma evaluated exactly with fractions
    fd, fr = Fraction(float(d)), Fraction(float(r))
    for xv in x[:: max(1, x.size // 1500)]:
        fx = Fraction(float(xv))
        p = float(np.float64(xv) * r)
        res = float(fx - Fraction(p) * fd)       # fma(-p, d, x), rounded once
        p2 = float(Fraction(res) * fr + Fraction(p))
        with np.errstate(all="ignore"):
            assert np.float32(p2).view(np.uint32) == (xv / d).astype(np.float32).view(np.uint32), (float(d), float(xv))


def test_the_advisors_counterexample():
    x, d = np.float32(9.108440018111311e-40), np.float32(1e5)
    r = np.float64(1.0) / np.float64(d)
    with np.errstate(all="ignore"):
        assert np.float32(np.float64(x) * r) != x / d
    assert guarded(np.array([x]), d, r)[0] == x / d


@pytest.mark.parametrize("res", [1600, 400, 7])
def test_identity_in_ieee_arithmetic(res):
    rng = np.random.default_rng(res)
    sig = (np.arange(1 << 23, dtype=np.uint32) | np.uint32(0x3F800000)).view(np.float32)
    for name, d in divisors(res).items():
        r = np.float64(1.0) / np.float64(d)
        xs = [np.ldexp(sig, e).astype(np.float32) for e in (0, -126, 100)]
        xs.append(rng.integers(0, 1 << 32, size=1 << 21, dtype=np.uint64).astype(np.uint32).view(np.float32))
        for x in xs:
            for s in (x, -x):
                with np.errstate(all="ignore"):
                    a = (s / d).astype(np.float32)
                    b = guarded(s, d, r)
                both_nan = np.isnan(a) & np.isnan(b)
                assert np.array_equal(a.view(np.uint32)[~both_nan], b.view(np.uint32)[~both_nan]), (res, name)


@pytest.mark.gpu
@pytest.mark.parametrize("res", [1600, 400, 200, 100, 1000, 3])
def test_identity_on_the_device(res, hip_lib):
    import fs
    from fs import _lib
    fs.runtime.init(gpu=0, dtype="f32")
    dev = fs.runtime.create_device((16, 8))
    try:
        for name, d in list(divisors(res).items()) + [("re1e8", np.float32(1e8)), ("tiny", np.float32(3.3e-38)), ("huge", np.float32(1e30))]:
            bad = ctypes.c_int(-1)
            _lib.call("fs_selftest_f64div", dev._ctx, float(d), ctypes.byref(bad))
            assert bad.value == 0, f"res {res} divisor {name} = {float(d)!r}: {bad.value} mismatches"
    finally:
        dev.close()
