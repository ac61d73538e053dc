"""The f64-multiply division of the f32 kernels (csrc/fs_device.h f64div): (float)((double)x * (1.0 / d)) == x / d for every f32 x.
CPU: the identity in numpy (IEEE arithmetic) over all 2^23 significands of several binades and random bit patterns, for the
loop-invariant divisors of several resolutions.  GPU: the same check on the device (fs_selftest_f64div)."""
import ctypes

import numpy as np
import pytest


def divisors(res, re=1.0e6):
    dx = 1.0 / res
    f = np.float32
    return {"dx": f(dx), "2dx": f(2.0 * dx), "dx**2": f(dx ** 2), "dx**3": f(dx ** 3), "dx*dx": f(dx) * f(dx), "6dx": f(6) * f(dx),
            "8dt": f(8) * f(0.05 / res), "re": f(re)}


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
                    b = (s.astype(np.float64) * r).astype(np.float32)
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
