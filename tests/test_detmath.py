"""The shared numerical contract (include/ppg_detmath.h, ppg_rng.h) against numpy's libm.

Tolerances are stated in ulps of the result / absolute error; the point of these functions is not libm
accuracy but bit-reproducibility across x86-64 and gfx950 (checked by tests/test_gpu_parity.py)."""
import ctypes as C

import numpy as np


def _eval(lib, op, a, b=None):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b if b is not None else np.zeros_like(a), np.float32)
    o0, o1 = np.zeros_like(a), np.zeros_like(a)
    fp = C.POINTER(C.c_float)
    rc = lib.ppgo_math_eval(op, C.c_uint32(a.size), a.ctypes.data_as(fp), b.ctypes.data_as(fp), o0.ctypes.data_as(fp), o1.ctypes.data_as(fp))
    assert rc == 0
    return o0, o1


def test_sincos(oracle_lib):
    x = np.linspace(-3.2, 7.0, 400001, dtype=np.float32)  # phi ranges used: [-pi/4, 3pi/4] and [0, 2pi]
    s, c = _eval(oracle_lib, 0, x)
    assert np.abs(s - np.sin(x.astype(np.float64))).max() < 1.5e-7
    assert np.abs(c - np.cos(x.astype(np.float64))).max() < 1.5e-7


def test_atan2(oracle_lib):
    rng = np.random.RandomState(1)
    y, x = rng.randn(200000).astype(np.float32), rng.randn(200000).astype(np.float32)
    a, _ = _eval(oracle_lib, 1, y, x)
    assert np.abs(a - np.arctan2(y.astype(np.float64), x.astype(np.float64))).max() < 4e-7  # < 2 ulp at pi
    a, _ = _eval(oracle_lib, 1, np.array([0, 1, -1, 0], np.float32), np.array([0, 0, 0, -1], np.float32))
    assert np.allclose(a, [0, np.pi / 2, -np.pi / 2, np.pi], atol=3e-7)


def test_exp(oracle_lib):
    x = np.linspace(-21, 21, 100001, dtype=np.float32)
    e, _ = _eval(oracle_lib, 2, x)
    ref = np.exp(x.astype(np.float64))
    assert (np.abs(e - ref) / ref).max() < 3e-7


def test_fixed_point_round_trip(oracle_lib):
    v = np.array([0.0, 1.0, 0.5, 0.3, 123456.789, 2.0 ** -25, 3 * 2.0 ** -25, 1e-9, 4.0e6], np.float32)
    r, _ = _eval(oracle_lib, 3, v)
    assert np.all(np.abs(r - v) <= 2.0 ** -25 + 1e-7 * v)  # resolution 2^-24, round-to-nearest-even
    assert r[0] == 0 and r[1] == 1 and r[2] == 0.5 and r[5] == 0.0 and r[6] == np.float32(2.0 ** -24 * 2)


def test_powi(oracle_lib):
    n = np.arange(0, 4000, 7).astype(np.float32)
    p, _ = _eval(oracle_lib, 5, np.full_like(n, 0.999), n)
    assert np.allclose(p, np.float64(np.float32(0.999)) ** n.astype(np.float64), rtol=1e-4)  # error grows ~ n * 2^-24; only feeds Adam's bias correction (GP:100)


def test_adam_learning_rate_is_the_references_double_expression(oracle_lib):
    """GP:100: learningRate * std::sqrt(1 - std::pow(beta2, iter)) / (1 - std::pow(beta1, iter)) — std::pow(float, int) promotes to double, so
    the reference evaluates it in double and rounds once (the float powers of round 4 were 3e-5 off at iter = 2).  The shared
    function must give that float: equal to numpy's double evaluation rounded to float, up to a rounding tie (none in this range)."""
    it = np.arange(1, 20001, dtype=np.float32)
    lr, _ = _eval(oracle_lib, 9, it)
    b1, b2, l0 = np.float64(np.float32(0.9)), np.float64(np.float32(0.999)), np.float64(np.float32(0.01))
    ref = (l0 * np.sqrt(1 - b2 ** it.astype(np.float64)) / (1 - b1 ** it.astype(np.float64))).astype(np.float32)
    assert np.array_equal(lr, ref)
    big = np.array([1 << 20, 1 << 24, (1 << 24) - 1], np.float32)  # the iteration count keeps growing over a render
    lr, _ = _eval(oracle_lib, 9, big)
    assert np.array_equal(lr, np.full(3, np.float32(0.01)))


def test_rng_uniform(oracle_lib):
    dims = np.tile(np.arange(8, dtype=np.float32), 50000)
    u, _ = _eval(oracle_lib, 4, dims, dims)
    assert u.min() >= 0.0 and u.max() < 1.0
    assert abs(u.mean() - 0.5) < 2e-3 and abs(u.var() - 1 / 12) < 1e-3
    per_dim = u.reshape(-1, 8)
    assert np.abs(np.corrcoef(per_dim.T) - np.eye(8)).max() < 0.02  # dimensions are decorrelated


def test_log_and_pow(oracle_lib):
    x = np.exp(np.linspace(-30, 30, 200001)).astype(np.float32)
    l, _ = _eval(oracle_lib, 6, x)
    ref = np.log(x.astype(np.float64))
    assert np.abs(l - ref).max() < 2.5e-6 and (np.abs(l - ref) / np.maximum(np.abs(ref), 1e-3)).max() < 3e-7 * 40  # <= 2 ulp of the result
    l, _ = _eval(oracle_lib, 6, np.array([1.0, 0.0, -1.0], np.float32))
    assert l[0] == 0.0 and l[1] == -np.inf and np.isnan(l[2])
    rng = np.random.RandomState(2)
    a = rng.uniform(1e-6, 1.0, 100000).astype(np.float32); b = rng.uniform(0.3, 1.1, 100000).astype(np.float32)  # pow(1 - u, fit), microfacet.h:596
    p, _ = _eval(oracle_lib, 7, a, b)
    ref = a.astype(np.float64) ** b.astype(np.float64)
    assert (np.abs(p - ref) / ref).max() < 2e-6
