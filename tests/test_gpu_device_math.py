"""GPU: the kernels' arithmetic building blocks, one value at a time, against the oracle's (bn_device_math_eval)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _eval(fn, x):
    import torch
    from benchnav_amd import _capi
    lib = _capi.load()
    xd = torch.from_numpy(np.ascontiguousarray(x, np.float32)).cuda()
    out = torch.empty_like(xd)
    torch.cuda.synchronize()
    _capi.check(lib.bn_device_math_eval(fn, C.c_void_p(xd.data_ptr()), C.c_void_p(out.data_ptr()), xd.numel(), C.c_void_p(0)))
    torch.cuda.synchronize()
    return out.cpu().numpy()


def test_sqrt_is_correctly_rounded_everywhere_it_is_used():
    rng = np.random.default_rng(0)
    x = np.concatenate([
        rng.random(1 << 21).astype(np.float32) * np.float32(1e4),                         # squared distances on a map
        np.exp(rng.uniform(np.log(1e-30), np.log(1e30), 1 << 20)).astype(np.float32),     # the whole normal range
        np.exp(rng.uniform(np.log(1e-45), np.log(1e-28), 1 << 16)).astype(np.float32),    # denormals and the slow path
        (np.arange(1, 4097, dtype=np.float32) ** 2), (np.arange(1, 4097, dtype=np.float32) ** 2 + 1),   # exact squares and neighbours
        np.array([0.0, np.inf, 1.0, 2.0, 2.0 ** -96, 2.0 ** -97, 2.0 ** -126, 1e-45, 3.4e38], np.float32)])
    got = _eval(0, x)
    want = np.sqrt(x.astype(np.float32))                       # IEEE correctly rounded float32
    assert np.array_equal(got, want), np.flatnonzero(got != want)[:10]


def test_the_short_sqrt_is_correctly_rounded_on_zero_and_normal_arguments():
    """sqrt_cr_normal (the latency kernel's stage cost): no scaling of small arguments, no inf / nan pass-through.  Valid for
    x == 0 and 2^-96 <= x < inf (below, the residuals of its correction step go denormal): squared distances between float
    positions on a map are zero or >= (2^-24 * 2^-10)^2 = 2^-68.  The same bits as IEEE sqrt there."""
    rng = np.random.default_rng(3)
    x = np.concatenate([
        rng.random(1 << 21).astype(np.float32) * np.float32(1e4),
        np.exp(rng.uniform(np.log(1e-28), np.log(3e38), 1 << 21)).astype(np.float32),
        (np.arange(1, 4097, dtype=np.float32) ** 2), (np.arange(1, 4097, dtype=np.float32) ** 2 + 1), (np.arange(2, 4097, dtype=np.float32) ** 2 - 1),
        np.array([0.0, 1.0, 2.0, 2.0 ** -96, 2.0 ** -90, 1e-20, 3.4e38], np.float32)])
    got = _eval(5, x)
    want = np.sqrt(x.astype(np.float32))
    assert np.array_equal(got, want), np.flatnonzero(got != want)[:10]


def test_sincos_and_wrap_equal_the_oracle_bit_for_bit():
    from oracle import oracle as O
    rng = np.random.default_rng(1)
    th = np.concatenate([rng.uniform(-np.pi - 0.2, np.pi + 0.2, 1 << 20), rng.uniform(-100, 100, 1 << 18),
                         np.array([0.0, np.pi, -np.pi, np.pi / 2, -np.pi / 2, 7.0, -7.0, 1e-8, 3.1415927, -3.1415927])]).astype(np.float32)
    s, c = O.sincos(th, O.TRIG_SPEC)
    assert np.array_equal(_eval(1, th), s) and np.array_equal(_eval(2, th), c)
    # heading wrap (robot_model.py:90): (theta + pi) % (2 pi) - pi with torch.remainder semantics, float32 constants
    pi32, two_pi32 = np.float32(np.pi), np.float32(2 * np.pi)
    a = (th + pi32).astype(np.float32)
    want = (np.mod(a, two_pi32).astype(np.float32) - pi32).astype(np.float32)
    assert np.array_equal(_eval(3, th), want)
    near = th[np.abs(th) <= np.pi + 0.15]
    a = (near + pi32).astype(np.float32)
    assert np.array_equal(_eval(4, near), (np.mod(a, two_pi32).astype(np.float32) - pi32).astype(np.float32))
