"""GPU: the device exp/log (roadsurf_amd/csrc/rs_math.hpp) must return the SAME BITS as
the libm functions the reference calls (glibc 2.35, FMA ifunc variants)."""
import ctypes as C

import numpy as np
import pytest

import oracle_helpers as oh
from roadsurf_amd import abi, lib

pytestmark = pytest.mark.gpu


def _device(fn, x, n=None):
    import torch
    from roadsurf_amd import device
    plan = device.Plan(256, abi.default_settings(10), abi.default_parameters(), 0)
    xd = torch.from_numpy(x).to(plan.device)
    yd = torch.empty_like(xd)
    lib.check(plan.L.rs_hip_test_math(plan._h, fn, x.size if n is None else n, C.c_void_p(xd.data_ptr()),
                                      C.c_void_p(yd.data_ptr())), "rs_hip_test_math")
    plan.sync()
    y = yd.cpu().numpy()
    plan.close()
    return y


def _libm(fn, x):
    port = oh.load("port")
    y = np.empty_like(x)
    port.oracle_libm_map.argtypes = [C.c_int, C.c_long, abi.c_double_p, abi.c_double_p]
    port.oracle_libm_map(fn, x.size, x.ctypes.data_as(abi.c_double_p), y.ctypes.data_as(abi.c_double_p))
    return y


def test_exp_bit_identical_to_libm():
    rs = np.random.RandomState(1)
    # Magnus arguments are in [-3.5, 2.5]; CalcPrecType's in [-280, 270]; relaxation's in [-12, 0]
    x = np.concatenate([rs.uniform(-4, 3, 500000), rs.uniform(-300, 300, 300000),
                        rs.uniform(-12, 0, 200000), rs.uniform(-1e-3, 1e-3, 50000),
                        np.array([0.0, -0.0, 1e-300, -1e-300, 1.0, -1.0, 1e-17, 511.9, -511.9])])
    y, ref = _device(0, x), _libm(0, x)
    bad = np.flatnonzero(y.view(np.int64) != ref.view(np.int64))
    assert bad.size == 0, (bad.size, x[bad[:5]], y[bad[:5]], ref[bad[:5]])


def test_log_bit_identical_to_libm():
    rs = np.random.RandomState(2)
    # model arguments: (1 + sqrt(1 - 16 Stab))/2 >= 1, typically < 10; both code paths
    x = np.concatenate([1.0 + rs.uniform(0, 9, 500000), 1.0 + 10.0 ** rs.uniform(-12, 0, 200000),
                        rs.uniform(0.9, 1.1, 200000), 10.0 ** rs.uniform(-300, 300, 100000),
                        np.array([1.0, 2.0, 0.5, 0.9375, 1.0644, 1.0645, 1.06494140625])])
    y, ref = _device(1, x), _libm(1, x)
    bad = np.flatnonzero(y.view(np.int64) != ref.view(np.int64))
    assert bad.size == 0, (bad.size, x[bad[:5]], y[bad[:5]], ref[bad[:5]])


def test_out_of_domain_arguments_are_sane():
    sp = _device(1, np.array([0.0, -1.0, np.inf, np.nan, 5e-324]))
    assert sp[0] == -np.inf and np.isnan(sp[1]) and sp[2] == np.inf and np.isnan(sp[3])
    assert abs(sp[4] - np.log(5e-324)) < 1e-12
    # |x| >= 512 is outside the model's domain (CheckValues keeps arguments within +-300):
    # documented saturation instead of glibc's gradual over/underflow
    se = _device(0, np.array([1000.0, -1000.0, np.nan, 600.0, -600.0, 511.0]))
    assert se[0] == np.inf and se[1] == 0.0 and np.isnan(se[2]) and se[3] == np.inf and se[4] == 0.0
    assert se[5] == _libm(0, np.array([511.0]))[0]


def test_bare_division_and_square_root_equal_ieee_on_random_operands():
    """rs_div (v_rcp_f64, one third-order refinement, quotient, exact remainder, final fma: seven
    instructions) and rs_sqrt against IEEE division / square root (numpy on the host) on 2 x 4 M operand
    pairs: exponents over the range the model's quantities live in (1e-12 .. 1e12, both signs), random
    mantissas, and structured ones - quotients next to 1, denominators next to powers of two, numerators
    that are small multiples of the denominator - where a last-bit error would show first."""
    rs = np.random.RandomState(7)
    n = 1 << 22
    a = rs.uniform(1.0, 2.0, n) * 10.0 ** rs.uniform(-12, 12, n) * rs.choice([-1.0, 1.0], n)
    b = rs.uniform(1.0, 2.0, n) * 10.0 ** rs.uniform(-12, 12, n) * rs.choice([-1.0, 1.0], n)
    # structured cases
    k = n // 8
    b[:k] = np.nextafter(2.0 ** rs.randint(-20, 20, k).astype(float), rs.choice([0.0, 1e300], k))
    a[k:2 * k] = b[k:2 * k] * (1.0 + rs.randint(-4, 5, k) * 2.0 ** -52)
    a[2 * k:3 * k] = b[2 * k:3 * k] * rs.randint(1, 1000, k)
    a[3 * k:4 * k] = 1.0                      # the reciprocals of the layer loop
    for rep in range(2):
        x = np.concatenate([a, b])
        got = _device(2, x, n)[:n]
        want = a / b
        bad = got.view(np.int64) != want.view(np.int64)
        assert not bad.any(), (int(bad.sum()), a[bad][:4], b[bad][:4], got[bad][:4], want[bad][:4])
        a, b = b, a                            # and the other way round
    x = np.abs(np.concatenate([a, b]))
    got = _device(3, x)
    want = np.sqrt(x)
    bad = got.view(np.int64) != want.view(np.int64)
    assert not bad.any(), (int(bad.sum()), x[bad][:4], got[bad][:4], want[bad][:4])
