"""The arithmetic contract on the device, primitive by primitive: the GPU's
min/max/sqrt/divide/round-to-int must equal the oracle's definitions bit for bit on special
values, and the kernels' short correctly-rounded sqrt must equal the generic one on EVERY float."""
import ctypes as C

import numpy as np
import pytest

from oracle import rm_oracle_np as onp
from ray_marching_amd import _ffi, renderer

pytestmark = pytest.mark.gpu
F = np.float32


@pytest.fixture(scope="module")
def res():
    r = renderer.RayMarchingResources(0)
    yield r
    r.close()


def test_short_sqrt_is_correctly_rounded_for_every_float(res):
    bad, first = C.c_uint64(1), C.c_uint32(0)
    _ffi.check(res._h, _ffi.hip_lib().rm_selftest_sqrt(res._h, C.byref(bad), C.byref(first)))
    assert bad.value == 0, "first mismatching bit pattern: 0x%08x" % first.value


def _special_values():
    sp = [0.0, -0.0, 1.0, -1.0, 0.5, 2.5, 1.5, -2.5, 3.5, 1e-45, -1e-45, 1e-38, 1.17549435e-38, 3.4028235e38,
          -3.4028235e38, np.inf, -np.inf, np.nan, 0.01, 100.0, 2147483520.0, 2147483648.0, -2147483648.0, 4294967296.0,
          8388607.5, -8388607.5, 16777216.0, 0.49999997, 1e-30, 7.888609e-31, 7.8886e-31, 2.0**-96, 2.0**-97]
    a, b = np.meshgrid(np.array(sp, dtype=F), np.array(sp, dtype=F))
    rng = np.random.default_rng(5)
    ra = rng.standard_normal(4096).astype(F) * F(10)
    rb = rng.standard_normal(4096).astype(F) * F(10)
    bits = rng.integers(0, 2**32, size=4096, dtype=np.uint64).astype(np.uint32)
    return (np.concatenate([a.ravel(), ra, bits.view(F)]).astype(F),
            np.concatenate([b.ravel(), rb, np.roll(bits, 1).view(F)]).astype(F))


def _same(got, exp):
    got, exp = np.asarray(got, F), np.asarray(exp, F)
    nan = np.isnan(got) & np.isnan(exp)
    return nan | (got.view(np.uint32) == exp.view(np.uint32))


def test_primitive_ops_match_the_oracle_definitions(res):
    a, b = _special_values()
    n = a.size
    out = np.zeros((8, n), dtype=F)
    _ffi.check(res._h, _ffi.hip_lib().rm_selftest_ops(res._h, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p),
                                                       out.ctypes.data_as(C.c_void_p), n))
    # The interpreter issues v_min_f32 / v_max_f32 directly only on arithmetic RESULTS, which are never
    # signalling NaNs; with a raw signalling NaN operand the instruction (IEEE mode) returns a quiet NaN
    # instead of the other operand, so those inputs are outside the direct forms' contract.
    def snan(x):
        u = x.view(np.uint32)
        return np.isnan(x) & ((u & np.uint32(0x00400000)) == 0)
    raw_snan = snan(a) | snan(b)
    assert raw_snan.any() and (~raw_snan).sum() > 5000
    with np.errstate(all="ignore"):
        exp_min, exp_max = onp.fmin(a, b), onp.fmax(a, b)
        checks = {
            "min (-0 < +0, NaN loses)": _same(out[0], exp_min),
            "max": _same(out[1], exp_max),
            "direct v_min_f32": _same(out[2], exp_min) | raw_snan,
            "direct v_max_f32(a,-b)": _same(out[3], onp.fmax(a, -b)) | raw_snan,
            "short sqrt": _same(out[4], np.sqrt(a)),
            "generic sqrt": _same(out[5], np.sqrt(a)),
            "divide": _same(out[6], a / b),
            "i32(round())": _same(out[7], onp.f2i(np.rint(a)).astype(F)),
        }
    for name, ok in checks.items():
        bad = np.nonzero(~ok)[0]
        assert bad.size == 0, "%s differs for a=%r b=%r" % (name, a[bad[:4]], b[bad[:4]])


def test_cross_lane_primitives_of_wave_level_culling(res):
    """wave_max_u32 (quad_perm / row_half_mirror / row_mirror + four readlanes) and wave_exclusive_min (row_shr 1,2,3,4,8 +
    row_bcast 15 / 31 + one shift) -- the DPP row operations wave-level culling builds its ball and its prefix bounds from --
    against numpy on random and adversarial waves: the maximum anywhere in the wave, a NaN among the inputs (its bit pattern
    wins the maximum: nothing may then be culled; the minimum skips it), +inf, runs of equal values, one value per lane."""
    rng = np.random.default_rng(99)
    waves = []
    for k in range(64):                                   # the maximum / a unique minimum at every lane position
        w = rng.uniform(1.0, 2.0, 64).astype(F)
        w[k] = 3.0
        w[(k * 7 + 3) % 64] = 0.25
        waves.append(w)
    for _ in range(200):
        w = np.abs(rng.normal(size=64)).astype(F) * F(rng.choice([1e-3, 1.0, 1e6]))
        sel = rng.random(64)
        w[sel < 0.05] = np.inf
        w[(sel > 0.05) & (sel < 0.08)] = 0.0
        if rng.random() < 0.3:
            w[rng.integers(64)] = np.nan
        waves.append(w)
    waves.append(np.full(64, np.inf, F))
    waves.append(np.arange(64, 0, -1).astype(F))         # strictly decreasing: every lane is a new minimum
    waves.append(np.arange(64).astype(F))
    a = np.ascontiguousarray(np.stack(waves))
    n = a.shape[0]
    out = np.zeros((2, n, 64), dtype=F)
    _ffi.check(res._h, _ffi.hip_lib().rm_selftest_wave(res._h, a.ctypes.data_as(C.c_void_p), n, out.ctypes.data_as(C.c_void_p)))
    exp_max = a.view(np.uint32).max(axis=1)               # by bit pattern: non-negative floats order like their bits, a NaN is above +inf
    assert (out[0].view(np.uint32) == exp_max[:, None]).all()
    with np.errstate(all="ignore"):
        run = np.fmin.accumulate(a, axis=1)              # fmin skips a NaN
    exp_min = np.concatenate([np.full((n, 1), np.inf, F), run[:, :-1]], axis=1)
    exp_min = np.where(np.isnan(exp_min), np.inf, exp_min).astype(F)     # (a wave that starts with a NaN: nothing below but NaNs)
    assert _same(out[1], exp_min).all(), np.argwhere(~_same(out[1], exp_min))[:5]
