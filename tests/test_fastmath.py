"""The environment prologue's own log / sincos / division-by-a-shared-divisor (csrc/fastmath.h), compiled for the host
and compared with libm and true division.  The device executes the same fma sequence."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def fm(tmp_path_factory):
    out = tmp_path_factory.mktemp("fastmath") / "libfastmath_host.so"
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-mfma", "-shared", "-fPIC", "-I", os.path.join(ROOT, "grid_fed_rl_gym_amd", "csrc"),
                    os.path.join(ROOT, "tests", "native", "fastmath_host.cpp"), "-o", str(out)], check=True)
    lib = ctypes.CDLL(str(out))
    dp = ctypes.POINTER(ctypes.c_double)
    lib.fm_log01.argtypes = [dp, dp, ctypes.c_long]
    lib.fm_sincos_turns.argtypes = [dp, dp, dp, ctypes.c_long]
    lib.fm_div_by.argtypes = [dp, dp, dp, ctypes.c_long]
    lib.fm_fmod_pos.argtypes = [dp, dp, dp, ctypes.c_long]
    return lib


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def test_log_of_a_uniform(fm):
    rng = np.random.default_rng(0)
    # the generator's uniforms: (k + 1/2) 2^-53, k < 2^53 -- plus the extremes and a log-uniform sweep down to 2^-53
    u = np.concatenate([(rng.integers(0, 1 << 53, 400000).astype(np.float64) + 0.5) / 9007199254740992.0,
                        np.exp(rng.uniform(np.log(2.0 ** -54), 0.0, 400000)),
                        [0.5 / 9007199254740992.0, 1.0 - 0.5 / 9007199254740992.0, 0.5, 0.70710678118654752, 0.7071067811865476, 1.0]])
    out = np.empty_like(u)
    fm.fm_log01(_p(u), _p(out), len(u))
    ref = np.log(u)
    err = np.abs(out - ref) / np.maximum(np.spacing(np.abs(ref)), 5e-324)
    assert np.max(err) <= 2.0, np.max(err)              # ulps (libm itself is within 1)
    assert out[-1] == 0.0


def test_sin_and_cos_of_a_fraction_of_a_turn(fm):
    rng = np.random.default_rng(1)
    t = np.concatenate([rng.uniform(0.0, 1.0, 500000), rng.uniform(-0.5, 0.5, 200000), rng.uniform(-1000.0, 1000.0, 100000),
                        np.arange(-16, 17) / 8.0, [0.5 / 9007199254740992.0, 1.0 - 2.0 ** -53]])
    s, c = np.empty_like(t), np.empty_like(t)
    fm.fm_sincos_turns(_p(t), _p(s), _p(c), len(t))
    # reference in extended precision: the argument 2 pi t is not a double
    tl = t.astype(np.longdouble)
    frac = tl - np.rint(tl)
    rs, rc = np.sin(2.0 * np.pi * frac.astype(np.longdouble)), np.cos(2.0 * np.pi * frac.astype(np.longdouble))
    assert np.max(np.abs(s - rs.astype(np.float64))) < 2.3e-16
    assert np.max(np.abs(c - rc.astype(np.float64))) < 2.3e-16
    exact = np.arange(-16, 17) / 8.0                     # multiples of an eighth of a turn
    k = len(t) - 2 - len(exact)
    assert np.all(np.abs(s[k:k + len(exact)][::2]) + np.abs(c[k:k + len(exact)][::2]) == 1.0)      # quarter turns: exactly 0 / +-1


def test_division_by_a_shared_divisor_is_correctly_rounded(fm):
    rng = np.random.default_rng(2)
    n = 2000000
    x = np.concatenate([rng.standard_normal(n) * 10.0 ** rng.uniform(-12, 12, n), [0.0, -0.0, np.inf, -np.inf, np.nan, 1e308, 5e-324]])
    d = np.concatenate([10.0 ** rng.uniform(-3, 9, n // 2), rng.uniform(0.5, 2.0, n // 2) * 2.0 ** rng.integers(-20, 40, n // 2),
                        [3.0, 3.0, 7.0, 7.0, 2.5, 1e-3, 3.0]])
    out = np.empty_like(x)
    fm.fm_div_by(_p(x), _p(d), _p(out), len(x))
    with np.errstate(all="ignore"):
        ref = x / d
    same = (out == ref) | (np.isnan(out) & np.isnan(ref))
    assert same[:n].all(), (x[:n][~same[:n]][:5], d[:n][~same[:n]][:5])
    assert same[n:n + 6].all()                           # zeros, infinities, NaN, a quotient that overflows


def test_remainder_of_a_non_negative_number_is_exact(fm):
    rng = np.random.default_rng(3)
    n = 1000000
    d = np.concatenate([np.full(n // 2, 24.0), rng.uniform(0.1, 100.0, n // 2)])
    x = np.concatenate([rng.uniform(0.0, 1e6, n // 2), rng.uniform(0.0, 1e9, n // 2)])
    x[:2000] = np.round(x[:2000] / 24.0) * 24.0 + rng.choice([0.0, -3.6e-12, 3.6e-12], 2000)      # at and around whole days
    x = np.abs(x)
    out = np.empty_like(x)
    fm.fm_fmod_pos(_p(x), _p(d), _p(out), len(x))
    assert np.array_equal(out, np.fmod(x, d))
