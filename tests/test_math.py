"""soc_math.h (the fp32 math the HIP kernels and the oracle's soc mode share) against
float64 references: accuracy bound in ulp, exactness of fmod/scale, special values."""
import numpy as np
import pytest

from hostprobe import HostProbe


def ulp_err(got, want64):
    want32 = want64.astype(np.float32)
    ulp = np.spacing(np.abs(want32)).astype(np.float64)
    ulp = np.maximum(ulp, np.float64(np.finfo(np.float32).tiny) * 2 ** -23)
    return np.abs(got.astype(np.float64) - want64) / ulp


@pytest.fixture(scope="module")
def hp():
    return HostProbe()


def test_exp(hp):
    x = np.concatenate([np.linspace(-86.9, 88.0, 200001), -np.logspace(-8, 1.9, 50001)]).astype(np.float32)
    assert ulp_err(hp.math("exp", x), np.exp(x.astype(np.float64))).max() < 1.5
    sp = hp.math("exp", np.array([-1000.0, -87.5, 0.0, 89.0, np.inf, -np.inf], np.float32))
    assert sp[0] == 0 and sp[1] == 0 and sp[2] == 1 and np.isinf(sp[3]) and np.isinf(sp[4]) and sp[5] == 0


def test_log(hp):
    x = np.concatenate([np.logspace(-37, 38, 200001), np.linspace(0.5, 2.0, 100001),
                        (np.arange(1, 2 ** 20, 97) * 2.0 ** -32)]).astype(np.float32)
    assert ulp_err(hp.math("log", x), np.log(x.astype(np.float64))).max() < 2.0
    sp = hp.math("log", np.array([0.0, 1.0, -1.0, np.inf], np.float32))
    assert sp[0] == -np.inf and sp[1] == 0.0 and np.isnan(sp[2]) and sp[3] == np.inf


def test_sincos(hp):
    x = np.linspace(-4 * np.pi, 4 * np.pi, 400001).astype(np.float32)
    s, c = hp.math("sin", x), hp.math("cos", x)
    # absolute accuracy (relative accuracy is lost near zeros of sin/cos, as for any fp32 routine)
    assert np.abs(s - np.sin(x.astype(np.float64))).max() < 2.5e-7
    assert np.abs(c - np.cos(x.astype(np.float64))).max() < 2.5e-7
    assert np.abs(s.astype(np.float64) ** 2 + c.astype(np.float64) ** 2 - 1).max() < 5e-7


def test_acos(hp):
    x = np.linspace(-1, 1, 400001).astype(np.float32)
    assert np.abs(hp.math("acos", x) - np.arccos(x.astype(np.float64))).max() < 5e-7
    sp = hp.math("acos", np.array([1.0, -1.0, 1.5, -1.5], np.float32))
    assert sp[0] == 0 and abs(sp[1] - np.pi) < 1e-6 and sp[2] == 0 and abs(sp[3] - np.pi) < 1e-6


def test_fmod1_exact(hp):
    rng = np.random.default_rng(2)
    x = np.concatenate([rng.uniform(-300, 300, 100000), [0.0, 1.0, -1.0, 255.99998, 1e-4, -1e-4]]).astype(np.float32)
    assert np.array_equal(hp.math("fmod1", x), np.fmod(x, np.float32(1.0)))


def test_sqrt_correctly_rounded(hp):
    x = np.random.default_rng(4).uniform(0, 1e6, 100000).astype(np.float32)
    assert np.array_equal(hp.math("sqrt", x), np.sqrt(x))


def test_expm1_pow15_logd(hp):
    x = -np.concatenate([np.logspace(-30, 1.9, 200001), np.linspace(0.3, 0.4, 20001), [0.0]]).astype(np.float32)
    assert ulp_err(hp.math("expm1", x), np.expm1(x.astype(np.float64))).max() < 2.0
    x = np.linspace(0.15, 2.8, 200001).astype(np.float32)          # 1+g^2-2g*cos, g=0.65
    assert ulp_err(hp.math("pow15", x), x.astype(np.float64) ** 1.5).max() < 1.5
    x = np.concatenate([np.logspace(-37, 38, 200001), 1.0 - np.logspace(-7, -0.01, 100001)]).astype(np.float32)
    want = np.log(x.astype(np.float64))
    assert np.array_equal(hp.math("logd", x), want.astype(np.float32))   # fp64 log rounded to fp32


def test_exp_small_is_exp_on_its_interval(hp):
    """soc_expf_small (used by the brick walk when every lane's argument is small) must give the
    bits of soc_expf on (-0.34, 0]."""
    x = -np.concatenate([np.linspace(0, 0.34, 2000001)[:-1], np.logspace(-30, -0.5, 200001), [0.0]]).astype(np.float32)
    x = x[x > -0.34]
    assert np.array_equal(hp.math("exp_small", x).view(np.uint32), hp.math("exp", x).view(np.uint32))


def test_division_by_cached_reciprocal_is_the_division(hp):
    """soc_div_by_rcp(n, u, 1/u) == n / u bit for bit over the ranges of GetStep
    (numerators (1+PEPS)-frac or -PEPS-frac, |u| in [5e-5, 1])."""
    rng = np.random.default_rng(7)
    m = 4000000
    u = np.exp(rng.uniform(np.log(4.9e-5), 0.0, m)).astype(np.float32) * rng.choice([-1.0, 1.0], m).astype(np.float32)
    frac = rng.uniform(0, 1, m).astype(np.float32)
    frac[: m // 8] = np.exp(rng.uniform(-60, 0, m // 8)).astype(np.float32)
    n = np.where(u > 0, np.float32(1.0 + 1.0e-4) - frac, np.float32(-1.0e-4) - frac).astype(np.float32)
    ones = (np.arange(100, 127, dtype=np.uint32)[:, None] << 23 | np.uint32(0x7fffff)).view(np.float32).ravel()   # all-ones significands
    u = np.concatenate([u, np.repeat(ones, 1000)])
    n = np.concatenate([n, rng.uniform(0.5, 1.0001, ones.size * 1000).astype(np.float32)])
    assert np.array_equal(hp.div_by_rcp(n, u).view(np.uint32), (n / u).view(np.uint32))


def test_atan2(hp):
    rng = np.random.default_rng(9)
    x = rng.standard_normal(400000).astype(np.float32)
    y = rng.standard_normal(400000).astype(np.float32)
    y[:1000] *= np.float32(1e-6)
    x[1000:2000] *= np.float32(1e-6)
    got = hp.atan2(y, x)
    want = np.arctan2(y.astype(np.float64), x.astype(np.float64))
    assert np.abs(got - want).max() < 4.0e-7                      # absolute: angles up to pi
    small = np.abs(want) < 0.5
    assert ulp_err(got[small], want[small]).max() < 2.5
    sp = hp.atan2(np.array([0.0, 1.0, -1.0, 0.0], np.float32), np.array([0.0, 0.0, 0.0, -1.0], np.float32))
    assert sp[0] == 0 and abs(sp[1] - np.pi / 2) < 1e-6 and abs(sp[2] + np.pi / 2) < 1e-6 and abs(sp[3] - np.pi) < 1e-6


def test_oracle_soc_mode_uses_this_header(hp, oracle_soc):
    x = np.random.default_rng(5).uniform(-20, 5, 20000).astype(np.float32)
    for fn, xx in (("exp", x), ("log", np.abs(x) + 1e-9), ("sin", x), ("cos", x), ("acos", np.clip(x / 20, -1, 1)),
                   ("expm1", -np.abs(x)), ("pow15", np.abs(x) + 0.1), ("logd", np.abs(x) + 1e-9)):
        assert np.array_equal(hp.math(fn, xx).view(np.uint32), oracle_soc.math(fn, xx).view(np.uint32))
