"""Device evaluation of tirt_math.h is bit-identical to the host evaluation (the premise of
fixed-seed image parity) and the RNG streams agree."""
import numpy as np
import pytest

import oracle_api as oa
from ti_raytrace_amd import _native

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(gpu_ctx_ok):
    c = _native.Context(0)
    yield c
    c.close()


def host_math(fn, x, y):
    L = oa.load()
    out = np.zeros_like(x)
    L.orc_kat_math(fn, x, y, out, x.size)
    return out


CASES = [
    (0, "sin", lambda r: (r.uniform(-7, 7, 200000).astype(np.float32), None)),
    (1, "cos", lambda r: (r.uniform(-7, 7, 200000).astype(np.float32), None)),
    (2, "exp", lambda r: (r.uniform(-100, 20, 200000).astype(np.float32), None)),
    (3, "log", lambda r: (np.abs(r.standard_cauchy(200000)).astype(np.float32) + np.float32(1e-30), None)),
    (4, "pow", lambda r: (r.uniform(0, 2, 200000).astype(np.float32), r.choice([2.4, 5.0, 1 / 2.4, 0.5, 3.0], 200000).astype(np.float32))),
    (5, "atan2", lambda r: (r.uniform(-1, 1, 200000).astype(np.float32), r.uniform(-1, 1, 200000).astype(np.float32))),
    (6, "acos", lambda r: (r.uniform(-1.001, 1.001, 200000).astype(np.float32), None)),
    (7, "sqrt", lambda r: (r.uniform(0, 1e6, 200000).astype(np.float32), None)),
    (8, "div", lambda r: (r.uniform(-10, 10, 200000).astype(np.float32), r.uniform(-10, 10, 200000).astype(np.float32))),
]


@pytest.mark.parametrize("fn,name,gen", CASES, ids=[c[1] for c in CASES])
def test_bit_exact(ctx, fn, name, gen):
    x, y = gen(np.random.RandomState(fn + 11))
    if y is None:
        y = np.zeros_like(x)
    dev = ctx.kat_math(fn, x, y)
    host = host_math(fn, x, y)
    same = (dev.view(np.uint32) == host.view(np.uint32)) | (np.isnan(dev) & np.isnan(host))
    assert same.all(), "%s: %d / %d differ, first %r" % (name, (~same).sum(), x.size, x[~same][:4])


def test_sincos_matches_separate_calls(ctx):
    x = np.random.RandomState(5).uniform(-7, 7, 100000).astype(np.float32)
    z = np.zeros_like(x)
    assert np.array_equal(ctx.kat_math(10, x, z).view(np.uint32), host_math(0, x, z).view(np.uint32))
    assert np.array_equal(ctx.kat_math(11, x, z).view(np.uint32), host_math(1, x, z).view(np.uint32))


def test_special_values(ctx):
    x = np.array([0.0, -0.001, 1.0, 0.0, np.inf, -1.0, 4.0], np.float32)
    y = np.array([2.4, 5.0, 0.0, 0.0, 2.0, 0.5, 0.5], np.float32)
    dev = ctx.kat_math(4, x, y)
    host = host_math(4, x, y)
    assert np.array_equal(dev.view(np.uint32)[~np.isnan(host)], host.view(np.uint32)[~np.isnan(host)])
    assert np.isnan(dev[np.isnan(host)]).all()


def test_rng_matches_host(ctx):
    seeds = np.arange(1000, dtype=np.uint32)
    pix = (seeds * np.uint32(7919)) ^ np.uint32(0xABCDEF)
    dev = ctx.kat_math(9, seeds.view(np.float32), pix.view(np.float32))
    L = oa.load()
    host = np.array([L.orc_kat_rand(int(s), int(p), 3, 5) for s, p in zip(seeds, pix)], np.float32)
    assert np.array_equal(dev, host)
    assert 0.0 <= dev.min() and dev.max() < 1.0
