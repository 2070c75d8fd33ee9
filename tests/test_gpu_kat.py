"""Known-answer tests per device function against the oracle's scalar entry points (bit-exact):
Disney.evaluate_pdf / sample, Glass.sample, UF.offset_ray at random and hand-made inputs; slabs /
Moller-Trumbore on hand-made rays through the batch closest-hit entry point."""
import numpy as np
import pytest

import oracle_api as oa
from ti_raytrace_amd import _native, Example, PT_RGB
from ti_raytrace_amd import SceneData as SCD

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(gpu_ctx_ok):
    c = _native.Context(0)
    yield c
    c.close()


def unit(r, n):
    v = r.normal(size=(n, 3)); return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)


def materials(r, n):
    m = np.zeros((n, 10), np.float32)
    m[:, 2:5] = r.uniform(0, 1, size=(n, 3))
    m[:, 5] = r.choice([0.0, 0.3, 1.0, 1.3, 1.5], n)          # metallic | ior
    m[:, 6] = r.choice([0.0, 0.001, 0.2, 0.5, 1.0, 5.0], n)   # roughness | extinction
    return m


def same_bits(a, b):
    return ((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all()


def test_disney_evaluate_pdf(ctx):
    r = np.random.RandomState(1); n = 20000
    m = materials(r, n); m[:, 5] = r.choice([0.0, 0.3, 1.0], n)
    N, V, L = unit(r, n), unit(r, n), unit(r, n)
    V[: n // 2] = np.where((np.sum(V[: n // 2] * N[: n // 2], 1) < 0)[:, None], -V[: n // 2], V[: n // 2])     # half in the upper hemisphere
    L[: n // 2] = np.where((np.sum(L[: n // 2] * N[: n // 2], 1) < 0)[:, None], -L[: n // 2], L[: n // 2])
    got = ctx.kat_brdf(0, np.concatenate([m, N, V, L], 1), 2)
    lib = oa.load(); want = np.zeros((n, 2), np.float32)
    for i in range(n):
        lib.orc_kat_disney(m[i], N[i], V[i], L[i], want[i])
    assert same_bits(got, want)
    assert (want[:, 1] > 0).mean() > 0.3 and (want[:, 1] < 0).mean() > 0.3


def test_disney_and_glass_sample(ctx):
    r = np.random.RandomState(2); n = 20000
    m = materials(r, n)
    d, N = unit(r, n), unit(r, n)
    rnd = r.uniform(0, 1, size=(n, 3)).astype(np.float32)
    lib = oa.load()
    md = m.copy(); md[:, 5] = r.choice([0.0, 0.5, 1.0], n)
    got = ctx.kat_brdf(1, np.concatenate([md, d, N, rnd], 1), 3)
    want = np.zeros((n, 3), np.float32)
    for i in range(n):
        lib.orc_kat_disney_sample(md[i], d[i], N[i], rnd[i], want[i])
    assert same_bits(got, want)
    mg = m.copy(); mg[:, 5] = r.choice([1.0, 1.3, 1.5, 2.4], n)
    got = ctx.kat_brdf(2, np.concatenate([mg, d, N, rnd[:, :1]], 1), 4)
    want = np.zeros((n, 4), np.float32)
    for i in range(n):
        lib.orc_kat_glass_sample(mg[i], d[i], N[i], float(rnd[i, 0]), want[i])
    assert same_bits(got, want)
    assert set(np.unique(want[:, 3]).tolist()) == {-1.0, 1.0}      # both reflection and refraction occur


def test_offset_ray(ctx):
    r = np.random.RandomState(3); n = 20000
    p = (r.normal(size=(n, 3)) * r.choice([1e-4, 1e-2, 1.0, 500.0], (n, 1))).astype(np.float32)
    p[:50] = 0.0; p[50:100, 0] = np.float32(1.0 / 256.0)
    nn = unit(r, n) * r.choice([-1.0, 1.0, 0.0], (n, 1)).astype(np.float32)
    got = ctx.kat_brdf(3, np.concatenate([p, nn], 1), 3)
    lib = oa.load(); want = np.zeros((n, 3), np.float32)
    for i in range(n):
        lib.orc_kat_offset_ray(p[i], nn[i], want[i])
    assert same_bits(got, want)


def test_hand_made_rays_on_two_triangles(gpu_ctx_ok):
    """slabs + Moller-Trumbore edge cases: edge and vertex hits, exact-t tie between two coplanar
    triangles (the later compact index wins), axis-parallel rays, origin on the plane, grazing."""
    ex = Example.example(16, 16, 4, 0)
    mat = SCD.Material(); mat.type = SCD.MAT_DISNEY; mat.setRough(0.5); mat.setColor([0.8, 0.8, 0.8, 1.0]); mat.alebdoTex = -1
    tris = np.array([[[0, 0, 0], [1, 0, 0], [0, 1, 0]],
                     [[1, 1, 0], [0, 1, 0], [1, 0, 0]],          # shares the diagonal edge with the first
                     [[0, 0, 0], [1, 0, 0], [0, 1, 0]],          # exact duplicate of the first: t ties
                     [[0, 0, -1], [1, 0, -1], [0, 1, -1]]], dtype=np.float64)
    ex.scene.add_mesh(tris, mat)
    ex.integrator = PT_RGB.PathTrace(16, 16, ex.cam, ex.scene, 64)
    ex.build_scene(); ex.frame_camera(0.8)
    o = oa.OracleScene(ex.scene, ex.cam); o.lbvh_build()
    rays = np.array([
        [0.25, 0.25, 1, 0, 0, -1],      # interior of the duplicated triangle (tie)
        [0.5, 0.5, 1, 0, 0, -1],        # on the shared diagonal edge
        [0, 0, 1, 0, 0, -1],            # exactly through a vertex
        [1, 1, 1, 0, 0, -1],            # far vertex
        [0.25, 0.25, 0, 0, 0, -1],      # origin on the plane: t = 0 is rejected, hits the triangle behind
        [0.25, 0.25, -2, 0, 0, 1],      # from below, back faces
        [-1, 0.25, 0, 1, 0, 0],         # in-plane, axis-parallel: grazing
        [2, 2, 1, 0, 0, -1],            # miss
        [0.3, 0.3, 1, 1e-7, 0, -1],     # |d.x| below the slabs parallel threshold after normalisation
    ], dtype=np.float32)
    rays[:, 3:6] /= np.linalg.norm(rays[:, 3:6], axis=1, keepdims=True)
    want, wprim, wcnt = o.closest_hit(rays, counts=True)
    for flags in (_native.TRAVERSE_EXHAUSTIVE | _native.COUNT_NODES, _native.TRAVERSE_ORDERED):
        got, gprim, gcnt = ex.scene.ctx.trace_closest(rays, 64, flags)
        assert np.array_equal(gprim, wprim), (gprim, wprim)
        assert same_bits(got[:, 0], want[:, 0])
        assert same_bits(got[wprim >= 0], want[wprim >= 0])
        if gcnt is not None:
            assert np.array_equal(gcnt, wcnt)
    assert wprim[0] in (0, 2) and wprim[7] == -1 and wprim[4] == 3
