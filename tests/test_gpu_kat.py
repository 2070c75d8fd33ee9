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


# ---- the device functions against values computed by the reference's OWN SOURCE TEXT -------------------------------------------
# tests/golden/refkat.npz: tools/refkat/make_refkat.py imports /root/reference/{UtilsFunc,Camera}.py, brdf/{Disney,Glass}.py through
# a stand-in for the taichi package (build container only) and calls their functions.  tests/test_refkat.py holds the oracle to the
# same numbers; see its header for what the tolerances mean (a transcription check: discrete outputs exact, values to ~1e-6 of
# the result's magnitude because sin / cos / pow come from different math libraries).
import os                                                                              # noqa: E402
from test_refkat import close, worst                                                   # noqa: E402

RG = np.load(os.path.join(os.path.dirname(__file__), "golden", "refkat.npz"))


def test_reference_text_disney_and_glass(ctx):                                          # brdf/Disney.py:17-108, brdf/Glass.py:9-59
    got = ctx.kat_brdf(0, RG["disney_in"], 2); want = RG["disney_evaluate_pdf"]
    assert np.array_equal(want[:, 1] < 0, got[:, 1] < 0)
    assert close(got[:, :1], want[:, :1], 2e-6), worst(got[:, :1], want[:, :1])
    assert close(got[:, 1:], want[:, 1:], 2e-6), worst(got[:, 1:], want[:, 1:])
    got = ctx.kat_brdf(1, RG["disney_sample_in"], 3); want = RG["disney_sample"]
    assert close(got, want[:, :3], 4e-6), worst(got, want[:, :3])
    got = ctx.kat_brdf(2, RG["glass_sample_in"], 4); want = RG["glass_sample"]
    assert np.array_equal(got[:, 3], want[:, 3])                                        # reflect / refract: the same decision for every sample
    assert close(got[:, :3], want[:, :3], 2e-6), worst(got[:, :3], want[:, :3])
    x = RG["glass_sample_in"]
    rows = np.concatenate([x[:, 10:16], RG["glass_lambda"][:, None], x[:, 16:17]], 1)
    got = ctx.kat_brdf(16, rows, 4); want = RG["glass_sample_lambda"]
    assert np.array_equal(got[:, 3], want[:, 3])
    assert close(got[:, :3], want[:, :3], 2e-6), worst(got[:, :3], want[:, :3])


def test_reference_text_utils(ctx):                                                     # UtilsFunc.py:305-523
    got = ctx.kat_brdf(3, RG["offset_ray_in"], 3)
    assert same_bits(got, RG["offset_ray"])
    got = ctx.kat_brdf(18, RG["slabs_in"], 2)
    assert np.array_equal(got[:, 0].astype(np.int32), RG["slabs"])
    assert np.array_equal(got[:, 1].astype(np.int32), RG["slabs"])                      # slabs_fast where k_trace uses it
    u = RG["u2"]; col = RG["colour_in"]; eta = RG["refract_in"][:, 6]
    assert close(ctx.kat_brdf(4, u, 3), RG["CosineSampleHemisphere"], 2e-6)
    assert close(ctx.kat_brdf(5, u, 2), RG["mapToDisk"], 2e-6)
    assert close(ctx.kat_brdf(6, u * np.float32([7.0, 3.0]), 1), RG["powerHeuristic"][:, None], 1e-6)
    x = RG["inverse_transform_in"].copy()
    x[:, 3:6] *= (np.float32(1.0) + np.float32(0.5) * (np.arange(len(x)) % 3).astype(np.float32))[:, None]
    assert close(ctx.kat_brdf(7, x, 3), RG["inverse_transform"], 2e-6)
    assert close(ctx.kat_brdf(8, col, 3), RG["srgb_to_lrgb"], 2e-6)
    assert close(ctx.kat_brdf(9, col * np.float32(1.5), 3), RG["lrgb_to_srgb"], 2e-6)
    assert close(ctx.kat_brdf(10, col * np.float32(4.0), 3), RG["tone_ACES"], 2e-6)
    assert close(ctx.kat_brdf(11, RG["refract_in"], 4), RG["refract"], 2e-6)
    assert close(ctx.kat_brdf(12, np.stack([u[:, 0], np.float32(1.0) / eta], 1), 1), RG["schlick"][:, None], 2e-6)
    assert close(ctx.kat_brdf(13, np.stack([u[:, 0], np.maximum(np.float32(0.001), u[:, 1])], 1), 1), RG["GTR2"][:, None], 2e-6)
    assert close(ctx.kat_brdf(14, u, 1), RG["smithG_GGX"][:, None], 2e-6)
    assert close(ctx.kat_brdf(15, u[:, :1] * np.float32(1.2) - np.float32(0.1), 1), RG["SchlickFresnel"][:, None], 2e-6)


def test_reference_text_camera(ctx):                                                    # Camera.py:122-142
    vi = RG["camera_view_inv"].reshape(-1); k = RG["camera_fx_fy_cx_cy"]; uv = RG["camera_uv"]; jit = RG["camera_jitter"]
    n = len(uv)
    rows = np.concatenate([np.tile(vi, (n, 1)), np.tile(k, (n, 1)), uv.astype(np.float32), np.zeros((n, 2), np.float32)], 1)
    assert close(ctx.kat_brdf(17, rows, 3), RG["camera_dir_frame0"], 2e-6)
    rows[:, 22:24] = jit - np.float32(0.5)
    assert close(ctx.kat_brdf(17, rows, 3), RG["camera_dir_jittered"], 2e-6)
