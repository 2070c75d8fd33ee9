"""Scene.closet_hit / closet_hit_shadow parity on primary-style rays (no RNG involved):
hit distance, primitive, position, normals and uv bit-identical to the oracle in both
visiting modes; exhaustive-mode N_box / N_leaf counts identical to the oracle's pop counts."""
import numpy as np
import pytest

import oracle_api as oa
from common import duplicate_code_scene, tiny_scene
from ti_raytrace_amd import scenes, _native

pytestmark = pytest.mark.gpu


def bits_equal(a, b):
    return (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))


def check_scene(ex, W, H, extra_rays=None, max_rays=40000):
    ex.build_scene()
    o = oa.OracleScene(ex.scene, ex.cam)
    o.lbvh_build()
    rays = oa.camera_rays(ex.cam, W, H)
    if extra_rays is not None:
        rays = np.concatenate([rays, extra_rays], axis=0)
    if rays.shape[0] > max_rays:
        rays = rays[np.random.RandomState(0).choice(rays.shape[0], max_rays, replace=False)]
    want, wprim, wcnt = o.closest_hit(rays, counts=True)
    ctx = ex.scene.ctx
    for flags in (_native.TRAVERSE_EXHAUSTIVE | _native.COUNT_NODES, _native.TRAVERSE_ORDERED | _native.COUNT_NODES,
                  _native.TRAVERSE_ORDERED):
        got, gprim, gcnt = ctx.trace_closest(rays, 64, flags)
        assert np.array_equal(gprim, wprim), "prim mismatch on %d rays (flags %d)" % ((gprim != wprim).sum(), flags)
        hit = wprim >= 0
        assert bits_equal(got[:, 0], want[:, 0]).all()
        assert bits_equal(got[hit], want[hit]).all(), "attribute mismatch (flags %d)" % flags
        if flags & _native.TRAVERSE_EXHAUSTIVE:
            assert np.array_equal(gcnt, wcnt), "N_box/N_leaf differ from the oracle's pop counts"
        elif gcnt is not None:
            assert (gcnt[:, 0] <= wcnt[:, 0]).all()            # ordered traversal never visits more
    # shadow variant returns (t, prim) of the closest hit
    st, sp, _ = o.shadow_hit(rays)
    gt, gp, _ = ctx.trace_shadow(rays, 64, 0)
    assert np.array_equal(gp, sp) and bits_equal(gt, st).all()
    return rays.shape[0], float((wprim >= 0).mean()), wcnt.mean(axis=0)


def random_rays(n, lo, hi, seed):
    r = np.random.RandomState(seed)
    o = r.uniform(lo, hi, size=(n, 3))
    d = r.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.concatenate([o, d], axis=1).astype(np.float32)


def test_cornell(gpu_ctx_ok):
    ex = scenes.cornell_box(96, 96, 4, device_id=0)
    # rays from inside the box in random directions + axis-parallel rays (slabs' special case)
    inside = random_rays(4000, 50, 500, 1); inside[:, 2] = -np.abs(inside[:, 2])
    axis = inside[:600].copy(); axis[:200, 3:6] = (1, 0, 0); axis[200:400, 3:6] = (0, -1, 0); axis[400:, 3:6] = (0, 0, 1)
    n, frac, cnt = check_scene(ex, 96, 96, np.concatenate([inside, axis]))
    assert frac > 0.5


def test_synthetic_small_and_duplicates(gpu_ctx_ok):
    check_scene(tiny_scene(3000, seed=5, W=64, H=64, spread=0.08, device_id=0), 64, 64, random_rays(3000, -1.5, 1.5, 2))
    check_scene(duplicate_code_scene(W=48, H=48, device_id=0), 48, 48, random_rays(2000, -1.5, 1.5, 3))


def test_single_triangle_and_sphere_only(gpu_ctx_ok):
    check_scene(tiny_scene(1, seed=9, W=32, H=32, spread=0.5, device_id=0), 32, 32, random_rays(500, -2, 4, 4))


def test_teapot_with_smooth_normals(gpu_ctx_ok):
    ex = scenes.single_model(64, 64, 4, device_id=0)
    ex.build_scene()
    o = oa.OracleScene(ex.scene, ex.cam)
    o.lbvh_build(); o.process_normal(ex.scene.vertex_index_np)
    rays = oa.camera_rays(ex.cam, 64, 64)
    want, wprim, _ = o.closest_hit(rays)
    got, gprim, _ = ex.scene.ctx.trace_closest(rays, 64, 0)
    assert np.array_equal(gprim, wprim)
    hit = wprim >= 0
    assert hit.mean() > 0.1
    assert bits_equal(got[hit], want[hit]).all()


def test_headline_100k_primary_rays(gpu_ctx_ok):
    ex = scenes.synthetic(256, 256, 4, device_id=0)
    n, frac, cnt = check_scene(ex, 256, 256, max_rays=30000)
    print("100k scene: hit fraction %.3f, exhaustive N_box %.1f N_leaf %.1f per primary ray" % (frac, cnt[0], cnt[1]))
    assert 0.3 < frac < 0.95
