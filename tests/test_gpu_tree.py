"""The traversal tree (ti_raytrace_amd/csrc/tirt_sah.hip): the binned-SAH binary tree the ordered traversal walks instead of
the reference's LBVH (accel/LBvh.py:229-467).  It may be any hierarchy over the same primitives -- every candidate hit is
verified against the reference tree (Scene.py:702-744) -- but it has to BE one: every primitive in exactly one leaf, every
box the union of its children's, pre-order layout.  And the option must not change a single bit of any result."""
import numpy as np
import pytest

import oracle_api as oa
from ti_raytrace_amd import scenes, _native
from common import duplicate_code_scene, tiny_scene
from test_gpu_trace import random_rays

pytestmark = pytest.mark.gpu


def prim_boxes(sc):
    P = sc.primitive_np
    V = sc.vertex_np[:, :3].astype(np.float32)
    n = P.shape[0]
    box = np.zeros((n, 6), np.float32)
    tri = P[:, 0] == 1
    vi = P[tri, 1]
    tv = np.stack([V[vi], V[vi + 1], V[vi + 2]], axis=1)
    box[tri, :3] = tv.min(axis=1); box[tri, 3:] = tv.max(axis=1)
    for i in np.flatnonzero(~tri):
        sh = np.asarray(sc.shape_np, np.float32).reshape(-1, 10)[P[i, 1]]
        box[i, :3] = sh[1:4] - sh[4]; box[i, 3:] = sh[1:4] + sh[4]
    return box


def check_tree(rows, box, shape_prims=None):
    n = box.shape[0]
    N = 2 * n - 1
    assert rows.shape == (N, 9)
    leaf = rows[:, 0] == 1.0
    assert leaf.sum() == n and ((rows[:, 0] == 0.0) | leaf).all()
    prims = rows[leaf, 1].astype(np.int64)
    assert np.array_equal(np.sort(prims), np.arange(n)), "every primitive in exactly one leaf"
    tri = np.ones(n, bool) if shape_prims is None else ~np.isin(np.arange(n), shape_prims)
    lb, pb = rows[leaf, 2:8], box[prims]
    assert np.array_equal(lb[tri[prims]], pb[tri[prims]]), "leaf boxes are the triangles' boxes"
    sh = ~tri[prims]          # analytic spheres: the tree holds their box padded by what the reference's sphere test can be off by (tirt_internal.h, sphere_pad)
    if sh.any():
        assert (lb[sh, :3] <= pb[sh, :3]).all() and (lb[sh, 3:] >= pb[sh, 3:]).all()
        r = 0.5 * (pb[sh, 3] - pb[sh, 0])
        diag = np.linalg.norm(rows[0, 5:8] - rows[0, 2:5])         # (the padded root: an upper bound of the scene diagonal the pad is sized by)
        pad_max = 1.0e-3 * r + 2.4e-7 * (16.0 * diag) ** 2 / r
        assert (np.abs(lb[sh] - pb[sh]) <= 1.01 * pad_max[:, None] + 1e-6).all(), "padded by no more than sphere_pad"

    # pre-order: subtree sizes bottom-up (children have larger indices than their parent)
    size = np.ones(N, np.int64)
    right = rows[:, 1].astype(np.int64)
    for i in range(N - 1, -1, -1):
        if not leaf[i]:
            l, r = i + 1, right[i]
            assert i + 1 < r < N, "right child after the left subtree"
            assert l + size[l] == r, "left subtree ends where the right one begins"
            size[i] = 1 + size[l] + size[r]
            assert np.array_equal(rows[i, 2:5], np.minimum(rows[l, 2:5], rows[r, 2:5]))
            assert np.array_equal(rows[i, 5:8], np.maximum(rows[l, 5:8], rows[r, 5:8]))
    assert size[0] == N
    return size


@pytest.mark.parametrize("make", [
    lambda: scenes.cornell_box(32, 32, 4, device_id=0),
    lambda: scenes.single_model(32, 32, 4, device_id=0),
    lambda: tiny_scene(3000, seed=5, W=32, H=32, spread=0.08, device_id=0),
    lambda: tiny_scene(2, seed=3, W=16, H=16, spread=0.5, device_id=0),
    lambda: tiny_scene(3, seed=4, W=16, H=16, spread=0.5, device_id=0),
    lambda: tiny_scene(5000, seed=6, W=16, H=16, spread=0.0001, device_id=0),     # 5000 primitives on a handful of points: range halving
    lambda: duplicate_code_scene(W=16, H=16, device_id=0),
])
def test_traversal_tree_is_a_tree(gpu_ctx_ok, make):
    ex = make(); ex.build_scene()
    sc = ex.scene
    rows = sc.ctx.traversal_tree_download(sc.primitive_count)
    check_tree(rows, prim_boxes(sc), np.flatnonzero(sc.primitive_np[:, 0] != 1))
    # with the option off the rows are the reference's compact_node
    ex2 = make(); ex2.scene.ctx.set_option("traversal_tree", 0); ex2.build_scene()
    lb = ex2.scene.ctx.traversal_tree_download(sc.primitive_count)
    _, _, compact = ex2.scene.ctx.lbvh_download(sc.primitive_count)
    assert np.array_equal(lb.view(np.uint32), compact.view(np.uint32))


def test_traversal_tree_headline_scene(gpu_ctx_ok):
    ex = scenes.synthetic(64, 64, 4, device_id=0); ex.build_scene()
    sc = ex.scene
    size = check_tree(sc.ctx.traversal_tree_download(sc.primitive_count), prim_boxes(sc), np.flatnonzero(sc.primitive_np[:, 0] != 1))
    # a top-down SAH tree of 100 001 primitives is shallow: depth well below the 64 levels after which ranges are halved
    depth = np.zeros(size.shape[0], np.int32)
    rows = sc.ctx.traversal_tree_download(sc.primitive_count)
    for i in range(size.shape[0]):
        if rows[i, 0] == 0.0:
            depth[i + 1] = depth[i] + 1; depth[int(rows[i, 1])] = depth[i] + 1
    assert depth.max() < 64


@pytest.mark.parametrize("make,W", [
    (lambda W: scenes.cornell_box(W, W, 4, device_id=0), 96),
    (lambda W: scenes.single_model(W, W, 4, device_id=0), 96),
    (lambda W: scenes.veach_bdpt(W, W, 4, device_id=0), 96),
    (lambda W: scenes.synthetic(W, W, 4, ntri=20000, device_id=0), 128),
])
def test_traversal_tree_option_changes_no_bit(gpu_ctx_ok, make, W):
    """films of 8 frames and the hit records of 60 000 rays, option on and off"""
    out = []
    for tree in (0, 1):
        ex = make(W); ex.scene.ctx.set_option("traversal_tree", tree); ex.build_scene()
        ctx = ex.scene.ctx
        ctx.pt_rgb_render(0, 8, 1, 15, 64, 0)
        film = ctx.film_download(W, W)[0]
        lo, hi = ex.scene.minboundarynp.reshape(-1), ex.scene.maxboundarynp.reshape(-1)
        ext = float((hi - lo).max())
        rays = np.concatenate([oa.camera_rays(ex.cam, W, W), random_rays(40000, float(lo.min()) - 0.2 * ext, float(hi.max()) + 0.2 * ext, 11)], axis=0)
        hit, prim, _ = ctx.trace_closest(rays, 64, 0)
        st, sp, _ = ctx.trace_shadow(rays, 64, 0)
        out.append((film, hit, prim, st, sp))
    for a, b in zip(*out):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_cost_optimal_collapse_changes_no_bit(gpu_ctx_ok, experiments_lib):
    """option wide_collapse = 1 (k_wide_dp): another grouping of the same binary tree into 4-wide nodes"""
    out = []
    W = 96
    for collapse in (0, 1):
        ex = scenes.veach_bdpt(W, W, 4, device_id=0); ex.scene.ctx.set_option("wide_collapse", collapse); ex.build_scene()
        ctx = ex.scene.ctx
        ctx.pt_rgb_render(0, 8, 1, 15, 64, 0)
        film = ctx.film_download(W, W)[0]
        rays = np.concatenate([oa.camera_rays(ex.cam, W, W), random_rays(40000, -3.0, 3.0, 12)], axis=0)
        hit, prim, _ = ctx.trace_closest(rays, 64, 0)
        out.append((film, hit, prim, ctx.bvh_info()["nodes"]))
    assert out[0][3] != out[1][3], "the two groupings differ"
    for a, b in zip(out[0][:3], out[1][:3]):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def _custom_scene(tris, W=24, H=24):
    from ti_raytrace_amd import Example, PT_RGB
    from ti_raytrace_amd import SceneData as SCD
    ex = Example.example(W, H, 4, 0)
    mat = SCD.Material()
    mat.type = SCD.MAT_DISNEY
    mat.setMetal(0.0); mat.setRough(0.5); mat.setColor([0.8, 0.8, 0.8, 1.0]); mat.alebdoTex = -1
    ex.scene.add_mesh(np.asarray(tris, np.float64), mat)
    ex.add_sphere_light(pos=(0.0, 3.0, 0.0), radius=0.75, emission=50.0)
    ex.integrator = PT_RGB.PathTrace(W, H, ex.cam, ex.scene, 64)
    return ex


@pytest.mark.parametrize("kind", ["exponential", "identical", "two_clusters"])
def test_traversal_tree_on_hostile_distributions(gpu_ctx_ok, kind):
    """Inputs a binned SAH build handles badly: centroids spaced exponentially (every split peels off a few primitives: the tree
    is as deep as the 64 levels after which ranges are halved), 3000 identical triangles (no plane separates anything: halving
    from the root), two far-apart clusters of very different size.  The tree must still be a tree and the hits the oracle's."""
    r = np.random.RandomState(7)
    if kind == "exponential":
        n = 300
        c = np.zeros((n, 3)); c[:, 0] = 1.05 ** np.arange(n) * 1e-3; c[:, 1] = r.uniform(-1, 1, n) * c[:, 0]
        tris = c[:, None, :] + r.uniform(-0.2, 0.2, (n, 3, 3)) * c[:, 0][:, None, None]
    elif kind == "identical":
        one = r.uniform(-1, 1, (3, 3))
        tris = np.repeat(one[None], 3000, axis=0)
    else:
        a = r.uniform(-1, 1, (5000, 1, 3)) * 0.01 + r.uniform(-0.001, 0.001, (5000, 3, 3))
        b = r.uniform(-1, 1, (40, 1, 3)) * 50.0 + 1000.0 + r.uniform(-5, 5, (40, 3, 3))
        tris = np.concatenate([a, b], axis=0)
    ex = _custom_scene(tris); ex.build_scene()
    sc = ex.scene
    check_tree(sc.ctx.traversal_tree_download(sc.primitive_count), prim_boxes(sc), np.flatnonzero(sc.primitive_np[:, 0] != 1))
    o = oa.OracleScene(sc, ex.cam); o.lbvh_build()
    lo, hi = np.asarray(tris).reshape(-1, 3).min(axis=0), np.asarray(tris).reshape(-1, 3).max(axis=0)
    org = r.uniform(lo - 0.1 * (hi - lo) - 1e-3, hi + 0.1 * (hi - lo) + 1e-3, (6000, 3))
    T = np.asarray(tris)[r.randint(0, len(tris), 6000)]                     # aimed at a point inside a triangle
    w = r.dirichlet((1.0, 1.0, 1.0), 6000)
    tgt = (T * w[:, :, None]).sum(axis=1)
    d = tgt - org; d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([org, d], axis=1).astype(np.float32)
    want, wprim, _ = o.closest_hit(rays, stack_size=16384)      # (5000 primitives in one Morton cell: a chain as deep in the reference tree)
    got, gprim, _ = sc.ctx.trace_closest(rays, 64, 0)
    assert (wprim >= 0).mean() > 0.02        # (aimed from far away in fp32: the small cluster is mostly missed, in the oracle as on the device)
    assert np.array_equal(gprim, wprim) and np.array_equal(got[:, 0].view(np.uint32), want[:, 0].view(np.uint32))
    assert sc.ctx.stats()["stack_overflow"] == 0
