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
    if not ex.cam.view_inv_np.any():          # plain Example.example: nobody framed the camera yet
        ex.frame_camera(0.8)
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
        # (no assertion on the ordered counts: far-away origins carry a wide grid margin and test MORE primitives than
        # the reference order does; the counts of rays from inside or near the scene are what bench.py reports)
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
    # The ordered traversal culls: for rays that start near the scene (camera rays) it must test far fewer boxes and primitives than
    # the reference's exhaustive order -- a bounded check that a regression in the culling (margins, cull distance, tree quality) trips.
    # (Far-away origins are exempt by design: wide grid margin, no distance culling beyond TR_FAR_RHO extents.)
    rays = oa.camera_rays(ex.cam, 256, 256)[::3]
    _, _, oc = ex.scene.ctx.trace_closest(rays, 64, _native.TRAVERSE_ORDERED | _native.COUNT_NODES)
    _, _, ec = ex.scene.ctx.trace_closest(rays, 64, _native.TRAVERSE_EXHAUSTIVE | _native.COUNT_NODES)
    on, ol = oc.mean(axis=0); en, el = ec.mean(axis=0)
    print("camera rays: ordered %.1f box tests / %.2f primitive tests per ray, reference order %.1f / %.2f" % (on, ol, en, el))
    assert ol <= 0.5 * el and on <= 0.6 * en and ol < 12.0 and on < 160.0


def test_deep_duplicate_chain_uses_the_spill_stack(gpu_ctx_ok):
    """48 triangles with the same centroid (one Morton code) form a 48-deep chain under the
    reference's duplicate rule (accel/LBvh.py:240-251): deeper than the 24-entry LDS stack, so the
    global spill tail of the traversal stack is exercised.  Hits must still equal the oracle's."""
    from ti_raytrace_amd import Example, PT_RGB
    from ti_raytrace_amd import SceneData as SCD
    W = H = 48
    ex = Example.example(W, H, 4, 0)
    mat = SCD.Material(); mat.type = SCD.MAT_DISNEY; mat.setRough(0.5); mat.setColor([0.8, 0.8, 0.8, 1.0]); mat.alebdoTex = -1
    r = np.random.RandomState(11)
    tris = []
    for k in range(48):                       # nested, slightly tilted triangles around the origin: centroid exactly (0,0,0)
        a = r.uniform(0.3, 1.0); th = r.uniform(0, 2 * np.pi); tilt = r.uniform(-0.3, 0.3)
        p = np.array([[np.cos(th + 2 * np.pi * j / 3) * a, np.sin(th + 2 * np.pi * j / 3) * a, 0.0] for j in range(3)])
        p[:, 2] = tilt * p[:, 0]
        p -= p.mean(axis=0, keepdims=True)
        tris.append(p)
    for k in range(40):                       # some ordinary geometry around it
        c = r.uniform(-1.5, 1.5, size=3); tris.append(c[None, :] + r.uniform(-0.2, 0.2, size=(3, 3)))
    ex.scene.add_mesh(np.asarray(tris), mat)
    ex.add_sphere_light(pos=(0.0, 3.0, 0.0), radius=0.5, emission=30.0)
    ex.integrator = PT_RGB.PathTrace(W, H, ex.cam, ex.scene, 64)
    n, frac, cnt = check_scene(ex, W, H, random_rays(4000, -1.5, 1.5, 5))
    o = oa.OracleScene(ex.scene, ex.cam); o.lbvh_build()
    codes = o.lbvh_get()[0][:, 0]
    runs = np.diff(np.flatnonzero(np.diff(codes) != 0)).max()
    assert runs >= 40                        # the chain exists
    st = ex.scene.ctx.stats()
    assert st["stack_overflow"] == 0
    # and the image still matches
    ex.integrator.render_frames(2)
    got = ex.integrator.hdr.to_numpy()
    want, ost = o.render(W, H, 0, 2, seed=ex.integrator.seed)
    assert ost["overflow"] == 0 and np.array_equal(got, want)


def test_stack_paging_with_a_minimal_lds_stack(gpu_ctx_ok):
    """trace_lds_depth=12 leaves 8 usable LDS entries before a lane pages its oldest 8 entries out to the
    global spill buffer (and back in when it pops the sentinel): on the 100k scene that happens all the
    time.  Hits, in both visiting modes, and a rendered film must not change."""
    ex = scenes.synthetic(128, 128, 4, device_id=0)
    ex.build_scene()
    ctx = ex.scene.ctx
    rays = np.concatenate([oa.camera_rays(ex.cam, 128, 128), random_rays(20000, -0.2, 1.2, 7)], axis=0)
    ref_t, ref_p, _ = ctx.trace_closest(rays, 64, 0)
    ex.integrator.render_frames(2)
    ref_film = ex.integrator.hdr.to_numpy().copy()
    ctx.film_clear()
    ctx.set_option("trace_lds_depth", 12)
    for flags in (0, _native.TRAVERSE_EXHAUSTIVE):
        got_t, got_p, _ = ctx.trace_closest(rays, 64, flags)
        assert np.array_equal(got_p, ref_p) and bits_equal(got_t, ref_t).all()
    ex.integrator.render_frames(2)
    assert np.array_equal(ex.integrator.hdr.to_numpy(), ref_film)
    assert ctx.stats()["stack_overflow"] == 0
    ctx.set_option("trace_lds_depth", 24)


def test_four_wide_nodes_on_unbalanced_trees(gpu_ctx_ok):
    """Scenes whose LBVH has leaves at every depth parity (2, 3, 5, 6, 7, 9 primitives): empty slots,
    leaf children next to collapsed ones, a root whose child is a leaf."""
    for n in (2, 3, 5, 6, 7, 9, 17):
        check_scene(tiny_scene(n, seed=20 + n, W=24, H=24, spread=0.6, device_id=0), 24, 24, random_rays(800, -2, 3, n))


def _grazing_rays(ex, n, seed):
    """Rays that run along the boundaries the quantised traversal must not get wrong: aimed exactly at triangle
    vertices and edge points (they graze the leaf boxes and every ancestor box the vertex is extreme in), rays inside
    the plane of axis-aligned triangles, far-away origins (the grid margin grows with the distance), origins exactly on
    triangle vertices, axis-parallel rays through vertices."""
    r = np.random.RandomState(seed)
    sc = ex.scene
    tri = sc.primitive_np[sc.primitive_np[:, 0] == 1]
    v = sc.vertex_np[:, :3].astype(np.float64)
    lo, hi = v.min(axis=0), v.max(axis=0)
    ext = float((hi - lo).max())
    pick = tri[r.randint(0, tri.shape[0], n), 1]
    corner = v[pick + r.randint(0, 3, n)]
    a, b = v[pick], v[pick + 1]
    w = r.uniform(0, 1, (n, 1))
    edge = a * w + b * (1 - w)
    rays = []
    for target in (corner, edge):
        for dist in (0.3 * ext, 3.0 * ext, 300.0 * ext, 30000.0 * ext):
            d = r.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
            o = target - d * dist
            rays.append(np.concatenate([o, d], axis=1))
    # axis-parallel rays through vertices, from outside and from the vertex itself
    for ax in range(3):
        d = np.zeros((n, 3)); d[:, ax] = r.choice([-1.0, 1.0], n)
        rays.append(np.concatenate([corner - d * 2.0 * ext, d], axis=1))
        rays.append(np.concatenate([corner, d], axis=1))
    # (Rays lying, to fp32 rounding, IN the plane of a triangle are left out on purpose: Moller-Trumbore then divides by a
    # determinant of rounding noise and the reference accepts a "hit" at an arbitrary distance -- a t that is not inside the
    # leaf's box, which no distance-culled traversal can reproduce.  DESIGN.md section 2 states this limit.)
    return np.concatenate(rays, axis=0).astype(np.float32)


def _leaf_box_fails_slabs(ex, rays, prim):
    """mask of the hit rays whose hit triangle's exact box does NOT pass the reference's `slabs` (UtilsFunc.py:494-523, fp32, 1 / d hoisted as the device does):
    the candidates k_trace accepts only after walking the leaf's ancestors (trace_leaf_step, tirt_internal.h) -- the path on which ROCm 7.2's register
    allocator once overwrote the primitive id (toolchain note)"""
    P, V = ex.scene.primitive_np, ex.scene.vertex_np[:, :3].astype(np.float32)
    hit = (prim >= 0) & (P[np.maximum(prim, 0), 0] == 1)
    vi = P[np.maximum(prim, 0), 1]
    tri = np.stack([V[vi], V[vi + 1], V[vi + 2]], axis=1)
    mn, mx = tri.min(axis=1), tri.max(axis=1)
    o, d = rays[:, :3].astype(np.float32), rays[:, 3:6].astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = (np.float32(1.0) / d).astype(np.float32)
        t1 = ((mn - o) * inv).astype(np.float32); t2 = ((mx - o) * inv).astype(np.float32)
    lo, hi = np.minimum(t1, t2), np.maximum(t1, t2)
    par = np.abs(d) < np.float32(0.000001)
    outside = par & ((o < mn) | (o > mx))
    lo = np.where(par, np.float32(0.0), lo); hi = np.where(par, np.float32(1.0e30), hi)
    tmin = np.maximum(lo.max(axis=1), np.float32(0.0)); tmax = np.minimum(hi.min(axis=1), np.float32(1000000.0))
    return hit & ((tmin > tmax) | outside.any(axis=1))


def test_hits_accepted_through_the_ancestor_walk(gpu_ctx_ok):
    """The rays of the grazing set whose hit lies OUTSIDE its leaf's exact box (to fp32 rounding): the ordered traversal accepts them only after `slabs` on every
    ancestor, and reads the primitive id again after that walk.  There must be such rays in the set (else the walk is untested), and on exactly those the whole
    hit record -- t, primitive id, position, both normals, uv: everything u and v go into -- equals the oracle's bit for bit (t / u / v / leaf live
    across the same walk that once cost the primitive id its register)."""
    n_walk = 0
    for make, W in ((lambda: scenes.cornell_box(48, 48, 4, device_id=0), 48), (lambda: tiny_scene(3000, seed=31, W=48, H=48, spread=0.08, device_id=0), 48)):
        ex = make(); ex.scene.setup_data_cpu()
        rays = _grazing_rays(ex, 900, 17)
        ex = make(); ex.build_scene()
        o = oa.OracleScene(ex.scene, ex.cam); o.lbvh_build()
        want, wprim, _ = o.closest_hit(rays)
        walk = _leaf_box_fails_slabs(ex, rays, wprim)
        got, gprim, _ = ex.scene.ctx.trace_closest(rays, 64, _native.TRAVERSE_ORDERED)
        print("rays %d, hits %d, accepted through the ancestor walk %d" % (len(rays), (wprim >= 0).sum(), walk.sum()))
        assert np.array_equal(gprim[walk], wprim[walk]) and bits_equal(got[walk], want[walk]).all()
        # the same rays as shadow rays (closest hit's t and primitive) and from the candidate-list side of the leaf step: nothing else runs that code
        st, sp, _ = o.shadow_hit(rays[walk]); gt, gp, _ = ex.scene.ctx.trace_shadow(rays[walk], 64, 0)
        assert np.array_equal(gp, sp) and bits_equal(gt, st).all()
        n_walk += int(walk.sum())
    assert n_walk >= 50, "the grazing rays no longer reach the ancestor walk: the test tests nothing"


def test_quantised_nodes_on_grazing_rays(gpu_ctx_ok):
    """The ordered traversal walks 16-bit quantised boxes that CONTAIN the reference's and re-checks the reference's
    own visiting condition before it accepts a hit (tirt_internal.h, BvhView): closest hits must equal the oracle's on
    rays chosen to sit on box boundaries -- Cornell (axis-aligned walls: zero-thickness boxes), random soup, Teapot."""
    for make, W in ((lambda: scenes.cornell_box(48, 48, 4, device_id=0), 48),
                    (lambda: tiny_scene(3000, seed=31, W=48, H=48, spread=0.08, device_id=0), 48),
                    (lambda: duplicate_code_scene(W=48, H=48, device_id=0), 48)):
        ex = make()
        ex.scene.setup_data_cpu()
        rays = _grazing_rays(ex, 900, 17)
        check_scene(make(), W, W, rays, max_rays=60000)


def test_quantised_nodes_headline_scene_far_and_near(gpu_ctx_ok):
    ex = scenes.synthetic(64, 64, 4, device_id=0)
    ex.scene.setup_data_cpu()
    rays = _grazing_rays(ex, 1500, 23)
    check_scene(scenes.synthetic(64, 64, 4, device_id=0), 64, 64, rays, max_rays=60000)


def test_very_long_duplicate_chain(gpu_ctx_ok):
    """600 triangles with one Morton code: a 600-deep chain in the reference tree (its own 64-entry stack overflows there; the
    oracle is given 2048 entries), ~200 levels of the collapsed 4-wide tree.  Build succeeds, hits equal the oracle's."""
    from ti_raytrace_amd import Example, PT_RGB
    from ti_raytrace_amd import SceneData as SCD
    W = H = 32
    ex = Example.example(W, H, 4, 0)
    mat = SCD.Material(); mat.type = SCD.MAT_DISNEY; mat.setRough(0.5); mat.setColor([0.8, 0.8, 0.8, 1.0]); mat.alebdoTex = -1
    r = np.random.RandomState(3)
    tris = []
    for k in range(600):
        a = r.uniform(0.2, 1.0); th = r.uniform(0, 2 * np.pi)
        p = np.array([[np.cos(th + 2 * np.pi * j / 3) * a, np.sin(th + 2 * np.pi * j / 3) * a, r.uniform(-0.3, 0.3)] for j in range(3)])
        p -= p.mean(axis=0, keepdims=True)
        tris.append(p)
    for k in range(30):
        c = r.uniform(-1.5, 1.5, size=3); tris.append(c[None, :] + r.uniform(-0.2, 0.2, size=(3, 3)))
    ex.scene.add_mesh(np.asarray(tris), mat)
    ex.add_sphere_light(pos=(0.0, 3.0, 0.0), radius=0.5, emission=30.0)
    ex.integrator = PT_RGB.PathTrace(W, H, ex.cam, ex.scene, 2048)
    ex.build_scene(); ex.frame_camera(0.8)
    o = oa.OracleScene(ex.scene, ex.cam); o.lbvh_build()
    rays = np.concatenate([oa.camera_rays(ex.cam, W, H), random_rays(3000, -1.5, 1.5, 9)], axis=0)
    want, wprim, _ = o.closest_hit(rays, stack_size=2048)
    for flags in (0, _native.TRAVERSE_EXHAUSTIVE):
        got, gprim, _ = ex.scene.ctx.trace_closest(rays, 2048, flags)
        assert np.array_equal(gprim, wprim) and bits_equal(got[:, 0], want[:, 0]).all()
    print("600-chain: hit fraction %.3f" % (wprim >= 0).mean())
    assert ex.scene.ctx.stats()["stack_overflow"] == 0 and (wprim >= 0).mean() > 0.05


def test_measurement_helpers(gpu_ctx_ok):
    ex = scenes.synthetic(32, 32, 4, ntri=5000, device_id=0)
    ex.build_scene()
    info = ex.scene.ctx.bvh_info()
    assert info["prim_bytes"] == 48 * 5001 and 1200 < info["nodes"] < 5001 and info["node_bytes"] == 64 * info["nodes"]
    assert info["nodes_in_lds"] == min(info["nodes"], 224)          # TR_TOP_SLOTS (tirt_internal.h)
    rate = ex.scene.ctx.micro_gather_rate(1 << 20, 200)
    assert 500.0 < rate < 40000.0                      # GB/s: a sane number, not a benchmark


def test_ordered_equals_exhaustive_on_two_million_stress_rays(gpu_ctx_ok):
    """tools/stress_ordered_vs_exhaustive.py as a test, on a 2 M-ray subset: the ordered traversal (quantised 4-wide nodes, distance
    culling, verified candidates) against the reference's exhaustive order on the device -- closest hit (primitive, t bit for bit)
    and the shadow query -- on random, aimed, nearly axis-parallel and grazing rays over four scenes.  The grazing set starts 0.3 to
    30 000 scene extents away: from hundreds of extents the reference's Moller-Trumbore distances are rounding noise and it takes
    "hits" in front of the real surface (19 of 48 M stress rays in round 2); rays that start more than TR_FAR_RHO extents from the
    grid therefore do not cull by distance (tirt_render.hip), which makes them the reference's, too."""
    n = 45000
    total = bad = 0
    for make in (lambda: scenes.cornell_box(32, 32, 4, device_id=0), lambda: scenes.single_model(32, 32, 4, device_id=0),
                 lambda: scenes.veach_bdpt(32, 32, 4, device_id=0, integrator="pt"), lambda: scenes.synthetic(32, 32, 4, device_id=0)):
        ex = make(); ex.build_scene(); ctx = ex.scene.ctx
        lo = ex.scene.minboundarynp[0].astype(np.float64); hi = ex.scene.maxboundarynp[0].astype(np.float64)
        ext = float((hi - lo).max()); ctr = 0.5 * (lo + hi)
        r = np.random.RandomState(11)
        kinds = []
        o = r.uniform(lo - 0.1 * ext, hi + 0.1 * ext, size=(n, 3)); d = r.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
        kinds.append(np.concatenate([o, d], 1))
        o = ctr + r.normal(size=(n, 3)) * 3 * ext; d = (ctr + r.uniform(-0.5, 0.5, (n, 3)) * ext) - o; d /= np.linalg.norm(d, axis=1, keepdims=True)
        kinds.append(np.concatenate([o, d], 1))
        d = r.normal(size=(n, 3)); d[:, r.randint(0, 3)] *= 1e-7; d /= np.linalg.norm(d, axis=1, keepdims=True)
        kinds.append(np.concatenate([r.uniform(lo, hi, size=(n, 3)), d], 1))
        kinds.append(_grazing_rays(ex, 26000, 43).astype(np.float64))          # 14 x 26 000 rays, half of them from 300 / 30 000 extents
        for rays in kinds:
            rays = rays.astype(np.float32)
            a, ap, _ = ctx.trace_closest(rays, 64, 0)
            b, bp, _ = ctx.trace_closest(rays, 64, _native.TRAVERSE_EXHAUSTIVE)
            sa, sap, _ = ctx.trace_shadow(rays, 64, 0)
            bad += int((ap != bp).sum() + (a[:, 0].view(np.uint32) != b[:, 0].view(np.uint32)).sum() + (sap != bp).sum())
            total += rays.shape[0]
    print("ordered vs exhaustive: %d rays, %d mismatches" % (total, bad))
    assert total >= 1900000 and bad == 0


def test_wave_timeline_of_a_counting_launch(gpu_ctx_ok):
    """tirt_trace_timeline (diagnostics, no reference counterpart): the armed counting launch records one (start, queue empty, end, hardware id)
    row per wave; five 256-thread blocks per CU means 20 waves on every CU, all alive together (round 3: the 512-thread blocks fitted two per CU)."""
    from ti_raytrace_amd import scenes, _native
    ex = scenes.synthetic(512, 512, 8, ntri=20000, device_id=0, seed=3)
    ctx = ex.scene.ctx
    ex.build_scene(); ctx.sync()
    ctx.set_option("time_kernels", 1)
    ctx.set_option("trace_timeline", 0)
    ctx.pt_rgb_render(0, 8, 1, 15, 64, _native.TRAVERSE_ORDERED | _native.COUNT_NODES); ctx.sync()
    tl = ctx.trace_timeline()
    ctx.set_option("time_kernels", 0); ctx.set_option("trace_timeline", -1)
    assert len(tl) > 0 and len(tl) % 4 == 0
    start, exh, end = tl[:, 0].astype(np.int64), tl[:, 1].astype(np.int64), tl[:, 2].astype(np.int64)
    assert (end >= start).all() and ((exh == 0) | ((exh >= start) & (exh <= end))).all()
    span = end.max() - start.min()
    life = (end - start).mean()
    hw = tl[:, 3]
    cu = ((hw >> np.uint64(32)) & np.uint64(0xf)) * np.uint64(1 << 16) + (hw & np.uint64(0xff00))          # XCC id, then CU / SH / SE bits of HW_ID
    per_cu = np.unique(cu, return_counts=True)[1]
    print("timeline: %d waves on %d CUs (waves per CU %d..%d), span %.1f us, mean wave life %.2f of the span" % (
        len(tl), len(per_cu), per_cu.min(), per_cu.max(), span * 1e-2, life / max(span, 1)))
    st = ctx.stats()
    assert st["diag_waves"] >= len(tl) and st["diag_wave_ticks"] > 0
    if len(tl) == 5 * 4 * len(per_cu):                 # a full grid: every CU runs its five blocks at once
        assert per_cu.min() == per_cu.max() == 20 and life > 0.4 * span          # (a small job: its tail is a large part of the launch)
