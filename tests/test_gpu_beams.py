"""Camera rays through the pixels' candidate lists (csrc/tirt_pvb.hip, option "primary_beams") against the ordinary bounce-0 launch of k_trace and
against the CPU oracle.  The lists only change WHICH kernel finds a camera ray's closest hit: every film below must come out bit for bit the same,
with the same ray counts, whether they are on or off -- on meshes, analytic spheres, env-lit scenes, odd film sizes, multi-rank tiles, after a
camera move, after a new scene in the same context, from far away (no lists) and through PT_Spec."""
import numpy as np
import pytest

import oracle_api as oa
from common import rel_l2, tiny_scene
from ti_raytrace_amd import scenes

pytestmark = pytest.mark.gpu


def film_and_counts(ex, W, H, frames, beams, render=None, opts=None, build=True):
    if build: ex.build_scene()
    ctx = ex.scene.ctx
    ctx.set_option("primary_beams", beams)
    ctx.set_option("primary_beams_min_frames", 1)
    for k, v in (opts or {}).items(): ctx.set_option(k, v)
    ctx.film_clear(); ctx.stats_reset()
    (render or (lambda c: c.pt_rgb_render(0, frames, 5, 15, 64, 0)))(ctx)
    film = ctx.film_download(W, H)[0]
    st = ctx.stats()
    return film, (st["rays_closest"], st["rays_shadow"], st["paths"], st["shaded"], st["stack_overflow"]), ctx.primary_beam_stats()


def same(a, b):
    return np.array_equal(a.view(np.uint32), b.view(np.uint32))


def one_primitive(W, H, n, kind):
    """A scene whose LBVH root IS its only leaf: k_trace goes straight to the primitive test and a hit outside the leaf's box has no ancestor to be
    refused by, so the list pass must not reject rays on the root's box either.  `tri`: one large triangle under the env map (no emitter: no NEE);
    `sphere`: the sphere light alone (the analytic-sphere leaf)."""
    from ti_raytrace_amd import Example, PT_RGB, SceneData as SCD
    ex = Example.example(W, H, n, 0)
    if kind == "tri":
        mat = SCD.Material(); mat.type = SCD.MAT_DISNEY; mat.setMetal(0.0); mat.setRough(0.5); mat.setColor([0.8, 0.7, 0.6, 1.0]); mat.alebdoTex = -1
        ex.scene.add_mesh(np.asarray([[[-1.0, -0.7, 0.1], [1.1, -0.6, -0.2], [0.05, 0.9, 0.3]]], np.float32), mat)
        ex.scene.add_env(scenes.asset("image", "env.png"), 2.0)
    else:
        ex.add_sphere_light(pos=(0.1, 0.2, -0.1), radius=0.75, emission=5.0)
    ex.integrator = PT_RGB.PathTrace(W, H, ex.cam, ex.scene, 64)
    return ex


SCENES = {
    "one_triangle": lambda W, H, n: one_primitive(W, H, n, "tri"),
    "one_sphere": lambda W, H, n: one_primitive(W, H, n, "sphere"),
    "cornell": lambda W, H, n: scenes.cornell_box(W, H, n, device_id=0),
    "teapot_glass_env": lambda W, H, n: scenes.single_model(W, H, n, device_id=0),
    "mesh_20k": lambda W, H, n: scenes.synthetic(W, H, n, ntri=20000, device_id=0),
    "sphere_env": lambda W, H, n: scenes.gallery_sphere(W, H, n, device_id=0),
    "soup_400": lambda W, H, n: tiny_scene(400, seed=21, W=W, H=H, spread=0.25, device_id=0),
}


@pytest.mark.parametrize("name", sorted(SCENES))
@pytest.mark.parametrize("size", [(128, 128), (50, 30)])
def test_beams_change_no_bit(gpu_ctx_ok, name, size):
    W, H = size
    frames = 12
    off, n_off, _ = film_and_counts(SCENES[name](W, H, frames), W, H, frames, 0)
    on, n_on, st = film_and_counts(SCENES[name](W, H, frames), W, H, frames, 1)
    print(name, size, st)
    assert st["rays"] == frames * W * H, "the camera rays did not go through the lists"
    assert st["pixels_with_list"] > 0          # (a 50 x 30 film of 20 000 triangles: most pixels see more leaves than a list holds, and are left to k_trace)
    assert n_on == n_off and n_on[4] == 0
    assert same(on, off)


def test_beams_against_the_oracle(gpu_ctx_ok):
    """not only equal to k_trace: the film of the 100k-triangle headline scene (reduced) and of the Cornell box against the CPU oracle, bit for bit"""
    for make, W, H, frames in ((lambda: scenes.synthetic(160, 160, 9, device_id=0), 160, 160, 9), (lambda: scenes.cornell_box(96, 96, 9, device_id=0), 96, 96, 9)):
        ex = make()
        got, n, st = film_and_counts(ex, W, H, frames, 1)
        o = oa.OracleScene(ex.scene, ex.cam); o.lbvh_build()
        want, ost = o.render(W, H, 0, frames, seed=5)
        assert st["rays"] == frames * W * H
        assert n[0] == ost["rays_closest"] and n[1] == ost["rays_shadow"]
        assert same(got, want), "rel-L2 %.3e" % rel_l2(got, want)


def test_beams_follow_the_camera_and_the_scene(gpu_ctx_ok):
    """the lists belong to (build, camera, film): a camera move, a new film size and a new scene in the SAME context each make new ones"""
    W = H = 96
    frames = 10
    ex = scenes.cornell_box(W, H, frames, device_id=0)
    a_on, _, _ = film_and_counts(ex, W, H, frames, 1)
    ex.cam.yaw += 0.4; ex.cam.update()                  # (pushes the new view to the context)
    moved_on, n1, st1 = film_and_counts(ex, W, H, frames, 1, build=False)
    moved_off, n0, _ = film_and_counts(ex, W, H, frames, 0, build=False)
    assert n1 == n0 and same(moved_on, moved_off)
    assert not same(moved_on, a_on), "the camera did not move: the test tests nothing"
    # another film size, same context
    ex.scene.ctx.film_create(64, 48, 0, 1, 4096)
    small_on, n1, _ = film_and_counts(ex, 64, 48, frames, 1, build=False)
    small_off, n0, _ = film_and_counts(ex, 64, 48, frames, 0, build=False)
    assert n1 == n0 and same(small_on, small_off)


def test_camera_moves_while_batches_are_in_flight(gpu_ctx_ok):
    """tirt_camera_set submits what is pending and does not wait: the new lists must not be written under the batches that still read the old ones.
    Four camera poses, 40 frames each into ONE film, nothing synchronised in between -- against the same sequence with the lists off."""
    W = H = 256
    films = []
    for beams in (0, 1):
        ex = scenes.synthetic(W, H, 160, ntri=20000, device_id=0)
        ex.build_scene()
        ctx = ex.scene.ctx
        ctx.set_option("primary_beams", beams)
        for pose in range(4):
            ctx.pt_rgb_render(40 * pose, 40, 7, 15, 64, 0)
            ex.cam.yaw += 0.15; ex.cam.update()
        films.append(ctx.film_download(W, H)[0])
        if beams: assert ctx.primary_beam_stats()["rays"] > 0
    assert same(films[0], films[1])


@pytest.mark.parametrize("name", ["soup", "cornell", "teapot", "slivers"])
def test_many_camera_poses(gpu_ctx_ok, name):
    """orbit angles, eye distances from inside the scene to seven extents away (beyond eight: no lists), three film shapes: film and ray counts with
    the lists on = off, every pose (tools/dbg/pvb_stress.py is the longer version)"""
    rng = np.random.default_rng(7)
    make = {"soup": lambda W, H: scenes.synthetic(W, H, 8, ntri=30000, device_id=0), "cornell": lambda W, H: scenes.cornell_box(W, H, 8, device_id=0),
            "teapot": lambda W, H: scenes.single_model(W, H, 8, device_id=0), "slivers": lambda W, H: tiny_scene(3000, seed=11, W=W, H=H, spread=0.9, device_id=0)}[name]
    for W, H in ((192, 192), (320, 64), (56, 200)):
        ex = make(W, H); ex.build_scene(); ctx = ex.scene.ctx
        ctx.set_option("primary_beams_min_frames", 1)
        for pose in range(8):
            ex.cam.yaw = float(rng.uniform(0, 6.28)); ex.cam.pitch = float(rng.uniform(-1.2, 1.2))
            ex.frame_camera(float(rng.choice([0.05, 0.2, 0.5, 0.8, 1.5, 3.0, 7.0])))
            res = []
            for beams in (0, 1):
                ctx.set_option("primary_beams", beams)
                ctx.film_clear(); ctx.stats_reset()
                ctx.pt_rgb_render(0, 8, 11 + pose, 15, 64, 0)
                st = ctx.stats()
                res.append((ctx.film_download(W, H)[0].view(np.uint32).copy(), st["rays_closest"], st["rays_shadow"], st["stack_overflow"]))
            assert res[0][1:] == res[1][1:], (W, H, pose, ex.cam.yaw, ex.cam.pitch, ex.cam.scale)
            assert np.array_equal(res[0][0], res[1][0]), (W, H, pose, ex.cam.yaw, ex.cam.pitch, ex.cam.scale)


def test_beams_with_tiles_of_several_ranks(gpu_ctx_ok):
    """a rank's tiles (round robin, ragged last tile, blocked and linear pixel order inside a tile): lists per LOCAL pixel"""
    W = H = 64
    frames = 9
    for tile in (100, 512, 1024):                        # 512, 1024: whole 8-pixel columns -> 8 x 8 pixel blocks inside a tile (tirt_film_create)
        for rank in range(3):
            films = []
            for beams in (0, 1):
                ex = scenes.synthetic(W, H, frames, ntri=5000, device_id=0)
                ex.build_scene()
                ctx = ex.scene.ctx
                ctx.film_create(W, H, rank, 3, tile)
                films.append(film_and_counts(ex, W, H, frames, beams, build=False))
            assert films[0][1] == films[1][1]
            assert same(films[0][0], films[1][0]), (tile, rank)


def test_far_camera_gets_no_lists(gpu_ctx_ok):
    """further than eight scene extents away k_trace stops culling by distance (tirt_render.hip, TR_FAR_RHO): no lists, every camera ray goes the ordinary way"""
    W = H = 64
    frames = 9
    res = []
    for beams in (0, 1):
        ex = tiny_scene(400, seed=3, W=W, H=H, spread=0.25, device_id=0)
        ex.build_scene()
        ex.frame_camera(40.0)
        res.append(film_and_counts(ex, W, H, frames, beams, build=False))
    assert res[1][2]["pixels_with_list"] == 0
    assert res[0][1] == res[1][1] and same(res[0][0], res[1][0])


def test_beams_in_pt_spec(gpu_ctx_ok):
    W = H = 96
    frames = 10
    r = lambda c: c.pt_spec_render(0, frames, 3, 10, 64, 0)
    off, n0, _ = film_and_counts(scenes.spectral_box(W, H, frames, device_id=0), W, H, frames, 0, render=r)
    on, n1, st = film_and_counts(scenes.spectral_box(W, H, frames, device_id=0), W, H, frames, 1, render=r)
    assert st["rays"] == frames * W * H and n0 == n1 and same(on, off)


def test_short_calls_skip_the_lists(gpu_ctx_ok):
    """a batch of fewer than primary_beams_min_frames frames (default 16: making the lists costs what they save on ~13 frames of a 1024^2 film) traces its
    camera rays the ordinary way"""
    W = H = 64
    ex = scenes.cornell_box(W, H, 4, device_id=0)
    ex.build_scene()
    ctx = ex.scene.ctx
    ctx.set_option("merge_paths", 0)
    ctx.pt_rgb_render(0, 15, 1, 15, 64, 0)
    assert ctx.primary_beam_stats()["rays"] == 0
    ctx.pt_rgb_render(15, 16, 1, 15, 64, 0)
    assert ctx.primary_beam_stats()["rays"] == 16 * W * H
