"""PT_RGB image parity, HIP wavefront pipeline vs the CPU oracle at the same counter-based
seed.  Tolerance (north star): relative L2 <= 1e-3.  Because both sides evaluate the same
fp32 operations in the same order, the images are expected to be bit-identical; the tests
assert the tolerance and report the identical-pixel count."""
import numpy as np
import pytest

import oracle_api as oa
from common import rel_l2, tiny_scene, spot_laser_scene
from ti_raytrace_amd import scenes, _native
from ti_raytrace_amd import UtilsFunc as UF

pytestmark = pytest.mark.gpu
TOL = 1e-3


def render_both(ex, W, H, frames, flags=0):
    ex.build_scene()
    ex.integrator.flags = flags
    o = oa.OracleScene(ex.scene, ex.cam)
    o.lbvh_build()
    return ex, o


def test_cornell_frames_and_counters(gpu_ctx_ok):
    W = H = 96
    ex, o = render_both(scenes.cornell_box(W, H, 8, device_id=0), W, H, 6)
    ctx = ex.scene.ctx
    ctx.stats_reset()
    ex.integrator.render_frames(6)
    got = ex.integrator.hdr.to_numpy()
    want, ost = o.render(W, H, 0, 6, seed=ex.integrator.seed)
    r = rel_l2(got, want)
    print("cornell 96^2 x6: rel-L2 %.3e, identical pixels %d/%d" % (r, (got == want).all(axis=2).sum(), W * H))
    assert r <= TOL
    st = ctx.stats()
    assert st["rays_closest"] == ost["rays_closest"] and st["rays_shadow"] == ost["rays_shadow"]
    assert st["shaded"] == ost["shaded"] and st["paths"] == ost["paths"] and st["stack_overflow"] == 0
    # tone map
    UF.tone_map(0.5, ex.integrator.hdr, ex.integrator.rgb_film)
    rgb = ex.integrator.rgb_film.to_numpy()
    assert rel_l2(rgb, o.tone_map(0.5, want)) <= TOL


def test_render_one_frame_at_a_time_equals_batch(gpu_ctx_ok):
    W = H = 48
    ex, o = render_both(scenes.cornell_box(W, H, 8, device_id=0), W, H, 5)
    for _ in range(5):
        ex.integrator.render()
        ex.cam.update_frame()
    a = ex.integrator.hdr.to_numpy()
    ex.scene.ctx.film_clear(); ex.cam.frame = 0
    ex.integrator.render_frames(5)
    b = ex.integrator.hdr.to_numpy()
    assert np.array_equal(a, b)
    want, _ = o.render(W, H, 0, 5, seed=ex.integrator.seed)
    assert rel_l2(a, want) <= TOL


def test_exhaustive_mode_counts_match_oracle(gpu_ctx_ok):
    W = H = 64
    ex, o = render_both(scenes.cornell_box(W, H, 4, device_id=0), W, H, 2)
    ctx = ex.scene.ctx
    ctx.stats_reset()
    ex.integrator.flags = _native.TRAVERSE_EXHAUSTIVE | _native.COUNT_NODES
    ex.integrator.render_frames(2)
    got = ex.integrator.hdr.to_numpy()
    want, ost = o.render(W, H, 0, 2, seed=ex.integrator.seed)
    assert rel_l2(got, want) <= TOL
    st = ctx.stats()
    for k in ("rays_closest", "rays_shadow", "box_closest", "leaf_closest", "box_shadow", "leaf_shadow", "shaded"):
        assert st[k] == ost[k], (k, st[k], ost[k])


def test_tiles_reassemble_to_the_full_frame(gpu_ctx_ok):
    """Multi-GPU sharding rule on one GPU: the tiles of N ranks, rendered one after the other,
    sum to the single-GPU film bit for bit (disjoint pixels, per-pixel RNG)."""
    W = H = 64
    ex = scenes.cornell_box(W, H, 4, device_id=0)
    ex.build_scene()
    ex.integrator.render_frames(3)
    full = ex.integrator.hdr.to_numpy()
    acc = np.zeros_like(full)
    for rank in range(3):
        ex.scene.ctx.film_create(W, H, rank, 3, 100)          # ragged: 4096 px in tiles of 100
        ex.scene.ctx.pt_rgb_render(0, 3, ex.integrator.seed, 15, 64, 0)
        part, _ = ex.scene.ctx.film_download(W, H)
        assert ((part != 0).any(axis=2) & (acc != 0).any(axis=2)).sum() == 0
        acc += part
    assert np.array_equal(acc, full)


def test_teapot_glass_env(gpu_ctx_ok):
    """BASELINE config 2 at reduced size: glass teapot, sphere light, env map, smooth normals."""
    W = H = 64
    ex = scenes.single_model(W, H, 4, device_id=0)
    ex.build_scene()
    o = oa.OracleScene(ex.scene, ex.cam)
    o.lbvh_build(); o.process_normal(ex.scene.vertex_index_np)
    ex.integrator.render_frames(3)
    got = ex.integrator.hdr.to_numpy()
    want, ost = o.render(W, H, 0, 3, seed=ex.integrator.seed)
    r = rel_l2(got, want)
    print("teapot 64^2 x3: rel-L2 %.3e, identical pixels %d/%d, oracle max stack %d"
          % (r, (got == want).all(axis=2).sum(), W * H, ost["max_stack"]))
    assert np.isfinite(want).all() == np.isfinite(got).all()
    m = np.isfinite(want).all(axis=2) & np.isfinite(got).all(axis=2)
    assert rel_l2(got[m], want[m]) <= TOL


def test_headline_scene_reduced(gpu_ctx_ok):
    """BASELINE config 3 geometry (100k triangles) at 128^2 x 2 frames."""
    W = H = 128
    ex, o = render_both(scenes.synthetic(W, H, 4, device_id=0), W, H, 2)
    ex.scene.ctx.stats_reset()
    ex.integrator.render_frames(2)
    got = ex.integrator.hdr.to_numpy()
    want, ost = o.render(W, H, 0, 2, seed=ex.integrator.seed)
    r = rel_l2(got, want)
    st = ex.scene.ctx.stats()
    print("100k 128^2 x2: rel-L2 %.3e, identical pixels %d/%d; rays %d+%d" %
          (r, (got == want).all(axis=2).sum(), W * H, st["rays_closest"], st["rays_shadow"]))
    assert r <= TOL
    assert st["rays_closest"] == ost["rays_closest"] and st["rays_shadow"] == ost["rays_shadow"]


def test_small_scene_many_bounces(gpu_ctx_ok):
    W = H = 40
    ex, o = render_both(tiny_scene(400, seed=21, W=W, H=H, spread=0.25, device_id=0), W, H, 4)
    ex.integrator.render_frames(4)
    got = ex.integrator.hdr.to_numpy()
    want, _ = o.render(W, H, 0, 4, seed=ex.integrator.seed)
    assert rel_l2(got, want) <= TOL


def test_cornell_full_config_matches_reference_out_png(gpu_ctx_ok):
    """The reference's own committed run (Main.py:15: Cornell, PT_RGB, 512^2, 512 spp, exposure
    0.5) repeated on the GPU and compared with its out.png: statistical parity (the Taichi RNG
    stream cannot be reproduced).  Mean colour within 1 %, 16x16-block means within 2 % rel-L2."""
    import os
    W = H = 512
    ex = scenes.cornell_box(W, H, 512, device_id=0)
    ex.build_scene()
    ex.render_all(batch=64)
    UF.tone_map(0.5, ex.integrator.hdr, ex.integrator.rgb_film)
    rgb = ex.integrator.rgb_film.to_numpy()
    img = np.transpose(rgb, (1, 0, 2))[::-1]
    ours = img.reshape(32, 16, 32, 16, 3).mean(axis=(1, 3))
    ref = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "out_png_blocks.npy"))
    mean_err = np.abs(ours.reshape(-1, 3).mean(0) - ref.reshape(-1, 3).mean(0)) / ref.reshape(-1, 3).mean(0)
    r = rel_l2(ours, ref)
    print("cornell 512^2 x512 vs out.png: mean err %s, block rel-L2 %.4f" % (np.round(mean_err, 4), r))
    assert (mean_err < 0.01).all() and r < 0.02


def test_full_size_ordered_equals_reference_order(gpu_ctx_ok):
    """BASELINE config 3 at its full 1024^2 size (4 frames): the product's ordered, t-culled
    traversal and the reference's exhaustive visiting order (verified against the oracle at
    reduced size above) produce the same film bit for bit, the same ray counts, and the
    2-lane / 4-lane batch overlap does not change a bit either."""
    W = H = 1024
    ex = scenes.synthetic(W, H, 8, device_id=0)
    ex.build_scene()
    ctx = ex.scene.ctx
    films, rays = [], []
    for flags, lanes, batch in ((0, 4, 1 << 20), (_native.TRAVERSE_EXHAUSTIVE, 1, 32 << 20), (0, 1, 32 << 20)):
        ctx.set_option("overlap_lanes", lanes)
        ctx.set_option("batch_paths", batch)          # 1 Mi paths per batch -> 4 batches rotate over the lanes
        ctx.film_clear(); ctx.stats_reset()
        ctx.pt_rgb_render(0, 4, 1, 15, 64, flags)
        films.append(ctx.film_download(W, H)[0])
        st = ctx.stats()
        rays.append((st["rays_closest"], st["rays_shadow"], st["shaded"], st["paths"]))
        assert st["stack_overflow"] == 0
    assert np.array_equal(films[0], films[1]) and np.array_equal(films[0], films[2])
    assert rays[0] == rays[1] == rays[2]
    assert np.isfinite(films[0]).all() and films[0].mean() > 0


def test_ordered_equals_reference_order_on_mesh_scenes(gpu_ctx_ok):
    """Same cross-check on real meshes: the Veach scene (11.5k triangles, glass, two emissive meshes, smooth
    normals) and the glass Teapot under the environment map (NaN shading normals included): ordered 4-wide
    traversal with bounded shadow rays == the reference's exhaustive order, film bits and ray counts."""
    for make in (lambda: scenes.veach_bdpt(384, 384, 8, device_id=0, integrator="pt"), lambda: scenes.single_model(384, 384, 8, device_id=0)):
        ex = make()
        ex.build_scene()
        ctx = ex.scene.ctx
        films, rays = [], []
        for flags in (0, _native.TRAVERSE_EXHAUSTIVE):
            ctx.film_clear(); ctx.stats_reset()
            ctx.pt_rgb_render(0, 6, ex.integrator.seed, 15, 64, flags)
            films.append(ctx.film_download(384, 384)[0])
            st = ctx.stats()
            rays.append((st["rays_closest"], st["rays_shadow"], st["shaded"], st["paths"]))
            assert st["stack_overflow"] == 0
        a, b = films[0].view(np.uint32), films[1].view(np.uint32)
        nan_same = np.isnan(films[0]) == np.isnan(films[1])
        assert nan_same.all() and ((a == b) | np.isnan(films[0])).all()
        assert rays[0] == rays[1] and rays[0][0] > 0


def test_odd_film_sizes_depths_and_frame_offsets(gpu_ctx_ok):
    """Ragged inputs: 1x1, 5x3 and 37x61 films (not multiples of the wave, block or tile size), MAX_DEPTH 1 / 2 / 15,
    renders that start at frame 7 on a film that holds frames 0..6 (bounce-0 state is implicit on the device,
    later bounces are not), tile size 16: all bit-identical to the oracle."""
    for (W, H, depth, tile) in ((1, 1, 15, 4096), (5, 3, 2, 4096), (37, 61, 1, 4096), (37, 61, 15, 16)):
        ex = scenes.cornell_box(W, H, 16, device_id=0, tile_size=tile) if tile != 4096 else scenes.cornell_box(W, H, 16, device_id=0)
        ex.build_scene()
        o = oa.OracleScene(ex.scene, ex.cam); o.lbvh_build()
        ctx = ex.scene.ctx
        seed = ex.integrator.seed
        ctx.pt_rgb_render(0, 7, seed, depth, 64, 0)
        ctx.pt_rgb_render(7, 3, seed, depth, 64, 0)
        got = ctx.film_download(W, H)[0]
        want, _ = o.render(W, H, 0, 10, seed=seed, max_depth=depth, tile_size=tile)
        assert np.array_equal(got, want), (W, H, depth, tile, rel_l2(got, want))


def test_abi_error_behaviour(gpu_ctx_ok):
    """Call-order and argument errors come back as negative codes with a message, never a crash."""
    ctx = _native.Context(0)
    with pytest.raises(_native.TirtError, match="not built"):
        ctx.pt_rgb_render(0, 1, 1)
    with pytest.raises(_native.TirtError, match="no primitives"):
        ctx.lbvh_build()
    v = np.zeros((3, 9), np.float32); m = np.zeros((1, 10), np.float32); s = np.zeros((1, 10), np.float32)
    good = np.array([[1, 0, 0]], np.int32)
    for bad in (np.array([[1, 1, 0]], np.int32), np.array([[1, 0, 5]], np.int32), np.array([[2, 3, 0]], np.int32)):
        with pytest.raises(_native.TirtError, match="out of range"):
            ctx.scene_upload(v, bad, m, s, np.zeros(1, np.int32), 0, np.zeros(3), np.ones(3))
    with pytest.raises(_native.TirtError, match="out of range"):
        ctx.scene_upload(v, good, m, s, np.array([7], np.int32), 1, np.zeros(3), np.ones(3))
    ctx.scene_upload(v, good, m, s, np.zeros(1, np.int32), 0, np.zeros(3), np.ones(3))
    with pytest.raises(_native.TirtError, match="bad size"):
        ctx.film_create(0, 16)
    with pytest.raises(_native.TirtError, match="unknown option"):
        ctx.set_option("nonsense", 1)
    with pytest.raises(_native.TirtError):
        _native.Context(9999)
    ctx.close()


def test_env_lit_scene_without_emitters(gpu_ctx_ok):
    """light_count == 0 with an environment map (the reference would index light[-1], Scene.py:423-428): defined as
    "no NEE sample" in the oracle and on the device; films, ray counts identical, no shadow rays at all."""
    from ti_raytrace_amd import Example, PT_RGB
    from ti_raytrace_amd import SceneData as SCD
    W = H = 48
    ex = Example.example(W, H, 4, 0)
    mat = SCD.Material(); mat.type = SCD.MAT_DISNEY; mat.setMetal(0.0); mat.setRough(0.5); mat.setColor([0.8, 0.6, 0.4, 1.0]); mat.alebdoTex = -1
    r = np.random.RandomState(5)
    tris = r.uniform(-1, 1, size=(300, 1, 3)) + r.uniform(-0.25, 0.25, size=(300, 3, 3))
    ex.scene.add_mesh(tris, mat)
    ex.scene.add_env(scenes.asset("image", "env.png"), 2.0)
    ex.integrator = PT_RGB.PathTrace(W, H, ex.cam, ex.scene, 64)
    ex.build_scene()
    ex.frame_camera(0.8)
    assert ex.scene.light_count == 0
    o = oa.OracleScene(ex.scene, ex.cam); o.lbvh_build()
    ctx = ex.scene.ctx
    ctx.stats_reset()
    ex.integrator.render_frames(3)
    got = ex.integrator.hdr.to_numpy()
    want, ost = o.render(W, H, 0, 3, seed=ex.integrator.seed)
    st = ctx.stats()
    assert np.array_equal(got, want), rel_l2(got, want)
    assert st["rays_shadow"] == 0 == ost["rays_shadow"] and st["rays_closest"] == ost["rays_closest"]
    assert np.isfinite(got).all() and got.mean() > 0
    with pytest.raises(_native.TirtError, match="no emitter"):
        ctx.bdpt_rgb_render(0, 1, 1)


@pytest.mark.parametrize("kinds,quad", [(("spot",), False), (("laser",), False), (("spot", "laser"), True)])
def test_spot_and_laser_emitters(gpu_ctx_ok, kinds, quad):
    """The two shape emitters without a surface (Scene.sample_li's SPOT / LASER branches, Scene.py:491-516; their point and normal,
    :413-418; area, :344-349): PT_RGB films bit-identical to the oracle, same ray counts; they light the scene (the spot's cone
    and the laser's disc on the floor) and are never hit themselves (intersect_prim returns INF for them, Scene.py:597-598)."""
    from common import spot_laser_scene
    W = H = 64
    ex = spot_laser_scene(W, H, kinds, device_id=0, with_quad_light=quad)
    ex.build_scene(); ex.scene.total_area(); ex.frame_camera(0.8)
    ctx = ex.scene.ctx
    o = oa.OracleScene(ex.scene, ex.cam); o.lbvh_build()
    _, _, compact = o.lbvh_get()
    assert np.array_equal(compact, ex.scene.bvh.compact_node.to_numpy())      # their leaf boxes are the reference's (0,0,0)-(0,0,0)
    ctx.stats_reset()
    ex.integrator.render_frames(6)
    got = ex.integrator.hdr.to_numpy()
    want, ost = o.render(W, H, 0, 6, seed=ex.integrator.seed)
    st = ctx.stats()
    assert np.array_equal(got, want, equal_nan=True), rel_l2(got, want)
    assert st["rays_closest"] == ost["rays_closest"] and st["rays_shadow"] == ost["rays_shadow"] and st["rays_shadow"] > 0
    assert np.isfinite(got).all() and got.mean() > 0.0
    if not quad:          # nothing else emits: everything the film shows was lit through sample_li's shape branch
        lit = (got.sum(axis=2) > 0).mean()
        print("%s only: %.1f %% of the pixels lit, mean %.4f" % (kinds[0], 100 * lit, got.mean()))
        assert 0.02 < lit < 0.98


def test_spot_and_laser_emitters_bdpt(gpu_ctx_ok):
    """Scene.sample_light's SPOT / LASER branches (Scene.py:449-472: the spot's direction through UF.mapToDisk and tan of its two
    angles, the laser's disc of start points) start BDPT's light sub-paths: film within 1e-5 of the oracle's (float-atomic splats),
    same ray counts."""
    from common import spot_laser_scene
    W = H = 40
    ex = spot_laser_scene(W, H, ("spot", "laser"), device_id=0, integrator="bdpt")
    ex.build_scene(); ex.scene.total_area(); ex.frame_camera(0.8)
    ctx = ex.scene.ctx
    o = oa.OracleScene(ex.scene, ex.cam); o.lbvh_build()
    ctx.stats_reset()
    ex.integrator.render_frames(3)
    got = ex.integrator.hdr.to_numpy()
    want, ost, _ = o.bdpt_render(ex.cam, W, H, 0, 3, seed=ex.integrator.seed)
    st = ctx.stats()
    assert rel_l2(got, want) <= 1e-5 and np.array_equal(np.isnan(got), np.isnan(want))
    assert st["rays_closest"] == ost["rays_closest"] and st["rays_shadow"] == ost["rays_shadow"]


def test_config2_full_size_tile_sample_against_the_oracle(gpu_ctx_ok):
    """BASELINE config 2 at its full 1024^2 x 64 spp (glass Teapot, sphere light, env map, smooth normals with the
    reference's NaN normals, Scene.py:377): three 2048-pixel runs of the film, re-rendered by the oracle, must equal the
    device film bit for bit -- NaN pixels included (a NaN sample makes a pixel NaN for good, PT_RGB.py:136) -- and the
    whole film's NaN-pixel count is reported."""
    W = H = 1024
    ex = scenes.single_model(W, H, 64, device_id=0)
    ex.build_scene()
    o = oa.OracleScene(ex.scene, ex.cam)
    o.lbvh_build(); o.process_normal(ex.scene.vertex_index_np)
    ex.render_all(batch=32)
    got = ex.integrator.hdr.to_numpy().reshape(-1, 3)
    nan_px = int(np.isnan(got).any(axis=1).sum())
    print("config 2, 1024^2 x 64 spp: %d NaN pixels of %d" % (nan_px, W * H))
    checked = nan_checked = 0
    for p0 in (300 * 1024 + 200, 512 * 1024 + 380, 700 * 1024 + 500):
        want, _ = o.render(W, H, 0, 64, seed=ex.integrator.seed, p_begin=p0, p_end=p0 + 2048)
        want = want.reshape(-1, 3)[p0:p0 + 2048]
        mine = got[p0:p0 + 2048]
        same = (mine.view(np.uint32) == want.view(np.uint32)) | (np.isnan(mine) & np.isnan(want))
        assert same.all(), (p0, int((~same).sum()))
        checked += 2048; nan_checked += int(np.isnan(want).any(axis=1).sum())
    print("  %d pixels x 64 spp identical to the oracle (%d of them NaN)" % (checked, nan_checked))
    assert (got[~np.isnan(got).any(axis=1)] >= 0).all()


def test_rccl_film_reduce_through_the_c_abi(gpu_ctx_ok):
    """tirt_comm_init / tirt_film_reduce (librccl loaded by the library itself, no torch): on this 1-GPU box a one-rank
    communicator -- the reduce must leave the film as it was, twice in a row, and a second context on the SAME device
    must be refused (one context per device in a communicator)."""
    W = H = 64
    ex = scenes.cornell_box(W, H, 4, device_id=0)
    ex.build_scene()
    ex.integrator.render_frames(2)
    before = ex.integrator.hdr.to_numpy().copy()
    comm = _native.Communicator([ex.scene.ctx])
    comm.film_reduce(0); comm.film_reduce(0)
    assert np.array_equal(ex.integrator.hdr.to_numpy(), before)
    other = _native.Context(0)
    with pytest.raises(_native.TirtError, match="already in a communicator|one context per device"):
        _native.Communicator([ex.scene.ctx, other])
    comm.close()
    with pytest.raises(_native.TirtError, match="one context per device"):
        _native.Communicator([ex.scene.ctx, other])
    other.close()
    # rendering goes on after the communicator is gone
    ex.integrator.render_frames(1)
    assert np.isfinite(ex.integrator.hdr.to_numpy()).all()


def test_example_loop_with_progressive_preview(gpu_ctx_ok, tmp_path):
    """The reference's host loop (example/Example.py:38-59): render() returns 1 until sample_count frames are in, tone-maps a
    preview on the way (headless: a PNG every `preview_every` frames instead of the ti.GUI blit), writes out.png at the end;
    previews must not disturb the film (the frame-at-a-time film equals the batch film)."""
    from PIL import Image
    W = H = 48
    ex = scenes.cornell_box(W, H, 6, device_id=0)
    ex.build_scene()
    ex.out_path = str(tmp_path / "out.png"); ex.preview_path = str(tmp_path / "preview.png"); ex.preview_every = 2
    n = 0
    while ex.render() == 1:
        n += 1
    assert n == 6 and ex.render() == 0
    assert Image.open(ex.preview_path).size == (W, H) and Image.open(ex.out_path).size == (W, H)
    a = ex.integrator.hdr.to_numpy()
    ex2 = scenes.cornell_box(W, H, 6, device_id=0); ex2.build_scene(); ex2.integrator.render_frames(6)
    assert np.array_equal(a, ex2.integrator.hdr.to_numpy())


def test_batch_planning_changes_no_bit(gpu_ctx_ok):
    """tirt_internal.h plan_batches: with the "job_frames" hint a job is cut into few large wavefront batches (here 104 frames of a 512^2 film
    = 27 M paths, above the 24 Mi-path threshold: 2 x 52 frames), without it into 32 Mi-path ones; calls
    may arrive frame by frame, in chunks that do not divide the plan, or all at once, and the hint may be wrong.  Same film, bit for bit."""
    W = H = 512
    frames = 104
    films = []
    for hint, chunk in ((0, frames), (frames, 8), (frames, frames), (frames // 2, 13), (4 * frames, 5)):
        ex = scenes.cornell_box(W, H, frames, device_id=0)
        ex.build_scene()
        ctx = ex.scene.ctx
        ctx.set_option("job_frames", hint)
        f = 0
        while f < frames:
            k = min(chunk, frames - f)
            ctx.pt_rgb_render(f, k, 1, 15, 64, 0)
            f += k
        films.append(ctx.film_download(W, H)[0])
        assert ctx.stats()["paths"] == frames * W * H
    for f in films[1:]:
        assert np.array_equal(f.view(np.uint32), films[0].view(np.uint32))


@pytest.mark.parametrize("integrator", ["pt_rgb", "pt_spec"])
def test_path_order_and_slice_layout_change_no_bit(gpu_ctx_ok, integrator):
    """tirt_internal.h TileMap::F / TraceArgs::slices_contig: the paths of a batch numbered pixel-block major instead of frame major, and k_trace's
    ray-fetch slices as contiguous stretches of the queue instead of interleaved chunks (each XCD then walks one region of the film): which lane
    traces which ray changes, nothing else -- the same film bit for bit, the same ray counts, alone and together, also when the tile has partial
    batches (7 frames cut 4 + 3) and when the local pixel count is not a multiple of 64 (frame-major order then)."""
    def run(W, H, opts, frames=7):
        ex = scenes.spectral_box(W, H, frames, device_id=0) if integrator == "pt_spec" else scenes.synthetic(W, H, frames, ntri=3000, device_id=0)
        ex.build_scene()
        ctx = ex.scene.ctx
        for k, v in opts.items(): ctx.set_option(k, v)
        ctx.set_option("batch_paths", 4 * W * H); ctx.set_option("merge_paths", 4 * W * H)
        ctx.stats_reset()
        if integrator == "pt_spec": ctx.pt_spec_render(0, frames, 3, 10, 64, 0)
        else: ctx.pt_rgb_render(0, frames, 3, 15, 64, 0)
        film = ctx.film_download(W, H)[0]
        st = ctx.stats()
        return film, (st["rays_closest"], st["rays_shadow"], st["paths"], st["shaded"])
    for W, H in ((96, 96), (50, 30)):
        base, n0 = run(W, H, {})
        for opts in ({"path_order_blocks": 0}, {"slices_contiguous": 1}, {"path_order_blocks": 1, "slices_contiguous": 1}):
            film, n = run(W, H, opts)
            assert n == n0, (opts, n, n0)
            assert np.array_equal(film.view(np.uint32), base.view(np.uint32)), opts


def test_blocked_tile_order_changes_no_bit(gpu_ctx_ok):
    """tirt_internal.h local_to_pixel: inside a tile of 8 whole columns the device walks 8 x 8 pixel blocks; any other tile size keeps the
    linear order.  Same film, and the tiles of three ranks still re-assemble to it."""
    W = H = 64
    films = []
    for tile_size in (8 * H, 100, 4096):                 # blocked | linear | blocked (the whole film is one tile)
        ex = scenes.cornell_box(W, H, 4, device_id=0)
        ex.integrator.tile_size = tile_size
        ex.build_scene()
        ex.integrator.render_frames(3)
        films.append(ex.integrator.hdr.to_numpy().copy())
    assert np.array_equal(films[0].view(np.uint32), films[1].view(np.uint32)) and np.array_equal(films[0].view(np.uint32), films[2].view(np.uint32))
    total = np.zeros_like(films[0])
    for rank in range(3):
        ex = scenes.cornell_box(W, H, 4, device_id=0)
        ex.integrator.tile_size = 8 * H; ex.integrator.tile_rank = rank; ex.integrator.tile_count = 3
        ex.build_scene()
        ex.integrator.render_frames(3)
        total += ex.integrator.hdr.to_numpy()
    assert np.array_equal(total.view(np.uint32), films[0].view(np.uint32))


def test_torch_cuda_still_comes_up_after_this_library(gpu_ctx_ok):
    """PyTorch-ROCm bundles a HIP runtime with the system one's SONAME: the copy loaded first serves the process, and torch.cuda does not come
    up on the system copy.  `_native.lib()` therefore loads torch's first (ti_raytrace_amd/_native.py) -- in a fresh interpreter, rendering
    with this package and THEN asking torch for the device must work (the multi-GPU path of bench.py needs both)."""
    import subprocess, sys, os
    code = ("from ti_raytrace_amd import scenes\n"
            "ex = scenes.cornell_box(32, 32, 2, device_id=0)\n"
            "ex.build_scene(); ex.integrator.render_frames(2)\n"
            "import torch\n"
            "assert torch.cuda.is_available(), 'torch.cuda lost the device'\n"
            "print(float(torch.ones(8, device='cuda:0').sum()))\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ); env.pop("TIRT_SYSTEM_HIP", None)
    pr = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert pr.returncode == 0 and pr.stdout.decode().strip().endswith("8.0"), pr.stdout.decode()[-800:]


@pytest.mark.parametrize("name", ["cornell", "sphere", "cornell_glass", "spot_laser"])
def test_device_film_equals_the_reference_text_film(gpu_ctx_ok, name):
    """tests/golden/refkat_render.npz: the film integrator/PT_RGB.py's own source text produces (executed as plain Python through the
    taichi stand-in of tools/refkat, build container only; tests/test_refkat.py has the details and holds the oracle to it)."""
    from test_refkat import GR, reference_text_scene, film_close
    ex, W, H, frames, seed = reference_text_scene(name, device_id=0)
    ex.integrator.seed = seed
    ex.build_scene()
    if name == "spot_laser":                 # (a bare Example.example: the scene classes of scenes.py do this in their build_scene)
        ex.scene.total_area(); ex.frame_camera(0.8)
    ex.integrator.render_frames(frames)
    got = ex.integrator.hdr.to_numpy()
    from test_refkat import GS
    rel, per = film_close(got, GS["render_spot_laser_film"] if name == "spot_laser" else GR["render_%s_film" % name])
    assert rel <= 1e-5 and per <= 1e-4, (rel, per)
