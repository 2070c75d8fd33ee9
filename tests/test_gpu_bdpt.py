"""BDPT_RGB (BASELINE config 5): HIP kernel vs the CPU restatement at the same
counter-based seed.  Parity is unpinned by the reference (no golden output exists for BDPT).
Light-tracing contributions are float atomics on other pixels, so frames agree up to the order of
those additions: tolerance = the north star's 1e-3 relative L2 (measured ~1e-7)."""
import numpy as np
import pytest

import oracle_api as oa
from common import rel_l2
from ti_raytrace_amd import scenes

pytestmark = pytest.mark.gpu


def run_both(ex, W, H, frames):
    ex.build_scene()
    o = oa.OracleScene(ex.scene, ex.cam)
    o.lbvh_build()
    if ex.scene.vertex_index_np is not None and type(ex).__name__ == "veach_bdpt":
        o.process_normal(ex.scene.vertex_index_np)
    ctx = ex.scene.ctx
    ctx.set_option("bdpt_state_fill", 2)          # vertex arrays poisoned with 0xFF before every batch: nothing may read an unwritten slot
    ctx.stats_reset()
    # frame by frame on the GPU, in one call on the oracle: the persistent per-pixel vertex state must carry over
    for _ in range(frames):
        ex.integrator.render()
        ex.cam.update_frame()
    got = ex.integrator.hdr.to_numpy()
    want, ost, _ = o.bdpt_render(ex.cam, W, H, 0, frames, seed=ex.integrator.seed)
    st = ctx.stats()
    return got, want, st, ost


def test_cornell_bdpt(gpu_ctx_ok):
    W = H = 40
    ex = scenes.cornell_box(W, H, 4, device_id=0)
    from ti_raytrace_amd import BDPT_RGB
    ex.integrator = BDPT_RGB.BDPT(W, H, ex.cam, ex.scene, 64)
    got, want, st, ost = run_both(ex, W, H, 3)
    r = rel_l2(got, want)
    print("cornell BDPT 40^2 x3: rel-L2 %.3e, mean %s" % (r, got.reshape(-1, 3).mean(0)))
    assert np.isfinite(got).all() and r <= 1e-3
    assert st["rays_closest"] == ost["rays_closest"] and st["rays_shadow"] == ost["rays_shadow"]


def test_veach_scene_bdpt(gpu_ctx_ok):
    """bdpt.obj (11 544 triangles, glass + two emissive meshes), smooth normals, 2 frames."""
    W = H = 32
    ex = scenes.veach_bdpt(W, H, 4, device_id=0)
    got, want, st, ost = run_both(ex, W, H, 2)
    m = np.isfinite(want).all(axis=2) & np.isfinite(got).all(axis=2)
    assert m.mean() > 0.98
    assert (np.isfinite(want).all(axis=2) == np.isfinite(got).all(axis=2)).all()
    r = rel_l2(got[m], want[m])
    print("veach BDPT 32^2 x2: rel-L2 %.3e over %d finite pixels" % (r, m.sum()))
    assert r <= 1e-3
    assert st["rays_closest"] == ost["rays_closest"] and st["rays_shadow"] == ost["rays_shadow"]


def test_bdpt_tiles_sum_to_full_frame(gpu_ctx_ok):
    """With pixel tiles every context splats into its own full-size film; the films add up."""
    W = H = 32
    ex = scenes.cornell_box(W, H, 4, device_id=0)
    from ti_raytrace_amd import BDPT_RGB
    ex.integrator = BDPT_RGB.BDPT(W, H, ex.cam, ex.scene, 64)
    ex.build_scene()
    ex.integrator.render_frames(2)
    full = ex.integrator.hdr.to_numpy()
    acc = np.zeros_like(full)
    for rank in range(2):
        ex.scene.ctx.film_create(W, H, rank, 2, 64)
        ex.scene.ctx.bdpt_rgb_render(0, 2, ex.integrator.seed)
        acc += ex.scene.ctx.film_download(W, H)[0]
    assert rel_l2(acc, full) <= 1e-5


def test_bounded_connection_rays_give_the_same_film(gpu_ctx_ok):
    """Connection rays cut off at their target distance ("bdpt_bounded", default) against the reference-style
    full closest-hit query, on the Veach scene at 192^2 x 4 frames (12 M connection rays): same ray counts,
    films equal up to the float-atomic order of the light-tracing splats."""
    W = H = 192
    films, stats = [], []
    for bounded in (1, 0):
        ex = scenes.veach_bdpt(W, H, 4, device_id=0)
        ex.build_scene()
        ctx = ex.scene.ctx
        ctx.set_option("bdpt_bounded", bounded)
        ctx.stats_reset()
        ex.integrator.render_frames(4)
        films.append(ex.integrator.hdr.to_numpy())
        stats.append(ctx.stats())
    assert stats[0]["rays_closest"] == stats[1]["rays_closest"] and stats[0]["rays_shadow"] == stats[1]["rays_shadow"]
    m = np.isfinite(films[0]).all(axis=2) & np.isfinite(films[1]).all(axis=2)
    assert (np.isfinite(films[0]).all(axis=2) == np.isfinite(films[1]).all(axis=2)).all() and m.mean() > 0.98
    assert rel_l2(films[0][m], films[1][m]) <= 1e-6


def _blocks(rgb, n=32):
    img = np.transpose(rgb, (1, 0, 2))[::-1]                  # film (i, j) -> PNG rows/cols (Example.write_png)
    b = img.shape[0] // n
    return img.reshape(n, b, n, b, 3).mean(axis=(1, 3))


def test_config5_matches_the_reference_gallery_render(gpu_ctx_ok):
    """BASELINE config 5 at its full size (veach_bdpt.py: bdpt.obj, BDPT_RGB, 512^2, exposure 0.5 of Example.py:43), 64 spp,
    against the reference's own render of it, image/veach-bdpt512.png (block means in tests/golden; the reference recorded
    neither seed nor sample count, so this is a statistical pin like the out.png one): mean colour within 2 %, 16x16-block
    means within 5 % relative L2.  The same scene through PT_RGB is compared with image/veach-pt512.png more loosely (the
    caustic paths PT finds rarely carry much of that image's energy: at 256 spp it still sits 10 % below the reference)."""
    import os
    from ti_raytrace_amd import UtilsFunc as UF
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    W = H = 512
    ex = scenes.veach_bdpt(W, H, 64, device_id=0)
    ex.build_scene()
    ex.integrator.render_frames(64)
    UF.tone_map(0.5, ex.integrator.hdr, ex.integrator.rgb_film)
    ours = _blocks(ex.integrator.rgb_film.to_numpy())
    ref = np.load(os.path.join(gold, "veach_bdpt512_blocks.npy"))
    mean_err = np.abs(ours.reshape(-1, 3).mean(0) - ref.reshape(-1, 3).mean(0)) / ref.reshape(-1, 3).mean(0)
    r = rel_l2(ours, ref)
    print("veach BDPT 512^2 x64 vs veach-bdpt512.png: mean err %s, block rel-L2 %.4f" % (np.round(mean_err, 4), r))
    assert (mean_err < 0.02).all() and r < 0.05
    ex = scenes.veach_bdpt(W, H, 256, device_id=0, integrator="pt")
    ex.build_scene()
    ex.render_all(batch=64)
    UF.tone_map(0.5, ex.integrator.hdr, ex.integrator.rgb_film)
    ours = _blocks(ex.integrator.rgb_film.to_numpy())
    ref = np.load(os.path.join(gold, "veach_pt512_blocks.npy"))
    mean_err = np.abs(ours.reshape(-1, 3).mean(0) - ref.reshape(-1, 3).mean(0)) / ref.reshape(-1, 3).mean(0)
    print("veach PT 512^2 x256 vs veach-pt512.png: mean err %s, block rel-L2 %.4f" % (np.round(mean_err, 4), rel_l2(ours, ref)))
    assert (mean_err < 0.20).all() and rel_l2(ours, ref) < 0.25


# (No Cornell-BDPT-vs-out.png test: with the reference's mis_weight comparing a material INDEX with MAT_DISNEY
# (integrator/BDPT_RGB.py:364,379,432 -- kept, see oracle.c) BDPT_RGB is not a consistent estimator of PT_RGB's image on scenes
# whose non-zero material indices are Disney materials: Cornell comes out ~2x brighter, in the oracle as on the device.  The
# reference's own BDPT render exists only for config 5, which is what is pinned above.)

def test_config5_full_size_frame_against_the_oracle(gpu_ctx_ok):
    """BASELINE config 5 at its full 512^2 (one frame = 262 144 eye + light sub-path pairs, 3.5 M rays): device film vs the
    oracle's -- same NaN pixels, rel-L2 <= 1e-3 on the finite ones (float-atomic order of the splats), same ray counts."""
    W = H = 512
    ex = scenes.veach_bdpt(W, H, 4, device_id=0)
    got, want, st, ost = run_both(ex, W, H, 1)
    gn, wn = np.isnan(got).any(axis=2), np.isnan(want).any(axis=2)
    print("config 5, 512^2 x 1 frame: %d NaN pixels (oracle %d), rays %d + %d" % (gn.sum(), wn.sum(), st["rays_closest"], st["rays_shadow"]))
    assert (gn == wn).all()
    m = np.isfinite(want).all(axis=2) & np.isfinite(got).all(axis=2)
    assert rel_l2(got[m], want[m]) <= 1e-3
    assert st["rays_closest"] == ost["rays_closest"] and st["rays_shadow"] == ost["rays_shadow"]


def test_bdpt_on_a_deep_duplicate_chain(gpu_ctx_ok):
    """Runs of identical Morton codes make right-deep chains under the reference's duplicate rule (accel/LBvh.py:240-251).
    BDPT's rays go through the same paged traversal stack as PT_RGB's (round 1's BDPT had its own 64-entry stack that dropped
    pushes silently): film and ray counts equal the oracle's, no overflow reported."""
    from common import duplicate_code_scene
    from ti_raytrace_amd import BDPT_RGB
    W = H = 32
    ex = duplicate_code_scene(W=W, H=H, device_id=0)
    ex.integrator = BDPT_RGB.BDPT(W, H, ex.cam, ex.scene, 64)
    ex.build_scene(); ex.frame_camera(0.8)
    o = oa.OracleScene(ex.scene, ex.cam); o.lbvh_build()
    ctx = ex.scene.ctx
    ctx.set_option("trace_lds_depth", 12)            # page the stack as early as possible
    ctx.stats_reset()
    ex.integrator.render_frames(3)
    got = ex.integrator.hdr.to_numpy()
    want, ost, _ = o.bdpt_render(ex.cam, W, H, 0, 3, seed=ex.integrator.seed)
    st = ctx.stats()                                 # raises TirtStackOverflow if a ray dropped a subtree
    m = np.isfinite(want).all(axis=2) & np.isfinite(got).all(axis=2)
    assert (np.isfinite(want).all(axis=2) == np.isfinite(got).all(axis=2)).all() and rel_l2(got[m], want[m]) <= 1e-3
    assert st["rays_closest"] == ost["rays_closest"] and st["rays_shadow"] == ost["rays_shadow"] and st["stack_overflow"] == 0


def test_bdpt_batching_and_lanes_change_nothing(gpu_ctx_ok):
    """6 frames as one batch on the main stream, as batches of one frame alternating between the two BDPT lanes (the `delta`
    memory of k_bd_delta and the running mean of the film are chained from batch to batch), and as two batches of three: the same
    ray counts and films within the float-atomic order of the splats; the oracle's frame-by-frame render agrees."""
    W = H = 48
    films, counts = [], []
    for items, lanes in ((1 << 29, 1), (W * H * 2, 4), (W * H * 6, 4)):        # one batch | 1 frame per lane batch | 3 frames per lane batch
        ex = scenes.veach_bdpt(W, H, 8, device_id=0)
        ex.build_scene()
        ctx = ex.scene.ctx
        ctx.set_option("bdpt_batch_items", items); ctx.set_option("overlap_lanes", lanes)
        ctx.stats_reset()
        ctx.bdpt_rgb_render(0, 6, 1)
        st = ctx.stats()
        films.append(ctx.film_download(W, H)[0]); counts.append((st["rays_closest"], st["rays_shadow"]))
    assert counts[0] == counts[1] == counts[2]
    m = np.isfinite(films[0]).all(axis=2)
    for f in films[1:]:
        assert (np.isfinite(f).all(axis=2) == m).all()
        assert rel_l2(f[m], films[0][m]) <= 1e-5
    o = oa.OracleScene(ex.scene, ex.cam); o.lbvh_build()
    o.process_normal(ex.scene.vertex_index_np)
    want, ost, _ = o.bdpt_render(ex.cam, W, H, 0, 6, seed=1)
    mm = m & np.isfinite(want).all(axis=2)
    assert mm.mean() > 0.95 and rel_l2(films[1][mm], want[mm]) <= 1e-3
    assert counts[1] == (ost["rays_closest"], ost["rays_shadow"])


def test_vertex_arrays_need_no_clearing(gpu_ctx_ok):
    """The per-item vertex arrays are not cleared between batches (the reference clears beta/type/fpdf/rpdf per frame,
    BDPT_RGB.py:95-103; here no read reaches a slot its item has not written, and BdStep::e_tail tells k_bd_delta about
    the one slot past the depth).  8 frames in batches of two on alternating lanes -- every batch after the first sees
    the previous batch's vertices -- with the arrays left as they are, zero-filled and filled with 0xFF (NaN floats,
    type/prim/mat/delta -1): same ray counts, same non-finite pixels, films within the float-atomic order of the splats."""
    W = H = 64
    films, counts = [], []
    for fill in (1, 0, 2):
        ex = scenes.veach_bdpt(W, H, 8, device_id=0)
        ex.build_scene()
        ctx = ex.scene.ctx
        ctx.set_option("bdpt_batch_items", W * H * 2); ctx.set_option("bdpt_state_fill", fill)
        ctx.stats_reset()
        ctx.bdpt_rgb_render(0, 8, 1)
        st = ctx.stats()
        films.append(ctx.film_download(W, H)[0]); counts.append((st["rays_closest"], st["rays_shadow"]))
    assert counts[0] == counts[1] == counts[2]
    m = np.isfinite(films[0]).all(axis=2)
    assert m.mean() > 0.95
    for f in films[1:]:
        assert (np.isfinite(f).all(axis=2) == m).all()
        assert rel_l2(f[m], films[0][m]) <= 1e-6


def test_bdpt_batches_shrink_to_the_free_memory(gpu_ctx_ok):
    """A BDPT call sizes its batches by `bdpt_batch_items` AND by what the device has free (tirt_bdpt.hip: hipMemGetInfo less 2 GB): with only
    6 GB to take -- option "bdpt_mem_budget", which caps what the call believes is free; nothing is allocated to squeeze it, other jobs on the
    device are left alone -- 64 frames of 256^2 (4 Mi items: 12 GB of wavefront state as one pair of batches) still render, in smaller
    batches, with the same ray counts and the same film up to the float-atomic order of the splats."""
    W = H = 256
    out = []
    for squeeze in (False, True):
        ex = scenes.veach_bdpt(W, H, 64, device_id=0)
        ex.build_scene()
        ctx = ex.scene.ctx
        if squeeze:
            ctx.set_option("bdpt_mem_budget", float(6 << 30))
        try:
            ctx.stats_reset()
            ctx.bdpt_rgb_render(0, 64, 1)
            st = ctx.stats()
            out.append((ctx.film_download(W, H)[0], st["rays_closest"], st["rays_shadow"]))
        finally:
            ctx.close()
    assert out[0][1:] == out[1][1:]
    m = np.isfinite(out[0][0]).all(axis=2)
    assert (np.isfinite(out[1][0]).all(axis=2) == m).all() and m.mean() > 0.95
    assert rel_l2(out[1][0][m], out[0][0][m]) <= 1e-5


@pytest.mark.parametrize("name", ["cornell", "cornell_glass", "spot_laser"])
def test_device_bdpt_film_equals_the_reference_text_film(gpu_ctx_ok, name):
    """tests/golden/refkat_bdpt.npz: the film integrator/BDPT_RGB.py's own source text produces (executed as plain Python through the taichi
    stand-in of tools/refkat, build container only; tests/test_refkat.py has the details and holds the oracle to it)."""
    from test_refkat import GB, reference_text_scene, film_close
    ex, W, H, frames, seed = reference_text_scene(name, device_id=0, bdpt=True)
    ex.integrator.seed = seed
    ex.build_scene()
    if name == "spot_laser":                 # (a bare Example.example: the scene classes of scenes.py do this in their build_scene)
        ex.scene.total_area(); ex.frame_camera(0.8)
    ex.integrator.render_frames(frames)
    got = ex.integrator.hdr.to_numpy()
    from test_refkat import GS
    rel, per = film_close(got, GS["bdpt_spot_laser_film"] if name == "spot_laser" else GB["bdpt_%s_film" % name])
    assert rel <= 1e-5 and per <= 1e-4, (rel, per)
