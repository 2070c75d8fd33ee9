"""BDPT_SPEC (integrator/BDPT_SPEC.py, the other half of SURVEY.md 8f rank 4; example/prism_rainbow.py): the HIP kernels
(tirt_bdpt.hip, template <bool SPEC>) against the CPU restatement at the same counter-based seed.  As for BDPT_RGB the
light-tracing contributions are float atomics on other pixels, so films agree up to the order of those additions: tolerance =
1e-3 relative L2 (measured ~1e-7); ray counts are equal."""
import numpy as np
import pytest

import oracle_api as oa
from common import rel_l2, spot_laser_scene
from ti_raytrace_amd import scenes, BDPT_SPEC

pytestmark = pytest.mark.gpu


def run_both(ex, W, H, frames, batch=False, frame_camera=False):
    ex.build_scene()
    if frame_camera:
        ex.scene.total_area(); ex.frame_camera(0.8)
    o = oa.OracleScene(ex.scene, ex.cam)
    o.lbvh_build()
    if getattr(ex.scene, "normals_processed", False):
        o.process_normal(ex.scene.vertex_index_np)
    o.set_spectral(ex.integrator.tables())
    ctx = ex.scene.ctx
    ctx.set_option("bdpt_state_fill", 2)          # vertex arrays poisoned with 0xFF before every batch: nothing may read an unwritten slot
    ctx.stats_reset()
    if batch:
        ex.integrator.render_frames(frames)
    else:                                # frame by frame: the persistent per-pixel `delta` memory must carry over
        for _ in range(frames):
            ex.integrator.render()
            ex.cam.update_frame()
    got = ex.integrator.hdr.to_numpy()
    want, ost, _ = o.bdpt_spec_render(ex.cam, W, H, 0, frames, seed=ex.integrator.seed, stack_size=1024)
    return got, want, ctx.stats(), ost


@pytest.mark.parametrize("with_sphere_light", [True, False])
def test_prism_rainbow(gpu_ctx_ok, with_sphere_light):
    """example/prism_rainbow.py: a laser (an emitter without a surface: only the light sub-path carries its light) through a glass
    prism (Glass.sample_lambda: the refraction depends on the sample's wavelength), with and without the sphere light above it."""
    W = H = 48
    ex = scenes.prism_rainbow(W, H, 4, device_id=0, with_sphere_light=with_sphere_light)
    got, want, st, ost = run_both(ex, W, H, 3)
    r = rel_l2(got, want)
    print("prism_rainbow (sphere light %s) 48^2 x3: rel-L2 %.3e, lit pixels %d / %d, max %.3g" % (with_sphere_light, r, int((want.sum(axis=2) > 0).sum()), W * H, float(want.max())))
    assert np.isfinite(got).all() and (want.sum(axis=2) > 0).sum() > 50 and r <= 1e-3
    assert st["rays_closest"] == ost["rays_closest"] and st["rays_shadow"] == ost["rays_shadow"]


def test_cornell_with_spot_and_laser_bdpt_spec(gpu_ctx_ok):
    """The Cornell box with its quad light, a spot and a laser through BDPT_SPEC, several frames in one call (batch path)."""
    W = H = 40
    ex = spot_laser_scene(W, H, device_id=0)
    ex.integrator = BDPT_SPEC.BDPT(W, H, ex.cam, ex.scene, 64)
    got, want, st, ost = run_both(ex, W, H, 3, batch=True, frame_camera=True)
    r = rel_l2(got, want)
    print("cornell + spot + laser BDPT_SPEC 40^2 x3: rel-L2 %.3e, mean %s" % (r, got.reshape(-1, 3).mean(0)))
    assert np.isfinite(got).all() and got.mean() > 0 and r <= 1e-3
    assert st["rays_closest"] == ost["rays_closest"] and st["rays_shadow"] == ost["rays_shadow"]


def test_rainbow_at_the_gallery_size(gpu_ctx_ok):
    """The structure pin of tests/test_bdpt_spec.py on the product path at the gallery image's own 512^2 (laser alone, the example's
    commented-out close-up camera), 64 samples per pixel."""
    import os
    W = H = 512
    ex = scenes.prism_rainbow(W, H, 64, device_id=0, with_sphere_light=False)
    ex.build_scene()
    ex.cam.yaw = 0.8; ex.cam.scale = 20.0; ex.cam.set_target(-50.0, 2.0, -93.0); ex.cam.update()
    ex.integrator.render_frames(64)
    hdr = ex.integrator.hdr.to_numpy()
    img = np.transpose(hdr, (1, 0, 2))[::-1].astype(np.float64)
    assert np.isfinite(img).all()
    lum = img.sum(axis=2)
    ref = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rainbow_blocks.npy")).astype(np.float64).sum(axis=2)

    def band(l, frac=0.2):
        ys, xs = np.nonzero(l > frac * l.max()); n = l.shape[0]
        return ys.min() / n, (ys.max() + 1) / n, xs.min() / n, (xs.max() + 1) / n
    blocks = lum.reshape(64, 8, 64, 8).mean(axis=(1, 3))
    y0, y1, x0, x1 = band(blocks); ry0, ry1, rx0, rx1 = band(ref)
    print("512^2 x 64: spectrum band rows %.2f-%.2f cols %.2f-%.2f (image/rainbow.png: %.2f-%.2f, %.2f-%.2f)" % (y0, y1, x0, x1, ry0, ry1, rx0, rx1))
    assert abs(0.5 * (y0 + y1) - 0.5 * (ry0 + ry1)) < 0.10 and abs(0.5 * (x0 + x1) - 0.5 * (rx0 + rx1)) < 0.10
    assert min(y1, ry1) - max(y0, ry0) > 0.25 and min(x1, rx1) - max(x0, rx0) > 0.02
    ys, xs = np.nonzero(lum > 0.2 * lum.max())
    prof = img[ys.min():ys.max() + 1].sum(axis=0); cols = np.arange(W)
    cr, cg, cb = [(cols * prof[:, k]).sum() / prof[:, k].sum() for k in range(3)]
    print("centroid columns: blue %.1f < green %.1f < red %.1f" % (cb, cg, cr))
    assert cb + 2 < cg < cr - 2


def test_device_bdpt_spec_film_equals_the_reference_text_film(gpu_ctx_ok):
    """tests/golden/refkat_spec.npz: the film integrator/BDPT_SPEC.py's own source text produces on example/prism_rainbow.py (16 x 16 x 4 frames; executed as
    plain Python through the taichi stand-in of tools/refkat, build container only; tests/test_refkat.py has the details and holds the oracle to it)."""
    from test_refkat import GS, prism_scene, prism_film_close
    ex, W, H, frames, seed = prism_scene(device_id=0)
    ex.integrator.seed = seed
    ex.build_scene()
    ex.integrator.render_frames(frames)
    got = ex.integrator.hdr.to_numpy()
    rel, per, n_ill = prism_film_close(got, GS["bdpt_spec_prism_film"])
    print("BDPT_SPEC prism: device vs reference text rel-L2 %.2e, worst value %.2e" % (rel, per))
    assert rel <= 1e-5 and per <= 1e-4
