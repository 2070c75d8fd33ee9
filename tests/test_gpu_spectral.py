"""Spectral path on the MI355X against the oracle: the device's optimiser for the RGB -> spectrum table
(spectrum/JakobSpecTable.py: tirt_spec_table_build) and PT_Spec on the wavefront (k_shade_spec / k_film_spec), fed by the same
host tables.  Everything is deterministic fp32 / fp64 arithmetic in the same order on both sides: bit-identical."""
import numpy as np
import pytest

import oracle_api as oa
from common import rel_l2
from ti_raytrace_amd import scenes, Example, PT_Spec, _native
from ti_raytrace_amd import SceneData as SCD

pytestmark = pytest.mark.gpu


def _oracle_for(ex):
    o = oa.OracleScene(ex.scene, ex.cam); o.lbvh_build()
    if getattr(ex.scene, "normals_processed", False):
        o.process_normal(ex.scene.vertex_index_np)
    o.set_spectral(ex.integrator.tables())
    return o


def test_spec_table_device_equals_oracle(gpu_ctx_ok):
    """3 x 64^3 cells x 3 coefficients from the device's Gauss-Newton (one thread per chain of cells, double precision) and from the
    oracle's: the same doubles, hence the same float32 table."""
    ex = scenes.spectral_box(32, 32, 4, device_id=0)
    ex.scene.setup_data_cpu(); ex.integrator.setup_data_cpu()
    it = ex.integrator
    it._d65_raw = it.d65.data_np.copy()
    xyz, d65 = it.data_np, it.d65_from_360()
    gs, gc = ex.scene.ctx.spec_table_build(64, xyz, d65)
    os_, oc = oa.spec_table_build(64, xyz, d65)
    same = (gc.view(np.uint32) == oc.view(np.uint32)).mean()
    print("table: %.4f %% of %d coefficients bit-identical, max |diff| %.3e" % (100 * same, gc.size, float(np.abs(gc - oc).max())))
    assert np.array_equal(gs, os_) and np.array_equal(gc, oc)
    assert np.isfinite(gc).all() and (gc != 0).mean() > 0.99


def test_spectral_box_film_is_the_oracles(gpu_ctx_ok):
    W = H = 96
    ex = scenes.spectral_box(W, H, 8, device_id=0)
    ex.build_scene()
    ctx = ex.scene.ctx
    o = _oracle_for(ex)
    ctx.stats_reset()
    ex.integrator.render_frames(5)
    got = ex.integrator.hdr.to_numpy()
    want, ost = o.spec_render(W, H, 0, 5, seed=ex.integrator.seed)
    st = ctx.stats()
    print("spectral box 96^2 x5: rel-L2 %.3e, identical pixels %d/%d" % (rel_l2(got, want), (got == want).all(axis=2).sum(), W * H))
    assert np.array_equal(got, want)
    for k in ("rays_closest", "rays_shadow", "shaded", "paths"):
        assert st[k] == ost[k], k
    # frame at a time == batch; tone map works on the spectral film as on any other
    ctx.film_clear(); ex.cam.frame = 0
    for _ in range(5):
        ex.integrator.render(); ex.cam.update_frame()
    assert np.array_equal(ex.integrator.hdr.to_numpy(), want)


def test_sky_dome_and_dispersive_glass(gpu_ctx_ok):
    """example/sky_dome.py (a mirror ball under the analytic sky: Hero.sky_sample on every miss) and a glass ball (Glass.sample_lambda:
    the Sellmeier index of the wavelength picked by Hero.get_rnd_hero) -- films bit-identical to the oracle's."""
    W = H = 64
    ex = scenes.sky_dome(W, H, 4, device_id=0)
    ex.build_scene()
    o = _oracle_for(ex)
    ex.integrator.render_frames(4)
    got = ex.integrator.hdr.to_numpy()
    want, _ = o.spec_render(W, H, 0, 4, seed=ex.integrator.seed)
    assert np.array_equal(got, want, equal_nan=True) and np.isfinite(got).mean() > 0.99 and got[np.isfinite(got)].mean() > 0
    ex = Example.example(W, H, 4, 0)
    ex.scene.add_obj(scenes.asset("model", "sphere.obj"))
    ex.scene.material_cpu[0].type = SCD.MAT_GLASS
    ex.scene.material_cpu[0].setIor(1.5); ex.scene.material_cpu[0].setExtinciton(5.0)
    ex.add_sphere_light(pos=(0.0, 4.0, 0.0), radius=0.8, emission=30.0)
    ex.integrator = PT_Spec.PathTrace(W, H, ex.cam, ex.scene, 64)
    ex.build_scene(); ex.scene.process_normal(); ex.scene.total_area(); ex.frame_camera(1.0)
    o = _oracle_for(ex)
    ex.scene.ctx.stats_reset()
    ex.integrator.render_frames(4)
    got = ex.integrator.hdr.to_numpy()
    want, ost = o.spec_render(W, H, 0, 4, seed=ex.integrator.seed)
    assert np.array_equal(got, want, equal_nan=True)
    assert ex.scene.ctx.stats()["rays_closest"] == ost["rays_closest"]


def test_spectral_tiles_reassemble(gpu_ctx_ok):
    W = H = 64
    full = None
    acc = np.zeros((W, H, 3), np.float32)
    for rank, count in ((0, 1), (0, 3), (1, 3), (2, 3)):
        ex = scenes.spectral_box(W, H, 4, device_id=0, tile_rank=rank, tile_count=count, tile_size=256)
        ex.build_scene()
        ex.integrator.render_frames(3)
        h = ex.integrator.hdr.to_numpy()
        if count == 1:
            full = h
        else:
            acc += h
    assert np.array_equal(acc, full)


# ---- the device against values computed by the reference's own source text (tests/golden/refkat_spec.npz, tools/refkat/make_refkat_spec.py) ----
import os
GS = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "refkat_spec.npz"))
SPEC_KATS = {0: 1, 1: 4, 2: 12, 3: 3, 4: 1, 5: 4, 6: 4, 7: 4, 8: 4, 9: 3, 10: 4, 11: 2}


@pytest.fixture(scope="module")
def spectral_box16_dev(gpu_ctx_ok):
    ex = scenes.spectral_box(16, 16, 4, device_id=0)
    ex.build_scene()                         # the device builds the RGB -> spectrum table (bit-identical to the oracle's, first test of this file)
    return ex


@pytest.mark.parametrize("which", sorted(SPEC_KATS))
def test_device_spectral_functions_equal_the_reference_text(spectral_box16_dev, which):
    """tirt_kat_spec: Spectrum.sample, HeroSample.*, Rgb2Spec.fetch / eval, the sky model, PathTrace.emission_to_rad / get_spec_power / AddSplat on the
    device, on the tables it was given, against the reference's text (which read the table this repo's generator makes): bit for bit."""
    ctx = spectral_box16_dev.scene.ctx
    got = ctx.kat_spec(which, GS["spec_k%d_in" % which], SPEC_KATS[which])
    want = GS["spec_k%d" % which]
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (which, float(np.abs(got - want).max()))


def test_device_pt_spec_film_equals_the_reference_text_film(spectral_box16_dev):
    """integrator/PT_Spec.py:189-279 executed from its source text (16 x 16 x 4 frames of example/spectral_box.py) against k_shade_spec / k_film_spec."""
    ex = spectral_box16_dev
    W, H, frames, seed = [int(x) for x in GS["render_spec_box_cfg"]]
    ctx = ex.scene.ctx
    ctx.film_clear()
    ctx.pt_spec_render(0, frames, seed, 10, 64, 0)
    got = ctx.film_download(W, H)[0]
    want = GS["render_spec_box_film"]
    rel = rel_l2(got, want)
    print("PT_Spec 16x16x4: device vs reference text rel-L2 %.2e" % rel)
    assert rel < 1e-5
