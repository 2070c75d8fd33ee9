"""Spectral path on the CPU: the oracle's restatement of spectrum/JakobSpecTable.py, Rgb2Spec, Spectrum,
HeroSample, sky/Sky.py and integrator/PT_Spec.py, fed by the host mirrors' tables.  The reference holds no test for any of this
and lacks the table file itself (spectrum/spec_table: .MISSING_LARGE_BLOBS), so the pins are the defining properties of the
pieces -- and the reference's gallery render image/spectral-cornellbox.png for the structure of the whole."""
import os

import numpy as np
import pytest

import oracle_api as oa
from ti_raytrace_amd import scenes, PT_Spec

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def spectral_box_oracle():
    W = H = 128
    ex = scenes.spectral_box(W, H, 64)
    ex.scene.setup_data_cpu(); ex.frame_camera(0.8)
    ex.integrator.setup_data_cpu()
    ex.integrator.setup_tables(lambda res, xyz, d65: oa.spec_table_build(res, xyz, d65))
    o = oa.OracleScene(ex.scene, ex.cam); o.lbvh_build(); o.process_normal(ex.scene.vertex_index_np)
    o.set_spectral(ex.integrator.tables())
    return ex, o


def _srgb_to_lrgb(c):
    c = np.asarray(c, np.float64)
    return np.where(c < 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4)


def test_rgb2spec_table_reproduces_its_colours(spectral_box_oracle):
    """What the Jakob-Hanika table is FOR: the sigmoid spectrum fetched for a colour, lit by D65 and seen by the CIE 1931 observer,
    is that colour again (the optimiser drives the CIE Lab error below 1e-3, JakobSpecTable.py:323).  Checked here in numpy with
    an independent restatement of Rgb2Spec.fetch / eval on the table the oracle's optimiser produced: pins the optimiser, the table
    layout (component order, scale axis, coefficient order), the CIE and D65 data and the xyz_to_srgb matrix in one go."""
    ex, _ = spectral_box_oracle
    it = ex.integrator
    res = it.rgb2spec.table_res
    scale = it.rgb2spec.table_scale_np.astype(np.float64)
    data = it.rgb2spec.table_data_np.astype(np.float64).reshape(3, res, res, res, 3)         # [l][k = z][j = y][i = x][coefficient]
    lam = np.arange(360.0, 831.0)
    xyz = it.data_np.astype(np.float64)
    d65 = it.d65_from_360().astype(np.float64)
    wgt = np.full(471, 3.0); wgt[3:-1:3] = 2.0; wgt[0] = wgt[-1] = 1.0                         # Simpson 3/8 (JakobSpecTable.py:336-343)
    white_Y = (xyz[:, 1] * d65 * wgt).sum()
    M = np.array([[3.240479, -1.537150, -0.498535], [-0.969256, 1.875991, 0.041556], [0.055648, -0.204043, 1.057311]])
    rng = np.random.RandomState(4)
    worst = 0.0
    for rgb in list(rng.uniform(0.02, 0.98, (60, 3))) + [np.array([0.8, 0.8, 0.8]), np.array([0.9, 0.1, 0.1]), np.array([0.05, 0.6, 0.1])]:
        l = int(np.argmax(rgb))
        order = {0: (1, 2, 0), 1: (2, 0, 1), 2: (0, 1, 2)}[l]                                   # Rgb2Spec.get_max_component (:50-74)
        x, y, z = rgb[order[0]], rgb[order[1]], rgb[order[2]]
        x, y = x * (res - 1) / z, y * (res - 1) / z
        xi, yi = min(int(x), res - 2), min(int(y), res - 2)
        zi = int(np.clip(np.searchsorted(scale, z, side="right") - 1, 0, res - 2))
        fx, fy, fz = x - xi, y - yi, (z - scale[zi]) / (scale[zi + 1] - scale[zi])
        c = np.zeros(3)
        for dz in (0, 1):
            for dy in (0, 1):
                for dx in (0, 1):
                    c += data[l, zi + dz, yi + dy, xi + dx] * ((fx if dx else 1 - fx) * (fy if dy else 1 - fy) * (fz if dz else 1 - fz))
        xx = (c[0] * lam + c[1]) * lam + c[2]
        spec = 0.5 * xx / np.sqrt(xx * xx + 1.0) + 0.5
        assert (spec >= 0).all() and (spec <= 1).all()
        XYZ = ((xyz * (spec * d65 * wgt)[:, None]).sum(axis=0)) / white_Y
        back = M @ XYZ
        worst = max(worst, float(np.abs(back - rgb).max()))
    print("rgb -> spectrum -> D65 x CIE 1931 -> rgb: worst component error %.2e over 63 colours" % worst)
    assert worst < 4e-3


def test_d65_is_normalised_and_tables_have_the_reference_shapes(spectral_box_oracle):
    ex, _ = spectral_box_oracle
    it = ex.integrator
    assert it.size == 471 and it.lambda_min == 360.0 and it.lambda_max == 830.0 and abs(it.lambda_range - 1.0) < 1e-12
    assert it.d65.size == 531 and it.white.size == it.red.size == it.green.size == 76
    # PathTrace.normalize_spec (PT_Spec.py:93-100): after the scaling the illuminant's Y is 1
    spec = it.d65
    it2 = PT_Spec.PathTrace(8, 8, ex.cam, ex.scene, 64); it2.setup_data_cpu()
    it2.d65.data_np = spec.data_np.copy(); it2.cal_white_point(it2.d65)
    assert abs(float(it2.d65.white_point_np[0, 1]) - 1.0) < 1e-4
    assert it.sky.configs_np.shape == (11, 9) and np.isfinite(it.sky.configs_np).all() and (it.sky.radiances_np > 0).all()
    assert abs(np.linalg.norm(it.sky.sun_dir_np[0]) - 1.0) < 1e-6


def test_spectral_cornell_box_against_the_reference_gallery_render(spectral_box_oracle):
    """example/spectral_box.py through the oracle, 128^2 x 64 spp, against image/spectral-cornellbox.png (block means,
    tests/golden/spectral_cornellbox_blocks.npy).  A STRUCTURE pin: the committed PT_Spec tints its light sample with the colour of
    the surface that was hit instead of the light's emission (PT_Spec.py:213, 248) -- next-event estimation is ~17x too dim and the
    box is lit by the paths that find the lamp by chance --, so the gallery render (sky 2-4x darker, walls 4x brighter than what the
    committed code produces) comes from another revision.  What must agree: the red wall is red and the green wall green through the
    tabulated reflectances red-spec / green-spec (MAT_SPECTRAL, alebdoTex 1 / 2), the sky surrounds the box with the gallery's
    gradient (bluish above, brownish below: sky/Sky.py through Hero.sky_sample), and inside the box the log luminance follows the
    gallery's."""
    ex, o = spectral_box_oracle
    W = H = 128
    hdr, st = o.spec_render(W, H, 0, 64, seed=1)
    assert np.isfinite(hdr).all() and st["overflow"] == 0 and st["rays_shadow"] > 0
    img = np.transpose(hdr, (1, 0, 2))[::-1].astype(np.float64)
    ours = img.reshape(32, 4, 32, 4, 3).mean(axis=(1, 3))
    ref = _srgb_to_lrgb(np.load(os.path.join(GOLD, "spectral_cornellbox_blocks.npy")))
    left, right = ours[10:22, 2:5].mean((0, 1)), ours[10:22, 27:30].mean((0, 1))
    rl, rr = ref[10:22, 2:5].mean((0, 1)), ref[10:22, 27:30].mean((0, 1))
    print("red wall %s (gallery %s), green wall %s (gallery %s)" % (left.round(4), rl.round(4), right.round(4), rr.round(4)))
    assert left[0] > 5 * left[1] and left[0] > 5 * left[2] and rl[0] > 5 * rl[1]
    assert right[1] > 3 * right[0] and right[1] > 3 * right[2] and rr[1] > 3 * rr[0]
    top, bottom = ours[0].mean(0), ours[31].mean(0)
    gt, gb = ref[0].mean(0), ref[31].mean(0)
    print("sky above %s (gallery %s), below %s (gallery %s)" % (top.round(4), gt.round(4), bottom.round(4), gb.round(4)))
    assert top[2] > top[0] and gt[2] > gt[0] and bottom[0] > bottom[2] and gb[0] > gb[2]           # bluish above, brownish below
    w = np.array([0.2126, 0.7152, 0.0722])
    inside = (slice(3, 29), slice(3, 29))
    a, b = np.log(1e-3 + ours[inside] @ w).ravel(), np.log(1e-3 + ref[inside] @ w).ravel()
    a -= a.mean(); b -= b.mean()
    c = float((a * b).sum() / np.sqrt((a * a).sum() * (b * b).sum()))
    print("log-luminance correlation inside the box %.3f" % c)
    assert c > 0.6


def test_spectral_render_is_deterministic_and_tiles_add_up(spectral_box_oracle):
    ex, o = spectral_box_oracle
    W = H = 128
    a, sa = o.spec_render(W, H, 0, 2, seed=5, p_begin=5000, p_end=9000, nthreads=1)
    b, sb = o.spec_render(W, H, 0, 2, seed=5, p_begin=5000, p_end=9000, nthreads=5)
    assert np.array_equal(a, b) and sa == sb
    h2, _ = o.spec_render(W, H, 0, 1, seed=5, p_begin=5000, p_end=9000)
    h3, _ = o.spec_render(W, H, 1, 1, seed=5, hdr=h2.copy(), p_begin=5000, p_end=9000)
    assert np.array_equal(h3, a)
