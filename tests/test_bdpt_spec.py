"""BDPT_SPEC on the CPU (oracle/oracle.c: bd_* with a `bd_spec`), SURVEY.md 8f rank 4.  The reference cannot run this integrator
(its spectrum/spec_table is missing) and holds no output of it except one gallery image: image/rainbow.png, example/prism_rainbow.py's
laser through a glass prism.  That image was taken with the close-up camera the example keeps as comments (prism_rainbow.py:58-60: yaw 0.8,
scale 20, target (-50, 2, -93)) and is black outside the spectrum on the far wall -- the laser alone.  It shows two overlapping spectra
(an earlier beam / prism set-up); the committed example gives one.  So this is a STRUCTURE pin: where the dispersed beam lands in the
frame, that nothing else is lit, and the order of the colours -- which constrains the laser emitter (Scene.sample_light's LASER branch:
an emitter only the light sub-path can see), Glass.sample_lambda's wavelength-dependent refraction through four glass interfaces, the
l = 6, e = 1 light-tracing connection with get_image_point, and AddSplat's wavelength -> sRGB."""
import os

import numpy as np
import pytest

import oracle_api as oa
from ti_raytrace_amd import scenes

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def rainbow_oracle():
    W = H = 96
    ex = scenes.prism_rainbow(W, H, 32, with_sphere_light=False)
    ex.scene.setup_data_cpu()
    ex.integrator.setup_data_cpu()
    ex.integrator.setup_tables(lambda res, xyz, d65: oa.spec_table_build(res, xyz, d65))
    ex.cam.yaw = 0.8; ex.cam.scale = 20.0; ex.cam.set_target(-50.0, 2.0, -93.0); ex.cam.update()
    o = oa.OracleScene(ex.scene, ex.cam); o.lbvh_build(); o.set_spectral(ex.integrator.tables())
    hdr, st, _ = o.bdpt_spec_render(ex.cam, W, H, 0, 32, seed=1, stack_size=1024)
    return np.transpose(hdr, (1, 0, 2))[::-1].astype(np.float64), st           # PNG row / column order


def _band(lum, frac=0.2):
    ys, xs = np.nonzero(lum > frac * lum.max())
    n = lum.shape[0]
    return ys.min() / n, (ys.max() + 1) / n, xs.min() / n, (xs.max() + 1) / n


def test_rainbow_lands_where_the_reference_image_has_it(rainbow_oracle):
    img, st = rainbow_oracle
    assert np.isfinite(img).all() and (img >= 0).all() and st["overflow"] == 0
    lum = img.sum(axis=2)
    ref = np.load(os.path.join(GOLD, "rainbow_blocks.npy")).astype(np.float64)
    y0, y1, x0, x1 = _band(lum)
    ry0, ry1, rx0, rx1 = _band(ref.sum(axis=2))
    print("spectrum band (rows, cols as fractions of the frame): ours %.2f-%.2f x %.2f-%.2f, image/rainbow.png %.2f-%.2f x %.2f-%.2f" % (y0, y1, x0, x1, ry0, ry1, rx0, rx1))
    # the same place in the frame: the boxes overlap in both directions and their centres are within a tenth of the frame
    assert abs(0.5 * (y0 + y1) - 0.5 * (ry0 + ry1)) < 0.10 and abs(0.5 * (x0 + x1) - 0.5 * (rx0 + rx1)) < 0.10
    assert min(y1, ry1) - max(y0, ry0) > 0.25 and min(x1, rx1) - max(x0, rx0) > 0.02
    # a tall thin band, as there: several times higher than wide
    assert (y1 - y0) > 3.0 * (x1 - x0) and (ry1 - ry0) > 3.0 * (rx1 - rx0)
    # and nothing else is lit: the reference image is black outside the band
    n = lum.shape[0]
    box = np.zeros_like(lum, bool)
    box[int((ry0 - 0.1) * n):int((ry1 + 0.1) * n), int((rx0 - 0.1) * n):int((rx1 + 0.1) * n)] = True
    inside = lum[box].sum() / lum.sum()
    print("energy inside the reference band's box (+- 0.1): %.4f" % inside)
    assert inside > 0.97


def test_rainbow_colour_order(rainbow_oracle):
    """Short wavelengths are refracted more: blue ends up on the left of the band, red on the right, green between -- in the reference image
    (its first spectrum) and here."""
    img, _ = rainbow_oracle
    lum = img.sum(axis=2)
    ys, xs = np.nonzero(lum > 0.2 * lum.max())
    prof = img[ys.min():ys.max() + 1].sum(axis=0)                      # [column][rgb]
    cols = np.arange(prof.shape[0])
    cr, cg, cb = [(cols * prof[:, k]).sum() / prof[:, k].sum() for k in range(3)]
    print("centroid columns of the band: blue %.2f < green %.2f < red %.2f (of %d)" % (cb, cg, cr, prof.shape[0]))
    assert cb + 0.5 < cg < cr - 0.5
    ref = np.load(os.path.join(GOLD, "rainbow_blocks.npy")).astype(np.float64)
    rl = ref.sum(axis=2)
    rys, rxs = np.nonzero(rl > 0.2 * rl.max())
    rp = ref[rys.min():rys.max() + 1, rxs.min():rxs.min() + 3].sum(axis=0)        # the first spectrum: three blocks
    assert np.argmax(rp[:, 2]) <= np.argmax(rp[:, 1]) <= np.argmax(rp[:, 0]) and np.argmax(rp[:, 2]) < np.argmax(rp[:, 0])


def test_bdpt_spec_is_deterministic_and_frames_accumulate(rainbow_oracle):
    W = H = 24
    ex = scenes.prism_rainbow(W, H, 4)
    ex.scene.setup_data_cpu(); ex.integrator.setup_data_cpu()
    ex.integrator.setup_tables(lambda res, xyz, d65: oa.spec_table_build(res, xyz, d65))
    ex.cam.scale = 10.0; ex.cam.set_target(0.0, 0.0, 0.0); ex.cam.update()
    o = oa.OracleScene(ex.scene, ex.cam); o.lbvh_build(); o.set_spectral(ex.integrator.tables())
    a, sa, _ = o.bdpt_spec_render(ex.cam, W, H, 0, 3, seed=5, stack_size=1024)
    b, sb, _ = o.bdpt_spec_render(ex.cam, W, H, 0, 3, seed=5, stack_size=1024)
    assert np.array_equal(a, b) and sa == sb
    # three frames in one call == frame after frame with the state carried over
    hdr = np.zeros((W, H, 3), np.float32); state = None
    for f in range(3):
        hdr, _, state = o.bdpt_spec_render(ex.cam, W, H, f, 1, seed=5, stack_size=1024, hdr=hdr, state=state)
    assert np.array_equal(hdr, a)
    c, _, _ = o.bdpt_spec_render(ex.cam, W, H, 0, 3, seed=6, stack_size=1024)
    assert not np.array_equal(a, c)
