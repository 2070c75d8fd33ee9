"""Structure pins of the HIP path against the reference's own gallery renders of example/single_model.py's sphere
(image/glass.png, metal.png, non-metal.png; block means in tests/golden/gallery_*_blocks.npy, generator make_golden.py).

What these constrain, and why they are correlations and not radiometric comparisons, is written in
tests/test_oracle_golden.py::test_oracle_against_the_reference_gallery_spheres and scenes.gallery_sphere: the gallery renders
predate the committed example (camera distance, light, yaw, tone curve).  Here the product path renders the three variants at
the gallery's own 512^2 with 256 spp -- smooth normals (a19), env lookup (a18), sphere light NEE + MIS (a14 sphere branch, a3,
a11 sphere, quirks B2 / B3), Disney metal / diffuse (a15) -- and a 2 048-pixel run of every film is compared with the oracle
bit for bit, so the pins hold for exactly the arithmetic the oracle restates."""
import os

import numpy as np
import pytest

import oracle_api as oa
from common import gallery_structure
from ti_raytrace_amd import scenes

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("variant,inside_min", [("glass", 0.35), ("metal", 0.88), ("non-metal", 0.90)])
def test_gallery_sphere_structure_at_full_size(gpu_ctx_ok, variant, inside_min):
    W = H = 512
    spp = 256
    ex = scenes.gallery_sphere(W, H, spp, variant=variant, device_id=0)
    ex.build_scene()
    ex.integrator.render_frames(spp)
    hdr = ex.integrator.hdr.to_numpy()
    ref = np.load(os.path.join(GOLD, "gallery_%s_blocks.npy" % variant.replace("-", "_")))
    inside, background = gallery_structure(hdr, ref)
    print("gallery %s 512^2 x %d: log-luminance correlation inside %.3f, background %.3f" % (variant, spp, inside, background))
    assert background > 0.93 and inside > inside_min
    # the same film, a run of 2 048 pixels through the middle of the sphere, from the oracle at 8 spp (seconds): bit-identical
    ex.scene.ctx.film_clear(); ex.cam.frame = 0
    ex.integrator.render_frames(8)
    got = ex.integrator.hdr.to_numpy().reshape(-1, 3)
    o = oa.OracleScene(ex.scene, ex.cam); o.lbvh_build(); o.process_normal(ex.scene.vertex_index_np)
    p0 = 256 * H + 200
    want, _ = o.render(W, H, 0, 8, seed=ex.integrator.seed, p_begin=p0, p_end=p0 + 2048)
    want = want.reshape(-1, 3)[p0:p0 + 2048]
    same = np.array_equal(got[p0:p0 + 2048], want, equal_nan=True)
    assert same, "film differs from the oracle on the sampled run (%d of 2048 pixels)" % int((got[p0:p0 + 2048] != want).any(axis=1).sum())
