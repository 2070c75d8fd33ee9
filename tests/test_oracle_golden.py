"""The oracle is pinned to the only two artefacts the reference's own hot path committed
(SURVEY.md 8c): nodelist.txt (exact LBVH of cornell_box.obj) and out.png (Cornell, PT_RGB,
512^2, 512 spp).  CPU only."""
import os

import numpy as np

import oracle_api as oa
from common import host_only, rel_l2
from ti_raytrace_amd import scenes, LBvh

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_nodelist_reproduced_line_by_line():
    ex = host_only(scenes.cornell_box(64, 64, 4))
    o = oa.OracleScene(ex.scene, ex.cam)
    assert o.lbvh_build() == ex.scene.primitive_count - 1
    _, _, compact = o.lbvh_get()
    ours = LBvh.format_nodelist(compact).strip().split("\n")
    ref = open(os.path.join(GOLD, "nodelist.txt")).read().strip().split("\n")
    assert len(ref) == 71 and len(ours) == 71
    assert ours == ref


def test_lbvh_structure_is_a_valid_tree():
    ex = host_only(scenes.cornell_box(64, 64, 4))
    o = oa.OracleScene(ex.scene, ex.cam)
    o.lbvh_build()
    morton, bvh, compact = o.lbvh_get()
    n = ex.scene.primitive_count
    assert (np.diff(morton[:, 0]) >= 0).all()                       # sorted
    assert sorted(morton[:, 1].tolist()) == list(range(n))          # a permutation
    leaves = compact[(compact[:, 0].astype(np.int32) & 1) == 1]
    assert sorted(leaves[:, 1].astype(np.int32).tolist()) == list(range(n))
    # parent box = union of children (compact layout: left = i+1, right = slot 1)
    for i in range(compact.shape[0]):
        if (int(compact[i, 0]) & 1) == 0:
            l, r = i + 1, int(compact[i, 1])
            assert np.array_equal(compact[i, 2:5], np.minimum(compact[l, 2:5], compact[r, 2:5]))
            assert np.array_equal(compact[i, 5:8], np.maximum(compact[l, 5:8], compact[r, 5:8]))


def test_cornell_image_matches_out_png_statistically():
    """128^2 x 48 spp vs the reference's 512^2 x 512 spp out.png: mean colour within 1.5 %,
    16x16-block means (of the 128^2 image = 64x64 blocks of out.png) within 6 % rel-L2.
    (SURVEY.md probe: 0.3 % / 2.6 % at 64 spp.)"""
    W = H = 128
    ex = host_only(scenes.cornell_box(W, H, 64))
    o = oa.OracleScene(ex.scene, ex.cam)
    o.lbvh_build()
    hdr, st = o.render(W, H, 0, 48, seed=1)
    assert st["overflow"] == 0
    rgb = o.tone_map(0.5, hdr)
    img = np.transpose(rgb, (1, 0, 2))[::-1]                          # ti.imwrite orientation
    ref = np.load(os.path.join(GOLD, "out_png_blocks.npy"))           # [32,32,3] blocks of 16 px (of 512)
    ours_blocks = img.reshape(8, 16, 8, 16, 3).mean(axis=(1, 3))      # 8x8 blocks of 16 px (of 128)
    ref_blocks = ref.reshape(8, 4, 8, 4, 3).mean(axis=(1, 3))
    mean_err = np.abs(img.reshape(-1, 3).mean(0) - ref.reshape(-1, 3).mean(0)) / ref.reshape(-1, 3).mean(0)
    assert (mean_err < 0.015).all(), mean_err
    assert rel_l2(ours_blocks, ref_blocks) < 0.06


def test_libm_oracle_agrees_with_shared_math_oracle():
    """Swapping tirt_math.h for libm changes individual paths (1-ulp differences) but not the
    image statistics -> the shared math does not bias the oracle."""
    W = H = 48
    ex = host_only(scenes.cornell_box(W, H, 16))
    a = oa.OracleScene(ex.scene, ex.cam)
    b = oa.OracleScene(ex.scene, ex.cam, libm=True)
    a.lbvh_build(); b.lbvh_build()
    assert np.array_equal(a.lbvh_get()[2], b.lbvh_get()[2])
    ha, _ = a.render(W, H, 0, 16)
    hb, _ = b.render(W, H, 0, 16)
    assert abs(ha.mean() - hb.mean()) / ha.mean() < 0.01
    assert (ha == hb).all(axis=2).mean() > 0.5        # most pixels are even bit-identical


def test_render_is_deterministic_and_tile_independent():
    W = H = 32
    ex = host_only(scenes.cornell_box(W, H, 4))
    o = oa.OracleScene(ex.scene, ex.cam)
    o.lbvh_build()
    full, _ = o.render(W, H, 0, 3, nthreads=3)
    again, _ = o.render(W, H, 0, 3, nthreads=1)
    assert np.array_equal(full, again)
    acc = np.zeros_like(full)
    for r in range(4):
        part, _ = o.render(W, H, 0, 3, tile_rank=r, tile_count=4, tile_size=64)
        acc += part
    assert np.array_equal(acc, full)
    # frames accumulate as a running mean: 2 + 1 frames == 3 frames
    h2, _ = o.render(W, H, 0, 2)
    h3, _ = o.render(W, H, 2, 1, hdr=h2.copy())
    assert np.array_equal(h3, full)


def test_bdpt_oracle_is_deterministic_and_stateful():
    """BDPT restatement (BASELINE config 5): finite, reproducible, and rendering frames 0..2 in one
    call equals 2 + 1 frames with the persistent per-pixel vertex state handed over."""
    W = H = 20
    ex = host_only(scenes.cornell_box(W, H, 4))
    o = oa.OracleScene(ex.scene, ex.cam)
    o.lbvh_build()
    a, st, _ = o.bdpt_render(ex.cam, W, H, 0, 3)
    b, _, _ = o.bdpt_render(ex.cam, W, H, 0, 3)
    assert np.array_equal(a, b) and np.isfinite(a).all() and a.mean() > 0.05
    assert st["rays_closest"] > 0 and st["rays_shadow"] > 0
    h2, _, state = o.bdpt_render(ex.cam, W, H, 0, 2)
    h3, _, _ = o.bdpt_render(ex.cam, W, H, 2, 1, hdr=h2.copy(), state=state)
    assert np.array_equal(h3, a)


def test_oracle_bdpt_against_the_reference_gallery_render(oracle_lib):
    """The oracle's BDPT_RGB restatement on config 5's scene (veach_bdpt.py: bdpt.obj, smooth normals, camera at
    0.5 x |diagonal|, exposure 0.5) at 128^2 x 8 spp against the reference's own render, image/veach-bdpt512.png
    (block means, tests/golden/veach_bdpt512_blocks.npy).  Statistical pin (seed and sample count of the reference run
    are not recorded; measured 4.6 % / 7.9 %): pins Scene.sample_light, the connection strategies, the MIS weights and
    the light-tracing splats through Camera.get_image_point -- a wrong weight or a mirrored splat moves the mean by
    tens of per cent (other image orientations: 60-70 % block error)."""
    import oracle_api as oa
    from common import rel_l2
    from ti_raytrace_amd import scenes
    W = H = 128
    ex = scenes.veach_bdpt(W, H, 8)
    ex.scene.setup_data_cpu(); ex.frame_camera(0.5)
    o = oa.OracleScene(ex.scene, ex.cam); o.lbvh_build(); o.process_normal(ex.scene.vertex_index_np)
    hdr, st, _ = o.bdpt_render(ex.cam, W, H, 0, 8, seed=1)
    rgb = o.tone_map(0.5, hdr)
    img = np.transpose(rgb, (1, 0, 2))[::-1]
    ours = img.reshape(32, 4, 32, 4, 3).mean(axis=(1, 3))
    ref = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "veach_bdpt512_blocks.npy"))
    mean_err = np.abs(ours.reshape(-1, 3).mean(0) - ref.reshape(-1, 3).mean(0)) / ref.reshape(-1, 3).mean(0)
    r = rel_l2(ours, ref)
    print("oracle BDPT 128^2 x8 vs veach-bdpt512.png: mean err %s, block rel-L2 %.4f" % (np.round(mean_err, 4), r))
    assert (mean_err < 0.08).all() and r < 0.12


def test_oracle_threading_does_not_change_the_film(oracle_lib):
    """The oracle hands out 64-pixel chunks through an atomic counter (bench.py's cpu_baseline leg runs it on every host core):
    pixels are independent, so 1 thread and 7 threads must produce the same film and the same counters."""
    import oracle_api as oa
    from ti_raytrace_amd import scenes
    W = H = 40
    ex = host_only(scenes.cornell_box(W, H, 4))
    o = oa.OracleScene(ex.scene, ex.cam); o.lbvh_build()
    a, sa = o.render(W, H, 0, 2, seed=3, nthreads=1)
    b, sb = o.render(W, H, 0, 2, seed=3, nthreads=7)
    assert np.array_equal(a, b) and sa == sb
    # a ragged pixel range with tiles
    c1, s1 = o.render(W, H, 0, 1, seed=3, p_begin=13, p_end=1207, tile_rank=1, tile_count=3, tile_size=50, nthreads=1)
    c2, s2 = o.render(W, H, 0, 1, seed=3, p_begin=13, p_end=1207, tile_rank=1, tile_count=3, tile_size=50, nthreads=5)
    assert np.array_equal(c1, c2) and s1 == s2 and s1["paths"] > 0


def test_oracle_against_the_reference_gallery_spheres(oracle_lib):
    """Structure pins on image/glass.png, metal.png, non-metal.png (example/single_model.py's sphere.obj and its two
    commented material variants, :27-31).  The gallery renders predate the committed example (camera distance, light,
    yaw and tone curve differ -- scenes.gallery_sphere states what was measured on the images), so what is compared is the
    correlation of log block luminance, 128^2 x 64 spp against the 512^2 gallery image:
      background (all three): the lat-long env lookup of PT_RGB.py:127-132 + Texture.texture2D through Camera's view matrix --
        0.95 here, 0.24 from the opposite yaw, < 0.5 for any other atan2 argument order;
      metal.png inside the sphere: Disney metal 1 / rough 0 = a mirror ball (brdf/Disney.py sample + evaluate_pdf) on smooth
        normals (process_normal, Scene.py:754-798): 0.91;
      non-metal.png inside: diffuse Disney under the env + the analytic sphere light (sample_li's sphere branch, quirks B2 / B3,
        MIS): 0.93;
      glass.png inside: 0.49 only, and no ior between 1.15 and 1.5 does better -- the gallery's glass is not the committed
        Glass.sample + extinction roulette (B13); reported, asserted loosely.  Glass stays pinned by this oracle only."""
    from common import gallery_structure
    W = H = 128
    got = {}
    for variant in ("glass", "metal", "non-metal"):
        ex = scenes.gallery_sphere(W, H, 64, variant=variant)
        ex.scene.setup_data_cpu(); ex.frame_camera()
        o = oa.OracleScene(ex.scene, ex.cam); o.lbvh_build(); o.process_normal(ex.scene.vertex_index_np)
        hdr, st = o.render(W, H, 0, 64, seed=1)
        ref = np.load(os.path.join(GOLD, "gallery_%s_blocks.npy" % variant.replace("-", "_")))
        got[variant] = gallery_structure(hdr, ref)
    print("gallery structure pins (inside, background):", {k: (round(a, 3), round(b, 3)) for k, (a, b) in got.items()})
    for variant in got:
        assert got[variant][1] > 0.92, (variant, got[variant])
    assert got["metal"][0] > 0.85 and got["non-metal"][0] > 0.88, got
    assert got["glass"][0] > 0.35, got
    # negative control: the committed example's yaw 0 looks at another part of the env
    ex = scenes.gallery_sphere(W, H, 16, variant="non-metal", yaw=0.0)
    ex.scene.setup_data_cpu(); ex.frame_camera()
    o = oa.OracleScene(ex.scene, ex.cam); o.lbvh_build(); o.process_normal(ex.scene.vertex_index_np)
    hdr, _ = o.render(W, H, 0, 16, seed=1)
    assert gallery_structure(hdr, np.load(os.path.join(GOLD, "gallery_non_metal_blocks.npy")))[1] < 0.5


def test_libm_oracle_has_the_same_lbvh_and_hit_records_on_the_teapot():
    """The oracle built on libm (-DORACLE_LIBM) instead of ti_raytrace_amd/csrc/tirt_math.h: everything that does not go through a
    transcendental function must come out bit-identical -- the whole LBVH (Morton codes, sort, topology, boxes, flatten), the
    smooth normals' finite entries and the 13-float closest-hit records of the Teapot's camera rays (t, position, both normals,
    uv; sqrt and division are correctly rounded on both sides).  Only sin/cos/exp/log/pow/atan2/acos differ (<= 1 ulp), i.e. the
    BSDF sampling and process_normal's acos-weighted angles."""
    W = H = 64
    ex = host_only(scenes.single_model(W, H, 4))
    a = oa.OracleScene(ex.scene, ex.cam)
    b = oa.OracleScene(ex.scene, ex.cam, libm=True)
    assert a.L.orc_uses_libm() == 0 and b.L.orc_uses_libm() == 1
    assert a.lbvh_build() == b.lbvh_build()
    for x, y in zip(a.lbvh_get(), b.lbvh_get()):
        assert np.array_equal(x, y)
    rays = oa.camera_rays(ex.cam, W, H)
    ha, pa, _ = a.closest_hit(rays)
    hb, pb, _ = b.closest_hit(rays)
    assert np.array_equal(pa, pb) and np.array_equal(ha, hb, equal_nan=True)
    assert (pa >= 0).mean() > 0.2
    # smooth normals go through acos: compare loosely, and count how many entries are even bit-identical
    a.process_normal(ex.scene.vertex_index_np); b.process_normal(ex.scene.vertex_index_np)
    va, vb = a.vertex(), b.vertex()
    fin = np.isfinite(va).all(axis=1) & np.isfinite(vb).all(axis=1)
    assert fin.mean() > 0.99 and np.abs(va[fin] - vb[fin]).max() < 1e-4


def test_oracle_spot_and_laser_emitters(oracle_lib):
    """The oracle's restatement of the shape emitters without a surface (Scene.py:344-349, 413-418, 449-472, 491-516) behaves as
    the reference's formulas say: a spot light under the ceiling lights the floor inside its cone at full strength up to x1, fading
    to nothing at x2; a laser lights a disc of its radius around its axis and nothing else; neither is ever hit by a ray."""
    from common import spot_laser_scene
    W = H = 48
    for kind in ("spot", "laser"):
        ex = spot_laser_scene(W, H, (kind,), with_quad_light=False)
        ex.scene.setup_data_cpu(); ex.frame_camera(0.8)
        o = oa.OracleScene(ex.scene, ex.cam); o.lbvh_build()
        # rays aimed at the emitter itself fly through it (intersect_prim: INF for these shapes)
        n = ex.scene.primitive_count
        pos = ex.scene.shape_np[-1, 1:4].astype(np.float64)
        org = np.array([278.0, 273.0, 700.0])
        d = pos - org; d /= np.linalg.norm(d)
        _, prim, _ = o.closest_hit(np.concatenate([org, d])[None].astype(np.float32))
        assert prim[0] != n - 1
        hdr, st = o.render(W, H, 0, 8, seed=1)
        assert np.isfinite(hdr).all() and st["rays_shadow"] > 0
        lit = hdr.sum(axis=2) > 0
        assert 0.02 < lit.mean() < 0.98, lit.mean()
    # the spot's fall-off, point by point, against the formula: directly below at full strength, beyond x2 nothing
    ex = spot_laser_scene(W, H, ("spot",), with_quad_light=False)
    ex.scene.setup_data_cpu(); ex.frame_camera(0.8)
    sh = ex.scene.shape_np[-1]
    assert int(sh[0]) == 3 and np.allclose(sh[4:7], [0.3, 0.6, 1.0]) and np.allclose(sh[7:10], [0, -1, 0])
