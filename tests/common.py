"""Shared scene builders for the tests (host side only; no device work here)."""
import numpy as np

from ti_raytrace_amd import scenes, Example, PT_RGB
from ti_raytrace_amd import SceneData as SCD


def host_only(ex, scale=0.8):
    """Run the host half of build_scene (packing + camera) without touching a device."""
    ex.scene.setup_data_cpu()
    ex.frame_camera(scale)
    return ex


def tiny_scene(ntri, seed=7, W=32, H=32, spread=0.2, with_light=True, device_id=None):
    ex = scenes.synthetic(W, H, 4, ntri=ntri, scene_seed=seed, spread=spread, device_id=device_id)
    return ex


def duplicate_code_scene(W=32, H=32, device_id=None):
    """Many triangles with identical centroids -> long runs of equal Morton codes (the
    reference's special case in determineRange, accel/LBvh.py:240-251)."""
    ex = Example.example(W, H, 4, device_id)
    rng = np.random.RandomState(3)
    base = rng.uniform(-1, 1, size=(40, 3))
    tris = []
    for c in base:
        reps = rng.randint(1, 7)
        for _ in range(reps):
            off = rng.uniform(-0.05, 0.05, size=(3, 3))
            off -= off.mean(axis=0, keepdims=True)        # same centroid -> same code
            tris.append(c[None, :] + off)
    mat = SCD.Material()
    mat.type = SCD.MAT_DISNEY
    mat.setMetal(0.0); mat.setRough(0.5); mat.setColor([0.8, 0.8, 0.8, 1.0]); mat.alebdoTex = -1
    ex.scene.add_mesh(np.asarray(tris), mat)
    ex.add_sphere_light(pos=(0.0, 3.0, 0.0), radius=0.75, emission=50.0)
    ex.integrator = PT_RGB.PathTrace(W, H, ex.cam, ex.scene, 64)
    return ex


def spot_laser_scene(W, H, kinds=("spot", "laser"), device_id=None, integrator="pt", with_quad_light=True):
    """Cornell box plus the two shape emitters that have no surface (SceneData.SHPAE_SPOT / SHPAE_LASER; the reference uses them in
    example/prism_rainbow.py:39-50 and samples them in Scene.sample_li / sample_light, Scene.py:449-472, 491-516): a spot light under
    the ceiling shining down (full intensity within 0.3 rad, fading out to 0.6 rad) and a laser beam of radius 40 aimed at the floor."""
    from ti_raytrace_amd import BDPT_RGB
    ex = Example.example(W, H, 4, device_id)
    ex.scene.add_obj(scenes.asset("model", "cornell_box.obj"))
    if not with_quad_light:                       # the box's own emissive quad becomes a grey Disney surface
        for m in ex.scene.material_cpu:
            if m.type == SCD.MAT_LIGHT:
                m.type = SCD.MAT_DISNEY; m.setMetal(0.0); m.setRough(0.5); m.setColor([0.7, 0.7, 0.7, 1.0])
    for kind in kinds:
        sh = SCD.Shape()
        mat = SCD.Material(); mat.type = SCD.MAT_LIGHT
        if kind == "spot":
            sh.type = SCD.SHPAE_SPOT
            sh.pos = [278.0, 520.0, -280.0]
            sh.setXita(0.3, 0.6); sh.setScale(1.0); sh.setNormal([0.0, -1.0, 0.0])
            mat.setColor([3.0e6, 3.0e6, 2.0e6])
        else:
            sh.type = SCD.SHPAE_LASER
            sh.pos = [120.0, 400.0, -100.0]
            sh.setRadius(40.0); sh.setNormal([0.25, -0.9, -0.35])
            mat.setColor([40.0, 10.0, 10.0])
        ex.scene.add_shape(sh, mat)
    if integrator == "bdpt":
        ex.integrator = BDPT_RGB.BDPT(W, H, ex.cam, ex.scene, 64)
    else:
        ex.integrator = PT_RGB.PathTrace(W, H, ex.cam, ex.scene, 64)
    return ex


def cornell_glass_wall(W, H, device_id=None, integrator="pt"):
    """The Cornell box with its red wall turned into a sheet of glass (ior 1.5, extinction 500 scene units): Disney surfaces, the quad
    light AND Glass.sample / the delta flags / the extinction roulette in one small scene (tools/refkat/make_refkat.py renders it through
    the reference's own source text; tests/test_refkat.py and the GPU tests render it through the oracle and the device)."""
    from ti_raytrace_amd import BDPT_RGB
    ex = scenes.cornell_box(W, H, 4, device_id=device_id)
    for m in ex.scene.material_cpu:
        if m.type == SCD.MAT_DISNEY and tuple(np.round(m.color[:3], 3)) == (1.0, 0.0, 0.0):
            m.type = SCD.MAT_GLASS; m.setIor(1.5); m.setExtinciton(500.0)
            break
    else:
        raise AssertionError("no red material in cornell_box.obj")
    if integrator == "bdpt":
        ex.integrator = BDPT_RGB.BDPT(W, H, ex.cam, ex.scene, 64)
    return ex


def refkat_lbvh_scene(which, device_id=None):
    """The two scenes tools/refkat/make_refkat.py --lbvh-only builds through the reference's accel/LBvh.py (tests/golden/refkat_lbvh.npz)"""
    if which == "duplicates":
        ex = duplicate_code_scene(16, 16, device_id=device_id)
        ex.add_sphere_light(pos=(0.3, 0.2, -0.4), radius=0.2, emission=10.0)
    else:
        ex = tiny_scene(700, seed=11, W=16, H=16, spread=0.2, device_id=device_id)
    return ex


def rel_l2(a, b):
    return float(np.sqrt(((a.astype(np.float64) - b) ** 2).sum() / max((b.astype(np.float64) ** 2).sum(), 1e-30)))


# ---- structure pins against the reference's gallery renders (tests/golden/gallery_*_blocks.npy) -------------------------
def _srgb_to_linear(c):
    c = np.asarray(c, np.float64)
    return np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4)


def gallery_structure(hdr_field, ref_blocks):
    """Correlation of log block luminance between a linear film (field orientation [W,H,3]) and the 32x32 block means of a
    gallery PNG, inside the sphere (radius 7 of 8 blocks) and over the background (outside the sphere and the light disc).
    Log luminance because the gallery images went through an unrecorded tone curve: a monotone curve keeps the ordering and,
    to first order, the correlation."""
    img = np.nan_to_num(np.transpose(np.asarray(hdr_field, np.float64), (1, 0, 2))[::-1])
    s = img.shape[0] // 32
    ours = img.reshape(32, s, 32, s, 3).mean(axis=(1, 3))
    w = np.array([0.2126, 0.7152, 0.0722])
    lo = np.log(1e-3 + ours @ w)
    lr = np.log(1e-3 + _srgb_to_linear(ref_blocks) @ w)
    yy, xx = np.mgrid[0:32, 0:32] + 0.5
    r2 = (xx - 16) ** 2 + (yy - 16) ** 2
    inside = r2 < 7.0 ** 2
    outside = (r2 > 9.5 ** 2) & (((xx - 16) ** 2 + (yy - 0.5) ** 2) > 4.0 ** 2)

    def corr(a, b):
        a = a - a.mean(); b = b - b.mean()
        return float((a * b).sum() / np.sqrt((a * a).sum() * (b * b).sum()))
    return corr(lo[inside], lr[inside]), corr(lo[outside], lr[outside])
