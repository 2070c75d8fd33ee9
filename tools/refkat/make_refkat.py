"""Golden vectors computed by the REFERENCE'S OWN SOURCE TEXT -- build container only (needs /root/reference).

    python tools/refkat/make_refkat.py            # writes tests/golden/refkat.npz (+ refkat_render.npz with --render)

The reference cannot run here: it is Python + Taichi DSL and Taichi is not installable.  But its `@ti.func`s are
plain Python once `taichi` / `taichi_glsl` resolve to the stand-in under tools/refkat/standin (identity decorators, an fp32 vector
class, numpy-backed fields, ti.random() from a queue).  This script imports /root/reference/{UtilsFunc,Camera,Scene}.py,
brdf/{Disney,Glass}.py and integrator/PT_RGB.py THROUGH that stand-in and calls their functions on seeded inputs; inputs and outputs
go to tests/golden/ as data.  tests/test_refkat.py (CPU, the oracle) and tests/test_gpu_kat.py (the HIP device functions through the
C-ABI) are compared with them.

What this pins and what it does not: every formula, constant, branch and operand order comes from the reference's text, executed
with one fp32 rounding per source-level operation -- a misreading of the reference in oracle.c AND tirt_device.h (which share
their author) shows up here.  It is a transcription check, not a run of the reference: Taichi's code generator (fast-math flags,
its own transcendental functions, its RNG) is not reproduced; sin / cos / pow / exp / atan2 / acos are evaluated in float64 and
rounded once, so values agree with the oracle to ~1e-6 relative, not bit for bit.

No reference source is copied: the reference is imported where it lies; only numbers are written."""
import argparse
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("TIRT_REFERENCE", "/root/reference")
sys.path[:0] = [os.path.join(HERE, "standin"), REF, os.path.join(REF, "brdf"), os.path.join(REF, "accel"),
                os.path.join(REF, "texture"), os.path.join(REF, "integrator")]
for _name in ("pywavefront", "trimesh", "cv2"):            # host-side loaders of the reference: not on the paths exercised here
    sys.modules.setdefault(_name, types.ModuleType(_name))

import taichi as ti                    # noqa: E402  (the stand-in)
import UtilsFunc as UF                 # noqa: E402
import Disney                          # noqa: E402
import Glass                           # noqa: E402
import Camera                          # noqa: E402

# `pow(a, b)` inside a ti.func is Taichi's f32 pow; the module-level name shadows the Python builtin (which would go through float64 too,
# but via numpy's float32 power).  Python-float module constants become fp32 so that `1.0 / UF.M_PIf` is an fp32 division as in Taichi.
for _m in (UF, Disney, Glass):
    _m.pow = ti.pow_
UF.M_PIf = np.float32(UF.M_PIf)
UF.INF_VALUE = np.float32(UF.INF_VALUE)
UF.EPS = np.float32(UF.EPS)

V = ti.Vector


def vec(a):
    return V([np.float32(x) for x in a])


def unit(r, n):
    v = r.normal(size=(n, 3))
    return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)


def materials(r, n):
    m = np.zeros((n, 10), np.float32)
    m[:, 2:5] = r.uniform(0, 1, size=(n, 3))
    m[:, 5] = r.choice([0.0, 0.3, 1.0], n)                       # metallic
    m[:, 6] = r.choice([0.0, 0.001, 0.2, 0.5, 1.0], n)           # roughness
    return m


def kat_functions(out):
    r = np.random.RandomState(20260929)
    n = 1500
    # ---- brdf/Disney.py ---------------------------------------------------------------------------
    m = materials(r, n)
    N, Vv, L = unit(r, n), unit(r, n), unit(r, n)
    up = slice(0, 2 * n // 3)            # two thirds with V and L in N's hemisphere (the branch that computes something)
    Vv[up] = np.where((np.sum(Vv[up] * N[up], 1) < 0)[:, None], -Vv[up], Vv[up])
    L[up] = np.where((np.sum(L[up] * N[up], 1) < 0)[:, None], -L[up], L[up])
    ev = np.zeros((n, 2), np.float32); pd = np.zeros(n, np.float32)
    for i in range(n):
        c, p = Disney.evaluate_pdf(vec(N[i]), vec(Vv[i]), vec(L[i]), m, i)
        ev[i] = (c, p)
        pd[i] = Disney.pdf(vec(N[i]), vec(Vv[i]), vec(L[i]), m, i)
    out.update(disney_in=np.concatenate([m, N, Vv, L], 1), disney_evaluate_pdf=ev, disney_pdf=pd)

    d = unit(r, n)
    rnd = r.uniform(0, 1, size=(n, 3)).astype(np.float32)
    sm = np.zeros((n, 4), np.float32)
    for i in range(n):
        ti.set_random([rnd[i, 0], rnd[i, 1], rnd[i, 2]])
        nd, io = Disney.sample(vec(d[i]), vec(N[i]), m, i)
        sm[i] = (nd.x, nd.y, nd.z, io)
    out.update(disney_sample_in=np.concatenate([m, d, N, rnd], 1), disney_sample=sm)

    # ---- brdf/Glass.py ----------------------------------------------------------------------------
    mg = m.copy(); mg[:, 5] = r.choice([1.0, 1.3, 1.5, 2.4], n); mg[:, 6] = r.choice([0.5, 2.0, 10.0], n)
    gs = np.zeros((n, 4), np.float32)
    for i in range(n):
        ti.set_random([rnd[i, 0]])
        nd, fb = Glass.sample(vec(d[i]), vec(N[i]), np.float32(1.0), mg, i)
        gs[i] = (nd.x, nd.y, nd.z, fb)
    lam = r.uniform(360.0, 830.0, n).astype(np.float32)
    gl = np.zeros((n, 4), np.float32)
    for i in range(n):
        ti.set_random([rnd[i, 0]])
        nd, fb = Glass.sample_lambda(vec(d[i]), vec(N[i]), np.float32(1.0), mg, i, lam[i])
        gl[i] = (nd.x, nd.y, nd.z, fb)
    out.update(glass_sample_in=np.concatenate([mg, d, N, rnd[:, :1]], 1), glass_sample=gs, glass_lambda=lam, glass_sample_lambda=gl)

    # ---- UtilsFunc.py -----------------------------------------------------------------------------
    p = (r.normal(size=(n, 3)) * r.choice([1e-4, 1e-2, 1.0, 500.0], (n, 1))).astype(np.float32)
    p[:20] = 0.0; p[20:40, 0] = np.float32(1.0 / 256.0)
    nn = (unit(r, n) * r.choice([-1.0, 1.0, 0.0], (n, 1))).astype(np.float32)
    off = np.zeros((n, 3), np.float32)
    for i in range(n):
        off[i] = UF.offset_ray(vec(p[i]), vec(nn[i])).to_numpy()
    out.update(offset_ray_in=np.concatenate([p, nn], 1), offset_ray=off)

    o = r.uniform(-2, 2, size=(n, 3)).astype(np.float32)
    dd = unit(r, n)
    dd[:200, r.randint(0, 3)] = 0.0                                   # axis-parallel components (the origin-in-slab branch)
    dd[200:300] *= np.float32(1e-7)                                   # whole direction below the 1e-6 threshold
    mn = r.uniform(-1, 0.5, size=(n, 3)).astype(np.float32)
    mx = mn + r.uniform(0.0, 1.5, size=(n, 3)).astype(np.float32)
    aim = mn[600:] + (mx[600:] - mn[600:]) * r.uniform(-0.1, 1.1, size=(n - 600, 3)).astype(np.float32)      # aimed at (or just past) the box
    dd[600:] = aim - o[600:]; dd[600:] /= np.linalg.norm(dd[600:], axis=1, keepdims=True)
    o[300:400] = (mn[300:400] + mx[300:400]) / 2                      # origins inside
    o[400:450] = mn[400:450]                                          # origins on a corner
    sl = np.zeros(n, np.int32)
    for i in range(n):
        sl[i] = UF.slabs(vec(o[i]), vec(dd[i]), vec(mn[i]), vec(mx[i]))
    out.update(slabs_in=np.concatenate([o, dd, mn, mx], 1), slabs=sl)

    q = r.uniform(-0.1, 1.1, size=(n, 3)).astype(np.float32)
    q[:8] = [[0, 0, 0], [1, 1, 1], [0.5, 0.5, 0.5], [1023 / 1024, 0, 0], [0, 1023.5 / 1024, 0], [0, 0, 0.999999], [1e-9, 1e-9, 1e-9], [2, -1, 0.25]]
    import io, contextlib
    mc = np.zeros(n, np.int32)
    with contextlib.redirect_stdout(io.StringIO()):                    # morton3D prints when the code is 0
        for i in range(n):
            mc[i] = UF.morton3D(q[i, 0], q[i, 1], q[i, 2])
    out.update(morton3d_in=q, morton3d=mc)
    a = r.randint(0, 1 << 30, n).astype(np.int64); b = r.randint(0, 1 << 30, n).astype(np.int64)
    b[:100] = a[:100]; b[100:200] = a[100:200] ^ (1 << r.randint(0, 30, 100))
    out.update(cub_in=np.stack([a, b], 1).astype(np.int32),
               common_upper_bits=np.array([UF.common_upper_bits(int(x), int(y)) for x, y in zip(a, b)], np.int32))

    u = r.uniform(0, 1, size=(n, 2)).astype(np.float32)
    u[:4] = [[0, 0], [0.5, 0.5], [1, 0], [0.25, 0.75]]
    csh = np.zeros((n, 3), np.float32); disk = np.zeros((n, 2), np.float32); ph = np.zeros(n, np.float32)
    for i in range(n):
        csh[i] = UF.CosineSampleHemisphere(u[i, 0], u[i, 1]).to_numpy()
        disk[i] = UF.mapToDisk(u[i, 0], u[i, 1])
        ph[i] = UF.powerHeuristic(u[i, 0] * np.float32(7.0), u[i, 1] * np.float32(3.0))
    it = np.zeros((n, 3), np.float32)
    for i in range(n):
        it[i] = UF.inverse_transform(vec(d[i]), vec(N[i] * np.float32(1.0 + 0.5 * (i % 3)))).to_numpy()
    col = r.uniform(0, 1, size=(n, 3)).astype(np.float32); col[:3] = [[0, 0, 0], [0.04045, 0.04, 0.05], [1, 1, 1]]
    s2l = np.zeros((n, 3), np.float32); l2s = np.zeros((n, 3), np.float32); aces = np.zeros((n, 3), np.float32)
    for i in range(n):
        s2l[i] = UF.srgb_to_lrgb(vec(col[i])).to_numpy()
        l2s[i] = UF.lrgb_to_srgb(vec(col[i] * np.float32(1.5))).to_numpy()
        aces[i] = UF.tone_ACES(vec(col[i] * np.float32(4.0))).to_numpy()
    eta = r.choice([1.5, 1 / 1.5, 1.3, 1 / 2.4], n).astype(np.float32)
    rf = np.zeros((n, 4), np.float32); sch = np.zeros(n, np.float32); g2 = np.zeros(n, np.float32); sg = np.zeros(n, np.float32); sf = np.zeros(n, np.float32)
    for i in range(n):
        Rv, suc = UF.refract(vec(d[i]), vec(N[i]), eta[i])
        rf[i] = (Rv.x, Rv.y, Rv.z, suc)
        sch[i] = UF.schlick(u[i, 0], np.float32(1.0) / eta[i])
        g2[i] = UF.GTR2(u[i, 0], max(np.float32(0.001), u[i, 1]))
        sg[i] = UF.smithG_GGX(u[i, 0], u[i, 1])
        sf[i] = UF.SchlickFresnel(u[i, 0] * np.float32(1.2) - np.float32(0.1))
    out.update(u2=u, CosineSampleHemisphere=csh, mapToDisk=disk, powerHeuristic=ph, inverse_transform_in=np.concatenate([d, N], 1),
               inverse_transform=it, colour_in=col, srgb_to_lrgb=s2l, lrgb_to_srgb=l2s, tone_ACES=aces,
               refract_in=np.concatenate([d, N, eta[:, None]], 1), refract=rf, schlick=sch, GTR2=g2, smithG_GGX=sg, SchlickFresnel=sf)

    # ---- Camera.py:122-142 ------------------------------------------------------------------------
    cam = Camera.Camera(64, 48, 16)
    cam.set_target(0.1, 0.2, -0.3)
    cam.set_view_point(0.6, 0.25, 0.0, 3.5)
    uv = np.stack([r.randint(0, 64, 400), r.randint(0, 48, 400)], 1).astype(np.int32)
    jit = r.uniform(0, 1, size=(400, 2)).astype(np.float32)
    rd0 = np.zeros((400, 3), np.float32); rd1 = np.zeros((400, 3), np.float32)
    for i in range(400):
        cam.frame_gpu[0] = 0
        rd0[i] = cam.get_ray_direction(int(uv[i, 0]), int(uv[i, 1])).to_numpy()
        cam.frame_gpu[0] = 3
        ti.set_random([jit[i, 0], jit[i, 1]])
        rd1[i] = cam.get_ray_direction(int(uv[i, 0]), int(uv[i, 1])).to_numpy()
    out.update(camera_view_inv=cam.view_inv.to_numpy()[0].astype(np.float32), camera_eye=cam.eye.to_numpy()[0].astype(np.float32),
               camera_fx_fy_cx_cy=np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32), camera_uv=uv, camera_jitter=jit,
               camera_dir_frame0=rd0, camera_dir_jittered=rd1)


# ---- the reference's whole integrator, executed from its source text ----------------------------------------------------------------
# integrator/PT_RGB.py:49-136 (`render`) with everything it calls -- Camera.get_ray_direction, Scene.closet_hit / closet_hit_shadow /
# intersect_prim / intersect_tri / sample_li / get_prim_random_point_normal / get_prim_area, Disney.*, Glass.*, UF.*, Texture.texture2D --
# run as plain Python over a small film.  Two things are supplied from this repo, because the reference has no reproducible
# equivalent: (1) ti.random() returns the counter-based tm_rand(seed, pixel, frame, dim) with the dimension decided by the CALL SITE
# (which reference function drew it, the how-many-th draw of that activation, and the `depth` of the render loop: SURVEY A.6 / tirt_math.h
# TM_DIM_* -- so the mapping "which random number feeds which decision" is pinned too); (2) sin / cos / exp / pow / atan2 / acos / sqrt
# are the shared polynomial kernels (through the oracle library's orc_kat_math), so that a film that differs does so because of the
# reference's formulas, not because of a different math library.  The scene arrays (vertex / primitive / material / shape / light rows,
# the compact LBVH nodes -- pinned by nodelist.txt --, camera matrices) are put straight into the reference classes' fields.
def render_reference_text(out, W, H, frames, seed, scene_name):
    import ctypes as C
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import oracle_api as oa
    from common import host_only
    from ti_raytrace_amd import scenes
    import Scene as RScene, PT_RGB as RPT, SceneData as RSCD       # the reference's modules (through the stand-in)
    for _m in (RScene, RPT):
        _m.pow = ti.pow_

    L = oa.load()

    def m1(fn):
        def f(x):
            xi = np.array([x], np.float32); o = np.zeros(1, np.float32)
            L.orc_kat_math(fn, xi, np.zeros(1, np.float32), o, 1)
            return o[0]
        return f

    def m2(fn):
        def f(x, y):
            o = np.zeros(1, np.float32)
            L.orc_kat_math(fn, np.array([x], np.float32), np.array([y], np.float32), o, 1)
            return o[0]
        return f
    ti.set_math({"sin": m1(0), "cos": m1(1), "exp": m1(2), "log": m1(3), "pow": m2(4), "atan2": m2(5), "acos": m1(6),
                 "tan": lambda x: np.float32(m1(0)(x) / m1(1)(x))})                  # tirt_math.h: tm_tan = tm_sin / tm_cos

    from common import cornell_glass_wall
    if scene_name == "cornell":
        ex = scenes.cornell_box(W, H, 4, device_id=None)
    elif scene_name == "cornell_glass":
        ex = cornell_glass_wall(W, H)
    elif scene_name == "spot_laser":                              # the emitters without a surface: Scene.sample_li's spot / laser branches (Scene.py:491-516)
        from common import spot_laser_scene
        ex = spot_laser_scene(W, H)
    else:                                                          # glass + disney + sphere light + env map + smooth normals
        ex = scenes.single_model(W, H, 4, model="sphere.obj", device_id=None)
    host_only(ex, 0.8)
    sc = ex.scene
    orc = oa.OracleScene(sc, ex.cam)
    orc.lbvh_build()
    if scene_name == "sphere":
        vi = np.ascontiguousarray(sc.vertex_index_np, np.int32)
        orc.L.orc_process_normal(orc.h, vi)                        # smooth normals (Scene.process_normal), pinned elsewhere
        sc.vertex_np[...] = orc.vertex()
    _, _, compact = orc.lbvh_get()

    # the reference's objects, fields filled directly
    rcam = Camera.Camera(W, H, 4)
    rcam.view.from_numpy(ex.cam.view_np); rcam.view_inv.from_numpy(ex.cam.view_inv_np); rcam.eye.from_numpy(ex.cam.eye_np)
    assert (rcam.fx, rcam.fy, rcam.cx, rcam.cy) == (ex.cam.fx, ex.cam.fy, ex.cam.cx, ex.cam.cy)
    rs = RScene.Scene()
    rs.material.from_numpy(sc.material_np); rs.vertex.from_numpy(sc.vertex_np); rs.primitive.from_numpy(sc.primitive_np)
    rs.shape.from_numpy(sc.shape_np); rs.light.from_numpy(sc.light_np.astype(np.int32))
    rs.light_count = sc.light_count; rs.primitive_count = sc.primitive_count; rs.env_power = np.float32(sc.env_power)
    rs.env.np_img = sc.env.np_img; rs.env.wid, rs.env.hgt = sc.env.np_img.shape; rs.env.buf.from_numpy(sc.env.np_img)

    class _B:
        pass
    rs.bvh = _B(); rs.bvh.compact_node = ti.Vector.field(RSCD.CPNOD_VEC_SIZE, dtype=ti.f32); rs.bvh.compact_node.from_numpy(compact)
    pt = RPT.PathTrace(W, H, rcam, rs, 64)
    pt.setup_data_cpu()

    # ti.random(): counter-based, dimension from the call site
    draws = {}
    SLOTS = {("Camera", "get_ray_direction"): [("abs", 0), ("abs", 1)],
             ("Glass", "sample"): [("b", 0)], ("Scene", "get_random_light_prim_index"): [("b", 0)],
             ("Scene", "get_prim_random_point_normal"): [("b", 1), ("b", 2)],
             ("Disney", "sample"): [("b", 3), ("b", 4), ("b", 5)], ("PT_RGB", "render"): [("b", 6)]}
    used = set()

    def rnd():
        f = sys._getframe(2)                                       # rnd <- ti.random <- the reference function that draws
        key = (f.f_globals["__name__"], f.f_code.co_name)
        if f.f_code not in draws:                                  # the k-th ti.random() of that function, by source line
            import inspect
            src, first = inspect.getsourcelines(f.f_code)
            draws[f.f_code] = [first + n for n, line in enumerate(src) if "ti.random()" in line and not line.lstrip().startswith("#")]
        k = draws[f.f_code].index(f.f_lineno)
        kind, slot = SLOTS[key][k]
        g = f
        while g.f_code.co_name != "render":
            g = g.f_back
        i, j = g.f_locals["i"], g.f_locals["j"]
        depth = g.f_locals.get("depth", 0)
        dim = slot if kind == "abs" else 2 + 8 * int(depth) + slot
        used.add((key, k))
        return L.orc_kat_rand(seed, int(i) * H + int(j), int(rcam.frame_gpu[0]), dim)
    ti.set_random(rnd)

    import io, contextlib
    for fr in range(frames):
        rcam.frame_gpu[0] = fr
        with contextlib.redirect_stdout(io.StringIO()):
            pt.render()
    film = pt.hdr.to_numpy().astype(np.float32)
    want, _ = orc.render(W, H, 0, frames, seed=seed)
    d = np.abs(film - want); ident = int((film.view(np.uint32) == want.view(np.uint32)).all(axis=2).sum())
    rel = float(np.sqrt(((film.astype(np.float64) - want) ** 2).sum() / max((want.astype(np.float64) ** 2).sum(), 1e-30)))
    print("%s %dx%d x %d frames: reference text vs oracle: rel-L2 %.3e, bit-identical pixels %d / %d, max abs %.3e, draw sites used %d"
          % (scene_name, W, H, frames, rel, ident, W * H, float(np.nanmax(d)), len(used)))
    out["render_%s_film" % scene_name] = film
    out["render_%s_cfg" % scene_name] = np.array([W, H, frames, seed], np.int64)
    ti.set_math({k: None for k in ()}); ti._math_impl.clear()


# ---- integrator/BDPT_RGB.py (BASELINE config 5's integrator), executed from its source text the same way -------------------------------
# render(): eye_path, light_path (Scene.sample_light), connect_path for every (e, l) with its Scene.closet_hit_shadow, mis_weight with its
# save / modify / restore of vertices through the temp arrays, the light-tracing splats through Camera.get_image_point -- and the
# reference's one-set-of-vertex-arrays-per-pixel that persists from frame to frame (the `delta` memory the device replays in k_bd_delta).
# ti.random() by call site again: which function draws (walking up to eye_path / light_path / connect_path for the depth or the eye
# vertex), the how-many-th draw of that function in bytecode order -> the oracle's BD_DIM_* schedule (oracle.c:1914-1917).
def _random_call_offsets(code):
    """bytecode offsets of the `ti.random()` calls of a function, in source order"""
    import dis
    offs, pending = [], False
    for ins in dis.get_instructions(code):
        if ins.opname in ("LOAD_METHOD", "LOAD_ATTR") and ins.argval == "random":
            pending = True
        elif pending and ins.opname.startswith("CALL"):
            offs.append(ins.offset); pending = False
        elif ins.opname in ("LOAD_GLOBAL", "LOAD_FAST") and pending and ins.argval != "ti":
            pending = False
    return offs


def render_bdpt_reference_text(out, W, H, frames, seed, scene_name):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import oracle_api as oa
    from common import host_only
    from ti_raytrace_amd import scenes
    import Scene as RScene, BDPT_RGB as RBD, SceneData as RSCD
    for _m in (RScene, RBD):
        _m.pow = ti.pow_
    L = oa.load()

    def m1(fn):
        def f(x):
            o = np.zeros(1, np.float32); L.orc_kat_math(fn, np.array([x], np.float32), np.zeros(1, np.float32), o, 1); return o[0]
        return f

    def m2(fn):
        def f(x, y):
            o = np.zeros(1, np.float32); L.orc_kat_math(fn, np.array([x], np.float32), np.array([y], np.float32), o, 1); return o[0]
        return f
    ti.set_math({"sin": m1(0), "cos": m1(1), "exp": m1(2), "log": m1(3), "pow": m2(4), "atan2": m2(5), "acos": m1(6),
                 "tan": lambda x: np.float32(m1(0)(x) / m1(1)(x))})                  # tirt_math.h: tm_tan = tm_sin / tm_cos
    from common import cornell_glass_wall
    if scene_name == "spot_laser":                                # Scene.sample_light's spot / laser branches (Scene.py:449-472) start light sub-paths here
        from common import spot_laser_scene
        ex = spot_laser_scene(W, H, integrator="bdpt")
    else:
        ex = scenes.cornell_box(W, H, 4, device_id=None) if scene_name == "cornell" else cornell_glass_wall(W, H, integrator="bdpt")
    host_only(ex, 0.8)
    sc = ex.scene
    orc = oa.OracleScene(sc, ex.cam)
    orc.lbvh_build()
    _, _, compact = orc.lbvh_get()
    rcam = Camera.Camera(W, H, 4)
    rcam.view.from_numpy(ex.cam.view_np); rcam.view_inv.from_numpy(ex.cam.view_inv_np); rcam.eye.from_numpy(ex.cam.eye_np)
    rs = RScene.Scene()
    rs.material.from_numpy(sc.material_np); rs.vertex.from_numpy(sc.vertex_np); rs.primitive.from_numpy(sc.primitive_np)
    rs.shape.from_numpy(sc.shape_np); rs.light.from_numpy(sc.light_np.astype(np.int32))
    rs.light_count = sc.light_count; rs.primitive_count = sc.primitive_count; rs.env_power = np.float32(sc.env_power)
    rs.env.np_img = sc.env.np_img; rs.env.wid, rs.env.hgt = sc.env.np_img.shape; rs.env.buf.from_numpy(sc.env.np_img)

    class _B:
        pass
    rs.bvh = _B(); rs.bvh.compact_node = ti.Vector.field(RSCD.CPNOD_VEC_SIZE, dtype=ti.f32); rs.bvh.compact_node.from_numpy(compact)
    bd = RBD.BDPT(W, H, rcam, rs, 64)
    bd.setup_data_cpu()

    offsets = {}
    used = set()
    EYE, LSTART, LIGHT, CONNECT = 16, 80, 96, 176             # oracle.c: BD_DIM_EYE / LSTART / LIGHT / CONNECT

    def rnd():
        f = sys._getframe(2)
        if f.f_code not in offsets:
            offsets[f.f_code] = _random_call_offsets(f.f_code)
        k = offsets[f.f_code].index(f.f_lasti)
        name = f.f_code.co_name
        chain = []
        g = f
        while g is not None and g.f_code.co_name != "render":
            chain.append(g); g = g.f_back
        names = [c.f_code.co_name for c in chain]
        i, j = g.f_locals["i"], g.f_locals["j"]

        def local(fn, var):
            return next(c for c in chain if c.f_code.co_name == fn).f_locals[var]
        if name == "get_ray_direction":
            dim = k
        elif name in ("eye_path", "light_path"):                 # the extinction roulette of that sub-path's vertex
            dim = (EYE if name == "eye_path" else LIGHT) + 8 * int(f.f_locals["depth"]) + 6
        elif name == "sample" and ("eye_path" in names or "light_path" in names):       # Glass.sample: slot 0; Disney.sample: slots 3, 4, 5
            base = (EYE + 8 * int(local("eye_path", "depth"))) if "eye_path" in names else (LIGHT + 8 * int(local("light_path", "depth")))
            dim = base + (0 if f.f_globals["__name__"] == "Glass" else 3 + k)
        elif "sample_light" in names:                                  # Scene.sample_light for the light sub-path's first vertex
            # sample_light's own draws in source order: the two of the cosine lobe, the two of the spot's disc, the laser's angle -- the laser takes
            # the slot the spot's first number has (oracle.c bd_sample_light: only one of the two branches runs)
            dim = LSTART + {"get_random_light_prim_index": 0, "get_prim_random_point_normal": 1 + k, "sample_light": (3, 4, 5, 6, 5)[k]}[name]
        elif "sample_li" in names and "connect_path" in names:         # the l == 1 connection at eye vertex e
            dim = CONNECT + 4 * int(local("connect_path", "e")) + {"get_random_light_prim_index": 0, "get_prim_random_point_normal": 1 + k}[name]
        else:
            raise RuntimeError("ti.random() from an unexpected place: %s" % names)
        used.add((name, k))
        return L.orc_kat_rand(seed, int(i) * H + int(j), int(rcam.frame_gpu[0]), dim)
    ti.set_random(rnd)

    import io, contextlib
    for fr in range(frames):
        rcam.frame_gpu[0] = fr
        with contextlib.redirect_stdout(io.StringIO()):
            bd.render()
    film = bd.hdr.to_numpy().astype(np.float32)
    want, st, _ = orc.bdpt_render(ex.cam, W, H, 0, frames, seed=seed)
    rel = float(np.sqrt(((film.astype(np.float64) - want) ** 2).sum() / max((want.astype(np.float64) ** 2).sum(), 1e-30)))
    per = np.abs(film.astype(np.float64) - want) / np.maximum(np.abs(want), 1e-3)
    print("BDPT %s %dx%d x %d frames: reference text vs oracle: rel-L2 %.3e, worst value %.3e, draw sites used %d, film mean %.4f"
          % (scene_name, W, H, frames, rel, per.max(), len(used), float(film.mean())))
    out["bdpt_%s_film" % scene_name] = film
    out["bdpt_%s_cfg" % scene_name] = np.array([W, H, frames, seed], np.int64)
    ti._math_impl.clear()


# ---- accel/LBvh.py: the reference's LBVH build executed from its source text ------------------------------------------------------------
# build_morton_3d, the 30 one-bit radix passes with their Blelloch scans (radix_sort_predicate / blelloch_scan_reduce / _downsweep /
# radix_sort_fill), build_lbvh with determineRange / findSplit (the duplicate-code rule included), gen_aabb until done, the recursive Python
# flatten -- on scenes nodelist.txt (35 primitives, no equal codes) does not reach: 154 primitives with runs of IDENTICAL Morton codes and two
# analytic spheres; 701 random primitives.  Struct-for loops run in index order here, in parallel in Taichi: every loop of this file is order-independent
# (scan steps touch disjoint pairs; gen_aabb is iterated to a fixed point).
def lbvh_reference_text(out, which):
    import tempfile
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import oracle_api as oa
    from common import duplicate_code_scene, host_only
    import LBvh as RL, SceneData as RSCD
    RL.min, RL.max = ti.vmin, ti.vmax
    from common import refkat_lbvh_scene
    ex = refkat_lbvh_scene(which)
    host_only(ex, 0.8)
    sc = ex.scene
    n = sc.primitive_count
    vertex = ti.Vector.field(RSCD.VER_VEC_SIZE, dtype=ti.f32); vertex.from_numpy(sc.vertex_np)
    shape = ti.Vector.field(RSCD.SHA_VEC_SIZE, dtype=ti.f32); shape.from_numpy(sc.shape_np)
    prim = ti.Vector.field(RSCD.PRI_VEC_SIZE, dtype=ti.i32); prim.from_numpy(sc.primitive_np)
    bvh = RL.Bvh(n, sc.minboundarynp, sc.maxboundarynp)
    cwd = os.getcwd()
    import io, contextlib
    with tempfile.TemporaryDirectory() as tmp:                     # (the reference writes nodelist.txt into the working directory)
        os.chdir(tmp)
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                bvh.setup_data_cpu()
                bvh.setup_data_gpu(vertex, shape, prim)
        finally:
            os.chdir(cwd)
    morton = bvh.morton_code_s.to_numpy().astype(np.int32)
    node = bvh.bvh_node.to_numpy().astype(np.float32)
    compact = bvh.compact_node.to_numpy().astype(np.float32)
    orc = oa.OracleScene(sc, ex.cam); orc.lbvh_build()
    om, ob, oc = orc.lbvh_get()
    dup = int((np.diff(morton[:, 0]) == 0).sum())
    print("LBVH %d primitives (%d adjacent equal Morton codes): reference text vs oracle: morton %s, bvh_node %s, compact_node %s"
          % (n, dup, np.array_equal(morton, om), np.array_equal(node.view(np.uint32), ob.view(np.uint32)), np.array_equal(compact.view(np.uint32), oc.view(np.uint32))))
    out.update({"lbvh_%s_morton" % which: morton, "lbvh_%s_bvh_node" % which: node, "lbvh_%s_compact_node" % which: compact, "lbvh_%s_n" % which: np.array([n, dup], np.int64)})


# ---- Scene.process_normal (Scene.py:754-798) and Scene.total_area (:747-750) from their source text ---------------------------------------
# The smooth-normal pass walks the LBVH per vertex (point-in-box, the reference's push order), adds angle x area weighted neighbour normals in
# that order, normalises -- including the NaNs it makes where acos gets a dot product just above 1 (Scene.py:377).  sphere.obj of single_model.py.
def normals_reference_text(out):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import oracle_api as oa
    from common import host_only
    from ti_raytrace_amd import scenes
    import Scene as RScene, SceneData as RSCD
    RScene.pow = ti.pow_
    L = oa.load()

    def m1(fn):
        def f(x):
            o = np.zeros(1, np.float32); L.orc_kat_math(fn, np.array([x], np.float32), np.zeros(1, np.float32), o, 1); return o[0]
        return f
    ti.set_math({"acos": m1(6)})
    ex = scenes.single_model(16, 16, 4, model="sphere.obj", device_id=None)
    host_only(ex, 0.8)
    sc = ex.scene
    orc = oa.OracleScene(sc, ex.cam); orc.lbvh_build()
    _, _, compact = orc.lbvh_get()
    rs = RScene.Scene()
    rs.material.from_numpy(sc.material_np); rs.vertex.from_numpy(sc.vertex_np); rs.primitive.from_numpy(sc.primitive_np)
    rs.shape.from_numpy(sc.shape_np); rs.light.from_numpy(sc.light_np.astype(np.int32)); rs.light_count = sc.light_count
    rs.vertex_index.from_numpy(np.ascontiguousarray(sc.vertex_index_np, np.int32))
    rs.smooth_normal.from_numpy(np.zeros((sc.vertex_count, 3), np.float32))
    rs.stack.from_numpy(np.zeros((sc.vertex_count, RScene.MAX_STACK_SIZE), np.int32))
    rs.light_area.from_numpy(np.zeros(1, np.float32))

    class _B:
        pass
    rs.bvh = _B(); rs.bvh.compact_node = ti.Vector.field(RSCD.CPNOD_VEC_SIZE, dtype=ti.f32); rs.bvh.compact_node.from_numpy(compact)
    before = sc.vertex_np.copy()
    rs.process_normal()
    rs.total_area()
    got = rs.vertex.to_numpy().astype(np.float32)
    orc.L.orc_process_normal(orc.h, np.ascontiguousarray(sc.vertex_index_np, np.int32))
    want = orc.vertex()
    nan_same = np.array_equal(np.isnan(got), np.isnan(want))
    fin = np.isfinite(want)
    print("process_normal %d vertices: reference text vs oracle: NaN pattern equal %s (%d NaN components), bit-identical rows %d / %d, max abs diff %.3e; total_area %r vs %r"
          % (len(got), nan_same, int(np.isnan(want).sum()), int(((got.view(np.uint32) == want.view(np.uint32)) | ~fin).all(axis=1).sum()), len(got),
             float(np.abs(np.where(fin, got - want, 0)).max()), float(rs.light_area[0]), float(orc.L.orc_total_area(orc.h))))
    out.update(normals_vertex_before=before.astype(np.float32), normals_vertex_after=got, normals_total_area=np.array([rs.light_area[0]], np.float32))
    ti._math_impl.clear()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "refkat.npz"))
    ap.add_argument("--render", action="store_true", help="also run integrator/PT_RGB.py's render from its source text (minutes)")
    ap.add_argument("--render-only", action="store_true")
    ap.add_argument("--lbvh-only", action="store_true", help="accel/LBvh.py's build from its source text")
    ap.add_argument("--bdpt-only", action="store_true", help="integrator/BDPT_RGB.py's render from its source text (minutes)")
    a = ap.parse_args()
    if not (a.render_only or a.bdpt_only or a.lbvh_only):
        out = {}
        kat_functions(out)
        np.savez_compressed(a.out, **out)
        print("wrote", a.out, "(%d arrays, %.1f KB)" % (len(out), os.path.getsize(a.out) / 1024))
    if a.lbvh_only:
        out = {}
        lbvh_reference_text(out, "duplicates")
        lbvh_reference_text(out, "random700")
        normals_reference_text(out)
        path = a.out.replace("refkat.npz", "refkat_lbvh.npz")
        np.savez_compressed(path, **out)
        print("wrote", path)
        return
    if a.bdpt_only:
        out = {}
        render_bdpt_reference_text(out, 16, 16, 4, 7, "cornell")
        render_bdpt_reference_text(out, 16, 16, 4, 7, "cornell_glass")
        path = a.out.replace("refkat.npz", "refkat_bdpt.npz")
        np.savez_compressed(path, **out)
        print("wrote", path)
        return
    if a.render or a.render_only:
        out = {}
        render_reference_text(out, 16, 16, 4, 7, "cornell")
        render_reference_text(out, 16, 16, 4, 7, "sphere")
        render_reference_text(out, 16, 16, 4, 7, "cornell_glass")
        path = a.out.replace("refkat.npz", "refkat_render.npz")
        np.savez_compressed(path, **out)
        print("wrote", path)


if __name__ == "__main__":
    main()
