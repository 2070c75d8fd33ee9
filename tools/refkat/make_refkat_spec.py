"""Golden vectors for the SPECTRAL path and the spot / laser emitters, computed by the reference's own source text -- build container only.

    python tools/refkat/make_refkat_spec.py            # writes tests/golden/refkat_spec.npz

Same method as make_refkat.py (which this imports for the stand-in set-up): the reference's spectrum/{HeroSample,Spectrum,Rgb2Spec}.py,
sky/Sky.py, integrator/{PT_Spec,BDPT_SPEC}.py and the spot / laser branches of Scene.py run as plain Python on seeded inputs; numbers only
are written.  One INPUT the reference repository lacks comes from this repo: spectrum/spec_table (.MISSING_LARGE_BLOBS) -- Rgb2Spec.load_table
is given the table the build's own generator makes (the oracle's restatement of spectrum/JakobSpecTable.py; the device builds the same bits).
Everything that READS the table -- Rgb2Spec.fetch / eval, HeroSample.srgb_to_spec, PT_Spec.emission_to_rad / get_spec_power -- is the
reference's text."""
import builtins
import contextlib
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_refkat as MR               # noqa: E402  (stand-in on sys.path, UF / Disney / Glass / Camera imported and patched)
from make_refkat import ti, UF, Glass, Camera, V, vec, ROOT, REF      # noqa: E402

sys.path[:0] = [os.path.join(REF, "spectrum"), os.path.join(REF, "sky")]
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]


@contextlib.contextmanager
def in_reference_dir():
    """the reference opens "spectrum/..." and "sky\\data.csv" relative to its checkout (the latter with a backslash: Windows)"""
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        yield
    finally:
        os.chdir(cwd)


def _open_any(path, mode="r", *a, **kw):
    return builtins.open(path.replace("\\", "/"), mode, *a, **kw)


def install_math(L):
    def m1(fn):
        def f(x):
            o = np.zeros(1, np.float32); L.orc_kat_math(fn, np.array([x], np.float32), np.zeros(1, np.float32), o, 1); return o[0]
        return f

    def m2(fn):
        def f(x, y):
            o = np.zeros(1, np.float32); L.orc_kat_math(fn, np.array([x], np.float32), np.array([y], np.float32), o, 1); return o[0]
        return f
    ti.set_math({"sin": m1(0), "cos": m1(1), "exp": m1(2), "log": m1(3), "pow": m2(4), "atan2": m2(5), "acos": m1(6)})


def reference_spectral_integrator(W, H, ex, which="pt", stack_size=64):
    """The reference's PT_Spec.PathTrace (or BDPT_SPEC.BDPT) set up from its own text on the scene of `ex` (host-side example of this repo whose
    packed arrays go straight into the reference's fields, as in make_refkat.render_reference_text)."""
    import oracle_api as oa
    import Sky as RSky
    import Rgb2Spec as RR2S
    import Spectrum as RSpectrum      # noqa: F401
    import HeroSample as RHero        # noqa: F401
    import Scene as RScene, SceneData as RSCD
    RSky.open = _open_any
    it = ex.integrator
    sc = ex.scene
    orc = oa.OracleScene(sc, ex.cam)
    orc.lbvh_build()
    if getattr(ex, "_smooth", True):
        orc.L.orc_process_normal(orc.h, np.ascontiguousarray(sc.vertex_index_np, np.int32))
        sc.vertex_np[...] = orc.vertex()
    _, _, compact = orc.lbvh_get()
    orc.set_spectral(it.tables())

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

    # the one missing input: spectrum/spec_table.  load_table's product (Rgb2Spec.py:14-37), from the table of this repo
    tbl_scale, tbl_data, res = it.rgb2spec.table_scale_np, it.rgb2spec.table_data_np, it.rgb2spec.table_res

    def load_table(self, table_path):
        self.table_res = res; self.table_size = res * res * res * 9
        self.dx = 3; self.dy = 3 * res; self.dz = 3 * res * res
        self.table_scale_np = tbl_scale.copy(); self.table_data_np = tbl_data.copy()
        ti.root.dense(ti.i, (self.table_size)).place(self.table_data)
        ti.root.dense(ti.i, (self.table_res)).place(self.table_scale)
    RR2S.Rgb2Spec.load_table = load_table
    if which == "pt":
        import PT_Spec as RPTS
        mod = RPTS
        with in_reference_dir():
            pt = RPTS.PathTrace(W, H, rcam, rs, stack_size)
            pt.setup_data_cpu()
            pt.setup_data_gpu()                        # sky.update() runs here, in Python floats: BEFORE `pow` of the modules becomes fp32
    else:
        import BDPT_SPEC as RBS
        mod = RBS
        with in_reference_dir():
            pt = RBS.BDPT(W, H, rcam, rs, stack_size)
            pt.setup_data_cpu()
            pt.setup_data_gpu()
    for m in (RSky, RR2S, mod, RScene):
        m.pow = ti.pow_
    RR2S.min, RR2S.max = ti.vmin, ti.vmax
    return pt, rs, rcam, orc


def spectral_kats(out):
    import oracle_api as oa
    from ti_raytrace_amd import scenes
    import HeroSample as RHero
    W = H = 16
    ex = scenes.spectral_box(W, H, 4)
    ex.scene.setup_data_cpu(); ex.frame_camera(0.8)
    ex.integrator.setup_data_cpu()
    ex.integrator.setup_tables(lambda res, xyz, d65: oa.spec_table_build(res, xyz, d65))
    L = oa.load()
    install_math(L)
    pt, rs, rcam, orc = reference_spectral_integrator(W, H, ex, "pt")
    t = ex.integrator.tables()

    # ---- the tables the reference's own set-up produced (setup_data_cpu / setup_data_gpu, Sky.update in Python floats, normalize_spec) ----
    ref_tables = {
        "sensor": pt.sensor.to_numpy().astype(np.float32).reshape(-1),
        "d65": pt.d65.data.to_numpy().astype(np.float32), "white": pt.white.data.to_numpy().astype(np.float32),
        "red": pt.red.data.to_numpy().astype(np.float32), "green": pt.green.data.to_numpy().astype(np.float32),
        "sky_cfg": pt.sky.configs.to_numpy().astype(np.float32).reshape(-1), "sky_rad": pt.sky.radiances.to_numpy().astype(np.float32),
        "sun_dir": pt.sky.sun_dir.to_numpy().astype(np.float32).reshape(-1),
        "spd_meta": np.array([[s.size, s.lambda_min, s.lambda_max, s.lambda_range] for s in (pt.d65, pt.white, pt.red, pt.green)], np.float64),
        "sensor_meta": np.array([pt.size, pt.lambda_min, pt.lambda_max, pt.lambda_range], np.float64),
    }
    mine = np.concatenate([ex.integrator.d65.data_np, ex.integrator.white.data_np, ex.integrator.red.data_np, ex.integrator.green.data_np])
    theirs = np.concatenate([ref_tables[k] for k in ("d65", "white", "red", "green")])
    print("tables: sensor %s, spectra %s (d65 after normalize_spec: max rel diff %.2e), sky configs %s (max rel %.2e), radiances %s, sun_dir %s"
          % (np.array_equal(ref_tables["sensor"], t["sensor"]), np.array_equal(theirs, mine),
             float(np.abs(ref_tables["d65"] - ex.integrator.d65.data_np).max() / np.abs(ref_tables["d65"]).max()),
             np.array_equal(ref_tables["sky_cfg"], t["sky_cfg"]), float(np.abs(ref_tables["sky_cfg"] - t["sky_cfg"]).max() / np.abs(t["sky_cfg"]).max()),
             np.array_equal(ref_tables["sky_rad"], t["sky_rad"]), np.allclose(ref_tables["sun_dir"], t["sun_dir"], rtol=0, atol=1e-7)))
    for k, v in ref_tables.items():
        out["spec_tables_" + k] = v

    r = np.random.RandomState(20260930)
    n = 600
    specs = [pt.d65, pt.white, pt.red, pt.green]
    # 0 Spectrum.sample
    kk = r.randint(0, 4, n); lam = r.uniform(280.0, 900.0, n).astype(np.float32)
    lam[:8] = [300.0, 830.0, 360.0, 380.0, 730.0, 780.0, 555.5, 299.99]
    o0 = np.array([specs[kk[i]].sample(lam[i]) for i in range(n)], np.float32)
    out.update(spec_k0_in=np.stack([kk.astype(np.float32), lam], 1), spec_k0=o0[:, None])
    # 1 HeroSample.sample
    lam0 = r.uniform(360.0, 460.0, n).astype(np.float32)
    o1 = np.array([RHero.sample(specs[kk[i]], lam0[i]).to_numpy() for i in range(n)], np.float32)
    out.update(spec_k1_in=np.stack([kk.astype(np.float32), lam0], 1), spec_k1=o1)
    # 2 HeroSample.sample_xyz on PathTrace.sample
    o2 = np.zeros((n, 12), np.float32)
    for i in range(n):
        x, y, z = RHero.sample_xyz(pt, lam0[i])
        o2[i] = np.concatenate([x.to_numpy(), y.to_numpy(), z.to_numpy()])
    out.update(spec_k2_in=lam0[:, None].copy(), spec_k2=o2)
    # 3 Rgb2Spec.fetch, 4 eval
    rgb = r.uniform(0, 1, (n, 3)).astype(np.float32)
    rgb[:10] = [[0, 0, 0], [1, 1, 1], [0.5, 0.5, 0.5], [1, 0, 0], [0, 1, 0], [0, 0, 1], [0.2, 0.2, 0.9], [0.9, 0.9, 0.2], [1e-7, 1e-6, 1e-7], [1.5, -0.2, 0.3]]
    o3 = np.array([pt.rgb2spec.fetch(vec(rgb[i])).to_numpy() for i in range(n)], np.float32)
    lam1 = r.uniform(360.0, 760.0, n).astype(np.float32)
    o4 = np.array([pt.rgb2spec.eval(vec(o3[i]), lam1[i]) for i in range(n)], np.float32)
    out.update(spec_k3_in=rgb, spec_k3=o3, spec_k4_in=np.concatenate([o3, lam1[:, None]], 1), spec_k4=o4[:, None])
    # 5 HeroSample.srgb_to_spec
    o5 = np.array([RHero.srgb_to_spec(pt.rgb2spec, vec(rgb[i]), lam0[i]).to_numpy() for i in range(n)], np.float32)
    out.update(spec_k5_in=np.concatenate([rgb, lam0[:, None]], 1), spec_k5=o5)
    # 6 HeroSample.sky_sample
    theta = r.uniform(0.0, 0.5 * 3.1415926, n).astype(np.float32); gamma = r.uniform(0.0, 3.14, n).astype(np.float32)
    lam2 = r.uniform(300.0, 460.0, n).astype(np.float32)
    o6 = np.array([RHero.sky_sample(pt.sky, theta[i], gamma[i], lam2[i]).to_numpy() for i in range(n)], np.float32)
    out.update(spec_k6_in=np.stack([theta, gamma, lam2], 1), spec_k6=o6)
    # 7 PathTrace.emission_to_rad
    em = (rgb * r.choice([0.0, 1.0, 17.0, 500.0], (n, 1))).astype(np.float32)
    o7 = np.array([pt.emission_to_rad(vec(em[i]), lam0[i]).to_numpy() for i in range(n)], np.float32)
    out.update(spec_k7_in=np.concatenate([em, lam0[:, None]], 1), spec_k7=o7)
    # 8 HeroSample.get_extinction_hero
    tt = r.uniform(0.0, 2000.0, n).astype(np.float32)
    o8 = np.array([RHero.get_extinction_hero(lam0[i], tt[i]).to_numpy() for i in range(n)], np.float32)
    out.update(spec_k8_in=np.stack([lam0, tt], 1), spec_k8=o8)
    # 9 PathTrace.AddSplat
    sp4 = r.uniform(0, 2, (n, 4)).astype(np.float32); coff = (1.0 / r.randint(1, 9, n)).astype(np.float32); hdr0 = r.uniform(0, 1, (n, 3)).astype(np.float32)
    o9 = np.zeros((n, 3), np.float32)
    for i in range(n):
        pt.hdr[0, 0] = vec(hdr0[i])
        pt.AddSplat(vec(sp4[i]), 0, 0, lam0[i], coff[i])
        o9[i] = pt.hdr[0, 0].to_numpy()
    pt.hdr[0, 0] = vec([0, 0, 0])
    out.update(spec_k9_in=np.concatenate([sp4, lam0[:, None], coff[:, None], hdr0], 1), spec_k9=o9)
    # 10 PathTrace.get_spec_power
    import SceneData as RSCD
    mats = np.zeros((n, 10), np.float32)
    mats[:, 0] = r.choice([RSCD.MAT_SPECTRAL, RSCD.MAT_DISNEY, RSCD.MAT_GLASS], n); mats[:, 1] = r.randint(0, 3, n); mats[:, 2:5] = rgb
    mf = ti.Vector.field(10, dtype=ti.f32); mf.from_numpy(mats)

    class _S:
        material = mf
    o10 = np.array([pt.get_spec_power(_S, i, lam0[i]).to_numpy() for i in range(n)], np.float32)
    out.update(spec_k10_in=np.concatenate([mats, lam0[:, None]], 1), spec_k10=o10)
    # 11 HeroSample.get_rnd_hero
    u = r.uniform(0, 1, n).astype(np.float32); u[:3] = [0.0, 0.25, 0.999999]
    o11 = np.zeros((n, 2), np.float32)
    for i in range(n):
        ti.set_random([u[i]])
        idx, lm = RHero.get_rnd_hero(lam0[i])
        o11[i] = (idx, lm)
    out.update(spec_k11_in=np.stack([u, lam0], 1), spec_k11=o11)

    # the oracle on the same inputs, now (tests/test_refkat.py asserts it; here: a report)
    for k, stride in ((0, 1), (1, 4), (2, 12), (3, 3), (4, 1), (5, 4), (6, 4), (7, 4), (8, 4), (9, 3), (10, 4), (11, 2)):
        got = orc.kat_spec(k, out["spec_k%d_in" % k], stride); want = out["spec_k%d" % k]
        den = np.maximum(np.abs(want), 1e-3 * max(float(np.abs(want).max()), 1e-30))
        print("kat %2d: oracle vs reference text: max rel %.2e, bit-identical %d / %d" % (k, float((np.abs(got - want) / den).max()), int((got.view(np.uint32) == want.view(np.uint32)).all(axis=1).sum()), n))
    return pt, rs, rcam, orc, ex


def render_pt_spec(out, pt, rs, rcam, orc, W, H, frames, seed):
    """integrator/PT_Spec.py:189-279 `render` from its source text; ti.random() by call site as in make_refkat.render_reference_text
    (dimension 4000 = the hero wavelength drawn in render, slot 7 = HeroSample.get_rnd_hero, slot 0 = Glass.sample_lambda)."""
    import oracle_api as oa
    L = oa.load()
    draws = {}
    SLOTS = {("Camera", "get_ray_direction"): [("abs", 0), ("abs", 1)], ("PT_Spec", "render"): [("abs", 4000)],
             ("Glass", "sample_lambda"): [("b", 0)], ("Scene", "get_random_light_prim_index"): [("b", 0)],
             ("Scene", "get_prim_random_point_normal"): [("b", 1), ("b", 2)],
             ("Disney", "sample"): [("b", 3), ("b", 4), ("b", 5)], ("HeroSample", "get_rnd_hero"): [("b", 7)]}
    used = set()

    def rnd():
        import inspect
        f = sys._getframe(2)
        key = (f.f_globals["__name__"], f.f_code.co_name)
        if f.f_code not in draws:
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
    for fr in range(frames):
        rcam.frame_gpu[0] = fr
        with contextlib.redirect_stdout(io.StringIO()):
            pt.render()
    film = pt.hdr.to_numpy().astype(np.float32)
    want, _ = orc.spec_render(W, H, 0, frames, seed=seed)
    rel = float(np.sqrt(((film.astype(np.float64) - want) ** 2).sum() / max((want.astype(np.float64) ** 2).sum(), 1e-30)))
    ident = int((film.view(np.uint32) == want.view(np.uint32)).all(axis=2).sum())
    print("PT_Spec spectral_box %dx%d x %d frames: reference text vs oracle: rel-L2 %.3e, bit-identical pixels %d / %d, max abs %.3e, draw sites %d, film mean %.4f"
          % (W, H, frames, rel, ident, W * H, float(np.abs(film - want).max()), len(used), float(film.mean())))
    out["render_spec_box_film"] = film
    out["render_spec_box_cfg"] = np.array([W, H, frames, seed], np.int64)


def render_bdpt_spec(out, W, H, frames, seed):
    """integrator/BDPT_SPEC.py:660-691 `render` from its source text on example/prism_rainbow.py (glass prism, sphere light, LASER): eye_path / light_path
    with Glass.sample_lambda (dispersion), Scene.sample_light for the light sub-path AND for the l == 1 connections (:605), connect_path / mis_weight,
    AddSplat through the CIE observer.  ti.random() by call site -> the oracle's BD_DIM_* schedule (oracle.c:1914-1917, 1996-1997)."""
    import oracle_api as oa
    from ti_raytrace_amd import scenes
    L = oa.load()
    install_math(L)
    ti._math_impl["tan"] = lambda x: np.float32(ti._math_impl["sin"](x) / ti._math_impl["cos"](x))
    ex = scenes.prism_rainbow(W, H, 4)
    ex.scene.setup_data_cpu(); ex.integrator.setup_data_cpu()
    ex.integrator.setup_tables(lambda res, xyz, d65: oa.spec_table_build(res, xyz, d65))
    ex.cam.scale = 10.0; ex.cam.set_target(0.0, 0.0, 0.0); ex.cam.update()
    ex._smooth = False
    bd, rs, rcam, orc = reference_spectral_integrator(W, H, ex, "bdpt", stack_size=1024)       # example/prism_rainbow.py:21
    offsets = {}
    used = set()
    EYE, LSTART, LIGHT, CONNECT_SPEC, LAMBDA = 16, 80, 96, 256, 2
    SL = {"get_random_light_prim_index": lambda k: 0, "get_prim_random_point_normal": lambda k: 1 + k, "sample_light": lambda k: (3, 4, 5, 6, 5)[k]}

    def rnd():
        f = sys._getframe(2)
        if f.f_code not in offsets:
            offsets[f.f_code] = MR._random_call_offsets(f.f_code)
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
        if name == "render":
            dim = LAMBDA
        elif name == "get_ray_direction":
            dim = k
        elif name in ("sample", "sample_lambda") and ("eye_path" in names or "light_path" in names):
            base = (EYE + 8 * int(local("eye_path", "depth"))) if "eye_path" in names else (LIGHT + 8 * int(local("light_path", "depth")))
            dim = base + (0 if f.f_globals["__name__"] == "Glass" else 3 + k)
        elif "sample_light" in names and "light_path" in names:
            dim = LSTART + SL[name](k)
        elif "sample_light" in names and "connect_path" in names:
            dim = CONNECT_SPEC + 8 * int(local("connect_path", "e")) + SL[name](k)
        else:
            raise RuntimeError("ti.random() from an unexpected place: %s" % names)
        used.add((name, k))
        return L.orc_kat_rand(seed, int(i) * H + int(j), int(rcam.frame_gpu[0]), dim)
    ti.set_random(rnd)
    # Ill-conditioned splats: a connection whose shadow ray is shorter than 1e-3 scene units (an eye vertex ON the sphere light connected to a
    # point of the same light a hair away: G = |cos cos| / t^2 with t ~ 5e-5 out of a sphere intersection that cancels five digits) turns one ulp
    # of its inputs into per cents of its value.  The pixels such splats land on are recorded: the tests hold them to 10 %, everything else to 1e-5.
    import Scene as RScene, BDPT_SPEC as RBS
    last_t, ill = {}, set()
    chs, add = RScene.Scene.closet_hit_shadow, RBS.BDPT.AddSplat

    def chs_logged(self, origin, direction, stack, i, j, MAX_SIZE):
        r = chs(self, origin, direction, stack, i, j, MAX_SIZE)
        last_t[(int(i), int(j))] = float(r[0])
        return r

    def add_logged(self, new_pos, Lambda, radiance):
        f = sys._getframe(1)
        ij = (int(f.f_locals["i"]), int(f.f_locals["j"]))
        if float(radiance) > 0.0 and int(f.f_locals["l"]) >= 1 and last_t.get(ij, 1.0) < 1e-3:
            ill.add(tuple(int(x) for x in new_pos.e))
        return add(self, new_pos, Lambda, radiance)
    RScene.Scene.closet_hit_shadow, RBS.BDPT.AddSplat = chs_logged, add_logged
    try:
        for fr in range(frames):
            rcam.frame_gpu[0] = fr
            with contextlib.redirect_stdout(io.StringIO()):
                bd.render()
    finally:
        RScene.Scene.closet_hit_shadow, RBS.BDPT.AddSplat = chs, add
    film = bd.hdr.to_numpy().astype(np.float32)
    out["bdpt_spec_prism_illcond"] = np.array(sorted(ill), np.int32).reshape(-1, 2)
    print("ill-conditioned splat pixels:", sorted(ill))
    want, st, _ = orc.bdpt_spec_render(ex.cam, W, H, 0, frames, seed=seed, stack_size=1024)
    rel = float(np.sqrt(((film.astype(np.float64) - want) ** 2).sum() / max((want.astype(np.float64) ** 2).sum(), 1e-30)))
    per = np.abs(film.astype(np.float64) - want) / np.maximum(np.abs(want), 1e-3 * float(np.abs(want).max()))
    print("BDPT_SPEC prism_rainbow %dx%d x %d frames: reference text vs oracle: rel-L2 %.3e, worst value %.3e, draw sites used %d (%s), film mean %.5f, lit pixels %d"
          % (W, H, frames, rel, per.max(), len(used), sorted(used), float(film.mean()), int((film.sum(axis=2) > 0).sum())))
    out["bdpt_spec_prism_film"] = film
    out["bdpt_spec_prism_cfg"] = np.array([W, H, frames, seed], np.int64)
    ti._math_impl.clear()


def main():
    out = {}
    pt, rs, rcam, orc, ex = spectral_kats(out)
    render_pt_spec(out, pt, rs, rcam, orc, 16, 16, 4, 7)
    ti._math_impl.clear()
    # the spot / laser emitters through the RGB integrators (make_refkat.py's machinery; keys render_spot_laser_*, bdpt_spot_laser_*)
    MR.render_reference_text(out, 16, 16, 4, 7, "spot_laser")
    MR.render_bdpt_reference_text(out, 16, 16, 4, 7, "spot_laser")
    render_bdpt_spec(out, 16, 16, 4, 7)
    path = os.path.join(ROOT, "tests", "golden", "refkat_spec.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "(%d arrays, %.1f KB)" % (len(out), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
