"""ctypes wrapper around oracle/liboracle.so -- the CPU restatement of the reference
(TEST INFRASTRUCTURE; see the header of oracle/oracle.c).  Imported only from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg.

It consumes exactly the packed arrays the host ``Scene`` produces (``vertex_np`` ...), so a
test drives the HIP path and the oracle from the same inputs.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_vp = C.c_void_p


class OrcStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "rays_closest", "rays_shadow", "box_closest", "leaf_closest", "box_shadow", "leaf_shadow",
        "shaded", "paths", "max_stack", "overflow")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


_libs = {}


def build():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


def build_native():
    """-O3 -march=native build made ON THE MACHINE THAT RUNS IT (bench.py's cpu_baseline leg; never shipped:
    oracle/_native/ is git- and gpurun-ignored).  Returns the path, or None when gcc is not there."""
    try:
        subprocess.run(["make", "-s", "-C", ORACLE_DIR, "native"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    except Exception:
        return None
    path = os.path.join(ORACLE_DIR, "_native", "liboracle_native.so")
    return path if os.path.exists(path) else None


def load(libm=False, native=False):
    name = "liboracle_libm.so" if libm else "liboracle.so"
    if native:
        name = "_native/liboracle_native.so"
    if name not in _libs:
        path = os.path.join(ORACLE_DIR, name)
        if native and build_native() is None:
            raise OSError("cannot build the native oracle")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_scene_create.restype = _vp
        L.orc_scene_create.argtypes = [_f32p, C.c_int, _i32p, C.c_int, _f32p, C.c_int, _f32p, C.c_int,
                                       _i32p, C.c_int, C.c_int, _f32p, _f32p]
        L.orc_scene_destroy.argtypes = [_vp]
        L.orc_env_set.argtypes = [_vp, _i32p, C.c_int, C.c_int, C.c_float]
        L.orc_camera_set.argtypes = [_vp, _f32p, _f32p, C.c_float, C.c_float, C.c_float, C.c_float]
        L.orc_lbvh_build.restype = C.c_int
        L.orc_lbvh_build.argtypes = [_vp]
        L.orc_lbvh_get.argtypes = [_vp, _i32p, _f32p, _f32p]
        L.orc_morton_codes.argtypes = [_vp, _i32p]
        L.orc_gen_aabb_rounds.restype = C.c_int
        L.orc_gen_aabb_rounds.argtypes = [_vp]
        L.orc_vertex_get.argtypes = [_vp, _f32p]
        L.orc_closest_hit_batch.argtypes = [_vp, _f32p, C.c_int, C.c_int, _f32p, _i32p, _vp]
        L.orc_shadow_hit_batch.argtypes = [_vp, _f32p, C.c_int, C.c_int, _f32p, _i32p, _vp]
        L.orc_pt_rgb_render.restype = C.c_int
        L.orc_pt_rgb_render.argtypes = [_vp, C.c_int, C.c_int, C.c_uint32, C.c_int, C.c_uint32, C.c_int, C.c_int,
                                        _f32p, C.c_long, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.POINTER(OrcStats)]
        L.orc_bdpt_create.restype = _vp
        L.orc_bdpt_create.argtypes = [C.c_int, C.c_int, _f32p]
        L.orc_bdpt_destroy.argtypes = [_vp]
        L.orc_bdpt_render.restype = C.c_int
        L.orc_bdpt_render.argtypes = [_vp, _vp, C.c_uint32, C.c_int, C.c_uint32, C.c_int, _f32p, _f32p, C.POINTER(OrcStats)]
        L.orc_bdpt_spec_render.restype = C.c_int
        L.orc_bdpt_spec_render.argtypes = [_vp, _vp, _vp, C.c_uint32, C.c_int, C.c_uint32, C.c_int, _f32p, _f32p, C.POINTER(OrcStats)]
        L.orc_tone_map.argtypes = [C.c_float, _f32p, _f32p, C.c_long]
        L.orc_total_area.restype = C.c_float
        L.orc_total_area.argtypes = [_vp]
        L.orc_process_normal.argtypes = [_vp, _i32p]
        L.orc_kat_disney.argtypes = [_f32p, _f32p, _f32p, _f32p, _f32p]
        L.orc_kat_disney_sample.argtypes = [_f32p, _f32p, _f32p, _f32p, _f32p]
        L.orc_kat_glass_sample.argtypes = [_f32p, _f32p, _f32p, C.c_float, _f32p]
        L.orc_kat_offset_ray.argtypes = [_f32p, _f32p, _f32p]
        L.orc_kat_slabs.restype = C.c_int
        L.orc_kat_slabs.argtypes = [_f32p, _f32p, _f32p, _f32p]
        L.orc_kat_morton3d.restype = C.c_int32
        L.orc_kat_morton3d.argtypes = [C.c_float, C.c_float, C.c_float]
        L.orc_kat_rand.restype = C.c_float
        L.orc_kat_rand.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_kat_math.argtypes = [C.c_int, _f32p, _f32p, _f32p, C.c_int]
        L.orc_uses_libm.restype = C.c_int
        L.orc_kat_spec.argtypes = [_vp, C.c_int, _f32p, _f32p]
        L.orc_spec_table_build.restype = C.c_int
        L.orc_spec_table_build.argtypes = [C.c_int, _f32p, _f32p, C.c_int, _f32p, _f32p, C.c_int]
        L.orc_spec_create.restype = _vp
        L.orc_spec_create.argtypes = [_f32p, C.c_int, C.c_float, C.c_float, C.c_float, _f32p, _i32p, _f32p, _f32p, _f32p,
                                      _f32p, _f32p, C.c_int, _f32p, _f32p, _f32p]
        L.orc_spec_destroy.argtypes = [_vp]
        L.orc_pt_spec_render.restype = C.c_int
        L.orc_pt_spec_render.argtypes = [_vp, _vp, C.c_int, C.c_int, C.c_uint32, C.c_int, C.c_uint32, C.c_int, C.c_int,
                                         _f32p, C.c_long, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(OrcStats)]
        _libs[name] = L
    return _libs[name]


class OracleScene:
    """One oracle scene built from a host ``ti_raytrace_amd.Scene`` after setup_data_cpu()."""

    def __init__(self, scene, cam=None, libm=False, native=False):
        self.L = load(libm, native)
        self.n = scene.primitive_count
        self.nv = scene.vertex_count
        self.N = 2 * self.n - 1
        self.h = self.L.orc_scene_create(
            np.ascontiguousarray(scene.vertex_np.reshape(-1)), self.nv,
            np.ascontiguousarray(scene.primitive_np.reshape(-1)), self.n,
            np.ascontiguousarray(scene.material_np.reshape(-1)), scene.material_np.shape[0],
            np.ascontiguousarray(scene.shape_np.reshape(-1)), scene.shape_np.shape[0],
            np.ascontiguousarray(scene.light_np), scene.light_np.shape[0], scene.light_count,
            np.ascontiguousarray(scene.minboundarynp.reshape(-1)), np.ascontiguousarray(scene.maxboundarynp.reshape(-1)))
        env = scene.env.np_img
        self.L.orc_env_set(self.h, np.ascontiguousarray(env.reshape(-1)), env.shape[0], env.shape[1], float(scene.env_power))
        if cam is not None:
            self.set_camera(cam)

    def close(self):
        if self.h:
            self.L.orc_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_camera(self, cam):
        self.L.orc_camera_set(self.h, np.ascontiguousarray(cam.view_inv_np[0].reshape(-1)),
                              np.ascontiguousarray(cam.eye_np[0]), cam.fx, cam.fy, cam.cx, cam.cy)

    def lbvh_build(self):
        return self.L.orc_lbvh_build(self.h)

    def lbvh_get(self):
        morton = np.zeros((self.n, 2), np.int32)
        bvh = np.zeros((self.N, 11), np.float32)
        compact = np.zeros((self.N, 9), np.float32)
        self.L.orc_lbvh_get(self.h, morton.reshape(-1), bvh.reshape(-1), compact.reshape(-1))
        return morton, bvh, compact

    def morton_codes(self):
        out = np.zeros((self.n, 2), np.int32)
        self.L.orc_morton_codes(self.h, out.reshape(-1))
        return out

    def gen_aabb_rounds(self):
        return self.L.orc_gen_aabb_rounds(self.h)

    def vertex(self):
        out = np.zeros((self.nv, 9), np.float32)
        self.L.orc_vertex_get(self.h, out.reshape(-1))
        return out

    def closest_hit(self, rays, stack_size=64, counts=False):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        nr = rays.shape[0]
        out = np.zeros((nr, 13), np.float32)
        prim = np.zeros(nr, np.int32)
        cnt = np.zeros((nr, 2), np.int32) if counts else None
        self.L.orc_closest_hit_batch(self.h, rays.reshape(-1), nr, stack_size, out.reshape(-1), prim,
                                     cnt.ctypes.data_as(_vp) if counts else None)
        return out, prim, cnt

    def shadow_hit(self, rays, stack_size=64, counts=False):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        nr = rays.shape[0]
        out = np.zeros(nr, np.float32)
        prim = np.zeros(nr, np.int32)
        cnt = np.zeros((nr, 2), np.int32) if counts else None
        self.L.orc_shadow_hit_batch(self.h, rays.reshape(-1), nr, stack_size, out, prim,
                                    cnt.ctypes.data_as(_vp) if counts else None)
        return out, prim, cnt

    def render(self, W, H, frame_begin, frame_count, seed=1, max_depth=15, stack_size=64, hdr=None,
               p_begin=0, p_end=None, tile_rank=0, tile_count=1, tile_size=4096, nthreads=None):
        if hdr is None:
            hdr = np.zeros((W, H, 3), np.float32)
        if p_end is None:
            p_end = W * H
        if nthreads is None:
            nthreads = os.cpu_count() or 1
        st = OrcStats()
        self.L.orc_pt_rgb_render(self.h, W, H, frame_begin, frame_count, seed, max_depth, stack_size,
                                 hdr.reshape(-1), p_begin, p_end, tile_rank, tile_count, tile_size,
                                 nthreads, C.byref(st))
        return hdr, st.as_dict()

    def bdpt_render(self, cam, W, H, frame_begin, frame_count, seed=1, stack_size=64, hdr=None, state=None):
        """BDPT_RGB.render x frame_count.  Returns (hdr, stats, state); pass `state` back to continue
        with the persistent per-pixel vertex arrays of the previous frames."""
        if hdr is None:
            hdr = np.zeros((W, H, 3), np.float32)
        if state is None:
            state = self.L.orc_bdpt_create(W, H, np.ascontiguousarray(cam.view_np[0].reshape(-1), np.float32))
        rad = np.zeros((W, H, 3), np.float32)
        st = OrcStats()
        self.L.orc_bdpt_render(self.h, state, frame_begin, frame_count, seed, stack_size, rad.reshape(-1), hdr.reshape(-1), C.byref(st))
        return hdr, st.as_dict(), state

    def bdpt_spec_render(self, cam, W, H, frame_begin, frame_count, seed=1, stack_size=64, hdr=None, state=None):
        """BDPT_SPEC.render x frame_count (set_spectral first).  Same conventions as bdpt_render."""
        if hdr is None:
            hdr = np.zeros((W, H, 3), np.float32)
        if state is None:
            state = self.L.orc_bdpt_create(W, H, np.ascontiguousarray(cam.view_np[0].reshape(-1), np.float32))
        rad = np.zeros((W, H, 3), np.float32)
        st = OrcStats()
        self.L.orc_bdpt_spec_render(self.h, self.spec, state, frame_begin, frame_count, seed, stack_size, rad.reshape(-1), hdr.reshape(-1), C.byref(st))
        return hdr, st.as_dict(), state

    def set_spectral(self, t):
        """t: PT_Spec.PathTrace.tables() -- the same arrays the device gets"""
        f = lambda k: np.ascontiguousarray(t[k], np.float32)
        self.spec = self.L.orc_spec_create(f("sensor"), int(t["n_sensor"]), t["s_min"], t["s_max"], t["s_range"], f("spd"),
                                           np.asarray(t["spd_n"], np.int32), np.asarray(t["spd_min"], np.float32),
                                           np.asarray(t["spd_max"], np.float32), np.asarray(t["spd_range"], np.float32),
                                           f("tbl_scale"), f("tbl_data"), int(t["tbl_res"]), f("sky_cfg"), f("sky_rad"),
                                           np.asarray(t["sun_dir"], np.float32))

    def spec_render(self, W, H, frame_begin, frame_count, seed=1, max_depth=10, stack_size=64, hdr=None,
                    p_begin=0, p_end=None, tile_rank=0, tile_count=1, tile_size=4096, nthreads=None):
        if hdr is None:
            hdr = np.zeros((W, H, 3), np.float32)
        if p_end is None:
            p_end = W * H
        if nthreads is None:
            nthreads = os.cpu_count() or 1
        st = OrcStats()
        self.L.orc_pt_spec_render(self.h, self.spec, W, H, frame_begin, frame_count, seed, max_depth, stack_size,
                                  hdr.reshape(-1), p_begin, p_end, tile_rank, tile_count, tile_size, nthreads, C.byref(st))
        return hdr, st.as_dict()

    def kat_spec(self, which, inp, out_stride):
        """orc_kat_spec row by row (set_spectral first): the spectral functions one by one, layouts in oracle.c"""
        inp = np.ascontiguousarray(inp, np.float32)
        out = np.zeros((inp.shape[0], out_stride), np.float32)
        for i in range(inp.shape[0]):
            self.L.orc_kat_spec(self.spec, int(which), inp[i], out[i])
        return out

    def tone_map(self, exposure, hdr):
        out = np.zeros_like(hdr)
        self.L.orc_tone_map(exposure, np.ascontiguousarray(hdr.reshape(-1)), out.reshape(-1), hdr.size // 3)
        return out

    def total_area(self):
        return float(self.L.orc_total_area(self.h))

    def process_normal(self, vertex_index):
        self.L.orc_process_normal(self.h, np.ascontiguousarray(vertex_index, np.int32))


def spec_table_build(res, cie_xyz, d65, nthreads=None, libm=False):
    """spectrum/JakobSpecTable.py in the oracle (double precision, pthreads): (scale[res], coeff[3*res^3*3]) as float32."""
    L = load(libm)
    scale = np.zeros(res, np.float32)
    coeff = np.zeros(9 * res ** 3, np.float32)
    rc = L.orc_spec_table_build(res, np.ascontiguousarray(cie_xyz, np.float32).reshape(-1), np.ascontiguousarray(d65, np.float32),
                                int(np.asarray(d65).size), scale, coeff, nthreads or (os.cpu_count() or 1))
    assert rc == 0
    return scale, coeff


def camera_rays(cam, W, H, pixels=None):
    """Frame-0 primary rays (no jitter) as the oracle/GPU generate them (Camera.py:122-142),
    in f32 numpy -- used to feed the batch closest-hit entry points.  Returns [n,6]."""
    f = np.float32
    if pixels is None:
        ii, jj = np.meshgrid(np.arange(W), np.arange(H), indexing="ij")
        ii, jj = ii.reshape(-1), jj.reshape(-1)
    else:
        ii, jj = pixels // H, pixels % H
    x = ((ii.astype(f) + f(0.0)) - f(cam.cx)) / f(cam.fx)
    y = ((jj.astype(f) + f(0.0)) - f(cam.cy)) / f(cam.fy)
    z = np.full_like(x, -1.0)
    M = cam.view_inv_np[0].astype(f)
    w = [((M[r, 0] * x + M[r, 1] * y) + M[r, 2] * z) + M[r, 3] * f(0.0) for r in range(3)]
    n2 = (w[0] * w[0] + w[1] * w[1]) + w[2] * w[2]
    inv = f(1.0) / np.sqrt(n2)
    d = np.stack([w[0] * inv, w[1] * inv, w[2] * inv], axis=1).astype(f)
    o = np.broadcast_to(cam.eye_np[0].astype(f), d.shape)
    return np.ascontiguousarray(np.concatenate([o, d], axis=1), dtype=f)
