"""ctypes binding of libtirt.so (the HIP library; C-ABI in include/tirt.h).

There is no CPU fallback: if the library is missing, or no HIP device can be opened, the
calls raise.  (The CPU oracle under oracle/ is test infrastructure and is never loaded
from here.)
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TIRT_LIB_PATH") or os.path.join(_HERE, "csrc", "libtirt.so")      # (TIRT_LIB_PATH: A/B builds of the same library, tools/ab*.sh)

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


class TirtError(RuntimeError):
    pass


class TirtStackOverflow(TirtError):
    """tirt_stats found rays whose traversal stack overflowed (TIRT_ERR_STACK): subtrees were dropped, the
    rendered result is wrong -- raise `stack_size`.  `.stats` holds the statistics that were read."""

    def __init__(self, msg, stats):
        TirtError.__init__(self, msg)
        self.stats = stats


ERR_STACK = -4


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "rays_closest", "rays_shadow", "box_closest", "leaf_closest", "box_shadow",
        "leaf_shadow", "shaded", "paths", "stack_overflow")] + [
        (n, C.c_double) for n in ("ms_build", "ms_render", "ms_trace_closest",
                                  "ms_trace_shadow", "ms_shade")] + [
        (n, C.c_uint64) for n in ("launches_trace_closest", "launches_trace_shadow",
                                  "launches_shade", "diag_it_node", "diag_lanes_node", "diag_it_leaf",
                                  "diag_lanes_leaf", "diag_refills", "diag_it_outer", "diag_wave_ticks", "diag_drain_ticks", "diag_waves", "launches_tail")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


TRAVERSE_ORDERED = 0
TRAVERSE_EXHAUSTIVE = 1
COUNT_NODES = 2

# name -> (restype, argtypes).  tests/test_abi.py checks every name against include/tirt.h.
_vp = C.c_void_p
SIGNATURES = {
    "tirt_last_error": (C.c_char_p, []),
    "tirt_version": (C.c_int, []),
    "tirt_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "tirt_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "tirt_destroy": (None, [_vp]),
    "tirt_sync": (C.c_int, [_vp]),
    "tirt_set_option": (C.c_int, [_vp, C.c_char_p, C.c_double]),
    "tirt_scene_upload": (C.c_int, [_vp, _f32p, C.c_int, _i32p, C.c_int, _f32p, C.c_int,
                                    _f32p, C.c_int, _i32p, C.c_int, C.c_int, _f32p, _f32p]),
    "tirt_material_upload": (C.c_int, [_vp, _f32p, C.c_int]),
    "tirt_env_upload": (C.c_int, [_vp, _i32p, C.c_int, C.c_int, C.c_float]),
    "tirt_lbvh_build": (C.c_int, [_vp]),
    "tirt_lbvh_download": (C.c_int, [_vp, _vp, _vp, _vp]),
    "tirt_traversal_tree_download": (C.c_int, [_vp, _vp]),
    "tirt_morton_download": (C.c_int, [_vp, _i32p]),
    "tirt_process_normal": (C.c_int, [_vp, _i32p]),
    "tirt_vertex_download": (C.c_int, [_vp, _f32p]),
    "tirt_total_area": (C.c_int, [_vp, C.POINTER(C.c_float)]),
    "tirt_camera_set": (C.c_int, [_vp, _f32p, _f32p, _f32p, C.c_float, C.c_float, C.c_float, C.c_float]),
    "tirt_film_create": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "tirt_film_clear": (C.c_int, [_vp]),
    "tirt_pt_rgb_render": (C.c_int, [_vp, C.c_uint32, C.c_int, C.c_uint32, C.c_int, C.c_int, C.c_int]),
    "tirt_bdpt_rgb_render": (C.c_int, [_vp, C.c_uint32, C.c_int, C.c_uint32]),
    "tirt_bdpt_spec_render": (C.c_int, [_vp, C.c_uint32, C.c_int, C.c_uint32]),
    "tirt_tone_map": (C.c_int, [_vp, C.c_float]),
    "tirt_film_download": (C.c_int, [_vp, _vp, _vp]),
    "tirt_film_export_device": (C.c_int, [_vp, _vp]),
    "tirt_film_import_device": (C.c_int, [_vp, _vp]),
    "tirt_trace_closest": (C.c_int, [_vp, _f32p, C.c_int, C.c_int, C.c_int, _f32p, _i32p, _vp]),
    "tirt_trace_shadow": (C.c_int, [_vp, _f32p, C.c_int, C.c_int, C.c_int, _f32p, _i32p, _vp]),
    "tirt_comm_init": (C.c_int, [C.POINTER(_vp), C.c_int]),
    "tirt_film_reduce": (C.c_int, [C.POINTER(_vp), C.c_int, C.c_int]),
    "tirt_comm_destroy": (C.c_int, [C.POINTER(_vp), C.c_int]),
    "tirt_bvh_info": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "tirt_micro_gather_rate": (C.c_int, [_vp, C.c_uint64, C.c_int, C.POINTER(C.c_double)]),
    "tirt_trace_timeline": (C.c_int, [_vp, C.POINTER(C.c_uint64), C.c_int, C.POINTER(C.c_int)]),
    "tirt_primary_beam_stats": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "tirt_stats": (C.c_int, [_vp, C.POINTER(Stats)]),
    "tirt_stats_reset": (C.c_int, [_vp]),
    "tirt_kat_math": (C.c_int, [_vp, C.c_int, _f32p, _f32p, _f32p, C.c_int]),
    "tirt_kat_brdf": (C.c_int, [_vp, C.c_int, _f32p, C.c_int, _f32p, C.c_int, C.c_int]),
    "tirt_kat_spec": (C.c_int, [_vp, C.c_int, _f32p, C.c_int, _f32p, C.c_int, C.c_int]),
    "tirt_obj_load": (C.c_int, [C.c_char_p, C.POINTER(_vp)]),
    "tirt_obj_free": (None, [_vp]),
    "tirt_obj_material_count": (C.c_int, [_vp]),
    "tirt_obj_material_info": (C.c_int, [_vp, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int),
                                         C.POINTER(C.c_int), C.POINTER(C.c_longlong)]),
    "tirt_obj_material_vertices": (C.c_int, [_vp, C.c_int, np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS"), C.c_longlong]),
}

class SpectralTables(C.Structure):
    """tirt_spectral_t (include/tirt.h)"""
    _fields_ = [("sensor", C.c_void_p), ("n_sensor", C.c_int), ("s_min", C.c_float), ("s_max", C.c_float), ("s_range", C.c_float),
                ("spd", C.c_void_p), ("spd_n", C.c_int * 4), ("spd_min", C.c_float * 4), ("spd_max", C.c_float * 4), ("spd_range", C.c_float * 4),
                ("tbl_scale", C.c_void_p), ("tbl_data", C.c_void_p), ("tbl_res", C.c_int),
                ("sky_cfg", C.c_void_p), ("sky_rad", C.c_void_p), ("sun_dir", C.c_float * 3)]


SIGNATURES.update({
    "tirt_spec_table_build": (C.c_int, [_vp, C.c_int, _f32p, _f32p, C.c_int, _f32p, _f32p]),
    "tirt_spectral_upload": (C.c_int, [_vp, C.POINTER(SpectralTables)]),
    "tirt_pt_spec_render": (C.c_int, [_vp, C.c_uint32, C.c_int, C.c_uint32, C.c_int, C.c_int, C.c_int]),
})

_lib = None


def _prefer_torch_hip_runtime():
    """PyTorch-ROCm ships its own HIP runtime under the SONAME of the system one (libamdhip64.so.7); whichever is loaded first serves the
    whole process, and torch.cuda does not come up on the system one ("No HIP GPUs are available" after this library initialised HIP).
    So torch's copy is loaded first -- without importing torch -- when there is one: libtirt.so runs on it as it does when the
    application imported torch before this package (bench.py, the multi-GPU path).  TIRT_SYSTEM_HIP=1 keeps the system runtime."""
    if os.environ.get("TIRT_SYSTEM_HIP", "0") not in ("", "0") or os.environ.get("TIRT_NO_ENV_TUNING", "0") not in ("", "0"):
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(path):
            C.CDLL(path, mode=C.RTLD_GLOBAL)
    except (OSError, ImportError, ValueError):
        pass                                   # no torch, or a torch without a bundled runtime: the system one


def _check_one_hip_runtime():
    """libtirt.so needs `libamdhip64.so.7` (DT_NEEDED, by SONAME).  The preload above only satisfies that when torch's bundled runtime carries
    the same SONAME; a wheel with another (or a hashed) name leaves the loader to pull in the system runtime as well -- two HIP runtimes in one
    process, each with its own devices, streams and allocations.  Say so instead of failing later in some unrelated call."""
    try:
        with open("/proc/self/maps") as f:
            libs = sorted({line.split()[-1] for line in f if "libamdhip64" in line})
    except OSError:
        return
    if len({os.path.realpath(p) for p in libs}) > 1:
        msg = ("two HIP runtimes are mapped into this process (%s): libtirt.so and PyTorch will not see each other's device memory; "
               "set TIRT_SYSTEM_HIP=1 (and import torch after this package), or make the SONAMEs agree" % ", ".join(libs))
        if os.environ.get("TIRT_ALLOW_TWO_HIP_RUNTIMES", "0") in ("", "0"):       # loud by default (round-4 review); a process that never hands device pointers across may opt out
            raise TirtError(msg + " -- or TIRT_ALLOW_TWO_HIP_RUNTIMES=1 if the two never exchange device pointers")
        import warnings
        warnings.warn(msg, RuntimeWarning)


def lib():
    """Load libtirt.so once.  Raises TirtError (never falls back) when it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TirtError(
                "HIP extension %s is missing -- build it with `python -c 'import "
                "__graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
                "There is no CPU fallback." % LIB_PATH)
        _prefer_torch_hip_runtime()
        try:
            handle = C.CDLL(LIB_PATH)
        except OSError as exc:
            raise TirtError("cannot load %s: %s" % (LIB_PATH, exc))
        _check_one_hip_runtime()
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(code):
    if code != 0:
        msg = lib().tirt_last_error()
        raise TirtError("libtirt error %d: %s" % (code, msg.decode() if msg else "?"))


def _ptr(arr):
    return None if arr is None else arr.ctypes.data_as(C.c_void_p)


class Context:
    """RAII wrapper around tirt_ctx*."""

    def __init__(self, device_id=0):
        self._h = C.c_void_p()
        check(lib().tirt_create(int(device_id), C.byref(self._h)))
        self.device_id = int(device_id)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().tirt_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        if not self._h.value:
            raise TirtError("context is closed")
        return self._h

    # thin typed wrappers ---------------------------------------------------------------
    def sync(self):
        check(lib().tirt_sync(self.handle))

    def set_option(self, name, value):
        check(lib().tirt_set_option(self.handle, name.encode(), float(value)))

    def scene_upload(self, vertex, primitive, material, shape, light, light_count, bmin, bmax):
        vertex = np.ascontiguousarray(vertex, np.float32)
        primitive = np.ascontiguousarray(primitive, np.int32)
        material = np.ascontiguousarray(material, np.float32)
        shape = np.ascontiguousarray(shape, np.float32)
        light = np.ascontiguousarray(light, np.int32)
        check(lib().tirt_scene_upload(
            self.handle, vertex.reshape(-1), vertex.shape[0], primitive.reshape(-1), primitive.shape[0],
            material.reshape(-1), material.shape[0], shape.reshape(-1), shape.shape[0],
            light.reshape(-1), light.shape[0], int(light_count),
            np.ascontiguousarray(bmin, np.float32).reshape(-1), np.ascontiguousarray(bmax, np.float32).reshape(-1)))

    def material_upload(self, material):
        material = np.ascontiguousarray(material, np.float32)
        check(lib().tirt_material_upload(self.handle, material.reshape(-1), material.shape[0]))

    def env_upload(self, packed, power):
        packed = np.ascontiguousarray(packed, np.int32)
        check(lib().tirt_env_upload(self.handle, packed.reshape(-1), packed.shape[0], packed.shape[1], float(power)))

    def lbvh_build(self):
        check(lib().tirt_lbvh_build(self.handle))

    def lbvh_download(self, n, want_morton=True, want_bvh=True, want_compact=True):
        N = 2 * n - 1
        morton = np.zeros((n, 2), np.int32) if want_morton else None
        bvh = np.zeros((N, 11), np.float32) if want_bvh else None
        compact = np.zeros((N, 9), np.float32) if want_compact else None
        check(lib().tirt_lbvh_download(self.handle, _ptr(morton), _ptr(bvh), _ptr(compact)))
        return morton, bvh, compact

    def traversal_tree_download(self, n):
        """rows [(2n-1), 9] of the tree the ordered traversal walks (tirt.h)"""
        rows = np.zeros((2 * n - 1, 9), np.float32)
        check(lib().tirt_traversal_tree_download(self.handle, _ptr(rows)))
        return rows

    def morton_download(self, n):
        out = np.zeros((n, 2), np.int32)
        check(lib().tirt_morton_download(self.handle, out.reshape(-1)))
        return out

    def process_normal(self, vertex_index):
        check(lib().tirt_process_normal(self.handle, np.ascontiguousarray(vertex_index, np.int32)))

    def vertex_download(self, nv):
        out = np.zeros((nv, 9), np.float32)
        check(lib().tirt_vertex_download(self.handle, out.reshape(-1)))
        return out

    def total_area(self):
        v = C.c_float(0.0)
        check(lib().tirt_total_area(self.handle, C.byref(v)))
        return float(v.value)

    def camera_set(self, view, view_inv, eye, fx, fy, cx, cy):
        check(lib().tirt_camera_set(
            self.handle, np.ascontiguousarray(view, np.float32).reshape(-1),
            np.ascontiguousarray(view_inv, np.float32).reshape(-1),
            np.ascontiguousarray(eye, np.float32).reshape(-1), float(fx), float(fy), float(cx), float(cy)))

    def film_create(self, W, H, tile_rank=0, tile_count=1, tile_size=4096):
        check(lib().tirt_film_create(self.handle, int(W), int(H), int(tile_rank), int(tile_count), int(tile_size)))

    def film_clear(self):
        check(lib().tirt_film_clear(self.handle))

    def pt_rgb_render(self, frame_begin, frame_count, seed, max_depth=15, stack_size=64, flags=0):
        check(lib().tirt_pt_rgb_render(self.handle, int(frame_begin), int(frame_count), int(seed),
                                       int(max_depth), int(stack_size), int(flags)))

    def bdpt_rgb_render(self, frame_begin, frame_count, seed):
        check(lib().tirt_bdpt_rgb_render(self.handle, int(frame_begin), int(frame_count), int(seed)))

    def bdpt_spec_render(self, frame_begin, frame_count, seed):
        check(lib().tirt_bdpt_spec_render(self.handle, int(frame_begin), int(frame_count), int(seed)))

    def pt_spec_render(self, frame_begin, frame_count, seed, max_depth=10, stack_size=64, flags=0):
        check(lib().tirt_pt_spec_render(self.handle, int(frame_begin), int(frame_count), int(seed),
                                        int(max_depth), int(stack_size), int(flags)))

    def spec_table_build(self, res, cie_xyz, d65):
        cie_xyz = np.ascontiguousarray(cie_xyz, np.float32).reshape(-1)
        d65 = np.ascontiguousarray(d65, np.float32)
        scale = np.zeros(res, np.float32)
        coeff = np.zeros(9 * res * res * res, np.float32)
        check(lib().tirt_spec_table_build(self.handle, int(res), cie_xyz, d65, int(d65.size), scale, coeff))
        return scale, coeff

    def spectral_upload(self, t):
        """t: the dict of PT_Spec.PathTrace.tables()"""
        keep = [np.ascontiguousarray(t[k], np.float32) for k in ("sensor", "spd", "tbl_scale", "tbl_data", "sky_cfg", "sky_rad")]
        st = SpectralTables()
        st.sensor, st.spd, st.tbl_scale, st.tbl_data, st.sky_cfg, st.sky_rad = [a.ctypes.data for a in keep]
        st.n_sensor = int(t["n_sensor"]); st.s_min = t["s_min"]; st.s_max = t["s_max"]; st.s_range = t["s_range"]
        for k in range(4):
            st.spd_n[k] = int(t["spd_n"][k]); st.spd_min[k] = t["spd_min"][k]; st.spd_max[k] = t["spd_max"][k]; st.spd_range[k] = t["spd_range"][k]
        st.tbl_res = int(t["tbl_res"])
        for k in range(3):
            st.sun_dir[k] = t["sun_dir"][k]
        check(lib().tirt_spectral_upload(self.handle, C.byref(st)))

    def tone_map(self, exposure):
        check(lib().tirt_tone_map(self.handle, float(exposure)))

    def film_download(self, W, H, want_hdr=True, want_rgb=False):
        hdr = np.zeros((W, H, 3), np.float32) if want_hdr else None
        rgb = np.zeros((W, H, 3), np.float32) if want_rgb else None
        check(lib().tirt_film_download(self.handle, _ptr(hdr), _ptr(rgb)))
        return hdr, rgb

    def film_export_device(self, dev_ptr):
        check(lib().tirt_film_export_device(self.handle, C.c_void_p(int(dev_ptr))))

    def film_import_device(self, dev_ptr):
        check(lib().tirt_film_import_device(self.handle, C.c_void_p(int(dev_ptr))))

    def trace_closest(self, rays, stack_size=64, flags=0):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        nr = rays.shape[0]
        out = np.zeros((nr, 13), np.float32)
        prim = np.zeros(nr, np.int32)
        counts = np.zeros((nr, 2), np.int32) if (flags & COUNT_NODES) else None
        check(lib().tirt_trace_closest(self.handle, rays.reshape(-1), nr, int(stack_size), int(flags),
                                       out.reshape(-1), prim, _ptr(counts)))
        return out, prim, counts

    def trace_shadow(self, rays, stack_size=64, flags=0):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        nr = rays.shape[0]
        out = np.zeros(nr, np.float32)
        prim = np.zeros(nr, np.int32)
        counts = np.zeros((nr, 2), np.int32) if (flags & COUNT_NODES) else None
        check(lib().tirt_trace_shadow(self.handle, rays.reshape(-1), nr, int(stack_size), int(flags),
                                      out, prim, _ptr(counts)))
        return out, prim, counts

    def bvh_info(self):
        out = (C.c_uint64 * 4)()
        check(lib().tirt_bvh_info(self.handle, out))
        return {"node_bytes": int(out[0]), "prim_bytes": int(out[1]), "nodes": int(out[2]), "nodes_in_lds": int(out[3])}

    def trace_timeline(self, max_waves=1 << 16):
        """Per-wave [start, queue empty, end, hw id] of the launch armed by set_option("trace_timeline", k): (n, 4) uint64."""
        import numpy as np
        out = np.zeros((max_waves, 4), np.uint64); n = C.c_int(0)
        check(lib().tirt_trace_timeline(self.handle, out.ctypes.data_as(C.POINTER(C.c_uint64)), int(max_waves), C.byref(n)))
        return out[:min(n.value, max_waves)]

    def primary_beam_stats(self):
        """Diagnostics of the camera rays' candidate lists (tirt.h, tirt_primary_beam_stats)."""
        out = (C.c_uint64 * 12)()
        check(lib().tirt_primary_beam_stats(self.handle, out))
        return {"pixels_with_list": int(out[0]), "leaves_listed": int(out[1]), "pixels_all_probes_hit": int(out[2]), "rays_to_k_trace": int(out[3]), "rays": int(out[4]),
                "list_builds": int(out[5]), "list_build_ms": out[6] * 1.0e-6, "list_builds_skipped": int(out[7]),
                "diag_leaf_steps": int(out[8]), "diag_rays_more_than_one_step": int(out[9]), "diag_lane_slots": int(out[10]), "diag_rays_more_than_two_steps": int(out[11])}

    def micro_gather_rate(self, working_set_bytes, iters=2000):
        v = C.c_double(0.0)
        check(lib().tirt_micro_gather_rate(self.handle, int(working_set_bytes), int(iters), C.byref(v)))
        return float(v.value)

    def stats(self):
        st = Stats()
        rc = lib().tirt_stats(self.handle, C.byref(st))
        if rc == ERR_STACK:
            msg = lib().tirt_last_error()
            raise TirtStackOverflow("libtirt error %d: %s" % (rc, msg.decode() if msg else "?"), st.as_dict())
        check(rc)
        return st.as_dict()

    def stats_reset(self):
        check(lib().tirt_stats_reset(self.handle))

    def kat_math(self, fn, x, y=None):
        x = np.ascontiguousarray(x, np.float32)
        y = np.ascontiguousarray(y if y is not None else np.zeros_like(x), np.float32)
        out = np.zeros_like(x)
        check(lib().tirt_kat_math(self.handle, int(fn), x, y, out, x.size))
        return out

    def kat_brdf(self, which, inp, out_stride):
        inp = np.ascontiguousarray(inp, np.float32)
        n, stride = inp.shape
        out = np.zeros((n, out_stride), np.float32)
        check(lib().tirt_kat_brdf(self.handle, int(which), inp.reshape(-1), stride, out.reshape(-1), out_stride, n))
        return out

    def kat_spec(self, which, inp, out_stride):
        inp = np.ascontiguousarray(inp, np.float32)
        n, stride = inp.shape
        out = np.zeros((n, out_stride), np.float32)
        check(lib().tirt_kat_spec(self.handle, int(which), inp.reshape(-1), stride, out.reshape(-1), out_stride, n))
        return out


def device_count():
    v = C.c_int(0)
    check(lib().tirt_device_count(C.byref(v)))
    return v.value


class Communicator:
    """RCCL communicator over several contexts of ONE process (tirt_comm_init / tirt_film_reduce)."""

    def __init__(self, contexts):
        self.contexts = list(contexts)
        self._arr = (C.c_void_p * len(self.contexts))(*[c.handle.value for c in self.contexts])
        check(lib().tirt_comm_init(self._arr, len(self.contexts)))

    def film_reduce(self, root=0):
        check(lib().tirt_film_reduce(self._arr, len(self.contexts), int(root)))

    def close(self):
        if self._arr is not None:
            lib().tirt_comm_destroy(self._arr, len(self.contexts))
            self._arr = None
