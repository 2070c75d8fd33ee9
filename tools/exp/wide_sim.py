"""Experiment (CPU only): what would 8-wide nodes buy k_trace?  The binary binned-SAH tree (tools/exp/sah_build.c) collapsed to 4-wide and 8-wide nodes,
walked near to far with distance culling by bounce-like rays (tools/exp/wide_sim.c): wide nodes visited and leaves tested per ray, and the LENGTH OF
THE DEPENDENT CHAIN of a ray (visits + leaf tests: what bounds the end of a launch), for children sorted by entry distance and in a fixed per-octant order,
with exact boxes and with boxes snapped to an 8-bit grid of the parent's box.
   python tools/exp/wide_sim.py [synthetic|veach|teapot]"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
from ti_raytrace_amd import scenes

which = sys.argv[1] if len(sys.argv) > 1 else "synthetic"
for src, so in (("sah_build.c", "/tmp/sah_build.so"), ("wide_sim.c", "/tmp/wide_sim.so")):
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tools/exp", src), "-lm"])
sah = C.CDLL("/tmp/sah_build.so"); sim = C.CDLL("/tmp/wide_sim.so")
if which == "synthetic":
    T = scenes.synthetic_triangles().astype(np.float32)
else:
    ex = (scenes.veach_bdpt if which == "veach" else scenes.single_model)(64, 64, 4, device_id=None)
    ex.scene.setup_data_cpu()
    P, V = ex.scene.primitive_np, ex.scene.vertex_np[:, :3].astype(np.float32)
    vi = P[P[:, 0] == 1, 1]
    T = np.stack([V[vi], V[vi + 1], V[vi + 2]], axis=1)
n = len(T)
r = np.random.RandomState(1)
k = r.randint(0, n, 200000); b = r.uniform(size=(len(k), 2)); b = np.where(b.sum(1, keepdims=True) > 1, 1 - b, b)
o = T[k, 0] + (T[k, 1] - T[k, 0]) * b[:, :1] + (T[k, 2] - T[k, 0]) * b[:, 1:]
d = r.normal(size=o.shape); d /= np.linalg.norm(d, axis=1, keepdims=True)
rays = np.concatenate([o + d * 1e-3 * np.ptp(T.reshape(-1, 3), axis=0).max(), d], 1).astype(np.float32)
boxes = np.concatenate([T.min(1), T.max(1)], 1).astype(np.float32)
N = 2 * n - 1
compact = np.zeros((N, 9), np.float32); csize = np.zeros(N, np.int32)
sah.sah_build(boxes.ctypes.data_as(C.c_void_p), n, compact.ctypes.data_as(C.c_void_p), csize.ctypes.data_as(C.c_void_p))
owner = np.arange(n, dtype=np.int32)
tris = np.ascontiguousarray(T.reshape(-1, 9))
print(which, n, "triangles")
for kk in (2, 4, 6, 8):
    for mode, quant, name in ((0, 0, "sorted, exact boxes"), (0, 255, "sorted, 8-bit boxes"), (1, 0, "octant order, exact"), (1, 255, "octant order, 8-bit")):
        if kk == 2 and (mode or quant): continue
        out = np.zeros(9); per = np.zeros(len(rays), np.int32)
        sim.simulate_wide(compact.ctypes.data_as(C.c_void_p), N, kk, mode, quant, owner.ctypes.data_as(C.c_void_p), tris.ctypes.data_as(C.c_void_p),
                          rays.ctypes.data_as(C.c_void_p), len(rays), out.ctypes.data_as(C.c_void_p), per.ctypes.data_as(C.c_void_p))
        print("k=%d %-22s nodes %6d  visits %.2f  leaf tests %.2f  children hit per visit %.2f  chain mean %.1f p99 %d p99.9 %d max %d" % (
            kk, name, int(out[8]), out[0], out[1], out[2], per.mean(), np.percentile(per, 99), np.percentile(per, 99.9), per.max()))
