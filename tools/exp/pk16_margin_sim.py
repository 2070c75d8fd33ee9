"""Round 6, VERDICT item 1 (b): what a WIDER conservative margin costs in node visits (CPU only).
A packed-fp16 box test (t = h * gA + gB for two children at once in v_pk_fma_f16) has 11 bits of mantissa: gB = (grid_min - o) / d rounded to fp16 is off by
up to |grid_min - o| * 2^-11 in POSITION (<= 4.9e-4 of the root box's extent for an origin inside it), h * gA by another |h * cell| * 2^-12 (<= 1.2e-4), the rounding
of the FMA's result by |t * d| * 2^-12 -- together ~8e-4 extents per plane against the 0.25 cells = 4.2e-6 extents of the fp32 test (tirt_render.hip, TR_GRID_AXIS).
The 4-wide SAH tree of tools/exp/sah_build.c walked as k_trace walks it (wide_sim.c: near to far, distance culled), every child box grown by the margin.
   python tools/exp/pk16_margin_sim.py [synthetic|veach|teapot]"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
from ti_raytrace_amd import scenes

which = sys.argv[1] if len(sys.argv) > 1 else "synthetic"
ntri = int(sys.argv[2]) if len(sys.argv) > 2 else 0
for src, so in (("sah_build.c", "/tmp/sah_build.so"), ("wide_sim.c", "/tmp/wide_sim.so")):
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tools/exp", src), "-lm"])
sah = C.CDLL("/tmp/sah_build.so"); sim = C.CDLL("/tmp/wide_sim.so")
sim.set_margin.argtypes = [C.c_float]
if which == "synthetic":
    T = (scenes.synthetic_triangles(ntri) if ntri else scenes.synthetic_triangles()).astype(np.float32)
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
extent = float(np.ptp(T.reshape(-1, 3), axis=0).max())
rays = np.concatenate([o + d * 1e-3 * extent, d], 1).astype(np.float32)
boxes = np.concatenate([T.min(1), T.max(1)], 1).astype(np.float32)
N = 2 * n - 1
compact = np.zeros((N, 9), np.float32); csize = np.zeros(N, np.int32)
sah.sah_build(boxes.ctypes.data_as(C.c_void_p), n, compact.ctypes.data_as(C.c_void_p), csize.ctypes.data_as(C.c_void_p))
owner = np.arange(n, dtype=np.int32)
tris = np.ascontiguousarray(T.reshape(-1, 9))
print(which, n, "triangles, extent %.3f; margins in root-box extents (one cell of the product's grid = 1.7e-5)" % extent)
base = None
for m in (0.0, 4.2e-6, 1e-4, 2e-4, 4e-4, 8e-4, 1.2e-3, 1.6e-3, 2.4e-3):
    sim.set_margin(m * extent)
    out = np.zeros(9); per = np.zeros(len(rays), np.int32)
    sim.simulate_wide(compact.ctypes.data_as(C.c_void_p), N, 4, 0, 0, owner.ctypes.data_as(C.c_void_p), tris.ctypes.data_as(C.c_void_p),
                      rays.ctypes.data_as(C.c_void_p), len(rays), out.ctypes.data_as(C.c_void_p), per.ctypes.data_as(C.c_void_p))
    if base is None: base = out.copy()
    print("margin %.1e  visits %.2f (%+.1f %%)  leaf tests %.2f (%+.1f %%)  chain mean %.1f" % (m, out[0], 100 * (out[0] / base[0] - 1), out[1], 100 * (out[1] / base[1] - 1), per.mean()))
