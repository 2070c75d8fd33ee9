"""Experiment (CPU only): what would splitting large / thin triangles into several references buy the traversal tree?
A binned-SAH tree (tools/exp/sah_build.c) over the primitives' boxes against the same builder over references -- each triangle cut
along its box's longest axis into pieces whose (clipped) boxes are at most `frac` of ... -- , both walked by the same ordered, t-culled
traversal (presplit_sim.c) with bounce-like rays: internal nodes visited and triangles tested per ray.
   python tools/exp/presplit_sim.py [synthetic|veach|teapot]"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
from ti_raytrace_amd import scenes

which = sys.argv[1] if len(sys.argv) > 1 else "synthetic"
for src, so in (("sah_build.c", "/tmp/sah_build.so"), ("presplit_sim.c", "/tmp/presplit_sim.so")):
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tools/exp", src), "-lm"])
sah = C.CDLL("/tmp/sah_build.so"); sim = C.CDLL("/tmp/presplit_sim.so")

if which == "synthetic":
    T = scenes.synthetic_triangles().astype(np.float32)
else:
    ex = (scenes.veach_bdpt if which == "veach" else scenes.single_model)(64, 64, 4, device_id=None)
    ex.scene.setup_data_cpu()
    P, V = ex.scene.primitive_np, ex.scene.vertex_np[:, :3].astype(np.float32)
    vi = P[P[:, 0] == 1, 1]
    T = np.stack([V[vi], V[vi + 1], V[vi + 2]], axis=1)
n = len(T)
print(which, n, "triangles")
r = np.random.RandomState(1)
# bounce-like rays: origins on random triangles (random barycentrics), directions uniform
k = r.randint(0, n, 200000); b = r.uniform(size=(len(k), 2)); b = np.where(b.sum(1, keepdims=True) > 1, 1 - b, b)
o = T[k, 0] + (T[k, 1] - T[k, 0]) * b[:, :1] + (T[k, 2] - T[k, 0]) * b[:, 1:]
d = r.normal(size=o.shape); d /= np.linalg.norm(d, axis=1, keepdims=True)
rays = np.concatenate([o + d * 1e-3 * np.ptp(T.reshape(-1, 3), axis=0).max(), d], 1).astype(np.float32)


def clip(poly, axis, lo, hi):
    """Sutherland-Hodgman against lo <= x[axis] <= hi"""
    for sgn, bound in ((1.0, lo), (-1.0, hi)):
        out = []
        for i in range(len(poly)):
            a, c = poly[i], poly[(i + 1) % len(poly)]
            da, dc = sgn * (a[axis] - bound) if sgn > 0 else bound - a[axis], sgn * (c[axis] - bound) if sgn > 0 else bound - c[axis]
            if da >= 0: out.append(a)
            if (da >= 0) != (dc >= 0):
                t = da / (da - dc); out.append(a + (c - a) * t)
        poly = out
        if not poly: break
    return poly


def make_refs(T, max_extent, max_parts=16):
    boxes, owner = [], []
    for ti in range(len(T)):
        work = [list(T[ti].astype(np.float64))]
        done = []
        while work:
            poly = work.pop()
            P = np.array(poly); mn, mx = P.min(0), P.max(0); ext = mx - mn
            ax = int(np.argmax(ext))
            if ext[ax] <= max_extent or len(done) + len(work) + 1 >= max_parts:
                done.append((mn, mx)); continue
            mid = 0.5 * (mn[ax] + mx[ax])
            a = clip(poly, ax, -1e30, mid); c = clip(poly, ax, mid, 1e30)
            for q in (a, c):
                if len(q) >= 3: work.append(q)
        for mn, mx in done:
            boxes.append(np.concatenate([mn, mx])); owner.append(ti)
    return np.array(boxes, np.float32), np.array(owner, np.int32)


def run(label, boxes, owner):
    m = len(boxes); N = 2 * m - 1
    compact = np.zeros((N, 9), np.float32); csize = np.zeros(N, np.int32)
    sah.sah_build(boxes.ctypes.data_as(C.c_void_p), m, compact.ctypes.data_as(C.c_void_p), csize.ctypes.data_as(C.c_void_p))
    out = np.zeros(3, np.float64); tt = np.zeros(len(rays), np.float32)
    sim.simulate(compact.ctypes.data_as(C.c_void_p), owner.ctypes.data_as(C.c_void_p), np.ascontiguousarray(T.reshape(-1, 9)).ctypes.data_as(C.c_void_p),
                 rays.ctypes.data_as(C.c_void_p), len(rays), out.ctypes.data_as(C.c_void_p), tt.ctypes.data_as(C.c_void_p))
    print("%-28s refs %7d (x%.2f)  internal visits %.2f  triangle tests %.2f  hit rate %.3f" % (label, m, m / n, out[0], out[1], out[2]))
    if m == n:
        for rel in (1e-3, 1e-2):
            o3 = np.zeros(3, np.float64)
            sim.simulate_plane_pretest(compact.ctypes.data_as(C.c_void_p), owner.ctypes.data_as(C.c_void_p), np.ascontiguousarray(T.reshape(-1, 9)).ctypes.data_as(C.c_void_p),
                                       rays.ctypes.data_as(C.c_void_p), len(rays), C.c_float(rel), o3.ctypes.data_as(C.c_void_p))
            print("   plane pre-test (crossing inside the leaf box's interval, widened by %g): triangle tests %.2f -> %.2f" % (rel, o3[0], o3[1]))
        for mode, name in ((0, "bounding sphere"), (1, "AABB")):
            for rel in (1e-3, 2e-2):
                o4 = np.zeros(3, np.float64)
                sim.simulate_point_pretest(compact.ctypes.data_as(C.c_void_p), owner.ctypes.data_as(C.c_void_p), np.ascontiguousarray(T.reshape(-1, 9)).ctypes.data_as(C.c_void_p),
                                           rays.ctypes.data_as(C.c_void_p), len(rays), C.c_float(rel), mode, o4.ctypes.data_as(C.c_void_p))
                print("   plane crossing in (0, best] and inside the %s (widened by %g): triangle tests %.2f -> %.2f, wrongly rejected hits %d" % (name, rel, o4[0], o4[1], int(o4[2])))
        for pk in (2, 3, 4):
            o2 = np.zeros(3, np.float64)
            sim.simulate_packets(compact.ctypes.data_as(C.c_void_p), csize.ctypes.data_as(C.c_void_p), pk, owner.ctypes.data_as(C.c_void_p),
                                 np.ascontiguousarray(T.reshape(-1, 9)).ctypes.data_as(C.c_void_p), rays.ctypes.data_as(C.c_void_p), len(rays), o2.ctypes.data_as(C.c_void_p))
            print("   leaf packets of <= %d: internal visits %.2f + packet visits %.2f, triangle tests %.2f" % (pk, o2[0], o2[1], o2[2]))
    return tt


base_boxes = np.concatenate([T.min(1), T.max(1)], 1).astype(np.float32)
t0 = run("one box per triangle", base_boxes, np.arange(n, dtype=np.int32))
ext = (T.max(1) - T.min(1)).max(1)
for q in (90, 50):
    lim = float(np.percentile(ext, q))
    bx, ow = make_refs(T, lim)
    t1 = run("longest extent <= p%d (%.4g)" % (q, lim), bx, ow)
    assert np.array_equal(t0, t1), "pre-splitting changed a hit"
