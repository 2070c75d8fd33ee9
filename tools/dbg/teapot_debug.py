import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from ti_raytrace_amd import scenes, _native
from test_gpu_trace import _grazing_rays
which = sys.argv[1] if len(sys.argv) > 1 else 'teapot'
ex = {'teapot': scenes.single_model, 'veach': scenes.veach_bdpt, 'cornell': scenes.cornell_box, 'synthetic': scenes.synthetic}[which](32, 32, 4, device_id=0); ex.build_scene(); ctx = ex.scene.ctx
n = (int(sys.argv[2]) if len(sys.argv) > 2 else 400000) // 14
rays = _grazing_rays(ex, n, 41)
a, ap, _ = ctx.trace_closest(rays, 64, 0)
b, bp, _ = ctx.trace_closest(rays, 64, _native.TRAVERSE_EXHAUSTIVE)
sa, sap, _ = ctx.trace_shadow(rays, 64, 0)
bad = np.flatnonzero((ap != bp) | (a[:, 0].view(np.uint32) != b[:, 0].view(np.uint32)) | (sap != bp))
print("bad", bad.size, "groups", np.bincount(bad // n, minlength=14))
nprim = ex.scene.primitive_count
for i in bad[:12]:
    print(i, "group", i // n, "o", rays[i, :3], "d", rays[i, 3:], "ordered", ap[i], a[i, 0], "exh", bp[i], b[i, 0], "shadow", sap[i], "sphere prim is", nprim - 1)
# which of the two is the reference's answer, and why: the oracle, and Moller-Trumbore's determinant for the triangle
import oracle_api as oa
o = oa.OracleScene(ex.scene, ex.cam); o.lbvh_build()
want, wprim, _ = o.closest_hit(rays[bad])
print("oracle prim", wprim, "t", want[:, 0])
v = ex.scene.vertex_np[:, :3].astype(np.float32)
P = ex.scene.primitive_np
for k, i in enumerate(bad):
    p = int(bp[i])
    if p < 0 or P[p, 0] != 1: continue
    vi = P[p, 1]; A, B, C = v[vi], v[vi + 1], v[vi + 2]
    ro = rays[i, :3].astype(np.float32); rd = rays[i, 3:].astype(np.float32)
    e1 = B - A; e2 = C - A; pv = np.cross(rd, e2); det = np.dot(e1, pv)
    nrm = np.cross(e1, e2)
    tv = [float(np.dot(x.astype(np.float64) - ro.astype(np.float64), rd.astype(np.float64))) for x in (A, B, C)]
    print(p, "det", det, "scale", np.linalg.norm(e1) * np.linalg.norm(pv), "cos(n,d)", np.dot(nrm, rd) / np.linalg.norm(nrm), "t of its vertices", tv)
