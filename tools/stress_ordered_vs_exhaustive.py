"""One-off stress of the ordered (quantised, verified) traversal against the reference-order traversal on the device:
millions of rays of several kinds per scene, closest hit (prim, t bits) and shadow query must agree.
python tools/stress_ordered_vs_exhaustive.py [rays_per_kind]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from ti_raytrace_amd import scenes, _native
from test_gpu_trace import _grazing_rays, random_rays

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
total = bad = 0
for name, make in (("cornell", lambda: scenes.cornell_box(32, 32, 4, device_id=0)), ("teapot", lambda: scenes.single_model(32, 32, 4, device_id=0)),
                   ("veach", lambda: scenes.veach_bdpt(32, 32, 4, device_id=0, integrator="pt")), ("synthetic100k", lambda: scenes.synthetic(32, 32, 4, device_id=0))):
    ex = make(); ex.build_scene(); ctx = ex.scene.ctx
    lo = ex.scene.minboundarynp[0].astype(np.float64); hi = ex.scene.maxboundarynp[0].astype(np.float64)
    ext = float((hi - lo).max()); ctr = 0.5 * (lo + hi)
    kinds = []
    r = np.random.RandomState(5)
    o = r.uniform(lo - 0.1 * ext, hi + 0.1 * ext, size=(n, 3)); d = r.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    kinds.append(("inside/random", np.concatenate([o, d], 1)))
    o = ctr + r.normal(size=(n, 3)) * 3 * ext; d = (ctr + r.uniform(-0.5, 0.5, (n, 3)) * ext) - o; d /= np.linalg.norm(d, axis=1, keepdims=True)
    kinds.append(("outside/aimed", np.concatenate([o, d], 1)))
    d = r.normal(size=(n, 3)); d[:, r.randint(0, 3)] *= 1e-7; d /= np.linalg.norm(d, axis=1, keepdims=True)
    kinds.append(("nearly axis-parallel", np.concatenate([r.uniform(lo, hi, size=(n, 3)), d], 1)))
    kinds.append(("grazing", _grazing_rays(ex, max(n // 14, 100), 41).astype(np.float64)))
    for kname, rays in kinds:
        rays = rays.astype(np.float32)
        t0 = time.time()
        a, ap, _ = ctx.trace_closest(rays, 64, 0)
        b, bp, _ = ctx.trace_closest(rays, 64, _native.TRAVERSE_EXHAUSTIVE)
        sa, sap, _ = ctx.trace_shadow(rays, 64, 0)
        mism = int((ap != bp).sum() + (a[:, 0].view(np.uint32) != b[:, 0].view(np.uint32)).sum() + (sap != bp).sum())
        total += rays.shape[0]; bad += mism
        print("%-14s %-22s %8d rays  hit %.2f  mismatches %d  (%.1f s)" % (name, kname, rays.shape[0], (bp >= 0).mean(), mism, time.time() - t0))
print("total %d rays, %d mismatches" % (total, bad))
