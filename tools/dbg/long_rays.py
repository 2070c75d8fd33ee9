"""Which rays are the long ones?  Bounce-1-like rays of the headline scene (origins = primary hit points, directions uniform) traced with per-ray
visit counts (ordered traversal): the distribution of node visits and what the rays of its tail have in common."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_api as oa
from ti_raytrace_amd import scenes, _native

W = 1024
ex = scenes.synthetic(W, W, 4, device_id=0)
ex.build_scene(); ctx = ex.scene.ctx
prim_rays = oa.camera_rays(ex.cam, W, W).astype(np.float32)
out, prim, _ = ctx.trace_closest(prim_rays, 64, 0)
hit = prim >= 0
pos = prim_rays[hit, :3] + prim_rays[hit, 3:] * out[hit, 0:1]
r = np.random.RandomState(1)
d = r.normal(size=pos.shape).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
rays = np.concatenate([pos + d * 1e-3, d], axis=1).astype(np.float32)
got, gp, cnt = ctx.trace_closest(rays, 64, _native.TRAVERSE_ORDERED | _native.COUNT_NODES)
nb = cnt[:, 0] / 4.0; nl = cnt[:, 1]
print("rays %d: node visits mean %.1f  p50 %.0f p90 %.0f p99 %.0f p99.9 %.0f p99.99 %.0f max %.0f; prim tests mean %.2f max %d" % (
    len(nb), nb.mean(), *np.percentile(nb, [50, 90, 99, 99.9, 99.99]), nb.max(), nl.mean(), nl.max()))
miss = gp < 0
print("misses %.3f of rays; node visits of misses mean %.1f, of hits mean %.1f" % (miss.mean(), nb[miss].mean(), nb[~miss].mean()))
tail = nb >= np.percentile(nb, 99.9)
print("tail (top 0.1 %%, %d rays): miss fraction %.3f; mean hit distance of tail hits %.3f vs all hits %.3f" % (
    tail.sum(), miss[tail].mean(), got[tail & ~miss, 0].mean() if (tail & ~miss).any() else -1, got[~miss, 0].mean()))
amin = np.abs(rays[:, 3:6]).min(axis=1)
print("smallest |direction component|: tail median %.4f, all median %.4f; tail fraction with a component < 0.01: %.3f (all: %.3f)" % (
    np.median(amin[tail]), np.median(amin), (amin[tail] < 0.01).mean(), (amin < 0.01).mean()))
share = np.sort(nb)[::-1].cumsum() / nb.sum()
print("share of all node visits in the longest 0.1 %% / 1 %% / 10 %% of rays: %.3f / %.3f / %.3f" % (share[len(nb) // 1000], share[len(nb) // 100], share[len(nb) // 10]))
# path length through the scene box of the tail rays
print("tail rays: travelled distance to hit / exit: hits %.2f" % (got[tail & ~miss, 0].mean() if (tail & ~miss).any() else -1))
for k in np.argsort(nb)[-5:]:
    print("  ray o=%s d=%s visits %.0f tests %d prim %d t %.3f" % (np.round(rays[k, :3], 3), np.round(rays[k, 3:], 4), nb[k], nl[k], gp[k], got[k, 0]))
