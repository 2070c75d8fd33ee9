"""Experiment: what would sorting secondary rays buy k_trace?  Bounce-1-like rays of the headline scene (origins = primary
hit points, directions uniform on the sphere) traced in pixel order, sorted by origin cell (+ direction octant), and shuffled.
Run under `rocprofv3 --kernel-trace` and read the k_trace<0,false,0> durations in launch order:
   primary, then each of [pixel order, origin-sorted, octant+origin-sorted, shuffled] twice."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_api as oa
from ti_raytrace_amd import scenes

W = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
ex = scenes.synthetic(W, W, 4, device_id=0)
ex.build_scene(); ctx = ex.scene.ctx
prim_rays = oa.camera_rays(ex.cam, W, W).astype(np.float32)
out, prim, _ = ctx.trace_closest(prim_rays, 64, 0)
hit = prim >= 0
t = out[hit, 0:1]
pos = prim_rays[hit, :3] + prim_rays[hit, 3:] * t
r = np.random.RandomState(1)
d = r.normal(size=pos.shape).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
rays = np.concatenate([pos + d * 1e-3, d], axis=1).astype(np.float32)
n = rays.shape[0]
print("secondary rays", n)

def part1by2(x):
    x = x.astype(np.uint64) & 0x3ff
    x = (x | (x << 16)) & 0x30000ff
    x = (x | (x << 8)) & 0x300f00f
    x = (x | (x << 4)) & 0x30c30c3
    x = (x | (x << 2)) & 0x9249249
    return x

lo, hi = rays[:, :3].min(axis=0), rays[:, :3].max(axis=0)
g = np.clip((rays[:, :3] - lo) / (hi - lo) * 1024.0, 0, 1023).astype(np.int64)
morton = part1by2(g[:, 0]) | (part1by2(g[:, 1]) << 1) | (part1by2(g[:, 2]) << 2)
octant = ((rays[:, 3] < 0).astype(np.uint64) | ((rays[:, 4] < 0).astype(np.uint64) << 1) | ((rays[:, 5] < 0).astype(np.uint64) << 2))
# coarse cell (5 bits per axis) then octant then fine cell: rays of one neighbourhood and one direction class together
coarse = morton >> 15
orders = {
    "pixel": np.arange(n),
    "origin": np.argsort(morton, kind="stable"),
    "coarse+octant+fine": np.argsort((coarse << 18) | (octant << 15) | (morton & 0x7fff), kind="stable"),
    "octant+origin": np.argsort((octant << 30) | morton, kind="stable"),
    "shuffled": r.permutation(n),
}
ref = None
for name, o in orders.items():
    for rep in range(2):
        got, gp, _ = ctx.trace_closest(rays[o], 64, 0)
    back = np.empty(n, np.int32); back[o] = gp
    if ref is None: ref = back
    print(name, "same hits:", bool(np.array_equal(back, ref)))
