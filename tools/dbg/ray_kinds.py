"""Node visits and primitive tests per ray by kind (closest / NEE shadow), ordered traversal, one 32-frame step of the headline scene."""
import sys
sys.path.insert(0, ".")
from ti_raytrace_amd import scenes, _native
ex = scenes.synthetic(1024, 1024, 64, ntri=100000, device_id=0, seed=1)
ctx = ex.scene.ctx
ex.build_scene(); ctx.sync()
ctx.pt_rgb_render(0, 32, 1, 15, 64, 0); ctx.sync()
ctx.stats_reset()
ctx.pt_rgb_render(0, 32, 1, 15, 64, _native.TRAVERSE_ORDERED | _native.COUNT_NODES); ctx.sync()
s = ctx.stats()
for k in ("closest", "shadow"):
    r = s["rays_" + k]
    print("%-8s rays %10d   node visits per ray %.2f   primitive tests per ray %.2f" % (k, r, s["box_" + k] / 4.0 / r, s["leaf_" + k] / r))
